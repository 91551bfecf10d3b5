// The step right after the MSM in one NIFS fold: witness / error-vector
// folding, reference src/plonk/mod.rs:1097-1134 (RelaxedPlonkWitness::fold)
//     W[i] = W1[i] + r * W2[i]                      (par_iter axpy)
//     E[i] = E[i] + sum_k r^(k+1) * T_k[i]          (K cross-term vectors)
// Element-wise over vectors that already live in HBM, one multiplication per 96 bytes moved:
// HBM-bound.  Each lane handles one element with two 16-byte loads per operand; results are
// canonical, so they equal the reference's bit for bit.
#pragma once
#include "field29.cuh"

static constexpr int FOLD_MAX_TERMS = 16;
struct FoldTerms {
    const unsigned char *t[FOLD_MAX_TERMS];
};

// multiplier-form constants (9 limbs padded to 48 B, value * 2^261): see ntt_kernels.cuh tw_load
template <class F> DEV Fe29<F> fold_const_load(const unsigned char *p) {
    const U4 *q = reinterpret_cast<const U4 *>(p);
    U4 a = q[0], b = q[1], c = q[2];
    Fe29<F> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
    F29_SET(r, 1.0);
    return r;
}
template <class F> DEV Fe<typename F::Sat> fold_canonical(const Fe29<F> &v) { return reduce_once(f29_pack(v)); }

// out[i] = a[i] + r * b[i]
template <class F>
KERNEL void k_fold_axpy(const unsigned char *__restrict__ a, const unsigned char *__restrict__ b, const unsigned char *__restrict__ r_mult,
                        uint64_t n, unsigned char *__restrict__ out) {
    using S = typename F::Sat;
    const Fe29<F> r = fold_const_load<F>(r_mult);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<S> x = fe_load<S>(a + i * 32), y = fe_load<S>(b + i * 32);
        Fe<S> t = fold_canonical(f29_mul(f29_unpack_canonical<F>(y), r));
        fe_store(out + i * 32, fe_add(x, t));
    }
}
// e[i] += sum_k powers[k] * T_k[i],  powers[k] = r^(k+1) in multiplier form (48 B each)
template <class F>
KERNEL void k_fold_error(unsigned char *e, const unsigned char *e_in /* may be e: element i is read before it is written */, FoldTerms terms, uint32_t K,
                         const unsigned char *__restrict__ powers, uint64_t n) {
    using S = typename F::Sat;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<S> acc = fe_load<S>(e_in + i * 32);
        for (uint32_t k = 0; k < K; k++) {
            Fe<S> t = fe_load<S>(terms.t[k] + i * 32);
            acc = fe_add(acc, fold_canonical(f29_mul(f29_unpack_canonical<F>(t), fold_const_load<F>(powers + (size_t)k * 48))));
        }
        fe_store(e + i * 32, acc);
    }
}

// out[i] = sum_k coeffs[k] * vecs[k][i]   (arbitrary coefficients, multiplier form, 48 B each):
// FoldedTrace's witness folding, reference src/nifs/protogalaxy/poly/folded_trace.rs:54-131 --
// per challenge X the cell L_0(X) * acc[col][row] + sum_j L_j(X) * trace_j[col][row].
template <class F>
KERNEL void k_lincomb(unsigned char *__restrict__ out, FoldTerms vecs, uint32_t K, const unsigned char *__restrict__ coeffs, uint64_t n) {
    using S = typename F::Sat;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<S> acc = fe_zero<S>();
        for (uint32_t k = 0; k < K; k++) {
            Fe<S> t = fe_load<S>(vecs.t[k] + i * 32);
            acc = fe_add(acc, fold_canonical(f29_mul(f29_unpack_canonical<F>(t), fold_const_load<F>(coeffs + (size_t)k * 48))));
        }
        fe_store(out + i * 32, acc);
    }
}

// M linear combinations of the same J vectors in one sweep: out_m[i] = sum_j coeffs[m][j] * vecs[j][i].  Every vector
// element is read once for all M outputs -- the interpolation step of the cross terms (CrossTermPlan in
// mira_amd/graph_evaluator.py: d - 1 outputs over d + 1 evaluations of the gate polynomial).
static constexpr int LINCOMB_MAX_OUTS = 8;
struct FoldOuts {
    unsigned char *t[LINCOMB_MAX_OUTS];
};
template <class F>
KERNEL void k_lincomb_multi(FoldOuts outs, uint32_t M, FoldTerms vecs, uint32_t J, const unsigned char *__restrict__ coeffs, uint64_t n) {
    using S = typename F::Sat;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<S> acc[LINCOMB_MAX_OUTS];
#pragma unroll
        for (int m = 0; m < LINCOMB_MAX_OUTS; m++) acc[m] = fe_zero<S>();
        for (uint32_t j = 0; j < J; j++) {
            const Fe29<F> t = f29_unpack_canonical<F>(fe_load<S>(vecs.t[j] + i * 32));
#pragma unroll
            for (int m = 0; m < LINCOMB_MAX_OUTS; m++)
                if ((uint32_t)m < M) acc[m] = fe_add(acc[m], fold_canonical(f29_mul(t, fold_const_load<F>(coeffs + ((size_t)m * J + j) * 48))));
        }
#pragma unroll
        for (int m = 0; m < LINCOMB_MAX_OUTS; m++)
            if ((uint32_t)m < M) fe_store(outs.t[m] + i * 32, acc[m]);
    }
}

// One round of ProtoGalaxy's weighted tree reduction (compute_F / compute_G, reference
// src/nifs/protogalaxy/poly/mod.rs:131-166, 263-290): node(left, right) at height j is
// left + right * w[p][j].  A workgroup folds 2^levels consecutive values of point p = blockIdx.y:
// each lane first folds 2^serial of them in registers, then the workgroup folds log2(blockDim.x)
// more levels through LDS.  in: point p's values start at element p * in_stride (0: every point
// reads the same leaves); w: [p][total_levels] canonical Montgomery elements.
static constexpr uint32_t TREE_MAX_SERIAL = 3;
template <class FP>
KERNEL void k_pow_tree(const unsigned char *__restrict__ in, uint64_t in_stride, uint32_t serial, uint32_t levels, uint32_t level0,
                       const unsigned char *__restrict__ w, uint32_t total_levels, unsigned char *__restrict__ out, uint64_t out_stride) {
    DYN_SHARED(unsigned char, red);
    const uint32_t p = blockIdx.y;
    const unsigned char *wp = w + ((size_t)p * total_levels + level0) * 32;
    const unsigned char *src = in + ((size_t)p * in_stride + ((size_t)blockIdx.x << levels) + ((size_t)threadIdx.x << serial)) * 32;
    Fe<FP> v[1u << TREE_MAX_SERIAL];
    const uint32_t cnt0 = 1u << serial;
    for (uint32_t q = 0; q < (1u << TREE_MAX_SERIAL); q++)
        if (q < cnt0) v[q] = fe_load<FP>(src + (size_t)q * 32);
    for (uint32_t j = 0, cnt = cnt0; j < serial; j++, cnt >>= 1) {
        const Fe<FP> wj = fe_load<FP>(wp + (size_t)j * 32);
        for (uint32_t q = 0; q < (1u << (TREE_MAX_SERIAL - 1)); q++)
            if (q < (cnt >> 1)) v[q] = fe_add(v[2 * q], fe_mul(v[2 * q + 1], wj));
    }
    Fe<FP> acc = v[0];
    fe_store(red + (size_t)threadIdx.x * 32, acc);
    __syncthreads();
    for (uint32_t j = serial, width = blockDim.x; j < levels; j++, width >>= 1) {
        const bool active = threadIdx.x < (width >> 1);
        if (active) acc = fe_add(fe_load<FP>(red + (size_t)(2 * threadIdx.x) * 32), fe_mul(fe_load<FP>(red + (size_t)(2 * threadIdx.x + 1) * 32), fe_load<FP>(wp + (size_t)j * 32)));
        __syncthreads();
        if (active) fe_store(red + (size_t)threadIdx.x * 32, acc);
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(out + ((size_t)p * out_stride + blockIdx.x) * 32, acc);
}
