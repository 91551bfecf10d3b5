// TEST-ONLY: XYZZ arithmetic over the 29-bit-limb field (curve29.cuh) against the saturated-limb
// implementation (curve.cuh), with bound tracking on (F29_TRACK asserts every precondition).
#define MIRA_CPU_EMU
#define F29_TRACK
#include "../../mira_amd/csrc/curve.cuh"
#include "../../mira_amd/csrc/curve29.cuh"
thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
pthread_barrier_t *emu_barrier = nullptr;
unsigned char *emu_dyn_shared = nullptr;

static uint64_t st = 99;
static uint64_t rnd() { st += 0x9E3779B97F4A7C15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <class F> Aff<typename F::Sat> gen_point(const Aff<typename F::Sat> &g, uint64_t k) {
    using S = typename F::Sat;
    Xyzz<S> acc = xyzz_identity<S>();
    for (int bit = 63; bit >= 0; bit--) { acc = xyzz_double(acc); if ((k >> bit) & 1) xyzz_add_affine(acc, g); }
    return xyzz_to_affine(acc);
}
// saturated R-form affine -> stored base format (canonical saturated R'-form), as k_convert_bases does
template <class F> void to_stored(const Aff<typename F::Sat> &p, unsigned char *out) {
    using S = typename F::Sat;
    if (aff_is_identity(p)) { memset(out, 0, 64); return; }
    Fe<S> x = reduce_once(f29_pack(f29_from_r256<F>(p.x))), y = reduce_once(f29_pack(f29_from_r256<F>(p.y)));
    fe_store(out, x); fe_store(out + 32, y);
}
template <class F> bool same(const Xyzz29<F> &a, const Xyzz<typename F::Sat> &b) {
    using S = typename F::Sat;
    unsigned char buf[128];
    xyzz29_export_r256(buf, a);
    Xyzz<S> e = xyzz_load<S>(buf);
    Aff<S> pa = xyzz_to_affine(e), pb = xyzz_to_affine(b);
    return fe_eq(pa.x, pb.x) && fe_eq(pa.y, pb.y);
}
template <class F> int run(const char *name, uint32_t gx, uint32_t gy_small, const Fe<typename F::Sat> *gy_full) {
    using S = typename F::Sat;
    Aff<S> g;
    Fe<S> t = fe_zero<S>(); t.l[0] = gx; g.x = fe_to_mont(t);
    if (gy_full) g.y = fe_to_mont(*gy_full); else { t.l[0] = gy_small; g.y = fe_to_mont(t); }
    int bad = 0;
    for (int it = 0; it < 300; it++) {
        Xyzz<S> acc = xyzz_identity<S>();
        Xyzz29<F> acc29 = xyzz29_identity<F>();
        Xyzz<S> other = xyzz_identity<S>();
        Xyzz29<F> other29 = xyzz29_identity<F>();
        Aff<S> last = g;
        for (int step = 0; step < 40; step++) {
            int mode = (int)(rnd() % 10);
            Aff<S> p = gen_point<F>(g, rnd() | 1);
            if (mode == 0) p = last;                                    // repeat: doubling path when acc == p
            if (mode == 1) { p = last; p.y = fe_neg(p.y); }             // opposite
            if (mode == 2) { p.x = fe_zero<S>(); p.y = fe_zero<S>(); }  // identity base
            if (mode == 3 && step > 0) {                                 // acc == p exactly
                p = xyzz_to_affine(acc);
            }
            bool neg = (rnd() & 1) != 0;
            unsigned char stored[64];
            to_stored<F>(p, stored);
            Aff29<F> p29 = aff29_load<F>(stored, neg);
            Aff<S> ps = p;
            if (neg) ps.y = fe_neg(ps.y);
            xyzz_add_affine(acc, ps);
            xyzz29_add_affine(acc29, p29);
            if (!same(acc29, acc)) { bad++; if (bad < 5) printf("%s madd mismatch it=%d step=%d mode=%d\n", name, it, step, mode); }
            last = p;
            if (mode == 4) { acc = xyzz_double(acc); acc29 = xyzz29_double(acc29); }
            if (mode == 5) { xyzz_add(other, acc); xyzz29_add(other29, acc29); }
            if (mode == 6) { xyzz_add(acc, other); xyzz29_add(acc29, other29); }
            if (mode == 7) { Xyzz<S> c = acc; xyzz_add(acc, c); Xyzz29<F> c29 = acc29; xyzz29_add(acc29, c29); }   // full add of equal points
            if (mode == 8) {   // store / load round trip of the partial-sum format
                unsigned char buf[XYZZ29_BYTES];
                xyzz29_store(buf, acc29);
                acc29 = xyzz29_load<F>(buf);
            }
            if (!same(acc29, acc) || !same(other29, other)) { bad++; if (bad < 5) printf("%s mismatch it=%d step=%d mode=%d\n", name, it, step, mode); }
        }
    }
    printf("%s: %s\n", name, bad ? "FAIL" : "ok");
    return bad;
}
int main() {
    Fe<FrP> gy = {{0x823f272cu, 0x833fc48du, 0xf1181294u, 0x2d270d45u, 0x06a45d63u, 0xcf135e75u, 0x2u, 0u}};
    int bad = run<Fq29>("bn256 (Fq29)", 1, 2, nullptr);
    bad |= run<Fr29>("grumpkin (Fr29)", 1, 0, &gy);
    return bad ? 1 : 0;
}
