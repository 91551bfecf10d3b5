// TEST-ONLY: checks the 9 x 29-bit field (field29.cuh) against the saturated field (field.cuh,
// itself checked against the oracle) on the host.  Build/run: see tests/test_field29_host.py
#define MIRA_CPU_EMU
#define F29_TRACK
#include "../../mira_amd/csrc/field29.cuh"
thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
pthread_barrier_t *emu_barrier = nullptr;
unsigned char *emu_dyn_shared = nullptr;

static uint64_t st = 0x1234567;
static uint64_t rnd() { st += 0x9E3779B97F4A7C15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <class F> Fe<typename F::Sat> rand_fe(int mode) {
    using S = typename F::Sat;
    Fe<S> v;
    for (int k = 0; k < 8; k += 2) { uint64_t x = rnd(); v.l[k] = (uint32_t)x; v.l[k + 1] = (uint32_t)(x >> 32); }
    if (mode == 1) for (int k = 0; k < 8; k++) v.l[k] = 0xFFFFFFFFu;
    if (mode == 2) for (int k = 1; k < 8; k++) v.l[k] = 0;
    if (mode == 3) { for (int k = 0; k < 8; k++) v.l[k] = S::P[k]; v.l[0] -= 1; return v; }   // P - 1
    if (mode == 4) return fe_zero<S>();
    for (int it = 0; it < 6; it++) { Fe<S> t; if (!sub_p(t, v)) v = t; }
    return v;
}

template <class F> int run(const char *name) {
    using S = typename F::Sat;
    int bad = 0;
    for (int it = 0; it < 200000; it++) {
        Fe<S> a = rand_fe<F>(it % 7), b = rand_fe<F>((it / 7) % 7), c = rand_fe<F>((it / 49) % 5);
        Fe29<F> A = f29_from_r256<F>(a), B = f29_from_r256<F>(b), C = f29_from_r256<F>(c);
        // round trip
        if (!fe_eq(f29_to_r256(A), a)) { bad++; if (bad < 5) printf("%s roundtrip fail it=%d\n", name, it); }
        // mul / sqr
        if (!fe_eq(f29_to_r256(f29_mul(A, B)), fe_mul(a, b))) { bad++; if (bad < 5) printf("%s mul fail it=%d\n", name, it); }
        if (!fe_eq(f29_to_r256(f29_sqr(A)), fe_sqr(a))) { bad++; if (bad < 5) printf("%s sqr fail it=%d\n", name, it); }
        // add / sub / dbl and chains as in the curve formulas: r^2 - ppp - 2q ; r * (q - x3) - y * ppp
        Fe29<F> AB = f29_mul(A, B), BC = f29_mul(B, C), AC = f29_mul(A, C);
        Fe<S> ab = fe_mul(a, b), bc = fe_mul(b, c), ac = fe_mul(a, c);
        if (!fe_eq(f29_to_r256(f29_add(AB, BC)), fe_add(ab, bc))) { bad++; if (bad < 5) printf("%s add fail\n", name); }
        if (!fe_eq(f29_to_r256(f29_sub<3>(AB, BC)), fe_sub(ab, bc))) { bad++; if (bad < 5) printf("%s sub fail\n", name); }
        Fe29<F> X3 = f29_sub<7>(f29_sqr(AB), f29_add(BC, f29_dbl(AC)));
        Fe<S> x3 = fe_sub(fe_sub(fe_sqr(ab), bc), fe_dbl(ac));
        if (!fe_eq(f29_to_r256(X3), x3)) { bad++; if (bad < 5) printf("%s x3 chain fail it=%d\n", name, it); }
        Fe29<F> Y3 = f29_sub<3>(f29_mul(AB, f29_sub<10>(AC, X3)), f29_mul(BC, X3));
        Fe<S> y3 = fe_sub(fe_mul(ab, fe_sub(ac, x3)), fe_mul(bc, x3));
        if (!fe_eq(f29_to_r256(Y3), y3)) { bad++; if (bad < 5) printf("%s y3 chain fail it=%d\n", name, it); }
        // subtraction of sums (3P-bounded subtrahend), product of un-carried sums
        Fe29<F> S2 = f29_add(AB, AC);
        if (!fe_eq(f29_to_r256(f29_sub<5>(BC, S2)), fe_sub(bc, fe_add(ab, ac)))) { bad++; if (bad < 5) printf("%s sub-of-sum fail\n", name); }
        Fe29<F> lazyA, lazyB;
        for (int i = 0; i < 9; i++) { lazyA.l[i] = AB.l[i] + AC.l[i]; lazyB.l[i] = BC.l[i] + AB.l[i]; }
        lazyA.bd = 4; lazyB.bd = 4;
        if (!fe_eq(f29_to_r256(f29_mul(lazyA, lazyB)), fe_mul(fe_add(ab, ac), fe_add(bc, ab)))) { bad++; if (bad < 5) printf("%s lazy mul fail\n", name); }
        // zero test
        Fe29<F> Z = f29_sub<3>(AB, AB);
        if (!f29_is_zero_mod_p<12>(Z)) { bad++; if (bad < 5) printf("%s zero test miss\n", name); }
        Fe29<F> Z2 = f29_sub<5>(f29_add(AB, BC), f29_add(BC, AB));
        if (!f29_is_zero_mod_p<12>(Z2)) { bad++; if (bad < 5) printf("%s zero test miss 2\n", name); }
        bool isz = fe_is_zero(fe_sub(ab, bc));
        if (f29_is_zero_mod_p<12>(f29_sub<3>(AB, BC)) != isz) { bad++; if (bad < 5) printf("%s zero test wrong\n", name); }
        // bounds: mul outputs loose
        Fe29<F> m = f29_mul(X3, Y3);
        for (int i = 0; i < 8; i++) if (m.l[i] > M29) { bad++; if (bad < 5) printf("%s limb bound\n", name); }
    }
    printf("%s: %s\n", name, bad ? "FAIL" : "ok");
    return bad;
}
int main() { return (run<Fq29>("Fq29") | run<Fr29>("Fr29")) ? 1 : 0; }
