// Development: the host-side field and point arithmetic of the commit epilogue (host_field.hpp) -- ns per Montgomery product
// in a dependent chain, and the Horner chain of one commit (W window sums, c doublings between them) in XYZZ -- and in Jacobian
// coordinates when host_field.hpp defines HOSTF_HAS_JACOBIAN (the variant of profiles/r03_h_small_commit_tail.txt; not shipped).   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench_hostf.hip -o tools/microbench_hostf
#include <chrono>
#include <cstdio>
#include <vector>

#include "../mira_amd/csrc/host_field.hpp"
using namespace hostf;
typedef FqP FP;

template <class Fn> static double ns_per(int n, Fn fn) {
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        fn(n);
        best = std::min(best, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / n);
    }
    return best;
}

int main() {
    HFe<FP> a = from_u64<FP>(12345), b = from_u64<FP>(987654321);
    HFe<FP> sink = a;
    {   // the dedicated squaring against the product, over a chain that visits unrelated values
        HFe<FP> x = a; int bad = 0;
        for (int i = 0; i < 100000; i++) { HFe<FP> s1 = sqr(x), s2 = mul(x, x); bad += memcmp(&s1, &s2, 32) != 0; x = add(mul(s1, b), a); }
        HFe<FP> top = {{P64<FP>(0) - 1, P64<FP>(1), P64<FP>(2), P64<FP>(3)}};
        HFe<FP> s1 = sqr(top), s2 = mul(top, top); bad += memcmp(&s1, &s2, 32) != 0;
        printf("sqr == mul(x, x) on 100001 values: %s\n", bad ? "NO" : "yes");
    }
    printf("mul, dependent chain      %.2f ns\n", ns_per(2000000, [&](int n) { HFe<FP> x = a; for (int i = 0; i < n; i++) x = mul(x, b); sink = x; }));
    printf("sqr, dependent chain      %.2f ns\n", ns_per(2000000, [&](int n) { HFe<FP> x = a; for (int i = 0; i < n; i++) x = sqr(x); sink = x; }));
    printf("add, dependent chain      %.2f ns\n", ns_per(2000000, [&](int n) { HFe<FP> x = a; for (int i = 0; i < n; i++) x = add(x, b); sink = x; }));
    printf("sub, dependent chain      %.2f ns\n", ns_per(2000000, [&](int n) { HFe<FP> x = a; for (int i = 0; i < n; i++) x = sub(b, x); sink = x; }));
    // a point: (1, 2) on y^2 = x^3 + 3
    HXyzz<FP> g = identity<FP>();
    g.x = from_u64<FP>(1); g.y = from_u64<FP>(2); g.zz = one<FP>(); g.zzz = one<FP>();
    std::vector<HXyzz<FP>> sums(32);
    sums[0] = g;
    for (int i = 1; i < 32; i++) sums[i] = add_pt(dbl_pt(sums[i - 1]), g);
    uint64_t out[8], out2[8];
    for (int c : {8, 16}) {
        const int W = 256 / c;
        double t = ns_per(200, [&](int n) {
            for (int it = 0; it < n; it++) {
                HXyzz<FP> acc = identity<FP>();
                for (int w = W - 1; w >= 0; w--) { for (int k = 0; k < c; k++) acc = dbl_pt(acc); acc = add_pt(acc, sums[w]); }
                to_affine(acc, out);
            }
        });
        printf("Horner c = %2d, XYZZ       %.2f us\n", c, t / 1e3);
#ifdef HOSTF_HAS_JACOBIAN
        double tj = ns_per(200, [&](int n) {
            for (int it = 0; it < n; it++) {
                HJac<FP> acc = jac_identity<FP>();
                for (int w = W - 1; w >= 0; w--) { for (int k = 0; k < c; k++) acc = dbl_jac(acc); acc = add_jac_xyzz(acc, sums[w]); }
                to_affine_jac(acc, out2);
            }
        });
        printf("Horner c = %2d, Jacobian   %.2f us   same point: %d\n", c, tj / 1e3, memcmp(out, out2, 64) == 0);
#endif
    }
    printf("(%llx)\n", (unsigned long long)sink.l[0]);
}
