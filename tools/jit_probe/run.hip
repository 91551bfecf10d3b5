// Development probe: time a precompiled code object's jit_eval kernel over 2^17 rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Cols { const unsigned char *p[44]; };
int main(int argc, char **argv) {
    const uint64_t n = 1 << 17;
    hipModule_t mod; hipFunction_t fn;
    CK(hipModuleLoad(&mod, argv[1]));
    CK(hipModuleGetFunction(&fn, mod, "jit_eval"));
    unsigned char *d; CK(hipMalloc(&d, 44 * n * 32));
    std::vector<unsigned char> h(44 * n * 32);
    for (size_t i = 0; i < h.size(); i++) h[i] = (i % 32 == 31) ? 0x10 : (unsigned char)rand();
    CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
    uint32_t *consts; CK(hipMalloc(&consts, 8 * 9 * 4)); CK(hipMemset(consts, 1, 8 * 9 * 4));
    unsigned char *out; CK(hipMalloc(&out, n * 32));
    Cols cols; for (int i = 0; i < 44; i++) cols.p[i] = d + (size_t)i * n * 32;
    uint64_t nrows = n;
    struct { Cols c; const uint32_t *k; unsigned char *o; uint64_t n; } args{cols, consts, out, nrows};
    size_t sz = sizeof(args);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int block : {64, 128, 256}) {
        for (int it = 0; it < 3; it++) CK(hipModuleLaunchKernel(fn, n / block, 1, 1, block, 1, 1, 0, 0, nullptr, cfg));
        CK(hipEventRecord(a, 0));
        for (int it = 0; it < 7; it++) CK(hipModuleLaunchKernel(fn, n / block, 1, 1, block, 1, 1, 0, 0, nullptr, cfg));
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%s block %d: 7 launches %.3f ms (%.3f each)\n", argv[1], block, ms, ms / 7);
    }
    return 0;
}
