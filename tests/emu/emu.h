// TEST-ONLY CPU emulation of the small subset of HIP the kernels use.  See
// mira_amd/csrc/platform.h.  Blocks run one after another; lanes of a block run either as a
// serial loop (LAUNCH: kernels without barriers) or as OS threads joined by a barrier
// (LAUNCH_BARRIER).  Atomics map to GCC __atomic builtins, so ThreadSanitizer sees them.
#pragma once
#include <pthread.h>

#include <algorithm>
#include <functional>
#include <thread>
#include <vector>

#define HD inline
#define DEV inline
#define KERNEL static
#define __shared__ static
#define __launch_bounds__(...)
#define __forceinline__ inline
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
extern pthread_barrier_t *emu_barrier;
extern unsigned char *emu_dyn_shared;

inline void __syncthreads() { if (emu_barrier) pthread_barrier_wait(emu_barrier); }
template <class T> inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicMax(T *p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
#define WAVE_SYNC() __syncthreads()   /* emulated lanes are OS threads: a wave-level exchange needs the block barrier (control flow is block-uniform wherever it is used) */
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void store_release_system(uint64_t *p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}

// quad29.cuh: emulated lanes are not in lockstep, so a lane cannot read its neighbours' registers;
// every lane computes all four products of a quad step itself (same values: the four lanes of a
// quad hold identical operands by contract)
static constexpr bool QUAD_COOPERATIVE = false;
inline unsigned quad_lane() { return threadIdx.x & 3u; }
template <int K> inline uint32_t quad_bcast(uint32_t v) { return v; }
#define DYN_SHARED(type, name) type *name = reinterpret_cast<type *>(emu_dyn_shared)

typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

template <class F> void emu_launch(bool barrier, dim3 grid, dim3 block, size_t shmem, F &&body) {
    gridDim = grid; blockDim = block;
    std::vector<unsigned char> sh(shmem + 16);
    emu_dyn_shared = sh.data();
    const unsigned nthreads = block.x * block.y * block.z;
    auto tid3 = [&](unsigned t) { return dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y)); };
    if (!barrier) {
        emu_barrier = nullptr;
        for (unsigned bz = 0; bz < grid.z; bz++)
            for (unsigned by = 0; by < grid.y; by++)
                for (unsigned bx = 0; bx < grid.x; bx++) {
                    blockIdx = dim3(bx, by, bz);
                    for (unsigned t = 0; t < nthreads; t++) { threadIdx = tid3(t); body(); }
                }
        return;
    }
    // barrier kernels: one OS thread per lane, created once per launch; the lanes walk the blocks
    // together (a block is finished by every lane before the next one starts, as LDS is reused)
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, nthreads);
    emu_barrier = &bar;
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            threadIdx = tid3(t);
            for (unsigned bz = 0; bz < grid.z; bz++)
                for (unsigned by = 0; by < grid.y; by++)
                    for (unsigned bx = 0; bx < grid.x; bx++) {
                        blockIdx = dim3(bx, by, bz);
                        body();
                        pthread_barrier_wait(&bar);
                    }
        });
    for (auto &x : th) x.join();
    pthread_barrier_destroy(&bar);
    emu_barrier = nullptr;
}
#define LAUNCH(kern, grid, block, shmem, stream, ...) \
    emu_launch(false, dim3(grid), dim3(block), (shmem), [&] { kern(__VA_ARGS__); })
#define LAUNCH_BARRIER(kern, grid, block, shmem, stream, ...) \
    emu_launch(true, dim3(grid), dim3(block), (shmem), [&] { kern(__VA_ARGS__); })
// blockDim-agnostic kernels run with a handful of lanes under emulation (OS threads are costly)
#define LAUNCH_BARRIER_FLEX(kern, grid, block, shmem, stream, ...) \
    emu_launch(true, dim3(grid), dim3(std::min<unsigned>((block), 8u)), (shmem), [&] { kern(__VA_ARGS__); })
