// Platform layer for the kernels in this directory.
//
// The product build is hipcc --offload-arch=gfx950 (everything below the #else).  The
// MIRA_CPU_EMU branch exists ONLY for tests/emu: it runs the same kernel sources on host
// threads (one OS thread per lane of a workgroup when the kernel uses a barrier) so kernel
// indexing can be checked, and sanitizers run, in a container without a GPU.  It is never
// part of libmira_gpu.so and mira_amd/ never loads it.
#pragma once
#ifdef __HIPCC_RTC__
// Runtime compilation of a specialised cross-term kernel (graph.hip, graph_jit.cuh): hiprtc brings the HIP device
// environment and the fixed-width integer types, but no libc headers; only the device-side arithmetic is needed.
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned long size_t;
#define HD __device__ __forceinline__
#define DEV __device__ __forceinline__
#else
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#endif

#if defined(__HIPCC_RTC__)
#elif !defined(MIRA_CPU_EMU)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#define DEV __device__ __forceinline__
#define KERNEL static __global__
#define LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (shmem), (stream), __VA_ARGS__)
#define LAUNCH_BARRIER LAUNCH
#define LAUNCH_BARRIER_FLEX LAUNCH   // kernel is correct for any blockDim (strided loops)
// lanes of one wave exchange data through LDS without a workgroup barrier: the DS unit serves a
// wave's instructions in order; the fence keeps the compiler from moving LDS accesses across it
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// DPP quads (quad29.cuh): lanes 4 i .. 4 i + 3 of a wave; quad_bcast<K> = every lane reads lane K of its quad
static constexpr bool QUAD_COOPERATIVE = true;
static __device__ __forceinline__ uint32_t quad_lane() { return threadIdx.x & 3u; }
template <int K> static __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xF, 0xF, true);   // quad_perm:[K,K,K,K]
}
// a word of mapped host memory the host spins on (ctx.h: rt_wait_flag)
static __device__ __forceinline__ void store_release_system(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
#define DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; type *name = reinterpret_cast<type *>(name##_raw)
#else
#include "../../tests/emu/emu.h"
#endif
