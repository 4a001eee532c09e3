// Sanitizer harness ONLY (tests/cpu_emu): a minimal stand-in for <hip/hip_runtime.h> that lets the device
// code of egdst_amd/csrc be compiled by g++ with -fsanitize=address,undefined and executed with
// ONE-THREAD workgroups (WAVE=GRID_BS=ENV_BS=1), so barriers and cross-lane operations are trivial.
// It exists to find out-of-bounds accesses and undefined behaviour in the kernels without risking a
// GPU fault.  It is never built into, imported by, or reachable from the egdst_amd package.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#ifdef EMU_PARALLEL_BLOCKS
#define __shared__ static thread_local
#else
#define __shared__ static
#endif
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static thread_local dim3 threadIdx(0, 0, 0);
static thread_local dim3 blockIdx(0, 0, 0);
static dim3 blockDim, gridDim;

using std::isfinite;
using std::max;
using std::min;

#include <pthread.h>
#include <thread>
#include <vector>
static pthread_barrier_t emu_barrier;
static int emu_block_threads = 1;
static inline void __syncthreads()
{
    if (emu_block_threads > 1) pthread_barrier_wait(&emu_barrier);
}
#ifndef WAVE
#define WAVE 1
#endif
#if WAVE == 1
template <class T> static inline T __shfl(T v, int) { return v; }
template <class T> static inline T __shfl_up(T v, int) { return v; }
template <class T> static inline T __shfl_xor(T v, int) { return v; }
static inline int __any(int x) { return x != 0; }
static inline unsigned long long __ballot(int x) { return x ? 1ull : 0ull; }
#else
// Real multi-lane waves: the WAVE consecutive threads of a workgroup exchange values through a per-wave slot
// array and a per-wave barrier.  Every lane of the wave must reach the same sequence of collectives (that is
// also what the GPU requires of the code paths that use them).
#define EMU_MAX_WAVES 64
static pthread_barrier_t emu_wave_barrier[EMU_MAX_WAVES];
static unsigned long long emu_wave_slot[EMU_MAX_WAVES][WAVE];
static inline void emu_exchange(unsigned long long mine, unsigned long long *all)
{
    const int w = threadIdx.x / WAVE, l = threadIdx.x % WAVE;
    emu_wave_slot[w][l] = mine;
    pthread_barrier_wait(&emu_wave_barrier[w]);
    for (int k = 0; k < WAVE; k++) all[k] = emu_wave_slot[w][k];
    pthread_barrier_wait(&emu_wave_barrier[w]);
}
template <class T> static inline unsigned long long emu_pack(T v)
{
    unsigned long long u = 0;
    memcpy(&u, &v, sizeof(T));
    return u;
}
template <class T> static inline T emu_unpack(unsigned long long u)
{
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}
template <class T> static inline T __shfl(T v, int src)
{
    unsigned long long all[WAVE];
    emu_exchange(emu_pack(v), all);
    return emu_unpack<T>(all[src % WAVE]);
}
template <class T> static inline T __shfl_up(T v, int d)
{
    unsigned long long all[WAVE];
    emu_exchange(emu_pack(v), all);
    const int l = threadIdx.x % WAVE;
    return l >= d ? emu_unpack<T>(all[l - d]) : v;
}
template <class T> static inline T __shfl_xor(T v, int o)
{
    unsigned long long all[WAVE];
    emu_exchange(emu_pack(v), all);
    return emu_unpack<T>(all[(threadIdx.x % WAVE) ^ o]);
}
static inline unsigned long long __ballot(int x)
{
    unsigned long long all[WAVE], m = 0;
    emu_exchange(x ? 1ull : 0ull, all);
    for (int k = 0; k < WAVE; k++) m |= (all[k] & 1ull) << k;
    return m;
}
static inline int __any(int x) { return __ballot(x) != 0ull; }
#endif
static inline void __threadfence_block() { __sync_synchronize(); }
static inline void __threadfence() { __sync_synchronize(); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int atomicCAS(int *p, int cmp, int val) { return __sync_val_compare_and_swap(p, cmp, val); }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __sync_fetch_and_add(p, v); }
static inline int atomicAdd(int *p, int v) { return __sync_fetch_and_add(p, v); }
static inline int atomicOr(int *p, int v) { return __sync_fetch_and_or(p, v); }
static inline int atomicExch(int *p, int v) { return __sync_lock_test_and_set(p, v); }
static inline int atomicMin(int *p, int v)
{
    int o = *p;
    while (v < o && !__sync_bool_compare_and_swap(p, o, v)) o = *p;
    return o;
}
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __sync_fetch_and_add(p, v); }

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1 };
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, int) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
typedef void *hipEvent_t;
typedef void *hipGraphExec_t;
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, int) { *e = (hipEvent_t)1; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, int) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return 0; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)malloc(n ? n : 1); return *p ? 0 : 1; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipHostMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 1; }
static inline hipError_t hipHostFree(void *p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, int)
{
    for (size_t r = 0; r < h; r++) memcpy((char *)d + r * dp, (const char *)s + r * sp, w);
    return 0;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipMemset2DAsync(void *d, size_t pitch, int v, size_t w, size_t h, hipStream_t)
{
    for (size_t r = 0; r < h; r++) memset((char *)d + r * pitch, v, w);
    return 0;
}

#if WAVE > 1
#define EMU_WAVE_INIT(n) for (unsigned w_ = 0; w_ < ((n) + WAVE - 1) / WAVE && w_ < EMU_MAX_WAVES; w_++) pthread_barrier_init(&emu_wave_barrier[w_], nullptr, WAVE)
#define EMU_WAVE_DONE(n) for (unsigned w_ = 0; w_ < ((n) + WAVE - 1) / WAVE && w_ < EMU_MAX_WAVES; w_++) pthread_barrier_destroy(&emu_wave_barrier[w_])
#else
#define EMU_WAVE_INIT(n)
#define EMU_WAVE_DONE(n)
#endif
// Default: workgroups run one after another; the threads of a workgroup run concurrently (std::thread) and
// meet at a pthread barrier, so ThreadSanitizer sees intra-workgroup races.  With EMU_PARALLEL_BLOCKS (and
// one-thread workgroups) all workgroups of a launch run concurrently instead: inter-workgroup races.
#ifdef EMU_PARALLEL_BLOCKS
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                 \
    do {                                                                            \
        dim3 g_ = (grid), b_ = (block);                                             \
        gridDim = g_;                                                               \
        blockDim = b_;                                                              \
        emu_block_threads = 1;                                                      \
        std::vector<std::thread> th_;                                               \
        for (unsigned by_ = 0; by_ < g_.y; by_++)                                   \
            for (unsigned bx_ = 0; bx_ < g_.x; bx_++)                               \
                th_.emplace_back([=]() {                                            \
                    blockIdx = dim3(bx_, by_, 0);                                   \
                    threadIdx = dim3(0, 0, 0);                                      \
                    kernel(__VA_ARGS__);                                            \
                });                                                                 \
        for (auto &t_ : th_) t_.join();                                             \
    } while (0)
#else
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                 \
    do {                                                                            \
        dim3 g_ = (grid), b_ = (block);                                             \
        gridDim = g_;                                                               \
        blockDim = b_;                                                              \
        emu_block_threads = (int)b_.x;                                              \
        if (b_.x > 1) pthread_barrier_init(&emu_barrier, nullptr, b_.x);            \
        EMU_WAVE_INIT(b_.x);                                                        \
        for (unsigned bz_ = 0; bz_ < g_.z; bz_++)                                   \
            for (unsigned by_ = 0; by_ < g_.y; by_++)                               \
                for (unsigned bx_ = 0; bx_ < g_.x; bx_++) {                         \
                    if (b_.x == 1) {                                                \
                        blockIdx = dim3(bx_, by_, bz_);                             \
                        threadIdx = dim3(0, 0, 0);                                  \
                        kernel(__VA_ARGS__);                                        \
                    } else {                                                        \
                        std::vector<std::thread> th_;                               \
                        for (unsigned tx_ = 0; tx_ < b_.x; tx_++)                   \
                            th_.emplace_back([=]() {                                \
                                blockIdx = dim3(bx_, by_, bz_);                     \
                                threadIdx = dim3(tx_, 0, 0);                        \
                                kernel(__VA_ARGS__);                                \
                            });                                                     \
                        for (auto &t_ : th_) t_.join();                             \
                    }                                                               \
                }                                                                   \
        if (b_.x > 1) pthread_barrier_destroy(&emu_barrier);                        \
        EMU_WAVE_DONE(b_.x);                                                        \
    } while (0)
#endif
