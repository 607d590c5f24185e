// Host stand-in for <hip/hip_runtime.h>, just large enough to run csrc/ssq_ridge.hip and
// csrc/ssq_kernels.hip on CPU threads (tests/emu/*_emu.cpp): one OS thread per work-item, pthread barriers
// for __syncthreads, a per-wavefront rendezvous for __ballot. TEST INFRASTRUCTURE ONLY --
// it checks the kernels' control flow and index arithmetic where no GPU is available.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>
#include <pthread.h>

// HIP's global-namespace device math that the kernels use unqualified
#include <algorithm>
using std::isinf; using std::isnan; using std::min; using std::max;
inline long long min(long long a, long b) { return a < b ? a : b; }
inline long long max(long long a, int b) { return a > b ? a : b; }
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
#define __align__(x) __attribute__((aligned(x)))

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(32) double4 { double x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) double2 { double x, y; };

typedef int hipError_t;
typedef void* hipStream_t;
constexpr hipError_t hipSuccess = 0;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 0;
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
// "device" memory is host memory
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };
constexpr int hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3;
inline hipError_t hipGetDeviceCount(int* c) { *c = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    __builtin_strcpy(p->name, "host emulation"); __builtin_strcpy(p->gcnArchName, "host");
    p->multiProcessorCount = 1; p->totalGlobalMem = 0;
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <typename P> hipError_t hipMalloc(P** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) {
    __builtin_memcpy(d, s, n); return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { __builtin_memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

namespace emu {
struct Block {
    pthread_barrier_t bar;
    std::vector<pthread_barrier_t> wave_bar;
    std::vector<unsigned long long> votes;      // one word per wavefront
    std::vector<int> xchg;                      // one word per lane (DPP moves)
};
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local Block* t_block;
}
#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)

inline void __syncthreads() { pthread_barrier_wait(&emu::t_block->bar); }

// ---- wavefront-level operations: every lane of the wavefront must take part (the kernels call
// them under wave-uniform control flow only), realised as rendezvous of the wave's 64 threads
inline void emu_wave_barrier() {
    pthread_barrier_wait(&emu::t_block->wave_bar[emu::t_threadIdx.x / 64]);
}
#define __builtin_amdgcn_wave_barrier emu_wave_barrier

// v_mov_b32_dpp row_shr:N (ctrl 0x110 + N): lane L of a 16-lane row reads lane L - N of the
// same row; without a source: 0 if bound_ctrl, else `old`
inline int emu_update_dpp(int old, int v, int ctrl, int, int, bool bound_ctrl) {
    emu::Block* b = emu::t_block;
    const unsigned w = emu::t_threadIdx.x / 64, lane = emu::t_threadIdx.x % 64;
    const int N = ctrl - 0x110;
    if (N < 1 || N > 15) abort();
    b->xchg[w * 64 + lane] = v;
    pthread_barrier_wait(&b->wave_bar[w]);
    const bool has = (int)(lane % 16) >= N;
    const int r = has ? b->xchg[w * 64 + lane - N] : (bound_ctrl ? 0 : old);
    pthread_barrier_wait(&b->wave_bar[w]);
    return r;
}
#define __builtin_amdgcn_update_dpp emu_update_dpp
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

inline float __log2f(float x) { return log2f(x); }          // v_log_f32 (1 ulp)
inline int __float_as_int(float f) { int i; __builtin_memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; __builtin_memcpy(&f, &i, 4); return f; }
inline int __double2loint(double d) { long long q; __builtin_memcpy(&q, &d, 8); return (int)(q & 0xffffffffll); }
inline int __double2hiint(double d) { long long q; __builtin_memcpy(&q, &d, 8); return (int)(q >> 32); }
inline double __hiloint2double(int hi, int lo) {
    long long q = ((long long)hi << 32) | (unsigned int)lo; double d; __builtin_memcpy(&d, &q, 8); return d;
}

// wave64 ballot: every lane of the (fully active) wavefront must call it
inline unsigned long long __ballot(bool pred) {
    emu::Block* b = emu::t_block;
    const unsigned w = emu::t_threadIdx.x / 64, lane = emu::t_threadIdx.x % 64;
    if (lane == 0) b->votes[w] = 0;
    pthread_barrier_wait(&b->wave_bar[w]);
    if (pred) __atomic_fetch_or(&b->votes[w], 1ull << lane, __ATOMIC_SEQ_CST);
    pthread_barrier_wait(&b->wave_bar[w]);
    const unsigned long long v = b->votes[w];
    pthread_barrier_wait(&b->wave_bar[w]);
    return v;
}

#define __builtin_amdgcn_ballot_w64 __ballot

template <typename K, typename... Args>
void emu_launch(K kernel, dim3 grid, dim3 block, Args... args) {
    const unsigned nt = block.x, nw = (nt + 63) / 64;
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            emu::Block blk;
            pthread_barrier_init(&blk.bar, nullptr, nt);
            blk.wave_bar.resize(nw); blk.votes.assign(nw, 0); blk.xchg.assign(nw * 64, 0);
            for (unsigned w = 0; w < nw; ++w) {
                const unsigned cnt = (w + 1) * 64 <= nt ? 64 : nt - w * 64;
                pthread_barrier_init(&blk.wave_bar[w], nullptr, cnt);
            }
            std::vector<std::thread> th;
            th.reserve(nt);
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([&, t] {
                    emu::t_threadIdx = dim3(t); emu::t_blockIdx = dim3(bx, by);
                    emu::t_blockDim = block; emu::t_gridDim = grid; emu::t_block = &blk;
                    kernel(args...);
                });
            for (auto& x : th) x.join();
            pthread_barrier_destroy(&blk.bar);
            for (auto& wb : blk.wave_bar) pthread_barrier_destroy(&wb);
        }
}
// (kernel), grid, block, dynamic LDS bytes, stream, args...
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    emu_launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)
