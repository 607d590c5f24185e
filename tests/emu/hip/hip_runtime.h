// Host stand-in for <hip/hip_runtime.h>, just large enough to run csrc/ssq_ridge.hip on
// CPU threads (tests/emu/ridge_emu.cpp): one OS thread per work-item, pthread barriers
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

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__

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

namespace emu {
struct Block {
    pthread_barrier_t bar;
    std::vector<pthread_barrier_t> wave_bar;
    std::vector<unsigned long long> votes;      // one word per wavefront
};
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local Block* t_block;
}
#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)

inline void __syncthreads() { pthread_barrier_wait(&emu::t_block->bar); }

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

template <typename K, typename... Args>
void emu_launch(K kernel, dim3 grid, dim3 block, Args... args) {
    const unsigned nt = block.x, nw = (nt + 63) / 64;
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            emu::Block blk;
            pthread_barrier_init(&blk.bar, nullptr, nt);
            blk.wave_bar.resize(nw); blk.votes.assign(nw, 0);
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
