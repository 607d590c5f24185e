// acc_read_probe.hip -- design aid (not shipped): read-side throughput of the reassignment
// kernel's tile access for two lane mappings at the same occupancy (38.4 KB LDS per workgroup):
//   A: wave = 4 columns x 16 rows per load instruction (32-byte runs; the shipped kernel)
//   B: wave = 16 columns x 4 rows per load instruction (128-byte runs)
// Loads only (Wx 8 B + bin map 2 B per point), U row batches in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PAT, int U, int WR>
__global__ __launch_bounds__(256) void rd(const float2* __restrict__ W, const unsigned short* __restrict__ K,
                                         float2* __restrict__ out, float2* __restrict__ Tout, int na, int n) {
    extern __shared__ float2 tile[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = gridDim.x >> 3;
    const int tile_id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile_id * 16 >= n) return;
    int col, row0;                       // this lane's column and its row inside a 16-row batch
    if (PAT == 0) { col = wave * 4 + (lane >> 4); row0 = lane & 15; }
    else { col = lane & 15; row0 = wave * 4 + (lane >> 4); }
    const int j = tile_id * 16 + col;
    float2 z[U]; unsigned short kk[U];
    auto req = [&](int u, int i) {
        z[u] = make_float2(0.f, 0.f); kk[u] = 0;
        if (i < na) { unsigned q = (unsigned)i * n + j; z[u] = W[q]; kk[u] = K[q]; }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) req(u, u * 16 + row0);
    float2 acc = make_float2(0.f, 0.f);
    for (int i0 = 0; i0 < na; i0 += 16 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc.x += z[u].x + kk[u]; acc.y += z[u].y;
            req(u, i0 + u * 16 + row0 + 16 * U);
        }
    }
    if (acc.x == 12345.f) { tile[threadIdx.x] = acc; out[j] = tile[0]; }
    if (WR) {                            // write-out of the tile, as the real kernel does
        __syncthreads();
        const int cc = threadIdx.x & 15, rr = threadIdx.x >> 4, jj = tile_id * 16 + cc;
        for (int k = rr; k < na; k += 16) Tout[(unsigned)k * n + jj] = tile[(k * 16 + cc) & 4095];
    }
}
template <int PAT, int U, int WR> void run(const float2* W, const unsigned short* K, float2* o, float2* T, int na, int n, int sig) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = ((n / 16 + 7) / 8) * 8; size_t lds = (size_t)na * 16 * 8;
    hipFuncSetAttribute((const void*)rd<PAT, U, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rd<PAT, U, WR>), dim3(grid, sig), dim3(256), lds, 0, W, K, o, T, na, n);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((rd<PAT, U, WR>), dim3(grid, sig), dim3(256), lds, 0, W, K, o, T, na, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("pattern %c U=%d write=%d: %.1f us  %.2f TB/s\n", PAT ? 'B' : 'A', U, WR, ms * 1e3, (double)na * n * (10 + 8 * WR) / (ms * 1e-3) / 1e12);
}
int main() {
    const int na = 300, n = 160000;
    float2* W; unsigned short* K; float2* o;
    hipMalloc(&W, (size_t)na * n * 8); hipMalloc(&K, (size_t)na * n * 2); hipMalloc(&o, (size_t)n * 8);
    hipMemset(W, 0, (size_t)na * n * 8); hipMemset(K, 0, (size_t)na * n * 2);
    // blockIdx.y ignored by the kernel (same data re-read): one signal's worth per launch
    float2* T; hipMalloc(&T, (size_t)na * n * 8);
    run<0, 8, 0>(W, K, o, T, na, n, 1); run<1, 8, 0>(W, K, o, T, na, n, 1);
    run<0, 8, 1>(W, K, o, T, na, n, 1); run<1, 8, 1>(W, K, o, T, na, n, 1);
    run<1, 16, 1>(W, K, o, T, na, n, 1); run<1, 4, 1>(W, K, o, T, na, n, 1);
    return 0;
}
