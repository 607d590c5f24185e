// stride_probe.hip -- how fast can 16-column x na-row tiles of a row-major (na, n)
// complex64 array be read, compared with the same bytes laid out tile-major?
// (design probe for the reassignment kernel's access pattern; not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void probe(const float2* __restrict__ W, float* __restrict__ out,
                                             long na, long n, int mode, int xcd) {
    extern __shared__ float dummy[];
    if (na < 0) dummy[threadIdx.x] = 1.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane >> 4, rl = lane & 15;
    long tile = blockIdx.x;
    if (xcd) { long per = gridDim.x >> 3; tile = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3); }
    const long col = wave * 4 + cl;
    float acc = 0.f;
    for (long i0 = 0; i0 < na; i0 += 64) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            long i = i0 + u * 16 + rl;
            v[u] = make_float2(0.f, 0.f);
            if (i < na) {
                long q = mode == 0 ? i * n + tile * 16 + col          // row-major, strided rows
                                   : (tile * na + i) * 16 + col;      // tile-major, contiguous
                v[u] = W[q];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}
int LDSB = 0;
int main(int argc, char** argv) {
    if (argc > 1) LDSB = atoi(argv[1]);
    int U = argc > 2 ? atoi(argv[2]) : 4; (void)U;
    const long na = 300, n = 160000, ntile = n / 16;
    float2* W; float* out;
    hipMalloc(&W, na * n * 8); hipMalloc(&out, ntile * 256 * 4);
    hipMemset(W, 0, na * n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int xcd = 0; xcd < 2; ++xcd) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(ntile), dim3(256), LDSB, 0, W, out, na, n, mode, xcd);
            hipEventRecord(e0);
            for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(probe, dim3(ntile), dim3(256), LDSB, 0, W, out, na, n, mode, xcd);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode=%s xcd_order=%d  %8.1f us  %6.2f TB/s\n", mode ? "tile-major" : "row-major ", xcd,
                   ms * 100, na * n * 8 / (ms / 10 * 1e-3) / 1e12);
        }
    return 0;
}
