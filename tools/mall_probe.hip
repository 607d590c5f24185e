// mall_probe.hip -- does data just written stay in the 256 MiB Infinity Cache for a
// strided tile read that follows immediately? (design probe, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void wr(float2* W, long total) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) W[t] = make_float2((float)t, 1.f);
}
__global__ __launch_bounds__(256) void rd(const float2* __restrict__ W, float* __restrict__ out, long na, long n) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane >> 4, rl = lane & 15;
    long per = gridDim.x >> 3; long tile = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const long col = wave * 4 + cl; float acc = 0.f;
    for (long i0 = 0; i0 < na; i0 += 64) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { long i = i0 + u * 16 + rl; v[u] = make_float2(0.f, 0.f); if (i < na) v[u] = W[i * n + tile * 16 + col]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const long n = 160000, ntile = n / 16;
    float2* W; float* out; hipMalloc(&W, 300 * n * 8); hipMalloc(&out, ntile * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (long na : {75L, 150L, 300L}) {
        float tot = 0;
        for (int rep = 0; rep < 6; ++rep) {
            hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, W, na * n);
            hipEventRecord(e0);
            hipLaunchKernelGGL(rd, dim3(ntile), dim3(256), 0, 0, W, out, na, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep) tot += ms;
        }
        printf("write %ld MB then strided read: %7.1f us  %5.2f TB/s\n", na * n * 8 / 1000000, tot / 5 * 1000, na * n * 8 / (tot / 5 * 1e-3) / 1e12);
    }
    return 0;
}
