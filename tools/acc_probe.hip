// acc_probe.hip -- ablation probe for the reassignment kernel (design aid, not shipped):
// same memory / LDS traffic as the 16-column tile kernel, with switchable stages.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE, int U, int TCOLS>
__global__ __launch_bounds__(TCOLS * 16) void acc(const float2* __restrict__ W, const unsigned short* __restrict__ K,
                                              const float* __restrict__ cst, float2* __restrict__ Tx, int na, int n) {
    extern __shared__ float2 tile[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane >> 4, rl = lane & 15;
    float2* slab = tile + wave * na * 4;
    const int per = gridDim.x >> 3;
    const int tile_id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile_id * TCOLS >= n) return;
    const int j = tile_id * TCOLS + wave * 4 + cl;
    if (MODE >= 1) { for (int t = lane; t < na * 4; t += 64) slab[t] = make_float2(0.f, 0.f); }
    float2 z[U]; unsigned short kk[U]; float wt[U];
    auto req = [&](int u, int i) {
        z[u] = make_float2(0.f, 0.f); kk[u] = 0xFFFF; wt[u] = 0.f;
        if (i < na) { unsigned q = (unsigned)i * n + j; z[u] = W[q]; kk[u] = K[q]; wt[u] = cst[i]; }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) req(u, u * 16 + rl);
    float2 acc = make_float2(0.f, 0.f);
    for (int i0 = 0; i0 < na; i0 += 16 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int i = i0 + u * 16 + rl;
            int k = kk[u] == 0xFFFF ? -1 : kk[u];
            float2 t = make_float2(z[u].x * wt[u], z[u].y * wt[u]);
            if (MODE >= 1) {
                if (k >= 0) {
                    float2* cell = slab + (k * 4 + ((cl + k) & 3));
                    float2 o = *cell;
                    if (MODE >= 3) {   // all-pairs fold, as in the real kernel (unordered variant)
#pragma unroll
                        for (int d = 1; d < 16; ++d) {
                            int ks = __shfl_up(k, d, 16);
                            float tx = __shfl_up(t.x, d, 16), ty = __shfl_up(t.y, d, 16);
                            if (rl >= d && ks == k) { o.x += tx; o.y += ty; }
                        }
                    }
                    o.x += t.x; o.y += t.y;
                    *cell = o;
                }
            } else { acc.x += t.x + k; acc.y += t.y; }
            req(u, i + 16 * U);
        }
    }
    if (MODE == 0) { if (acc.x == 12345.f) Tx[j] = acc; return; }
    if (MODE >= 2) {
        __syncthreads();
        const int cc = threadIdx.x % TCOLS, rr = threadIdx.x / TCOLS, jj = tile_id * TCOLS + cc;
        const float2* ws = tile + (cc >> 2) * na * 4;
        for (int k = rr; k < na; k += 16) Tx[(unsigned)k * n + jj] = ws[k * 4 + (((cc & 3) + k) & 3)];
    }
}
template <int MODE, int U, int TCOLS> void run(const float2* W, const unsigned short* K, const float* c, float2* T, int na, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = ((n / TCOLS + 7) / 8) * 8; size_t lds = (size_t)na * TCOLS * 8;
    hipFuncSetAttribute((const void*)acc<MODE, U, TCOLS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((acc<MODE, U, TCOLS>), dim3(grid), dim3(TCOLS * 16), lds, 0, W, K, c, T, na, n);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((acc<MODE, U, TCOLS>), dim3(grid), dim3(TCOLS * 16), lds, 0, W, K, c, T, na, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("MODE=%d U=%d TC=%d  %8.1f us\n", MODE, U, TCOLS, ms * 100);
}
int main() {
    const int na = 300, n = 160000;
    float2 *W, *T; unsigned short* K; float* c;
    hipMalloc(&W, (size_t)na * n * 8); hipMalloc(&T, (size_t)na * n * 8); hipMalloc(&K, (size_t)na * n * 2); hipMalloc(&c, na * 4);
    hipMemset(W, 0, (size_t)na * n * 8); hipMemset(c, 0, na * 4);
    unsigned short* hk = (unsigned short*)malloc((size_t)na * n * 2);
    srand(1); for (size_t q = 0; q < (size_t)na * n; ++q) { int i = q / n; int k = i + (rand() % 33) - 16; hk[q] = k < 0 ? 0 : (k >= na ? na - 1 : k); }
    hipMemcpy(K, hk, (size_t)na * n * 2, hipMemcpyHostToDevice);
    run<0, 8, 16>(W, K, c, T, na, n);
    run<2, 8, 16>(W, K, c, T, na, n);
    run<3, 8, 16>(W, K, c, T, na, n);
    run<0, 8, 32>(W, K, c, T, na, n);
    run<2, 8, 32>(W, K, c, T, na, n);
    run<3, 8, 32>(W, K, c, T, na, n);
    run<0, 8, 64>(W, K, c, T, na, n);
    run<2, 8, 64>(W, K, c, T, na, n);
    run<3, 8, 64>(W, K, c, T, na, n);
    return 0;
}
