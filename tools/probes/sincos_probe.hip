// accuracy of v_sin_f32 / v_cos_f32 (argument in revolutions) on the M-point unit circle
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(float* s, float* c, int M) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    float x = (float)p * (1.0f / (float)M);
    s[p] = __builtin_amdgcn_sinf(x);
    c[p] = __builtin_amdgcn_cosf(x);
}
int main() {
    const int M = 1 << 18;
    float *ds, *dc; hipMalloc(&ds, M * 4); hipMalloc(&dc, M * 4);
    hipLaunchKernelGGL(k, dim3(M / 256), dim3(256), 0, 0, ds, dc, M);
    std::vector<float> s(M), c(M);
    hipMemcpy(s.data(), ds, M * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, M * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0;
    for (int p = 0; p < M; ++p) {
        double a = 2 * M_PI * p / M;
        es = fmax(es, fabs(s[p] - sin(a))); ec = fmax(ec, fabs(c[p] - cos(a)));
    }
    printf("max abs err: sin %.3e cos %.3e (float eps 5.96e-8)\n", es, ec);
    return 0;
}
