// store_rate_probe.hip -- how many bytes per clock can ONE CU store, by store width?
// (round 4: both tile kernels lose 40-50 us when their Wx stores are removed although the chip is
// far from its HBM write rate, and their time per tile does not depend on how many CUs run.)
// Persistent workgroups of 12 wavefronts; every wavefront streams stores to its own region.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_rate_probe.hip -o tools/probes/store_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// MODE 0: 4 B per lane; 1: 8 B per lane (512-B runs); 2: 16 B per lane (1-KB runs); 3: 16 B per lane, odd lanes off;
// 4: 8 B per lane as two 256-B runs 1.28 MB apart (tile2_kernel's Wx store); 5: 8 B per lane, rows 1.28 MB apart per instruction
template <int MODE>
__global__ __launch_bounds__(768) void stores(char* out, long bytes_per_wave, int iters) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* base = out + ((long)blockIdx.x * 12 + wv) * bytes_per_wave;
    const float v = (float)lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { *reinterpret_cast<float*>(base + (long)it * 256 + lane * 4) = v; }
        else if (MODE == 1) { f2v q = {v, v}; *reinterpret_cast<f2v*>(base + (long)it * 512 + lane * 8) = q; }
        else if (MODE == 2) { f4v q = {v, v, v, v}; *reinterpret_cast<f4v*>(base + (long)it * 1024 + lane * 16) = q; }
        else if (MODE == 3) { f4v q = {v, v, v, v}; if (!(lane & 1)) *reinterpret_cast<f4v*>(base + (long)it * 1024 + lane * 16) = q; }
        else if (MODE == 4) { f2v q = {v, v}; *reinterpret_cast<f2v*>(base + (long)(it & 63) * 256 + (long)(it >> 6) * 32768 + (lane >> 5) * 16384 + (lane & 31) * 8) = q; }
        else { f2v q = {v, v}; *reinterpret_cast<f2v*>(base + ((long)it * 1280000 % bytes_per_wave / 512 * 512) + lane * 8) = q; }
    }
}

template <int MODE> static void run(char* d, int ncu, const char* name, int width) {
    const int iters = 4096;
    const long bpw = (long)iters * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int g : {ncu, ncu / 4, ncu / 16}) {
        hipLaunchKernelGGL(stores<MODE>, dim3(g), dim3(768), 0, 0, d, bpw, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(stores<MODE>, dim3(g), dim3(768), 0, 0, d, bpw, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)g * 12 * iters * width;
        printf("%-58s %3d CUs  %7.3f ms  %6.2f TB/s  %6.1f B/clk/CU  %6.1f clk per wavefront store\n", name, g, ms,
               bytes / (ms * 1e-3) / 1e12, bytes / g / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (12.0 * iters));
    }
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    char* d; hipMalloc(&d, (size_t)ncu * 12 * 4096 * 1024 + (1 << 20));
    printf("%s\n", pr.name);
    run<0>(d, ncu, "4 B per lane (256-B runs)", 256);
    run<1>(d, ncu, "8 B per lane (512-B runs)", 512);
    run<2>(d, ncu, "16 B per lane (1-KB runs)", 1024);
    run<3>(d, ncu, "16 B per lane, every other lane (512 B per instruction)", 512);
    run<4>(d, ncu, "8 B per lane, two 256-B runs per instruction", 512);
    run<5>(d, ncu, "8 B per lane, 512-B runs 1.28 MB apart", 512);
    return 0;
}
