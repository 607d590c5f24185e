// stride_store_probe.hip -- what does ONE CU pay per wavefront store (or load) that goes to "another row"?
// (round 4: store_rate_probe: 16.5 clk per 512-byte wavefront store when a wavefront's stores are
// contiguous, 47 clk when they are 1.28 MB apart -- on 16 CUs as on 64; the tile kernels do ~650 such
// row visits per 64-column tile.) 16 CUs x 12 wavefronts; every wavefront cycles over `nrows` rows
// `stride` bytes apart, moving 512 bytes to the right after every full cycle.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stride_store_probe.hip -o tools/probes/stride_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// MODE 0: store 8 B per lane; 1: nontemporal store 8 B; 2: store 16 B per lane (1 KB per visit);
// 3: two 8-B stores side by side per visit (1 KB per visit); 4: load 8 B per lane
template <int MODE>
__global__ __launch_bounds__(768) void visits(char* buf, long stride, int nrows, int iters, long wave_bytes, float* sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* base = buf + ((long)blockIdx.x * 12 + wv) * wave_bytes;
    const float v = (float)lane;
    float acc = 0.f;
    int r = 0; long col = 0;
    for (int it = 0; it < iters; ++it) {
        char* p = base + (long)r * stride + col;
        if (MODE == 0) { f2v q = {v, v}; *reinterpret_cast<f2v*>(p + lane * 8) = q; }
        else if (MODE == 1) { f2v q = {v, v}; __builtin_nontemporal_store(q, reinterpret_cast<f2v*>(p + lane * 8)); }
        else if (MODE == 2) { f4v q = {v, v, v, v}; *reinterpret_cast<f4v*>(p + lane * 16) = q; }
        else if (MODE == 3) { f2v q = {v, v}; *reinterpret_cast<f2v*>(p + lane * 8) = q; *reinterpret_cast<f2v*>(p + 512 + lane * 8) = q; }
        else { f2v q = *reinterpret_cast<const f2v*>(p + lane * 8); acc += q.x; }
        if (++r == nrows) { r = 0; col += (MODE == 2 || MODE == 3) ? 1024 : 512; }
    }
    if (acc == 1.2345f) *sink = acc;
}

template <int MODE> static void run(char* d, float* sink, const char* name, long stride, int nrows) {
    const int iters = 8192, g = 16;
    const long wave_bytes = (long)nrows * stride + (1 << 20);     // every wavefront its own rows
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(visits<MODE>, dim3(g), dim3(768), 0, 0, d, stride, nrows, iters, wave_bytes, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(visits<MODE>, dim3(g), dim3(768), 0, 0, d, stride, nrows, iters, wave_bytes, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s stride %9ld B x %4d rows  %7.3f ms  %6.1f clk per visit per CU\n", name, stride, nrows, ms, ms * 1e-3 * 2.4e9 / (12.0 * iters));
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const size_t total = (size_t)16 * 12 * ((size_t)300 * 1280000 + (2 << 20));
    char* d; if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    float* sink; hipMalloc(&sink, 4);
    printf("%s; buffer %zu MB\n", pr.name, total >> 20);
    for (long s : {512L, 4096L, 16384L, 65536L, 262144L, 1280000L, 2097152L}) run<0>(d, sink, "store 8 B/lane", s, 32);
    for (int n : {2, 4, 8, 16, 64, 300}) run<0>(d, sink, "store 8 B/lane", 1280000L, n);
    run<1>(d, sink, "nontemporal store 8 B/lane", 1280000L, 300);
    run<2>(d, sink, "store 16 B/lane (1 KB per visit)", 1280000L, 300);
    run<3>(d, sink, "two 8-B stores side by side (1 KB per visit)", 1280000L, 300);
    run<4>(d, sink, "load 8 B/lane", 1280000L, 300);
    run<4>(d, sink, "load 8 B/lane", 512L, 32);
    run<4>(d, sink, "load 8 B/lane", 65536L, 32);
    return 0;
}
