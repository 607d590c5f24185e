// store_ack_probe.hip -- how long until a wavefront's store is acknowledged (vmcnt drops), and how long a
// load issued behind it takes -- on an idle chip and while every other CU streams writes?
// (round 4: vector memory operations retire in order on gfx9; both tile kernels lose 40-60 us when
// their Wx stores are removed.)  Workgroup 0 measures, the others make traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_ack_probe.hip -o tools/probes/store_ack_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));

// MODE: 0 plain store, 1 nontemporal store, 2 sc0 sc1 store
template <int MODE>
__global__ __launch_bounds__(768) void probe(char* buf, long region, int iters, int traffic, unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* base = buf + ((long)blockIdx.x * 12 + wv) * region;
    f2v q = {(float)lane, 1.f};
    if (blockIdx.x != 0) {                       // background: stream stores (and a few loads) to private regions
        if (!traffic) return;
        float acc = 0.f;
        for (int it = 0; it < iters * 64; ++it) {
            char* p = base + ((long)it * 512) % region;
            *reinterpret_cast<f2v*>(p + lane * 8) = q;
            if ((it & 7) == 7) acc += reinterpret_cast<const f2v*>(p + lane * 8 - 2048 * (it >= 4))->x;
        }
        if (acc == 1.2345f) *sink = acc;
        return;
    }
    unsigned long long t_store = 0, t_load = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        char* p = base + (long)it * 1280000 % region / 512 * 512;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == 0) *reinterpret_cast<f2v*>(p + lane * 8) = q;
        else if (MODE == 1) __builtin_nontemporal_store(q, reinterpret_cast<f2v*>(p + lane * 8));
        else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p + lane * 8), "v"(q) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        const f2v r = *reinterpret_cast<const f2v*>(p + 65536 + lane * 8);
        acc += r.x;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        t_store += t1 - t0; t_load += t2 - t1;
    }
    if (threadIdx.x == 0) { out[0] = t_store / iters; out[1] = t_load / iters; }
    if (acc == 1.2345f) *sink = acc;
}
template <int MODE> static void run(char* d, unsigned long long* o, float* sink, const char* name, int ncu) {
    for (int traffic = 0; traffic < 2; ++traffic) {
        hipLaunchKernelGGL(probe<MODE>, dim3(ncu), dim3(768), 0, 0, d, 8L << 20, 2048, traffic, o, sink);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
        printf("%-22s %-28s store -> vmcnt(0): %6llu clk   load after it: %6llu clk\n", name,
               traffic ? "all other CUs streaming" : "idle chip", h[0], h[1]);
    }
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    char* d; hipMalloc(&d, (size_t)ncu * 12 * (8L << 20) + (1 << 20));
    unsigned long long* o; hipMalloc(&o, 64); float* sink; hipMalloc(&sink, 4);
    printf("%s (counter: s_memtime ticks)\n", pr.name);
    run<0>(d, o, sink, "plain store", ncu);
    run<1>(d, o, sink, "nontemporal store", ncu);
    run<2>(d, o, sink, "sc0 sc1 store", ncu);
    return 0;
}
