// write_pattern_probe.hip -- what HBM write rate does the tile kernel's store pattern get?
// (round 4: with all arithmetic removed the tile kernel is no faster, without its Wx stores it is 50 us
// faster: it behaves as if bound by ~3.3 TB/s of writes.) Persistent workgroups (one per CU, 12
// wavefronts) walk column tiles of a (rows x N) complex64 array as the tile kernel does: for every row
// of a tile one chunk of CH bytes, rows 1.28 MB apart. Variants: chunk width, which workgroups hold
// neighbouring tiles (all XCDs interleaved, or 32 consecutive tiles per XCD), plain / nontemporal
// stores; plus a plain streaming write of the same volume as the reference point.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/write_pattern_probe.hip -o tools/probes/write_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int NW = 12;
constexpr long N = 160000, NA = 300;

template <int LPR, bool NT>   // LPR = lanes per row chunk: 64 lanes x 8 B = 512 B; chunk = CH bytes = (CH / 8) lanes
__global__ __launch_bounds__(64 * NW) void tile_writes(float2* out, int nsig, int cols, int xcd_blocked) {
    // a wavefront writes `64 / cols` ... keep it simple: cols = columns per tile (32, 64, 128, 256);
    // a wavefront instruction covers 64 consecutive columns of one row, or two rows of 32 columns
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ntx = (int)((N + cols - 1) / cols), ntot = ntx * nsig;
    const int nb = gridDim.x, b = blockIdx.x;
    const int per = (ntot + nb - 1) / nb;
    for (int j = 0; j < per; ++j) {
        int t;
        if (xcd_blocked) {
            const int x = b & 7, s = b >> 3, sl = nb >> 3;      // XCD x, slot s of sl
            t = (j * 8 + x) * sl + s;
        } else t = j * nb + b;
        if (t >= ntot) continue;
        const int sg = t / ntx, tx = t - sg * ntx;
        float2* base = out + (long)sg * NA * N + (long)tx * cols;
        const float2 v = make_float2((float)t, (float)lane);
        if (cols >= 64) {
            const int parts = cols / 64;
            for (int k = wv; k < NA * parts; k += NW) {
                const int row = k / parts, part = k - row * parts;
                const long col = (long)tx * cols + part * 64 + lane;
                if (col < N) {
                    float2* p = base + (long)row * N + part * 64 + lane;
                    if (NT) { f2v q = {v.x, v.y}; __builtin_nontemporal_store(q, reinterpret_cast<f2v*>(p)); } else *p = v;
                }
            }
        } else {   // 32 columns: two rows per instruction
            for (int k = wv; k < NA / 2; k += NW) {
                const int row = 2 * k + (lane >> 5);
                const long col = (long)tx * cols + (lane & 31);
                if (col < N) {
                    float2* p = base + (long)row * N + (lane & 31);
                    if (NT) { f2v q = {v.x, v.y}; __builtin_nontemporal_store(q, reinterpret_cast<f2v*>(p)); } else *p = v;
                }
            }
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void stream_writes(float4* out, long n16) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        const float4 v = make_float4((float)i, 1.f, 2.f, 3.f);
        if (NT) { f4v q = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(q, reinterpret_cast<f4v*>(out + i)); } else out[i] = v;
    }
}
__global__ __launch_bounds__(256) void stream_reads(const float4* in, long n16, float* sink) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        const float4 v = in[i]; acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) *sink = acc;
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount, nsig = 16;
    const size_t bytes = (size_t)nsig * NA * N * 8;
    float2* d; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
    float* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-72s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
    };
    printf("%s, %d CUs; %zu MB per pass\n", pr.name, ncu, bytes >> 20);
    timeit("streaming write, 16 B per lane, plain", [&] { hipLaunchKernelGGL(stream_writes<false>, dim3(ncu * 8), dim3(256), 0, 0, (float4*)d, (long)(bytes / 16)); });
    timeit("streaming write, 16 B per lane, nontemporal", [&] { hipLaunchKernelGGL(stream_writes<true>, dim3(ncu * 8), dim3(256), 0, 0, (float4*)d, (long)(bytes / 16)); });
    timeit("streaming read, 16 B per lane", [&] { hipLaunchKernelGGL(stream_reads, dim3(ncu * 8), dim3(256), 0, 0, (const float4*)d, (long)(bytes / 16), sink); });
    const int colsv[] = {32, 64, 128, 256, 512};
    for (int ci = 0; ci < 5; ++ci)
        for (int xb = 0; xb < 2; ++xb)
            for (int nt = 0; nt < 2; ++nt) {
                char name[128];
                snprintf(name, sizeof name, "tile walk, %4d-byte chunks, %s, %s", colsv[ci] * 8,
                         xb ? "32 neighbouring tiles per XCD" : "neighbouring tiles on different XCDs", nt ? "nontemporal" : "plain");
                const int cols = colsv[ci];
                if (nt) timeit(name, [&] { hipLaunchKernelGGL((tile_writes<0, true>), dim3(ncu), dim3(64 * NW), 0, 0, d, nsig, cols, xb); });
                else timeit(name, [&] { hipLaunchKernelGGL((tile_writes<0, false>), dim3(ncu), dim3(64 * NW), 0, 0, d, nsig, cols, xb); });
            }
    return 0;
}
