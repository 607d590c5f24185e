// lds_atomic_probe.hip -- how fast does one CU's LDS perform read-modify-write operations on a tile
// shared by 12 wavefronts? (round 4: the unordered reassignment with ds_add_f32 measured 2.5x slower
// than the ticketed one; this probe separates the instruction's own rate from everything else.)
// One 768-thread workgroup per CU, 154 KB of LDS, every wavefront issues ITER operations of one kind
// on cells (bin * 64 + lane) -- lane = column, as in the tile kernel -- with `bin` either random per
// lane, the same for the wavefront, or the same for all wavefronts.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_atomic_probe.hip -o tools/probes/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NW = 12, NA = 150, ITER = 4096;

enum Op { ADD_F32 = 0, ADD_U32, ADD_U64, ADD_RTN_U32, PLAIN_RMW, ADD_F64, CAS_B64, ADD_U64_X2, WRITE_B64, READ_B64, PK_ADD_BF16, NOPS };
static const char* names[] = {"ds_add_f32 x2 (8 B cell)", "ds_add_u32 x2 (8 B cell)", "ds_add_u64 x1 (8 B cell)",
                              "ds_add_rtn_u32 x2", "ds_read_b64 + 2 add + ds_write_b64 (racy)", "ds_add_f64 x1",
                              "ds_cmpst_rtn_b64 x1", "ds_add_u64 x2 (16 B cell)", "ds_write_b64 x1", "ds_read_b64 x1",
                              "ds_pk_add_bf16 x2"};

template <int OP, int PAT>
__global__ __launch_bounds__(64 * NW) void probe(unsigned long long* out, int cell_bytes) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (NA + 1) * 64 * cell_bytes / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = 0;
    __syncthreads();
    unsigned st = 12345u + 747796405u * (PAT == 0 ? threadIdx.x : PAT == 1 ? wv : 0) + blockIdx.x;
    const unsigned base = (unsigned)(size_t)lds;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
        st = st * 1664525u + 1013904223u;
        const unsigned bin = (st >> 8) % NA;
        const unsigned addr = base + (bin * 64 + lane) * cell_bytes;
        const float v = __uint_as_float(0x3f800000u | (st & 0xffff));
        const unsigned vi = st & 0xff;
        const unsigned long long vl = st;
        if (OP == ADD_F32) {
            asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %2 offset:4" :: "v"(addr), "v"(v), "v"(v) : "memory");
        } else if (OP == ADD_U32) {
            asm volatile("ds_add_u32 %0, %1\n\tds_add_u32 %0, %2 offset:4" :: "v"(addr), "v"(vi), "v"(vi) : "memory");
        } else if (OP == ADD_U64) {
            asm volatile("ds_add_u64 %0, %1" :: "v"(addr), "v"(vl) : "memory");
        } else if (OP == ADD_RTN_U32) {
            unsigned r0, r1;
            asm volatile("ds_add_rtn_u32 %0, %2, %3\n\tds_add_rtn_u32 %1, %2, %3 offset:4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1) : "v"(addr), "v"(vi) : "memory");
            acc += (float)(r0 + r1);
        } else if (OP == PLAIN_RMW) {
            float2 t;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr) : "memory");
            t.x += v; t.y += v;
            asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(t) : "memory");
        } else if (OP == ADD_F64) {
            const double d = (double)v;
            asm volatile("ds_add_f64 %0, %1" :: "v"(addr), "v"(d) : "memory");
        } else if (OP == CAS_B64) {
            unsigned long long r, cmp = 0ull;
            asm volatile("ds_cmpst_rtn_b64 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(cmp), "v"(vl) : "memory");
            acc += (float)(unsigned)r;
        } else if (OP == ADD_U64_X2) {
            asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %1 offset:8" :: "v"(addr), "v"(vl) : "memory");
        } else if (OP == WRITE_B64) {
            asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(vl) : "memory");
        } else if (OP == READ_B64) {
            unsigned long long r;
            asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr) : "memory");
            if ((it & 15) == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += (float)(unsigned)r; }
        } else if (OP == PK_ADD_BF16) {
            asm volatile("ds_pk_add_bf16 %0, %1\n\tds_pk_add_bf16 %0, %1 offset:4" :: "v"(addr), "v"(vi) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; }
    if (acc == 1.2345f) out[1] = 1;
}

template <int OP, int PAT>
static void run(unsigned long long* d_out, int cell_bytes, int ncu) {
    auto k = probe<OP, PAT>;
    const size_t ldsb = (size_t)(NA + 1) * 64 * cell_bytes;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(ncu), dim3(64 * NW), ldsb, 0, d_out, cell_bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(ncu), dim3(64 * NW), ldsb, 0, d_out, cell_bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    const int nops = (OP == ADD_U64 || OP == ADD_F64 || OP == CAS_B64 || OP == WRITE_B64 || OP == READ_B64) ? 1 : (OP == PLAIN_RMW ? 2 : 2);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)NW * ITER * nops);
    printf("%-44s pat %d  %8.3f ms  %7.1f cycles (2.4 GHz) per wave-instruction per CU   [counter %llu]\n",
           names[OP], PAT, ms, cyc, h[0]);
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    printf("%s, %d CUs; %d wavefronts x %d iterations per CU; pat 0 = bin random per lane, 1 = per wavefront, 2 = one for all\n",
           pr.name, ncu, NW, ITER);
#define ALLPAT(OP, CB) run<OP, 0>(d, CB, ncu); run<OP, 1>(d, CB, ncu); run<OP, 2>(d, CB, ncu);
    ALLPAT(ADD_F32, 8) ALLPAT(ADD_U32, 8) ALLPAT(ADD_U64, 8) ALLPAT(ADD_RTN_U32, 8) ALLPAT(PLAIN_RMW, 8)
    ALLPAT(ADD_F64, 8) ALLPAT(CAS_B64, 8) ALLPAT(WRITE_B64, 8) ALLPAT(READ_B64, 8) ALLPAT(PK_ADD_BF16, 8)
    run<ADD_U64_X2, 0>(d, 16, ncu); run<ADD_U64_X2, 1>(d, 16, ncu);
    return 0;
}
