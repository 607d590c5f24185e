// ssq_common.h -- internal helpers shared by the translation units of libssq_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "../../include/ssq_hip.h"

// make the compiler treat a value as lane-dependent (keeps a load of a lane-independent
// address on the vector memory path)
#ifndef SSQ_OPAQUE_V
#define SSQ_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif
// ... and as a wave-uniform value the compiler may not fold or hoist (it stays in a scalar register)
// occupancy the compiler must plan a kernel's registers for (wavefronts per SIMD: min, max)
#ifndef SSQ_WAVES_PER_EU
#define SSQ_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
#ifndef SSQ_OPAQUE_S
#define SSQ_OPAQUE_S(x) asm volatile("" : "+s"(x))
// nothing moves across this point in the instruction scheduler (bounds how far ahead an unrolled loop's loads are hoisted)
#define SSQ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#ifndef SSQ_LDS_ADD_F64
// LDS float64 add without a return value at byte offset `off` of the workgroup's LDS
#define SSQ_LDS_ADD_F64(base, off, val) asm volatile("ds_add_f64 %0, %1" :: "v"((unsigned)(size_t)(base) + (unsigned)(off)), "v"(val) : "memory")
// ... at an LDS byte address worked out by the caller (SSQ_LDS_ADDR of the array, once, + its offsets), plus a
// constant: the pointer-to-address conversion above costs a null test and two adds per use
#define SSQ_LDS_ADDR(ptr) ((unsigned)(size_t)(ptr))
#define SSQ_LDS_ADD_F64_AT(addr, o, val) asm volatile("ds_add_f64 %0, %1 offset:%2" :: "v"(addr), "v"(val), "n"(o) : "memory")
// every LDS operation of this wavefront done, then the workgroup's barrier -- without the wait for
// vector memory that __syncthreads() implies (the loads in flight belong to the next tile)
#define SSQ_WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// a table the kernel never writes, read with a wavefront-uniform index: the constant address space
// makes the compiler fetch it through the scalar cache (s_load) instead of the vector memory path
#define SSQ_CONST_PTR(T, p) reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p))
#endif

// Packed float32 multiply-add with one half of a register pair broadcast to both lanes of the
// packed operation (v_pk_fma_f32 op_sel): acc.xy += w.xy * s.x  /  acc.xy += w.xy * s.y, and the
// same for the multiply. hipcc materialises the broadcast with two v_mov per operand instead
// (34 of 131 instructions of the tile kernel's inner row), hence the explicit forms; and
// ds_bpermute_b32 with its immediate offset, which hipcc re-adds in a VGPR per use.
#ifndef SSQ_PK_DEFINED
typedef float ssq_f2 __attribute__((ext_vector_type(2)));
#define SSQ_PK_MUL_LO(d, w, s) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(w), "v"(s))
#define SSQ_PK_MUL_HI(d, w, s) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(w), "v"(s))
#define SSQ_PK_FMA_LO(acc, w, s) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(s))
#define SSQ_PK_FMA_HI(acc, w, s) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(s))
#define SSQ_BPERMUTE_OFF(d, addr, v, off) asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(d) : "v"(addr), "v"(v), "n"(off))
// bitfield insert: d = (m & a) | (~m & b) in one instruction (the compiler spells the select out as four)
#define SSQ_BFI(d, m, a, b) asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b))
// complex product d = a b of two (re, im) register pairs in two packed instructions:
//   t = (a.x b.x, a.x b.y);  d = (-a.y b.y + t.x, a.y b.x + t.y)
// (hipcc spells a complex product out as two multiplies and two multiply-adds, or as packed
// operations fed by v_mov shuffles; -DSSQ_NO_CMUL_PK restores that form for A/B builds).
// The s_nop is the wait state gfx950 requires between a transcendental instruction (v_sin_f32,
// v_cos_f32: twiddles) and a vector instruction that reads its result: the compiler inserts it for
// its own instructions and cannot see into this block -- without it the product read stale
// registers whenever the twiddle came straight out of v_sin / v_cos (caught by the GPU suite).
#define SSQ_CMUL_PK(d, a, b) do { ssq_f2 t_;                                                               \
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t_) : "v"(a), "v"(b));                  \
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                       \
        : "=v"(d) : "v"(a), "v"(b), "v"(t_)); } while (0)
// eight taps in one block: A = sum_t w_t.x s_t, D = sum_t w_t.y s_t (s_t = a complex sample, w_t = a pair of real
// weights), each component the same chain of multiply-adds as SSQ_PK_MUL/FMA_LO/HI build. One asm statement
// because the compiler pads every pair of dependent packed instructions that sit in separate statements with an
// s_nop (it cannot see that the instruction between them is one); here each accumulator's instructions alternate
// with the other's, which is the one wait state the packed forwarding needs.
#define SSQ_TAPS8(A, D, w, s0, s1, s2, s3, s4, s5, s6, s7)                                                    \
    asm volatile("v_pk_mul_f32 %0, %2, %10 op_sel_hi:[0,1]\n\t"                                               \
                 "v_pk_mul_f32 %1, %2, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"                                  \
                 "v_pk_fma_f32 %0, %3, %11, %0 op_sel_hi:[0,1,1]\n\t"                                         \
                 "v_pk_fma_f32 %1, %3, %11, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                          \
                 "v_pk_fma_f32 %0, %4, %12, %0 op_sel_hi:[0,1,1]\n\t"                                         \
                 "v_pk_fma_f32 %1, %4, %12, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                          \
                 "v_pk_fma_f32 %0, %5, %13, %0 op_sel_hi:[0,1,1]\n\t"                                         \
                 "v_pk_fma_f32 %1, %5, %13, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                          \
                 "v_pk_fma_f32 %0, %6, %14, %0 op_sel_hi:[0,1,1]\n\t"                                         \
                 "v_pk_fma_f32 %1, %6, %14, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                          \
                 "v_pk_fma_f32 %0, %7, %15, %0 op_sel_hi:[0,1,1]\n\t"                                         \
                 "v_pk_fma_f32 %1, %7, %15, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                          \
                 "v_pk_fma_f32 %0, %8, %16, %0 op_sel_hi:[0,1,1]\n\t"                                         \
                 "v_pk_fma_f32 %1, %8, %16, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                          \
                 "v_pk_fma_f32 %0, %9, %17, %0 op_sel_hi:[0,1,1]\n\t"                                         \
                 "v_pk_fma_f32 %1, %9, %17, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]"                                \
                 : "=&v"(A), "=&v"(D)                                                                         \
                 : "v"((w)[0]), "v"((w)[1]), "v"((w)[2]), "v"((w)[3]), "v"((w)[4]), "v"((w)[5]), "v"((w)[6]),  \
                   "v"((w)[7]), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7))
// The same for TWO points that share their samples (neighbouring columns of one row, tile3_kernel): weights wa / wb,
// four accumulators taken in turn -- every instruction's operands are three instructions old.
#define SSQ_TAPS_T_(t, ta, tb, ts)                                                                            \
                 "v_pk_fma_f32 %0, %" #ta ", %" #ts ", %0 op_sel_hi:[0,1,1]\n\t"                               \
                 "v_pk_fma_f32 %1, %" #ta ", %" #ts ", %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                \
                 "v_pk_fma_f32 %2, %" #tb ", %" #ts ", %2 op_sel_hi:[0,1,1]\n\t"                               \
                 "v_pk_fma_f32 %3, %" #tb ", %" #ts ", %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define SSQ_TAPS8X2(A0, D0, A1, D1, wa, wb, s0, s1, s2, s3, s4, s5, s6, s7)                                    \
    asm volatile("v_pk_mul_f32 %0, %4, %20 op_sel_hi:[0,1]\n\t"                                               \
                 "v_pk_mul_f32 %1, %4, %20 op_sel:[1,0] op_sel_hi:[1,1]\n\t"                                  \
                 "v_pk_mul_f32 %2, %12, %20 op_sel_hi:[0,1]\n\t"                                              \
                 "v_pk_mul_f32 %3, %12, %20 op_sel:[1,0] op_sel_hi:[1,1]\n\t"                                 \
                 SSQ_TAPS_T_(1, 5, 13, 21) SSQ_TAPS_T_(2, 6, 14, 22) SSQ_TAPS_T_(3, 7, 15, 23)                 \
                 SSQ_TAPS_T_(4, 8, 16, 24) SSQ_TAPS_T_(5, 9, 17, 25) SSQ_TAPS_T_(6, 10, 18, 26)                \
                 SSQ_TAPS_T_(7, 11, 19, 27)                                                                   \
                 : "=&v"(A0), "=&v"(D0), "=&v"(A1), "=&v"(D1)                                                 \
                 : "v"((wa)[0]), "v"((wa)[1]), "v"((wa)[2]), "v"((wa)[3]), "v"((wa)[4]), "v"((wa)[5]),        \
                   "v"((wa)[6]), "v"((wa)[7]), "v"((wb)[0]), "v"((wb)[1]), "v"((wb)[2]), "v"((wb)[3]),        \
                   "v"((wb)[4]), "v"((wb)[5]), "v"((wb)[6]), "v"((wb)[7]),                                    \
                   "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7))
// flip a wave-uniform 0/1 and take the wavefront's issue priority from it (2 or 0): the compiler's form of
// the same costs two more scalar instructions per use
#define SSQ_PRIO_TOGGLE(p) asm volatile("s_xor_b32 %0, %0, 1\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\t"  \
                                        "s_setprio 2\n\ts_branch 2f\n1:\n\ts_setprio 0\n2:" : "+s"(p) :: "scc")
#define SSQ_LDS_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SSQ_LDS_WAITN(n) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(n) : "memory")
#endif

namespace ssq {

void set_error(const char* fmt, ...);

#define SSQ_CHECK_HIP(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            ssq::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                           __FILE__, __LINE__);                                    \
            return -2;                                                             \
        }                                                                          \
    } while (0)

#define SSQ_REQUIRE(cond, ...)                                                     \
    do {                                                                           \
        if (!(cond)) {                                                             \
            ssq::set_error(__VA_ARGS__);                                           \
            return -1;                                                             \
        }                                                                          \
    } while (0)

#define SSQ_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t _e = hipGetLastError();                                         \
        if (_e != hipSuccess) {                                                    \
            ssq::set_error("kernel launch failed: %s (%s:%d)",                     \
                           hipGetErrorString(_e), __FILE__, __LINE__);             \
            return -3;                                                             \
        }                                                                          \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// A cached plan owns device workspaces that every execute reuses. Executes are asynchronous
// and torch's side streams do not synchronise with each other, so two executes of one plan on
// different streams (or from two host threads) would overwrite each other's workspace. `enter`
// makes the new stream wait for the plan's previous execute (one event, recorded by `leave`);
// the mutex serialises the host side for the duration of the enqueue.
struct PlanOrder {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool used = false;
    int enter(hipStream_t s) {
        mu.lock();
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; return 0; }
        if (used && last != s) (void)hipStreamWaitEvent(s, ev, 0);
        return 0;
    }
    void leave(hipStream_t s) {
        if (ev) { (void)hipEventRecord(ev, s); used = true; last = s; }
        mu.unlock();
    }
    void destroy() { if (ev) (void)hipEventDestroy(ev); ev = nullptr; }
};
// per-row reassignment weights on the device: a new buffer for every new content (an execute
// already enqueued keeps reading the one it was given); all freed with the plan
struct WeightVersions {
    std::vector<void*> bufs;
    int upload(void** current, const void* host, size_t bytes) {
        void* p = nullptr;
        if (bufs.size() >= 32) {                 // bounded: recycle after draining the device
            (void)hipDeviceSynchronize();
            for (size_t i = 0; i + 1 < bufs.size(); ++i) (void)hipFree(bufs[i]);
            void* keep = bufs.back(); bufs.clear(); bufs.push_back(keep);
        }
        SSQ_CHECK_HIP(hipMalloc(&p, bytes ? bytes : 1));
        SSQ_CHECK_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
        bufs.push_back(p);
        *current = p;
        return 0;
    }
    void destroy() { for (void* p : bufs) (void)hipFree(p); bufs.clear(); }
};

template <typename T> struct cplx { T re, im; };

// reassignment parameters as the kernels consume them
struct SsqParams {
    double p[5];      // see SSQ_GRID_* in ssq_hip.h
    double gamma;
    int    grid;
    int    flipud;
    int    cst_f64;
    int    cst_uniform;   // all per-row weights equal (scalar `const`): kernels read one value
    // float32 screening of the bin map (ssq_point_math.inl: bin_screen_f32): the bin
    // is first estimated in float32; only points whose estimate lies within `guard`
    // bins of a rounding boundary are re-evaluated with the exact double sequence.
    float  pf[5];     // p[] rounded to float (pf[1], pf[2], pf[3] hold reciprocals)
    float  guard;     // >= 0.25 disables screening
};

// fill the float32 screening constants from p[]/grid
static inline void finalize_params(SsqParams& sp) {
    // abs. error budget of the float32 chain (log2 w - vlmin): v_log_f32 is good to 1 ulp
    // (<= 1e-6 for |log2 w| < 16), float(vlmin) and the subtraction round by <= 5e-7
    // each -> 2e-6; budgeted 2.5x. Relative terms are added per point.
    const double slack = 5e-6;
    double g;
    if (sp.grid == SSQ_GRID_LIN) {
        sp.pf[0] = (float)sp.p[0]; sp.pf[1] = (float)(1.0 / sp.p[1]);
        g = 1e-6;                         // relative part is added per point
    } else if (sp.grid == SSQ_GRID_LOG) {
        sp.pf[0] = (float)sp.p[0]; sp.pf[1] = (float)(1.0 / sp.p[1]);
        g = slack / sp.p[1];
    } else {
        sp.pf[0] = (float)sp.p[0]; sp.pf[1] = (float)sp.p[1];
        sp.pf[2] = (float)(1.0 / sp.p[2]); sp.pf[3] = (float)(1.0 / sp.p[3]);
        sp.pf[4] = (float)sp.p[4];
        double dmin = sp.p[2] < sp.p[3] ? sp.p[2] : sp.p[3];
        g = slack / dmin;
    }
    if (!(g < 0.25)) g = 1.0;
    sp.guard = (float)g;
}

// ---- launchers implemented in ssq_kernels.hip, used by the plans --------------
// SSQ_TILE_ORDER=ordered: reassignment sums in the CPU path's order (bit-exact kernels)
bool reassign_ordered();
// bin source for the accumulate kernel
enum BinSrc { BIN_FROM_DWX = 0, BIN_FROM_W = 1, BIN_FROM_KIDX = 2 };

// Tx <- accumulate(Wx, src) ; src is dWx (complex), w (real) or kidx (uint16)
int launch_accumulate(int dtype, int binsrc, const void* Wx, const void* src,
                      const void* Sfs, void* Tx, const void* cst,
                      const SsqParams& sp, int64_t batch, int64_t na, int64_t n,
                      int32_t* kmap, hipStream_t stream);

}  // namespace ssq
