// Host stand-in for <hip/hip_runtime.h>, just large enough to run ssqueezepy_amd/csrc on the
// CPU: every work-item of a workgroup is a fiber (ucontext) of one OS thread, scheduled
// wavefront by wavefront; __syncthreads and the wavefront operations (ballot, DPP moves,
// wave_barrier) are the points where a fiber yields. The workgroups of a launch are spread
// over the host's cores (block-scope and dynamic LDS arrays are thread_local).
// TEST INFRASTRUCTURE ONLY -- it checks the kernels' control flow, LDS layout and index
// arithmetic where no GPU is available; the product never includes it.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <ucontext.h>

// HIP's global-namespace device math that the kernels use unqualified
using std::isinf; using std::isnan; using std::min; using std::max;
inline long long min(long long a, long b) { return a < b ? a : b; }
inline long long max(long long a, int b) { return a > b ? a : b; }

#define SSQ_OPAQUE_V(x) ((void)(x))
#define SSQ_OPAQUE_S(x) ((void)(x))
#define SSQ_SCHED_FENCE() ((void)0)
#define SSQ_WAVES_PER_EU(lo, hi)
// stand-ins of the inline-assembly forms of csrc/ssq_common.h
inline int emu_ds_bpermute(int byte_addr, int v);
#define SSQ_PK_DEFINED 1
typedef float ssq_f2 __attribute__((ext_vector_type(2)));
#define SSQ_PK_MUL_LO(d, w, s) do { (d).x = (w).x * (s).x; (d).y = (w).y * (s).x; } while (0)
#define SSQ_PK_MUL_HI(d, w, s) do { (d).x = (w).x * (s).y; (d).y = (w).y * (s).y; } while (0)
#define SSQ_PK_FMA_LO(acc, w, s) do { (acc).x = __builtin_fmaf((w).x, (s).x, (acc).x); (acc).y = __builtin_fmaf((w).y, (s).x, (acc).y); } while (0)
#define SSQ_PK_FMA_HI(acc, w, s) do { (acc).x = __builtin_fmaf((w).x, (s).y, (acc).x); (acc).y = __builtin_fmaf((w).y, (s).y, (acc).y); } while (0)
#define SSQ_TAPS8(A, D, w, s0, s1, s2, s3, s4, s5, s6, s7) do {                                               \
    const ssq_f2 s_[8] = {s0, s1, s2, s3, s4, s5, s6, s7};                                                     \
    (A).x = (w)[0].x * s_[0].x; (A).y = (w)[0].x * s_[0].y; (D).x = (w)[0].y * s_[0].x; (D).y = (w)[0].y * s_[0].y; \
    for (int t_ = 1; t_ < 8; ++t_) {                                                                          \
        (A).x = __builtin_fmaf((w)[t_].x, s_[t_].x, (A).x); (A).y = __builtin_fmaf((w)[t_].x, s_[t_].y, (A).y); \
        (D).x = __builtin_fmaf((w)[t_].y, s_[t_].x, (D).x); (D).y = __builtin_fmaf((w)[t_].y, s_[t_].y, (D).y); \
    } } while (0)
#define SSQ_TAPS8X2(A0, D0, A1, D1, wa, wb, s0, s1, s2, s3, s4, s5, s6, s7) do {                                 \
    SSQ_TAPS8(A0, D0, wa, s0, s1, s2, s3, s4, s5, s6, s7); SSQ_TAPS8(A1, D1, wb, s0, s1, s2, s3, s4, s5, s6, s7); } while (0)
#define SSQ_BPERMUTE_OFF(d, addr, v, off) ((d) = emu_ds_bpermute((addr) + (off), (v)))
#define SSQ_CMUL_PK(d, a, b) do { const float tx_ = (a).x * (b).x, ty_ = (a).x * (b).y;                      \
    (d).x = __builtin_fmaf(-(a).y, (b).y, tx_); (d).y = __builtin_fmaf((a).y, (b).x, ty_); } while (0)
#define SSQ_BFI(d, m, a, b) ((d) = ((m) & (a)) | (~(m) & (b)))
#define SSQ_LDS_WAIT() ((void)0)
// LDS float64 add (ds_add_f64) and the LDS-only workgroup barrier of tile2_kernel
#define SSQ_LDS_ADD_F64(base, off, val) (*reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(base) + (off)) += (val))
namespace ssq { extern thread_local unsigned char lds_raw[]; }
#define SSQ_LDS_ADDR(ptr) ((unsigned)(reinterpret_cast<unsigned char*>(ptr) - ssq::lds_raw))
#define SSQ_LDS_ADD_F64_AT(addr, o, val) (*reinterpret_cast<double*>(ssq::lds_raw + (addr) + (o)) += (val))
#define SSQ_PRIO_TOGGLE(p) ((p) ^= 1)
#define SSQ_WG_BARRIER() __syncthreads()
#define SSQ_CONST_PTR(T, p) reinterpret_cast<const T*>(p)
#define SSQ_LDS_WAITN(n) ((void)0)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __align__(x) __attribute__((aligned(x)))
#ifdef EMU_STATIC_SHARED
#define __shared__ static thread_local      // block-scope LDS arrays: one copy per OS thread
#else
#define __shared__ thread_local             // `extern __shared__` dynamic LDS: emu_globals.cpp
#endif

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(32) double4 { double x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) double2 { double x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) int2 { int x, y; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline double2 make_double2(double x, double y) { return double2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }

// ------------------------------------------------------------------ runtime API ("device"
// memory is host memory, streams and events are no-ops)
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
constexpr hipError_t hipSuccess = 0;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 0;
constexpr int hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3;
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem;
                         size_t maxSharedMemoryPerMultiProcessor; };
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* c) { *c = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    strcpy(p->name, "host emulation"); strcpy(p->gcnArchName, "host");
    p->multiProcessorCount = 3; p->totalGlobalMem = 0; p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <typename P> hipError_t hipMalloc(P** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
template <typename P> hipError_t hipMallocAsync(P** p, size_t n, hipStream_t) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipFreeAsync(void* p, hipStream_t) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
constexpr int hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, int) { *e = (hipEvent_t)1; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
typedef void* hipGraph_t; typedef void* hipGraphExec_t;
constexpr int hipStreamCaptureModeRelaxed = 2;
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 1; }      // no graphs under the emulator
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 1; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 1; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
constexpr int hipStreamNonBlocking = 1;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// ------------------------------------------------------------------ fibers
namespace emu {
enum Wait { RUNNABLE = 0, AT_WAVE = 1, AT_BLOCK = 2, FINISHED = 3, AT_SPIN = 4 };
struct Fiber {
    ucontext_t ctx;
    void* stack = nullptr;
    int wait = RUNNABLE;
    unsigned tid = 0;
    unsigned gen = 0;                        // wavefront operations executed so far
};
struct Wave {                                // rotating slots: see __ballot below
    unsigned long long votes[3];
    int xchg[3][64];
};
struct Worker {                              // one per OS thread: the workgroup it is running
    ~Worker() { for (auto& f : fibers) if (f.stack) munmap(f.stack, 256 * 1024); }
    ucontext_t sched;
    std::vector<Fiber> fibers;               // grown to the largest workgroup seen
    std::vector<Wave> waves;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    dim3 blockIdx_, blockDim_, gridDim_;
};
extern thread_local Worker t_worker;
constexpr size_t STACK_BYTES = 256 * 1024;

inline void yield(int why) {
    Worker& w = t_worker;
    Fiber* f = w.cur;
    f->wait = why;
    swapcontext(&f->ctx, &w.sched);
}
inline void fiber_main() {
    Worker& w = t_worker;
    (*w.body)();
    w.cur->wait = FINISHED;
    swapcontext(&w.cur->ctx, &w.sched);      // never resumed
}
inline void resume(Worker& w, Fiber& f) {
    w.cur = &f;
    f.wait = RUNNABLE;
    swapcontext(&w.sched, &f.ctx);
}
// run one workgroup of `nt` work-items to completion on the calling OS thread
inline void run_block(const std::function<void()>& body, dim3 bidx, dim3 bdim, dim3 gdim) {
    Worker& w = t_worker;
    const unsigned nt = bdim.x, nw = (nt + 63) / 64;
    if (w.fibers.size() < nt) {
        const size_t old = w.fibers.size();
        w.fibers.resize(nt);
        for (size_t i = old; i < nt; ++i) {
            w.fibers[i].stack = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE,
                                     MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (w.fibers[i].stack == MAP_FAILED) { perror("emu: mmap"); abort(); }
        }
    }
    if (w.waves.size() < nw) w.waves.resize(nw);
    for (unsigned i = 0; i < nw; ++i) memset(&w.waves[i], 0, sizeof(Wave));
    w.body = &body; w.blockIdx_ = bidx; w.blockDim_ = bdim; w.gridDim_ = gdim;
    for (unsigned t = 0; t < nt; ++t) {
        Fiber& f = w.fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK_BYTES; f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_main, 0);
        f.wait = RUNNABLE; f.tid = t; f.gen = 0;
    }
    // Wavefronts of a workgroup run one after another between two __syncthreads; a kernel that
    // is free of LDS races gives the same result in any order. SSQ_EMU_ORDER=reverse|shuffle
    // changes the order (shuffle: a new pseudo-random order every phase) to expose a missing
    // barrier, which the default order could hide.
    static const int order_mode = [] {
        const char* e = getenv("SSQ_EMU_ORDER");
        return !e ? 0 : (strcmp(e, "reverse") == 0 ? 1 : (strcmp(e, "shuffle") == 0 ? 2 : 0));
    }();
    static const bool lanes_reversed = [] { const char* e = getenv("SSQ_EMU_LANES"); return e && strcmp(e, "reverse") == 0; }();
    std::vector<unsigned> order(nw);
    unsigned long long rng = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)bidx.x * 7919u + bidx.y);
    for (;;) {                               // one iteration = one __syncthreads phase
        bool any_alive = false;
        for (unsigned i = 0; i < nw; ++i) order[i] = order_mode == 1 ? nw - 1 - i : i;
        if (order_mode == 2)
            for (unsigned i = nw; i > 1; --i) {
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                std::swap(order[i - 1], order[(rng >> 33) % i]);
            }
        for (unsigned oi = 0; oi < nw; ++oi) {
            const unsigned wv = order[oi];
            const unsigned lo = wv * 64, hi = std::min(nt, lo + 64);
            for (;;) {                       // sweep the wave until it reaches a block barrier or polls
                // lanes that are not parked at a wavefront operation run to their next yield
                for (unsigned i = lo; i < hi; ++i) {
                    // SSQ_EMU_LANES=reverse: the lanes of a wavefront in descending order (code
                    // that is correct only in lockstep, without a wave_barrier, shows up)
                    const unsigned t = lanes_reversed ? hi - 1 - (i - lo) : i;
                    Fiber& f = w.fibers[t];
                    if (f.wait == RUNNABLE) resume(w, f);
                }
                unsigned n_wave = 0, n_spin = 0, n_block = 0;
                for (unsigned t = lo; t < hi; ++t) {
                    const int s = w.fibers[t].wait;
                    n_wave += s == AT_WAVE; n_spin += s == AT_SPIN; n_block += s == AT_BLOCK;
                }
                // Some lanes poll LDS for another wavefront's progress: the others must run first.
                // (Lanes of one wavefront may leave a polling loop in different sweeps -- a lane that
                // runs before the lane whose store it polls for -- so lanes already at the next
                // wavefront operation stay parked there until the pollers have caught up.)
                if (n_spin) break;
                if (!n_wave) break;
                if (n_block) {
                    fprintf(stderr, "emu: wavefront %u of workgroup (%u,%u,%u) diverged: some lanes "
                            "wait at a wavefront operation, others at __syncthreads\n", wv, bidx.x, bidx.y, bidx.z);
                    abort();
                }
                // every live lane is at the wavefront operation: complete it
                for (unsigned i = lo; i < hi; ++i) {
                    const unsigned t = lanes_reversed ? hi - 1 - (i - lo) : i;
                    Fiber& f = w.fibers[t];
                    if (f.wait == AT_WAVE) resume(w, f);
                }
            }
            for (unsigned t = lo; t < hi; ++t) any_alive |= w.fibers[t].wait == AT_BLOCK || w.fibers[t].wait == AT_SPIN;
        }
        if (!any_alive) break;
        // wavefronts polling LDS for another wavefront's progress (s_sleep) run again first; the
        // ones at __syncthreads wait until nobody is polling any more
        bool spinning = false;
        for (unsigned t = 0; t < nt; ++t) spinning |= w.fibers[t].wait == AT_SPIN;
        for (unsigned t = 0; t < nt; ++t)
            if (w.fibers[t].wait == (spinning ? AT_SPIN : AT_BLOCK)) w.fibers[t].wait = RUNNABLE;
    }
}
}  // namespace emu

#define threadIdx (dim3(emu::t_worker.cur->tid))
#define blockIdx (emu::t_worker.blockIdx_)
#define blockDim (emu::t_worker.blockDim_)
#define gridDim (emu::t_worker.gridDim_)

inline void __syncthreads() { emu::yield(emu::AT_BLOCK); }
// s_sleep inside a polling loop: let the other wavefronts of the workgroup run
inline void emu_s_sleep(int) { emu::yield(emu::AT_SPIN); }
#define __builtin_amdgcn_s_sleep emu_s_sleep

// ---- wavefront-level operations. Every live lane of a wavefront executes the same sequence
// of them (the kernels call them under wave-uniform control flow only). A lane publishes its
// contribution in slot gen % 3, yields, and reads the slot after every lane has run up to the
// same point; slot (gen + 2) % 3 -- last used two operations ago, no lane can still need it --
// is cleared for later use.
inline void emu_wave_barrier() {
    emu::Worker& w = emu::t_worker;
    const unsigned g = w.cur->gen++;
    emu::yield(emu::AT_WAVE);
    w.waves[w.cur->tid / 64].votes[(g + 2) % 3] = 0;
}
#define __builtin_amdgcn_wave_barrier emu_wave_barrier

inline unsigned long long __ballot(bool pred) {
    emu::Worker& w = emu::t_worker;
    emu::Fiber* f = w.cur;
    emu::Wave& wv = w.waves[f->tid / 64];
    const unsigned g = f->gen++, lane = f->tid % 64;
    if (pred) wv.votes[g % 3] |= 1ull << lane;
    emu::yield(emu::AT_WAVE);
    wv.votes[(g + 2) % 3] = 0;
    return wv.votes[g % 3];
}
#define __builtin_amdgcn_ballot_w64 __ballot

// v_mov_b32_dpp row_shr:N (ctrl 0x110 + N): lane L of a 16-lane row reads lane L - N of the
// same row; without a source: 0 if bound_ctrl, else `old`
inline int emu_update_dpp(int old, int v, int ctrl, int, int, bool bound_ctrl) {
    emu::Worker& w = emu::t_worker;
    emu::Fiber* f = w.cur;
    emu::Wave& wv = w.waves[f->tid / 64];
    const unsigned g = f->gen++, lane = f->tid % 64;
    const int N = ctrl - 0x110;
    const bool quad = ctrl >= 0 && ctrl <= 0xFF;              // quad_perm: lane L reads lane (L & ~3) + sel[L & 3]
    if (!quad && (N < 1 || N > 15)) { fprintf(stderr, "emu: unsupported dpp_ctrl %#x\n", ctrl); abort(); }
    wv.xchg[g % 3][lane] = v;
    emu::yield(emu::AT_WAVE);
    wv.votes[(g + 2) % 3] = 0;
    if (quad) return wv.xchg[g % 3][(lane & ~3u) + ((ctrl >> (2 * (lane & 3))) & 3)];
    const bool has = (int)(lane % 16) >= N;
    return has ? wv.xchg[g % 3][lane - N] : (bound_ctrl ? 0 : old);
}
#define __builtin_amdgcn_update_dpp emu_update_dpp

// ds_bpermute_b32: lane L reads the value lane (byte_addr / 4) % 64 contributed
inline int emu_ds_bpermute(int byte_addr, int v) {
    emu::Worker& w = emu::t_worker;
    emu::Fiber* f = w.cur;
    emu::Wave& wv = w.waves[f->tid / 64];
    const unsigned g = f->gen++, lane = f->tid % 64;
    wv.xchg[g % 3][lane] = v;
    emu::yield(emu::AT_WAVE);
    wv.votes[(g + 2) % 3] = 0;
    return wv.xchg[g % 3][((unsigned)byte_addr / 4) % 64];
}
#define __builtin_amdgcn_ds_bpermute emu_ds_bpermute
#define __builtin_amdgcn_readfirstlane(x) (x)     // used on wavefront-uniform values only
// LDS float atomic (ds_add_f32): the work-items of a workgroup are fibers of ONE OS thread
inline float atomicAdd(float* p, float v) { float old = *p; *p = old + v; return old; }
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_nontemporal_store(v, p) (*(p) = (v))      // (a cache hint; host movnt wants 16-byte alignment)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
// memory fences: the work-items of a workgroup are fibers of one OS thread
#define __builtin_amdgcn_fence(order, scope, ...) ((void)0)
// v_sin_f32 / v_cos_f32: argument in revolutions
inline float emu_sinf_rev(float x) { return (float)sin(6.283185307179586 * (double)x); }
inline float emu_cosf_rev(float x) { return (float)cos(6.283185307179586 * (double)x); }
#define __builtin_amdgcn_sinf emu_sinf_rev
#define __builtin_amdgcn_cosf emu_cosf_rev

inline float __log2f(float x) { return log2f(x); }          // v_log_f32 (1 ulp)
inline int __ffs(int x) { return __builtin_ffs(x); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline int __double2loint(double d) { long long q; memcpy(&q, &d, 8); return (int)(q & 0xffffffffll); }
inline int __double2hiint(double d) { long long q; memcpy(&q, &d, 8); return (int)(q >> 32); }
inline double __hiloint2double(int hi, int lo) {
    long long q = ((long long)hi << 32) | (unsigned int)lo; double d; memcpy(&d, &q, 8); return d;
}

// ------------------------------------------------------------------ launch
namespace ssq { extern thread_local unsigned char lds_raw[], smem[]; }   // emu_globals.cpp
namespace emu {
constexpr size_t DYN_LDS_BYTES = 160 * 1024;
// SSQ_EMU_LDSGUARD=1: the part of the dynamic LDS buffers beyond what the launch asked for is
// filled with a pattern before every workgroup and checked after it (out-of-bounds writes)
inline void lds_guard(size_t asked, bool check, dim3 b) {
    for (unsigned char* buf : {ssq::lds_raw, ssq::smem}) {
        if (!check) { memset(buf + asked, 0xA5, DYN_LDS_BYTES - asked); continue; }
        for (size_t i = asked; i < DYN_LDS_BYTES; ++i)
            if (buf[i] != 0xA5) {
                fprintf(stderr, "emu: workgroup (%u,%u,%u) wrote dynamic LDS at byte %zu, beyond the %zu "
                        "bytes of its launch\n", b.x, b.y, b.z, i, asked);
                abort();
            }
    }
}
}  // namespace emu

template <typename K, typename... Args>
void emu_launch(K kernel, dim3 grid, dim3 block, size_t dyn_lds, Args... args) {
    static const bool guard = getenv("SSQ_EMU_LDSGUARD") != nullptr;
    if (dyn_lds > emu::DYN_LDS_BYTES) { fprintf(stderr, "emu: %zu bytes of dynamic LDS requested\n", dyn_lds); abort(); }
    const std::function<void()> kbody = [=] { kernel(args...); };
    const std::function<void()>& body = kbody;
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nblocks) return;
            const unsigned bx = (unsigned)(b % grid.x), by = (unsigned)((b / grid.x) % grid.y),
                           bz = (unsigned)(b / ((size_t)grid.x * grid.y));
            if (guard) emu::lds_guard(dyn_lds, false, dim3(bx, by, bz));
            emu::run_block(body, dim3(bx, by, bz), block, grid);
            if (guard) emu::lds_guard(dyn_lds, true, dim3(bx, by, bz));
        }
    };
    static const unsigned ncpu = [] {
        const char* e = getenv("SSQ_EMU_THREADS");
        unsigned n = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        return n < 1 ? 1u : (n > 16 ? 16u : n);
    }();
    const unsigned nth = (unsigned)std::min<size_t>(ncpu, nblocks);
    if (nth <= 1) { work(); return; }
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nth; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
}
// (kernel), grid, block, dynamic LDS bytes, stream, args...
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    emu_launch(kernel, dim3(grid), dim3(block), (size_t)(lds), __VA_ARGS__)
