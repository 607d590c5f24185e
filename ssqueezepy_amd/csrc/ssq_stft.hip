// ssq_stft.hip -- STFT / synchrosqueezed-STFT plan (C ABI: ssq_stft_*).
//
// Data flow per signal (reference: ssqueezepy/_stft.py:127-147,166-170,
// _ssq_stft.py:88-122):
//   x (N) --pad_kernel--> xp (N + n_fft - 1)
//   xp --frame_window_kernel--> frames[c][r] = xp[frame c, sample r] * window[r]
//        (and the same with diff_window), frame-major so each transform is contiguous
//   frames --rocFFT R2C, batch = n_hops, output stride n_hops--> Sx[f][c], dSx[f][c]
//        (the strided store writes the (rows, n_hops) layout directly: no transpose)
//   Sx, dSx --accumulate_tile_kernel (STFT form)--> Tx
// float32 with a power-of-two n_fft in [128, 2048] takes the fused kernel instead:
// framing, both windows and both transforms in ONE launch for the whole batch -- the
// two real frames a = frame*window, b = frame*diff_window ride one complex LDS FFT as
// a + ib and are separated by Hermitian symmetry (stft_fused_kernel below).
// Compiled with -ffp-contract=off (see ssq_kernels.hip).
#include "ssq_common.h"
#include "ssq_fft.h"
#include "ssq_ldsfft.h"
#include "ssq_stft.h"
#include <vector>
#include <cmath>
#include <algorithm>
#include <rocfft/rocfft.h>

namespace ssq {

#include "ssq_point_math.inl"

#ifdef STFT_STAMPS
__device__ unsigned long long g_stft_prof[16];
#define ST_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&g_stft_prof[i], t_ - tprev); tprev = t_; } while (0)
#else
#define ST_STAMP(i) do {} while (0)
#endif
typedef float ssq_f4 __attribute__((ext_vector_type(4)));
typedef float ssq_f4a4 __attribute__((ext_vector_type(4), aligned(4)));     // 16 bytes at a sample's 4-byte boundary

template <typename T>
__global__ __launch_bounds__(256) void frame_window_kernel(
    const T* __restrict__ xp, const T* __restrict__ window, const T* __restrict__ diff_window,
    T* __restrict__ frames, T* __restrict__ dframes, int64_t n_fft, int64_t n_hops, int64_t hop,
    int64_t s20, int64_t s21, int modulated) {
    const int64_t total = n_fft * n_hops;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t % n_fft, c = t / n_fft;
        int64_t start = hop * c, s;
        if (!modulated) s = start + r;
        else if (r < s20) s = start + s21 + r;
        else s = start + (r - s20);
        T v = xp[s];
        frames[t] = v * window[r];
        if (dframes) dframes[t] = v * diff_window[r];
    }
}

// ---- fused framing + window + FFT (float32, n_fft = L a power of two) -----------------
// One item = G = 4096/L consecutive frames of one signal. The forward transform is
// taken as conj(IFFT(conj(.))) with the inverse LDS FFT of ssq_ldsfft.h: input
// a - ib, output Z' with FFT(a + ib) = conj(Z'). With A = FFT(a), B = FFT(b) Hermitian,
//   A[f] = (Z[f] + conj(Z[L-f])) / 2,   B[f] = (Z[f] - conj(Z[L-f])) / (2i),  f <= L/2.
// Round 6 (profiles/r6_ab_history.txt "r6w".."r6y"): the Tx planes take the FFT buffer's place (four workgroups per CU);
// the item's samples come through LDS in one coalesced pass; the tables an item needs (window pairs, row frequencies,
// weights) are asked for a phase ahead of their use. Persistent workgroups that fetch the next item's samples while they
// transform the current one were built and measured slower (the loop costs 40 registers: three workgroups per CU).
// What the kernel is NOT bound by, each measured ("r6y8".."r6y11"): a fifth workgroup per CU (the last row of the planes
// kept in registers, LDS exactly 32 KB: no change), its 32-byte output pieces (the same bytes as full lines: -4 %), the
// transform's LDS bank conflicts (padded columns: +2 %). The counters put the vector ALU at ~60 % and the LDS pipe at
// ~58 % busy: what is left is instruction count.

template <int L, int G, int R1, int R2, int R3, bool REASSIGN, bool CST64>
__global__ __launch_bounds__(NT) void stft_fused_kernel(StftFusedArgs A, SsqParams sp) {
    // the frames' Tx: a real and an imaginary plane of (L/2 + 1) x G float64 cells. The planes take the FFT buffer's
    // place once its last reader is done (the workgroup's LDS stays at 32 KB + 16 G bytes: four workgroups per CU)
    constexpr int CELLS = REASSIGN ? (L / 2 + 1) * G : 1;
    constexpr int RAW = REASSIGN && 2 * CELLS > D_POINTS ? 2 * CELLS : D_POINTS;
    __shared__ double raw[RAW];
    c32* const buf = reinterpret_cast<c32*>(raw);
    double* const txt = raw;
    float* const sm = reinterpret_cast<float*>(raw);
    static_assert(sizeof(double) * RAW <= 40 * 1024, "stft_fused_kernel: static LDS beyond a quarter of a gfx950 CU's 160 KB");
    constexpr int RL = (R3 > 1) ? R3 : R2;
    constexpr int NI = ((L / 2 + 1) * G + NT - 1) / NT;       // epilogue points per work-item
    using w_t = typename std::conditional<CST64, double, float>::type;
    const int tid = threadIdx.x;
    const bool deriv = A.dSx != nullptr || A.kidx != nullptr || REASSIGN;
    // The item's frames overlap (hop < L): their samples, hop (G - 1) + L of them, come in ONE coalesced pass -- global
    // -> LDS (the FFT buffer, idle until the first pass) -> registers -- instead of 16 four-byte loads per work-item
    // that touch G separate runs each. Frames too far apart for the buffer, and the padded-copy route, read global
    // memory directly as before.
    const int span = A.hop * (G - 1) + L;
    const bool staged = A.x != nullptr && span <= 2 * D_POINTS;
    // modulated: the frame is rotated by n_fft / 2 (utils/stft_utils.py:76-82) -- L is even: an XOR
    const int rot = A.modulated ? L / 2 : 0;
    // the items (signal, group of G frames), frames fastest. Workgroups are dealt to the 8 XCDs round-robin: each XCD
    // takes one contiguous range of items, so that the G*8-byte pieces of an output line meet in one L2 before they leave
    const int ng = (int)((A.n_hops + G - 1) / G);
    const int64_t total = (int64_t)ng * A.batch;
    int64_t item = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (A.xcd) {
        const int64_t nwg = (int64_t)gridDim.x * gridDim.y, per = nwg >> 3;       // (the launch pads nwg to a multiple of 8)
        item = (item & 7) * per + (item >> 3);
    }
    if (item >= total) return;
#ifdef STFT_STAMPS
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    {
        const int b = (int)(item / ng), c0 = (int)(item % ng) * G;
        const int t0 = A.hop * c0 - A.n1;                      // the first sample's index in the signal
        // (frames inside the signal read it directly -- all but the items at its two ends, padded on the fly)
        const bool inside = A.x != nullptr && t0 >= 0 && t0 + span - 1 < A.n;
        c32 z[PPT];
        if (staged) {
            // (the window pairs are asked for before the samples: their latency passes behind the staging barrier --
            // phase stamps, "r6x": reading them after it was a quarter of a workgroup's life)
            constexpr int NB = PPT / R1, STR = L / R1;
            float2 wv[PPT];
#pragma unroll
            for (int it = 0; it < NB; ++it) {
#pragma unroll
                for (int k = 0; k < R1; ++k) {
                    const int r = (tid + it * NT) / G + k * STR;
                    if (deriv && A.wd) wv[it * R1 + k] = A.wd[r];
                    else wv[it * R1 + k] = make_float2(A.window[r], 0.f);
                }
            }
            const float* xb = A.x + (int64_t)b * A.n;
            if (inside) {
                for (int i = tid * 4; i < span; i += NT * 4) {
                    if (i + 4 <= span) *reinterpret_cast<ssq_f4*>(sm + i) = *reinterpret_cast<const ssq_f4a4*>(xb + t0 + i);
                    else for (int q = i; q < span; ++q) sm[q] = xb[t0 + q];
                }
            } else {
                for (int i = tid; i < span; i += NT) {
                    const int src = stft_pad_source(t0 + i, A.n, A.padtype);
                    sm[i] = src < 0 ? 0.f : xb[src];
                }
            }
            __syncthreads();
            ST_STAMP(0);
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                const int idx = tid + it * NT, g = idx % G, u = idx / G;
                const bool live = c0 + g < A.n_hops;
                const float* fr = sm + A.hop * g;
#pragma unroll
                for (int k = 0; k < R1; ++k) {
                    const int r = u + k * STR;                // sample of the frame
                    const float v = live ? fr[r ^ rot] : 0.f;
                    z[it * R1 + k] = {v * wv[it * R1 + k].x, -(v * wv[it * R1 + k].y)};
                }
            }
            __syncthreads();                                   // the samples are in registers: the buffer is the FFT's
            ST_STAMP(1);
        } else {
            const float* xp = A.xp + (int64_t)b * A.padlen;
            const float* xs = A.x + (int64_t)b * A.n - A.n1;
            constexpr int NB = PPT / R1, STR = L / R1;
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                const int idx = tid + it * NT, g = idx % G, u = idx / G;
                const int c = c0 + g;
#pragma unroll
                for (int k = 0; k < R1; ++k) {
                    const int r = u + k * STR;                // sample of the frame
                    const int s = r ^ rot;
                    float a = 0.f, bb = 0.f;
                    if (c < A.n_hops) {
                        float v;
                        if (inside) v = xs[A.hop * c + s];     // (item-uniform: every sample of its frames exists)
                        else if (A.x) {                        // the signal's ends: padded on the fly
                            const int src = stft_pad_source(A.hop * c + s - A.n1, A.n, A.padtype);
                            v = src < 0 ? 0.f : A.x[(int64_t)b * A.n + src];
                        } else v = xp[(int64_t)A.hop * c + s];
                        a = v * A.window[r];
                        if (deriv) bb = v * A.diff_window[r];
                    }
                    z[it * R1 + k] = {a, -bb};
                }
            }
#ifdef STFT_STAMPS
            __builtin_amdgcn_s_waitcnt(0); ST_STAMP(1);
#endif
        }
        // the rows' frequencies of the epilogue's points (and the weight of the reference's linear grid -- a weight
        // vector is read where it is used) are asked for here, a transform ahead of their use
        float sf[REASSIGN ? NI : 1]; w_t cw0 = 0;
        if constexpr (REASSIGN) {
#pragma unroll
            for (int it = 0; it < NI; ++it) sf[it] = A.Sfs[min((tid + it * NT) / G, L / 2)];
            cw0 = ((const w_t*)A.cst)[0];
        }
        // (FRESH: a barrier has just passed; the twiddles asked for ahead of each pass' barrier measured no gain here)
        lds_ifft<L, G, R1, R2, R3, false, true>(z, buf, A.ftw, tid);
        __syncthreads();                              // last pass' LDS reads are done
        ST_STAMP(2);
        {
            constexpr int NB = PPT / RL, STR = L / RL;
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
                for (int k = 0; k < RL; ++k) buf[(u + k * STR) * G + g] = z[it * RL + k];
            }
        }
        __syncthreads();
        ST_STAMP(3);
        const int64_t base = (int64_t)b * A.rows * A.n_hops;
        if constexpr (!REASSIGN) {
            for (int i = tid; i < (L / 2 + 1) * G; i += NT) {
                const int f = i / G, g = i % G, c = c0 + g;
                if (c >= A.n_hops) continue;
                const c32 P = buf[f * G + g], Q = buf[((L - f) & (L - 1)) * G + g];
                const int64_t q = base + (int64_t)f * A.n_hops + c;
                const float sr = 0.5f * (P.x + Q.x), si = 0.5f * (Q.y - P.y);
                A.Sx[q] = make_float2(sr, si);
                if (!deriv) continue;
                const float dr = -0.5f * (P.y + Q.y), di = 0.5f * (Q.x - P.x);
                if (A.dSx) A.dSx[q] = make_float2(dr, di);
                if (A.kidx) {   // the fused kernels' rule (algos.py:956-984): |Sx| > gamma, then the bin
                    const int64_t omax = A.rows - 1;
                    unsigned short kk = 0xFFFFu;
                    if (mag_gt(sr, si, A.gamma)) {
                        const int64_t kb = bin_of_point(dr, di, sr, si, true, A.Sfs[f], sp, omax);
                        kk = (unsigned short)(sp.flipud ? omax - kb : kb);
                    }
                    A.kidx[q] = kk;
                }
            }
        } else {
            // every point of the item's frames leaves the FFT buffer for registers (its Sx on the way to HBM, its bin
            // beside it) before the buffer becomes the Tx planes. No early-outs: a point outside the frames, or below
            // gamma, carries the bin 0xFFFF
            float vr[NI], vi[NI]; unsigned kq[NI];
            const int omax = (int)A.rows - 1;
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int i = tid + it * NT, f = min(i / G, L / 2), g = i % G, c = c0 + g;
                const bool valid = i < (L / 2 + 1) * G && c < A.n_hops;
                const c32 P = buf[f * G + g], Q = buf[((L - f) & (L - 1)) * G + g];
                const float sr = 0.5f * (P.x + Q.x), si = 0.5f * (Q.y - P.y);
                const float dr = -0.5f * (P.y + Q.y), di = 0.5f * (Q.x - P.x);
                if (valid) {
                    A.Sx[base + (int64_t)f * A.n_hops + c] = make_float2(sr, si);
                    if (A.dSx) A.dSx[base + (int64_t)f * A.n_hops + c] = make_float2(dr, di);
                }
                // the fused kernels' rule (algos.py:956-984): |Sx| > gamma, then the bin
                const bool on = valid && mag_gt(sr, si, A.gamma);
                const int kb = bin_of_point_stft(dr, di, sr, si, sf[it], sp, omax, on);
                kq[it] = on ? (unsigned)(sp.flipud ? omax - kb : kb) : 0xFFFFu;
                vr[it] = sr; vi[it] = si;
                if (it % 3 == 2) SSQ_SCHED_FENCE();            // (three points' LDS reads in flight, not all nine)
            }
            ST_STAMP(4);
            __syncthreads();                              // the buffer's last read
            for (int i = tid; i < 2 * CELLS; i += NT) txt[i] = 0.0;
            __syncthreads();
            ST_STAMP(5);
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                if (kq[it] == 0xFFFFu) continue;
                const int g = (tid + it * NT) % G;
                // the term in the CPU path's arithmetic (float32 product; float64 with a float64 weight vector), the
                // sum in float64
                const w_t cw = A.cst_uniform ? cw0 : ((const w_t*)A.cst)[min((tid + it * NT) / G, L / 2)];
                const double tr = (double)((w_t)vr[it] * cw), ti = (double)((w_t)vi[it] * cw);
                const unsigned off = (kq[it] * G + (unsigned)g) * 8u;
                SSQ_LDS_ADD_F64(txt, off, tr);
                SSQ_LDS_ADD_F64(txt, off + (unsigned)CELLS * 8u, ti);
            }
            __syncthreads();
            ST_STAMP(6);
            for (int i = tid; i < (L / 2 + 1) * G; i += NT) {
                const int f = i / G, g = i % G, c = c0 + g;
                if (c >= A.n_hops) continue;
                A.Tx[base + (int64_t)f * A.n_hops + c] = make_float2((float)txt[i], (float)txt[CELLS + i]);
            }
            ST_STAMP(7);
#ifdef STFT_STAMPS
            __builtin_amdgcn_s_waitcnt(0);
            ST_STAMP(8);
#endif
        }
    }
}

// one workgroup per item; the grid is two-dimensional only to hold more than 2^31 - 1 of them
template <int L, int G, int R1, int R2, int R3>
static int launch_stft_fused(const StftFusedArgs& A, const SsqParams& sp, int64_t batch, hipStream_t stream) {
    SSQ_REQUIRE(!A.Tx || A.rows == L / 2 + 1, "fused reassignment: %lld rows, transform of %d", (long long)A.rows, L);
    static const bool remap = [] { const char* e = getenv("SSQ_DEBUG_STFT_XCD"); return !e || atoi(e) != 0; }();
    StftFusedArgs B = A;
    B.batch = (int)batch;
    int64_t total = (int64_t)((A.n_hops + G - 1) / G) * batch;
    B.xcd = remap && total >= 64;
    if (B.xcd) total = (total + 7) & ~(int64_t)7;
    // (grid.x a multiple of 8 when the items are remapped, so that the product is one too)
    const int64_t gx = std::min<int64_t>(total, (int64_t)1 << 20), gy = (total + gx - 1) / gx;
    SSQ_REQUIRE(gy <= 65535, "stft_fused_kernel: %lld items", (long long)total);
    dim3 grid((unsigned)gx, (unsigned)gy);
    if (!A.Tx)
        hipLaunchKernelGGL((stft_fused_kernel<L, G, R1, R2, R3, false, false>), grid, dim3(NT), 0, stream, B, sp);
    else if (sp.cst_f64)
        hipLaunchKernelGGL((stft_fused_kernel<L, G, R1, R2, R3, true, true>), grid, dim3(NT), 0, stream, B, sp);
    else
        hipLaunchKernelGGL((stft_fused_kernel<L, G, R1, R2, R3, true, false>), grid, dim3(NT), 0, stream, B, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}


// R2C with transposed (strided) output: transform c writes bin f at out[f*n_hops + c]
struct StridedR2C {
    rocfft_plan plan = nullptr; rocfft_execution_info info = nullptr;
    void* work = nullptr; size_t work_bytes = 0;
    int create(int dtype, size_t n_fft, size_t n_hops) {
        if (fft_global_setup()) return -4;
        rocfft_plan_description desc = nullptr;
        if (rocfft_plan_description_create(&desc) != rocfft_status_success) { set_error("rocfft desc"); return -4; }
        size_t offs = 0, in_stride = 1, out_stride = n_hops;
        if (rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real,
                rocfft_array_type_hermitian_interleaved, &offs, &offs, 1, &in_stride, n_fft, 1,
                &out_stride, 1) != rocfft_status_success) { set_error("rocfft layout"); return -4; }
        rocfft_status st = rocfft_plan_create(&plan, rocfft_placement_notinplace,
                rocfft_transform_type_real_forward,
                dtype == SSQ_F32 ? rocfft_precision_single : rocfft_precision_double, 1, &n_fft,
                n_hops, desc);
        rocfft_plan_description_destroy(desc);
        if (st != rocfft_status_success) { set_error("rocfft_plan_create (stft) failed: %d", (int)st); return -4; }
        rocfft_plan_get_work_buffer_size(plan, &work_bytes);
        rocfft_execution_info_create(&info);
        if (work_bytes) {
            SSQ_CHECK_HIP(hipMalloc(&work, work_bytes));
            rocfft_execution_info_set_work_buffer(info, work, work_bytes);
        }
        return 0;
    }
    int execute(void* in, void* out, hipStream_t stream) {
        rocfft_execution_info_set_stream(info, stream);
        void* ins[1] = {in}; void* outs[1] = {out};
        rocfft_status st = rocfft_execute(plan, ins, outs, info);
        if (st != rocfft_status_success) { set_error("rocfft_execute (stft) failed: %d", (int)st); return -4; }
        return 0;
    }
    void destroy() {
        if (info) rocfft_execution_info_destroy(info);
        if (plan) rocfft_plan_destroy(plan);
        if (work) (void)hipFree(work);
        info = nullptr; plan = nullptr; work = nullptr;
    }
};

}  // namespace ssq

using namespace ssq;

struct ssq_stft_plan {
    ssq_stft_desc d;
    int64_t padlen = 0, n1 = 0, n2 = 0, rows = 0, n_hops = 0;
    int rsize() const { return d.dtype == SSQ_F32 ? 4 : 8; }
    void* window = nullptr; void* diff_window = nullptr;
    void* xp = nullptr; void* frames = nullptr; void* dframes = nullptr; void* dSx_ws = nullptr;
    StridedR2C fft;
    bool fused = false; void* ftw = nullptr;      // fused float32 path (power-of-two n_fft)
    void* wd = nullptr;                           // ... its (window, diff_window) pairs, one 8-byte load per sample
    // fused float32 path for the other sizes (prime factors <= 31): mixed-radix LDS transform
    bool generic_fused = false; int gen_radix[GEN_MAX_PASSES] = {0}; int gen_npass = 0, gen_G = 0;
    unsigned short* kidx = nullptr;               // bin map of the fused ssq_stft form
    bool have_ssq = false; SsqParams sp{}; void* cst = nullptr; void* Sfs = nullptr;   // current entries of
    WeightVersions weights, freqs;                                                      // these
    PlanOrder order;
    bool executed = false;
};

extern "C" {

int ssq_stft_plan_create(ssq_stft_plan** out, const ssq_stft_desc* desc) {
    SSQ_REQUIRE(out && desc, "ssq_stft_plan_create: null pointer");
    const ssq_stft_desc& d = *desc;
    SSQ_REQUIRE(d.dtype == SSQ_F32 || d.dtype == SSQ_F64, "bad dtype %d", d.dtype);
    SSQ_REQUIRE(d.n >= 1 && d.n_fft >= 1 && d.hop_len >= 1, "bad sizes n=%lld n_fft=%lld hop=%lld",
                (long long)d.n, (long long)d.n_fft, (long long)d.hop_len);
    SSQ_REQUIRE(d.window, "window must not be null");
    SSQ_REQUIRE(d.padtype >= SSQ_PAD_ZERO && d.padtype <= SSQ_PAD_WRAP, "bad padtype %d", d.padtype);
    auto* pl = new ssq_stft_plan();
    pl->d = d;
    if (pl->d.max_batch < 1) pl->d.max_batch = 1;
    // padsignal(x, padlength = N + n_fft - 1): even total -> split evenly, odd -> left
    // gets the extra sample (utils/common.py:116-124)
    pl->padlen = d.n + d.n_fft - 1;
    int64_t tot = pl->padlen - d.n;
    pl->n2 = tot / 2;
    pl->n1 = (tot % 2 == 0) ? pl->n2 : pl->n2 + 1;
    pl->rows = d.n_fft / 2 + 1;
    pl->n_hops = (pl->padlen - d.n_fft) / d.hop_len + 1;
    const int rs = pl->rsize();
    int rc = 0;
#define TRYA(p, bytes) do { if (hipMalloc((void**)&(p), (bytes)) != hipSuccess) { set_error("hipMalloc failed (stft plan)"); ssq_stft_plan_destroy(pl); return -2; } } while (0)
    TRYA(pl->window, (size_t)d.n_fft * rs);
    SSQ_CHECK_HIP(hipMemcpy(pl->window, d.window, (size_t)d.n_fft * rs, hipMemcpyHostToDevice));
    if (d.diff_window) {
        TRYA(pl->diff_window, (size_t)d.n_fft * rs);
        SSQ_CHECK_HIP(hipMemcpy(pl->diff_window, d.diff_window, (size_t)d.n_fft * rs, hipMemcpyHostToDevice));
    }
    TRYA(pl->xp, (size_t)pl->d.max_batch * pl->padlen * rs);
#undef TRYA
    const bool pow2 = (d.n_fft & (d.n_fft - 1)) == 0;
    // (SSQ_DEBUG_STFT_MIXED=1: the mixed-radix kernel for the powers of two as well -- A/B aid)
    const bool prefer_mixed = getenv("SSQ_DEBUG_STFT_MIXED") && atoi(getenv("SSQ_DEBUG_STFT_MIXED")) != 0;
    if (d.dtype == SSQ_F32 && pow2 && d.n_fft >= 128 && d.n_fft <= 2048 && !getenv("SSQ_DEBUG_STFT_GENERIC") && !prefer_mixed) {
        std::vector<float> tw((size_t)2 * d.n_fft);
        for (int64_t q = 0; q < d.n_fft; ++q) {
            double ang = 2.0 * 3.14159265358979323846 * (double)q / (double)d.n_fft;
            tw[2 * q] = (float)cos(ang); tw[2 * q + 1] = (float)sin(ang);
        }
        if (hipMalloc(&pl->ftw, tw.size() * 4) != hipSuccess) { set_error("hipMalloc failed (stft plan)"); ssq_stft_plan_destroy(pl); return -2; }
        SSQ_CHECK_HIP(hipMemcpy(pl->ftw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
        if (d.diff_window) {
            std::vector<float> wd((size_t)2 * d.n_fft);
            for (int64_t q = 0; q < d.n_fft; ++q) {
                wd[2 * q] = ((const float*)d.window)[q]; wd[2 * q + 1] = ((const float*)d.diff_window)[q];
            }
            if (hipMalloc(&pl->wd, wd.size() * 4) != hipSuccess) { set_error("hipMalloc failed (stft plan)"); ssq_stft_plan_destroy(pl); return -2; }
            SSQ_CHECK_HIP(hipMemcpy(pl->wd, wd.data(), wd.size() * 4, hipMemcpyHostToDevice));
        }
        pl->fused = true;
    } else if (d.dtype == SSQ_F32 && !getenv("SSQ_DEBUG_STFT_GENERIC")
               && stft_generic_plan(d.n_fft, pl->gen_radix, &pl->gen_npass, &pl->gen_G)) {
        std::vector<float> tw((size_t)2 * d.n_fft);
        for (int64_t q = 0; q < d.n_fft; ++q) {
            double ang = 2.0 * 3.14159265358979323846 * (double)q / (double)d.n_fft;
            tw[2 * q] = (float)cos(ang); tw[2 * q + 1] = (float)sin(ang);
        }
        if (hipMalloc(&pl->ftw, tw.size() * 4) != hipSuccess) { set_error("hipMalloc failed (stft plan)"); ssq_stft_plan_destroy(pl); return -2; }
        SSQ_CHECK_HIP(hipMemcpy(pl->ftw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
        pl->generic_fused = true;
    }
    // The framing workspace (two n_fft x n_hops arrays) and the rocFFT plan belong to the unfused route only: a fused
    // plan does without them (at the reference's hop-1 benchmark shape they would be 2 x 383 MB). The derivative's
    // workspace (dSx not returned, and no bin map in its place) is allocated at the first execute that needs it.
    if (!pl->fused && !pl->generic_fused) {
        const size_t fb = (size_t)d.n_fft * pl->n_hops * rs;
        if (hipMalloc(&pl->frames, fb) != hipSuccess || hipMalloc(&pl->dframes, fb) != hipSuccess) {
            set_error("hipMalloc failed (stft plan)"); ssq_stft_plan_destroy(pl); return -2;
        }
        rc = pl->fft.create(d.dtype, (size_t)d.n_fft, (size_t)pl->n_hops);
        if (rc) { ssq_stft_plan_destroy(pl); return rc; }
    }
    pl->d.window = nullptr; pl->d.diff_window = nullptr;
    *out = pl;
    return 0;
}

void ssq_stft_plan_destroy(ssq_stft_plan* pl) {
    if (!pl) return;
    pl->fft.destroy();
    pl->weights.destroy(); pl->freqs.destroy(); pl->order.destroy();
    void* ptrs[] = {pl->window, pl->diff_window, pl->xp, pl->frames, pl->dframes, pl->dSx_ws,
                    pl->ftw, pl->kidx, pl->wd};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete pl;
}

int ssq_stft_plan_set_ssq(ssq_stft_plan* pl, const void* Sfs, int grid, const double* params,
                          const void* cst, int cst_f64, int flipud, double gamma) {
    SSQ_REQUIRE(pl && Sfs && params && cst, "ssq_stft_plan_set_ssq: null pointer");
    SSQ_REQUIRE(grid >= SSQ_GRID_LOG && grid <= SSQ_GRID_LIN, "unknown grid kind %d", grid);
    for (int t = 0; t < 5; ++t) pl->sp.p[t] = params[t];
    pl->sp.grid = grid; pl->sp.flipud = flipud ? 1 : 0; pl->sp.gamma = gamma;
    pl->sp.cst_f64 = (cst_f64 && pl->d.dtype == SSQ_F32) ? 1 : 0;
    {
        const bool wide = cst_f64 || pl->d.dtype == SSQ_F64;
        bool uni = true;
        for (int64_t i = 1; i < pl->rows && uni; ++i)
            uni = wide ? ((const double*)cst)[i] == ((const double*)cst)[0]
                       : ((const float*)cst)[i] == ((const float*)cst)[0];
        pl->sp.cst_uniform = uni ? 1 : 0;
    }
    finalize_params(pl->sp);
    const int rs = pl->rsize();
    std::lock_guard<std::mutex> lock(pl->order.mu);
    int rc = pl->weights.upload(&pl->cst, cst, (size_t)pl->rows * ((cst_f64 || rs == 8) ? 8 : 4));
    if (rc) return rc;
    rc = pl->freqs.upload(&pl->Sfs, Sfs, (size_t)pl->rows * rs);
    if (rc) return rc;
    pl->have_ssq = true;
    return 0;
}

#ifdef STFT_STAMPS
int ssq_debug_stft_prof(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssq::g_stft_prof), sizeof(unsigned long long) * 16);
}
#endif
const char* ssq_stft_plan_algo(const ssq_stft_plan* pl) {
    return !pl ? "" : pl->fused ? "fused" : pl->generic_fused ? "fused-mixed-radix" : "rocfft";
}

int ssq_stft_plan_shape(const ssq_stft_plan* pl, int64_t* rows, int64_t* n_hops) {
    SSQ_REQUIRE(pl, "ssq_stft_plan_shape: null plan");
    if (rows) *rows = pl->rows;
    if (n_hops) *n_hops = pl->n_hops;
    return 0;
}

}  // extern "C"

template <typename T>
static int stft_execute_t(ssq_stft_plan* pl, const void* x, int64_t batch, void* Sx, void* dSx,
                          void* Tx, void* w, hipStream_t stream) {
    const ssq_stft_desc& d = pl->d;
    const int64_t rows = pl->rows, n_hops = pl->n_hops, n_fft = d.n_fft;
    const bool deriv = dSx || Tx || w;
    SSQ_REQUIRE(!deriv || pl->diff_window, "derivative outputs need a diff_window");
    // (the power-of-two fused kernel pads on the fly; the other routes read a padded copy)
    const bool pad_in_kernel = sizeof(T) == 4 && pl->fused && d.n + pl->n1 + pl->n2 < ((int64_t)1 << 30);
    int rc = 0;
    if (!pad_in_kernel) rc = ssq_pad_signal(d.dtype, x, pl->xp, batch, d.n, pl->n1, pl->n2, d.padtype, stream);
    if (rc) return rc;
    const int64_t s20 = (n_fft + 1) / 2, s21 = (n_fft % 2 == 1) ? s20 - 1 : s20;
    // (decided below: whether the derivative is stored at all; its workspace exists from the first call that needs one)
    auto dsx_workspace = [&]() -> void* {
        if (!pl->dSx_ws) {
            if (hipMalloc(&pl->dSx_ws, (size_t)pl->d.max_batch * rows * n_hops * sizeof(T) * 2) != hipSuccess) return nullptr;
        }
        return pl->dSx_ws;
    };
    T* dS = (T*)dSx;
    // fused ssq_stft form: Tx wanted, neither dSx nor w -> the kernel emits 2-byte bins
    // instead of the 8-byte derivative (the reassignment then reads 10 instead of 16 B/pt)
    // ... and, unless the ordered sums are asked for (SSQ_TILE_ORDER=ordered), sums Tx itself
    bool use_kidx = false, fused_tx = false;
    if constexpr (sizeof(T) == 4) {
        if ((pl->fused || pl->generic_fused) && Tx && !w && !dSx && rows < 65535) {
            use_kidx = true;
            // (round 6, the Tx planes in the FFT buffer's place, four workgroups per CU: config 3 at 512 signals 1.87 ms
            // against 2.19 with the separate pass, 64 signals 0.26 against 0.31, hop 1 0.83 against 1.02 -- profiles/
            // r6_ab_history.txt "r6w"; the separate pass stays for the ordered sums; SSQ_DEBUG_STFT_FUSED_TX=0/1 forces)
            const char* fe = getenv("SSQ_DEBUG_STFT_FUSED_TX");        // (read at every call, like SSQ_TILE_ORDER: tests switch it)
            const int force = (fe && *fe) ? atoi(fe) : -1;
            fused_tx = !reassign_ordered() && rows == n_fft / 2 + 1 && force != 0;
            if (!fused_tx && !pl->kidx)
                SSQ_CHECK_HIP(hipMalloc((void**)&pl->kidx, (size_t)pl->d.max_batch * rows * n_hops * 2));
        }
    }
    if (deriv && !dSx && !use_kidx) {              // the derivative is needed (phase transform / bins) but not returned
        dS = (T*)dsx_workspace();
        SSQ_REQUIRE(dS, "hipMalloc failed (stft derivative workspace)");
    }
    if constexpr (sizeof(T) == 4) {
        if (pl->fused) {
            StftFusedArgs A;
            A.xp = (const float*)pl->xp; A.window = (const float*)pl->window;
            A.diff_window = (const float*)pl->diff_window; A.ftw = (const c32*)pl->ftw;
            A.Sx = (float2*)Sx; A.dSx = (deriv && !use_kidx) ? (float2*)dS : nullptr;
            A.kidx = (use_kidx && !fused_tx) ? pl->kidx : nullptr; A.Sfs = (const float*)pl->Sfs; A.gamma = pl->sp.gamma;
            A.Tx = fused_tx ? (float2*)Tx : nullptr; A.cst = pl->cst; A.cst_uniform = pl->sp.cst_uniform;
            A.padlen = pl->padlen; A.n_hops = n_hops; A.rows = rows;
            A.hop = (int)d.hop_len; A.s20 = (int)s20; A.s21 = (int)s21; A.modulated = d.modulated;
            A.x = pad_in_kernel ? (const float*)x : nullptr; A.n = (int)d.n; A.n1 = (int)pl->n1; A.padtype = d.padtype;
            A.wd = (const float2*)pl->wd;
            switch (n_fft) {
                case 128: rc = launch_stft_fused<128, 32, 16, 8, 1>(A, pl->sp, batch, stream); break;
                case 256: rc = launch_stft_fused<256, 16, 16, 16, 1>(A, pl->sp, batch, stream); break;
                case 512: rc = launch_stft_fused<512, 8, 8, 8, 8>(A, pl->sp, batch, stream); break;
                case 1024: rc = launch_stft_fused<1024, 4, 16, 8, 8>(A, pl->sp, batch, stream); break;
                default: rc = launch_stft_fused<2048, 2, 16, 16, 8>(A, pl->sp, batch, stream); break;
            }
            if (rc) return rc;
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (pl->generic_fused) {
            StftFusedArgs A;
            A.xp = (const float*)pl->xp; A.window = (const float*)pl->window;
            A.diff_window = (const float*)pl->diff_window; A.ftw = nullptr;
            A.Sx = (float2*)Sx; A.dSx = (deriv && !use_kidx) ? (float2*)dS : nullptr;
            A.kidx = (use_kidx && !fused_tx) ? pl->kidx : nullptr; A.Sfs = (const float*)pl->Sfs; A.gamma = pl->sp.gamma;
            A.Tx = fused_tx ? (float2*)Tx : nullptr; A.cst = pl->cst; A.cst_uniform = pl->sp.cst_uniform;
            A.padlen = pl->padlen; A.n_hops = n_hops; A.rows = rows;
            A.hop = (int)d.hop_len; A.s20 = (int)s20; A.s21 = (int)s21; A.modulated = d.modulated; A.xcd = 0;
            A.x = nullptr; A.n = (int)d.n; A.n1 = (int)pl->n1; A.padtype = d.padtype; A.wd = nullptr; A.batch = (int)batch;
            rc = launch_stft_generic(A, pl->sp, (const c32*)pl->ftw, (int)n_fft, pl->gen_radix, pl->gen_npass, pl->gen_G,
                                     batch, stream);
            if (rc) return rc;
        }
    }
    for (int64_t b = 0; b < ((pl->fused || pl->generic_fused) ? 0 : batch); ++b) {
        const T* xp = (const T*)pl->xp + (size_t)b * pl->padlen;
        int64_t total = n_fft * n_hops;
        unsigned g = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
        hipLaunchKernelGGL((frame_window_kernel<T>), dim3(g), dim3(256), 0, stream, xp,
                           (const T*)pl->window, (const T*)pl->diff_window, (T*)pl->frames,
                           deriv ? (T*)pl->dframes : (T*)nullptr, n_fft, n_hops, d.hop_len, s20, s21,
                           d.modulated);
        SSQ_LAUNCH_CHECK();
        T* Sx_b = (T*)Sx + (size_t)b * rows * n_hops * 2;
        rc = pl->fft.execute(pl->frames, Sx_b, stream);
        if (rc) return rc;
        if (deriv) {
            rc = pl->fft.execute(pl->dframes, dS + (size_t)b * rows * n_hops * 2, stream);
            if (rc) return rc;
        }
    }
    if (w) {
        rc = ssq_phase_stft(d.dtype, Sx, dS, pl->Sfs, w, batch, rows, n_hops, pl->sp.gamma, stream);
        if (rc) return rc;
    }
    if (Tx && !fused_tx) {
        if (w)
            rc = launch_accumulate(d.dtype, BIN_FROM_W, Sx, w, nullptr, Tx, pl->cst, pl->sp, batch,
                                   rows, n_hops, nullptr, stream);
        else if (use_kidx)
            rc = launch_accumulate(d.dtype, BIN_FROM_KIDX, Sx, pl->kidx, nullptr, Tx, pl->cst, pl->sp, batch,
                                   rows, n_hops, nullptr, stream);
        else
            rc = launch_accumulate(d.dtype, BIN_FROM_DWX, Sx, dS, pl->Sfs, Tx, pl->cst, pl->sp, batch,
                                   rows, n_hops, nullptr, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int ssq_stft_execute(ssq_stft_plan* pl, const void* x, int64_t batch, void* Sx, void* dSx,
                                void* Tx, void* w, void* stream) {
    SSQ_REQUIRE(pl && x && Sx, "ssq_stft_execute: null pointer");
    SSQ_REQUIRE(batch >= 1 && batch <= pl->d.max_batch, "batch %lld outside [1, %lld]",
                (long long)batch, (long long)pl->d.max_batch);
    SSQ_REQUIRE(!(Tx || w) || pl->have_ssq, "Tx / w requested but ssq parameters were not set");
    hipStream_t st = as_stream(stream);
    pl->order.enter(st);
    const int rc = pl->d.dtype == SSQ_F32 ? stft_execute_t<float>(pl, x, batch, Sx, dSx, Tx, w, st)
                                          : stft_execute_t<double>(pl, x, batch, Sx, dSx, Tx, w, st);
    pl->executed = true;
    pl->order.leave(st);
    return rc;
}
