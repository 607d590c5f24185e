// ssq_cwt_blocks.hip -- the CWT fast path: overlap-save "zoom" iFFT, LDS-resident,
// fused with the unpad / phase-transform / bin-map epilogue.  gfx950; float32 (tuned, with
// a lean fused-ssq instantiation and four-step exact kernels) and float64.
//
// Math (see ssqueezepy_amd/_blocks.py for the derivation and the host planning):
// a row whose impulse response fits +-m samples is evaluated block by block; block b
// of class (P, m, V) covers padded samples t0 = n1 - m + b*V ... t0 + P and is exact
// on its central V samples. With X_b = FFT_P(block) and the wavelet's P-grid samples
// psi[kappa] (the M-grid bank at every (M/P)-th bin), block output sample
//   t = q*R' + c   (R' = P/L', q in [0, L'), c in [0, R'))
// is   y[t] = sum_kappa  psi[kappa] X_b[kappa] e^{2i pi kappa c / P} / P  *  e^{2i pi kappa q / L'}
// i.e. entry q of an L'-point inverse FFT whose input at (kappa mod L') is the band
// entry times a per-column twiddle. The derivative row uses inputs times 1j*xi/dt.
//
// One workgroup (256 threads) = one (row, block, group of G adjacent columns):
// D = L'*G = 4096 complex points, 16 per thread, laid out in LDS as [q][g] with the
// column index fastest -- so every LDS access of the Stockham passes is
// conflict-free for G >= 16, and the final outputs (held in registers) go to HBM as
// runs of G adjacent time samples (128-byte segments for G = 16). Per workgroup:
// 32 KiB of LDS for the FFT + up to 17 KiB of staged band / twiddle powers; 140-160 VGPRs
// -> 3 workgroups per CU. HBM traffic: Wx (8 B/pt) + bin map
// (2 B/pt) written once; the inputs (band of X_b, psi, twiddles: a few KiB per
// workgroup) come from L2. No intermediate array is ever written.
//
// Compiled with -ffp-contract=off because the epilogue computes bin indices with the
// exact operation sequence of the CPU path (ssq_point_math.inl); the FFT butterflies
// use explicit fmaf.
#include "ssq_common.h"
#include "ssq_blocks.h"
#include "ssq_ldsfft.h"
#include <cmath>

namespace ssq {

#include "ssq_point_math.inl"

// ---- per-point output of the fused epilogues (unpadded Wx [, dWx, w, bin map])
struct EmitRow {
    float2* W; float2* D; float* w; unsigned short* k;
    float rs; bool scale; double gamma; int64_t omax;
    float m2hi, m2lo;          // |Wx|^2 screens of mag_gt (ssq_point_math.inl)
    int fx, fa;                // flipud as (k ^ fx) + fa: (-1, omax + 1) or (0, 0)
};
__device__ __forceinline__ EmitRow make_emit_row(float* Wx, float* dWx, float* w, unsigned short* kidx,
                                                 const float* row_scale, int sig, int ksig, int row,
                                                 int64_t na, int64_t N, double gamma, int flipud) {
    EmitRow e;
    const int64_t base = ((int64_t)sig * na + row) * N;
    e.W = reinterpret_cast<float2*>(Wx) + base;
    e.D = dWx ? reinterpret_cast<float2*>(dWx) + base : nullptr;
    e.w = w ? w + base : nullptr;
    e.k = kidx ? kidx + (int64_t)ksig * na * N + kidx_index(row, 0, na, N) : nullptr;
    e.scale = row_scale != nullptr;
    e.rs = row_scale ? row_scale[row] : 1.f;
    e.gamma = gamma; e.omax = na - 1;
    const float g2 = (float)(gamma * gamma);
    e.m2hi = g2 * 1.000004f; e.m2lo = g2 * 0.999996f;
    e.fx = flipud ? -1 : 0; e.fa = flipud ? (int)na : 0;
    return e;
}
// Stores one point. Returns true when its bin could not be decided by the float32
// screens: the caller then runs `emit_point_exact` for it outside the unrolled loop
// (one copy of the double-precision code per kernel instead of one per point -- the
// unrolled epilogue stays small enough for the instruction cache).
// LEAN: the fused ssq_cwt form (Wx + bin map only) -- the kernels are instantiated
// separately for it so that its epilogue carries no code for the optional outputs, and
// its bin screen is the branch-free one, specialised per grid kind (GRID).
template <bool LEAN, int GRID>
__device__ __forceinline__ bool emit_point(const EmitRow& e, int j, c32 W, c32 D, const SsqParams& sp) {
    float c = W.x, d = W.y, a = D.x, b = D.y;
    if constexpr (LEAN) {                    // lean kernels run without row scaling
        e.W[(unsigned)j] = make_float2(c, d);
        const float m2 = c * c + d * d, num = b * c - a * d;
        const bool above = m2 > e.m2hi, below = m2 < e.m2lo;
        // hardware reciprocal: relative error of w under 3e-7 (see bin_of_point)
        const float w32 = fabsf(num * __builtin_amdgcn_rcpf(m2 * 6.2831855f));
        bool ok;
        const int kb = bin_screen_cwt<GRID>(w32, sp, (int)e.omax, ok);
        const int kf = (kb ^ e.fx) + e.fa;
        e.k[(unsigned)j] = (unsigned short)(above ? kf : 0xFFFF);   // overwritten by the exact path if undecided
        return !(below | (above & ok));
    } else {
        c = c * e.rs; d = d * e.rs; a = a * e.rs; b = b * e.rs;        // rs == 1 (exact) when unscaled
        e.W[j] = make_float2(c, d);
        if (e.D) e.D[j] = make_float2(a, b);
        if (e.w) {
            float wv;
            if (mag_lt(c, d, (float)e.gamma)) wv = INFINITY;
            else wv = (float)fabs(phase_ratio(a, b, c, d));
            e.w[j] = wv;
        }
        if (e.k) {
            const int above = mag_gt_screen(c, d, e.gamma);
            if (above < 0) return true;
            unsigned short kk = 0xFFFFu;
            if (above) {
                const int kb = bin_of_point_screen(a, b, c, d, sp, (int)e.omax);
                if (kb == -2) return true;
                kk = (unsigned short)(sp.flipud ? (int)e.omax - kb : kb);
            }
            e.k[j] = kk;
        }
        return false;
    }
}
// run the epilogue specialised for the grid kind (lean kernels) or generically
template <int V> struct GridTag { static constexpr int value = V; };
template <bool LEAN, typename F>
__device__ __forceinline__ void dispatch_grid(int grid, F&& f) {
    if constexpr (LEAN) {
        if (grid == SSQ_GRID_LOG) f(GridTag<SSQ_GRID_LOG>{});
        else if (grid == SSQ_GRID_LOG_PIECEWISE) f(GridTag<SSQ_GRID_LOG_PIECEWISE>{});
        else f(GridTag<SSQ_GRID_LIN>{});
    } else {
        f(GridTag<-1>{});
    }
}
__device__ __forceinline__ void emit_point_exact(const EmitRow& e, unsigned short* kout, c32 W, c32 D,
                                                 const SsqParams& sp) {
    const float c = W.x * e.rs, d = W.y * e.rs, a = D.x * e.rs, b = D.y * e.rs;
    unsigned short kk = 0xFFFFu;
    if (mag_of(c, d) > e.gamma) {
        const int64_t kb = bin_of_point_exact(a, b, c, d, sp, e.omax);
        kk = (unsigned short)(sp.flipud ? e.omax - kb : kb);
    }
    *kout = kk;
}
struct BlockArgs {
    const int4* items;                 // (row, block, c0, class)
    const BlockRowDev* rows;
    const BlockClassDev* classes;
    const float* pbank;
    const float* pxi;
    const c32* ctw;                    // per-class column twiddles e^{2 pi i q / P}
    const c32* ftw;                    // e^{2 pi i q / L}
    const c32* xb;                     // block spectra, all classes, all signals
    const float* row_scale;
    float* Wx; float* dWx; float* w; unsigned short* kidx;
    int64_t M, N, na, n_items;
    double h;                          // 2 pi / M
    float inv_dt;
    double gamma;
    int sig;                           // first signal of the launch (blockIdx.y adds to it)
};

// The body of one workgroup (item `item_idx` of the class's list; blockIdx.y = signal of the launch). Its LDS
// arrays come from the caller: blockzoom_kernel sizes them for its class, blockzoom_multi_kernel -- every class
// in ONE launch, for transforms too small to fill the GPU per class -- shares one set between the classes.
// NOD: the call asks for Wx alone (a plain cwt: no dWx, no w, no bin map) -- the derivative's inputs, its transform and
// its share of the epilogue are compiled out (config 1's block rows 27 -> 17 us).
template <int L, int G, int R1, int R2, int R3, bool LEAN, bool NOD = false>
__device__ __forceinline__ void blockzoom_body(const BlockArgs& A, const SsqParams& sp, int item_idx,
                                               c32* __restrict__ buf, c32* __restrict__ spow,
                                               c32* __restrict__ wrapf, c32* __restrict__ bandW,
                                               c32* __restrict__ bandD) {
    constexpr int RL = (R3 > 1) ? R3 : R2;         // last radix
    const int tid = threadIdx.x;
    const int4 item = A.items[item_idx];
    const int row = item.x, blk = item.y, c0 = item.z;
    const BlockRowDev r = A.rows[row];
    const BlockClassDev cl = A.classes[item.w];
    const int P = (int)cl.P, Rp = P / L;
    const int sig = A.sig + (int)blockIdx.y;
    const c32* xb = A.xb + cl.xb_off + ((int64_t)sig * cl.nb + blk) * cl.xb_stride;
    const c32* ctw = A.ctw + cl.ctw_off;
    const float* psi = A.pbank + r.pb_off;
    const float* pxi = A.pxi + r.pb_off;

    // ---- prologue: pass-1 inputs of both transforms, in registers.
    // Input slot q of column c holds  psi[kap] X[kap] e^{2 pi i kap c / P} / P  with
    // kap = klo + ((q - klo) mod L). A thread's R1 slots are q = u + t*STR, so its bins
    // advance by STR (minus L on wrap-around): the column twiddle is one table gather
    // for t = 0 times a power of s_c = e^{2 pi i STR c / P} (and the wrap factor
    // e^{-2 pi i L c / P}), and those powers -- R1 x G values per workgroup -- are staged
    // in LDS once. 8x fewer scattered table gathers than one per point (the gathers,
    // not the arithmetic, bounded this stage).
    // (spow: R1 * G entries, wrapf: G)
    // The band itself (psi X / P and its derivative multiple) is the same for all G
    // columns of the workgroup: for the small transforms it is formed once per
    // workgroup in LDS instead of once per point from global memory (bandW, bandD: L entries).
    constexpr bool STAGE = (L <= 512);
    const float invP = 1.0f / (float)P;             // exact (P is a power of two)
    // Round 5, the large classes (no band staging): every global load of the prologue -- the staged twiddles, the
    // work-item's first column twiddle and the band's three loads per point -- is issued up front, into registers,
    // and only then are the staged values written to LDS and the barrier taken: the band's latency runs under the
    // staging instead of starting behind its barrier. Block rows 69.4 -> 64.0 us (an ablation without the band's
    // loads runs at 51.8: profiles/r5_ab_history.txt). -DSSQ_BLOCK_EARLY=0: loads where they are used, as before.
#ifndef SSQ_BLOCK_EARLY
#define SSQ_BLOCK_EARLY 1
#endif
    constexpr bool EARLY = !STAGE && SSQ_BLOCK_EARLY && (R1 * G <= NT);
    float e_p[EARLY ? PPT : 1], e_m[EARLY ? PPT : 1];
    c32 e_X[EARLY ? PPT : 1], e_cw[EARLY ? PPT / R1 : 1];
    if constexpr (EARLY) {
        constexpr int NBe = PPT / R1, STR = L / R1;
        c32 st_v = {0.f, 0.f}, st_w = {0.f, 0.f};
        if (tid < R1 * G) {
            const unsigned t = (unsigned)(tid / G), col = (unsigned)(c0 + tid % G);
            st_v = ctw[(t * (unsigned)STR * col) & (unsigned)(P - 1)];
        }
        if (tid < G) st_w = ctw[(0u - (unsigned)L * (unsigned)(c0 + tid)) & (unsigned)(P - 1)];
#pragma unroll
        for (int it = 0; it < NBe; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
            const int off0 = (u - r.klo) & (L - 1);
            e_cw[it] = ctw[((unsigned)(r.klo + off0) * (unsigned)(c0 + g)) & (unsigned)(P - 1)];
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int off = (off0 + k * STR) & (L - 1);
                const bool in = off < r.KP;              // slots past the band: zeros
                e_p[it * R1 + k] = in ? psi[off] : 0.f;
                e_X[it * R1 + k] = in ? xb[r.klo + off] : c32{0.f, 0.f};
                e_m[it * R1 + k] = in ? pxi[off] : 0.f;
            }
        }
        if (tid < R1 * G) spow[tid] = st_v;
        if (tid < G) wrapf[tid] = st_w;
    } else {
        constexpr int STR = L / R1;
        for (int i = tid; i < R1 * G; i += NT) {
            const unsigned t = (unsigned)(i / G), col = (unsigned)(c0 + i % G);
            spow[i] = ctw[(t * (unsigned)STR * col) & (unsigned)(P - 1)];
        }
        if (tid < G) {
            const unsigned col = (unsigned)(c0 + tid);
            wrapf[tid] = ctw[(0u - (unsigned)L * col) & (unsigned)(P - 1)];
        }
        if constexpr (STAGE) {
            for (int off = tid; off < L; off += NT) {
                c32 b = {0.f, 0.f}, d = {0.f, 0.f};
                if (off < r.KP) {
                    const float p = psi[off] * invP;
                    const c32 X = xb[r.klo + off];
                    b = {p * X.x, p * X.y};
                    const float mm = pxi[off] * A.inv_dt;   // 1j*xi/dt, xi as the reference stores it
                    d = {-(b.y * mm), b.x * mm};
                }
                bandW[off] = b; bandD[off] = d;
            }
        }
    }
    __syncthreads();
    c32 zw[PPT], zd[PPT];
    {
        constexpr int NB = PPT / R1, STR = L / R1;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
            const unsigned col = (unsigned)(c0 + g);
            const int off0 = (u - r.klo) & (L - 1);
            const c32 cw0 = EARLY ? e_cw[EARLY ? it : 0] : ctw[((unsigned)(r.klo + off0) * col) & (unsigned)(P - 1)];
            const c32 wf = wrapf[g];
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int offu = off0 + k * STR;            // < 2L
                const int off = offu & (L - 1);             // band element at FFT slot q
                c32 z = {0.f, 0.f}, dz = {0.f, 0.f};
                if constexpr (STAGE) {
                    // zero-padded band: slots past KP hold zeros, no branch needed
                    c32 cw = (k == 0) ? cw0 : cmul_v(cw0, spow[k * G + g]);
                    if (offu >= L) cw = cmul_v(cw, wf);
                    z = cmul_v(bandW[off], cw);
                    dz = cmul_v(bandD[off], cw);
                } else if (EARLY || off < r.KP) {            // (EARLY: slots past the band were loaded as zeros)
                    const float p = (EARLY ? e_p[EARLY ? it * R1 + k : 0] : psi[off]) * invP;
                    const c32 X = EARLY ? e_X[EARLY ? it * R1 + k : 0] : xb[r.klo + off];
                    c32 cw = (k == 0) ? cw0 : cmul_v(cw0, spow[k * G + g]);
                    if (offu >= L) cw = cmul_v(cw, wf);
                    const c32 bz = {p * X.x, p * X.y};
                    z = cmul_v(bz, cw);
                    const float mm = (EARLY ? e_m[EARLY ? it * R1 + k : 0] : pxi[off]) * A.inv_dt;
                    dz = {-(z.y * mm), z.x * mm};
                }
                zw[it * R1 + k] = z; zd[it * R1 + k] = dz;
            }
        }
    }
    // (twiddles requested ahead of each pass' barrier; nothing has touched `buf` before the first transform:
    // 58.9 -> 56.7 -> 56.2 us at config 2, round 5)
    lds_ifft<L, G, R1, R2, R3, true, true>(zw, buf, A.ftw, tid);
    if constexpr (!NOD) lds_ifft<L, G, R1, R2, R3, true>(zd, buf, A.ftw, tid);

    // ---- epilogue: unpad, store, phase transform, bin map
    constexpr int NB = PPT / RL, STR = L / RL;
    const EmitRow er = make_emit_row(A.Wx, A.dWx, A.w, A.kidx, A.row_scale, sig, (int)blockIdx.y, row, A.na, A.N, A.gamma, sp.flipud);
    const int m = (int)cl.m, N = (int)A.N;
    const int jbase = blk * (int)cl.V - m + c0;
    // valid outputs of the block: samples [m, m + V) that fall inside the signal
    const unsigned span = (unsigned)max(0, min((int)cl.V, N - blk * (int)cl.V));
    unsigned pend = 0;                              // slots whose bin needs the exact map
    auto emit_all = [&](auto grid_tag) {
        constexpr int GRID = decltype(grid_tag)::value;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const int dcol = (u + k * STR) * Rp + g + (c0 - m);   // sample index inside the block - margin
                if ((unsigned)dcol >= span) continue;               // margin or past the signal's end
                const int j = dcol + blk * (int)cl.V;               // output column
                if constexpr (NOD) {                                 // Wx alone (rs == 1, exact, when unscaled)
                    er.W[j] = make_float2(zw[it * RL + k].x * er.rs, zw[it * RL + k].y * er.rs);
                    continue;
                }
                if (emit_point<LEAN, GRID>(er, j, zw[it * RL + k], NOD ? c32{0.f, 0.f} : zd[it * RL + k], sp))
                    pend |= 1u << (it * RL + k);
            }
        }
    };
    dispatch_grid<LEAN>(sp.grid, emit_all);
    // rare: points inside a screen's guard band. One pending point per lane and round,
    // picked with compile-time slot indices (zw/zd stay in registers).
    while (__builtin_amdgcn_ballot_w64(pend != 0)) {
        const unsigned low = pend & (0u - pend);
        pend ^= low;
        c32 W = {0.f, 0.f}, D = {0.f, 0.f};
        int j = 0;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < RL; ++k)
                if (low == (1u << (it * RL + k))) {
                    W = zw[it * RL + k]; D = NOD ? c32{0.f, 0.f} : zd[it * RL + k]; j = jbase + (u + k * STR) * Rp + g;
                }
        }
        if (low) emit_point_exact(er, er.k + j, W, D, sp);
    }
}

template <int L, int G, int R1, int R2, int R3, bool LEAN, bool NOD = false>
__global__ __launch_bounds__(NT) void blockzoom_kernel(BlockArgs A, SsqParams sp) {
    __shared__ c32 buf[D_POINTS];
    __shared__ c32 spow[R1 * G];
    __shared__ c32 wrapf[G];
    __shared__ c32 bandW[(L <= 512) ? L : 1], bandD[(L <= 512) ? L : 1];
    blockzoom_body<L, G, R1, R2, R3, LEAN, NOD>(A, sp, (int)blockIdx.x, buf, spow, wrapf, bandW, bandD);
}

// every class of a plan in one launch: workgroup b belongs to the class whose item range holds b
struct BlockMultiArgs {
    BlockArgs A;                       // (items, n_items, ftw: filled per class inside)
    const int4* items[5]; const c32* ftw[5];
    int first[6];                      // first workgroup of class s; first[5] = grid size
};
template <bool LEAN, bool NOD = false>
__global__ __launch_bounds__(NT) void blockzoom_multi_kernel(BlockMultiArgs M, SsqParams sp) {
    __shared__ c32 buf[D_POINTS];
    __shared__ c32 spow[512];          // max R1 * G (16 x 32)
    __shared__ c32 wrapf[32];
    __shared__ c32 bandW[512], bandD[512];
    const int b = (int)blockIdx.x;
    int s = 0;
#pragma unroll
    for (int k = 1; k < 5; ++k) s += b >= M.first[k];
    BlockArgs A = M.A;
    A.items = M.items[s]; A.ftw = M.ftw[s];
    const int it = b - M.first[s];
    switch (s) {
        case 0: blockzoom_body<128, 32, 16, 8, 1, LEAN, NOD>(A, sp, it, buf, spow, wrapf, bandW, bandD); break;
        case 1: blockzoom_body<256, 16, 16, 16, 1, LEAN, NOD>(A, sp, it, buf, spow, wrapf, bandW, bandD); break;
        case 2: blockzoom_body<512, 8, 8, 8, 8, LEAN, NOD>(A, sp, it, buf, spow, wrapf, bandW, bandD); break;
        case 3: blockzoom_body<1024, 4, 16, 8, 8, LEAN, NOD>(A, sp, it, buf, spow, wrapf, bandW, bandD); break;
        default: blockzoom_body<2048, 2, 16, 16, 8, LEAN, NOD>(A, sp, it, buf, spow, wrapf, bandW, bandD); break;
    }
}

// ======================================================================= float64 block rows
// The same block decomposition in double precision: 2048 points per 128-thread workgroup
// (32 KiB of LDS, 16 points per thread), twiddles and spectra in double, the exact
// (double) bin map for every point. No lean variant, no LDS band staging.
struct BlockArgs64 {
    const int4* items; const BlockRowDev* rows; const BlockClassDev* classes;
    const double* pbank; const double* pxi;
    const c64* ctw; const c64* ftw; const c64* xb;
    const double* row_scale;
    double* Wx; double* dWx; double* w; unsigned short* kidx;
    int64_t M, N, na, n_items;
    double inv_dt, gamma;
    int sig;
};

// (two wavefronts per SIMD: left alone the compiler takes 276-280 registers -- one wavefront per SIMD -- and config 5's
// block rows run at 10.5 ms instead of 8.1; at 256 it spills ~90 bytes per lane)
// NOD: Wx alone (a plain cwt), as for the float32 kernels: the derivative's transform and outputs compiled out.
template <int L, int G, int R1, int R2, int R3, bool NOD = false>
__global__ __launch_bounds__(FftGeom<double>::NT) SSQ_WAVES_PER_EU(2, 2) void blockzoom_f64_kernel(BlockArgs64 A, SsqParams sp) {
    constexpr int NT64 = FftGeom<double>::NT;
    __shared__ c64 buf[FftGeom<double>::D];
    __shared__ c64 spow[R1 * G];
    __shared__ c64 wrapf[G];
    constexpr int RL = (R3 > 1) ? R3 : R2;
    const int tid = threadIdx.x;
    const int4 item = A.items[blockIdx.x];
    const int row = item.x, blk = item.y, c0 = item.z;
    const BlockRowDev r = A.rows[row];
    const BlockClassDev cl = A.classes[item.w];
    const int P = (int)cl.P, Rp = P / L;
    const int sig = A.sig + (int)blockIdx.y;
    const c64* xb = A.xb + cl.xb_off + ((int64_t)sig * cl.nb + blk) * cl.xb_stride;
    const c64* ctw = A.ctw + cl.ctw_off;
    const double* psi = A.pbank + r.pb_off;
    const double* pxi = A.pxi + r.pb_off;
    const double invP = 1.0 / (double)P;
    // (round 6, as in the float32 kernels since round 5: the band's three loads per point are issued BEFORE the barrier
    // behind the staged twiddles -- two dependent round trips become one; config 5 "r6y21")
    constexpr int NBP = PPT / R1, STRP = L / R1;
    double bp[PPT], bm[PPT]; c64 bX[PPT];
#pragma unroll
    for (int it = 0; it < NBP; ++it) {
        const int u = (tid + it * NT64) / G;
        const int off0 = (u - r.klo) & (L - 1);
#pragma unroll
        for (int k = 0; k < R1; ++k) {
            const int off = (off0 + k * STRP) & (L - 1);
            const int oc = off < r.KP ? off : 0;             // unconditional, clamped loads
            bp[it * R1 + k] = psi[oc]; bX[it * R1 + k] = xb[r.klo + oc]; bm[it * R1 + k] = pxi[oc];
        }
    }
    {
        constexpr int STR = L / R1;
        for (int i = tid; i < R1 * G; i += NT64) {
            const unsigned t = (unsigned)(i / G), col = (unsigned)(c0 + i % G);
            spow[i] = ctw[(t * (unsigned)STR * col) & (unsigned)(P - 1)];
        }
        if (tid < G) {
            const unsigned col = (unsigned)(c0 + tid);
            wrapf[tid] = ctw[(0u - (unsigned)L * col) & (unsigned)(P - 1)];
        }
    }
    c64 cw0s[NBP];
#pragma unroll
    for (int it = 0; it < NBP; ++it) {
        const int idx = tid + it * NT64, g = idx % G, u = idx / G;
        const int off0 = (u - r.klo) & (L - 1);
        cw0s[it] = ctw[((unsigned)(r.klo + off0) * (unsigned)(c0 + g)) & (unsigned)(P - 1)];
    }
    __syncthreads();
    c64 zw[PPT], zd[PPT];
    {
        constexpr int NB = PPT / R1, STR = L / R1;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT64, g = idx % G, u = idx / G;
            const int off0 = (u - r.klo) & (L - 1);
            const c64 cw0 = cw0s[it];
            const c64 wf = wrapf[g];
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int offu = off0 + k * STR, off = offu & (L - 1);
                const bool in = off < r.KP;
                const double p = bp[it * R1 + k] * invP;
                const c64 X = bX[it * R1 + k];
                const double mm = bm[it * R1 + k] * A.inv_dt;
                c64 cw = (k == 0) ? cw0 : cmul(cw0, spow[k * G + g]);
                if (offu >= L) cw = cmul_v(cw, wf);
                const c64 bz = {p * X.x, p * X.y};
                const c64 zz = cmul_v(bz, cw);
                c64 z = {0.0, 0.0}, dz = {0.0, 0.0};
                if (in) { z = zz; dz = {-(zz.y * mm), zz.x * mm}; }
                zw[it * R1 + k] = z; zd[it * R1 + k] = dz;
            }
        }
    }
    lds_ifft<L, G, R1, R2, R3, false, true>(zw, buf, A.ftw, tid);      // (the buffer is untouched so far: no barrier in front)
    if constexpr (!NOD) lds_ifft<L, G, R1, R2, R3>(zd, buf, A.ftw, tid);

    constexpr int NB = PPT / RL, STR = L / RL;
    const int64_t base = ((int64_t)sig * A.na + row) * A.N;
    double2* Wo = reinterpret_cast<double2*>(A.Wx) + base;
    double2* Do = A.dWx ? reinterpret_cast<double2*>(A.dWx) + base : nullptr;
    double* wo = A.w ? A.w + base : nullptr;
    unsigned short* ko = A.kidx ? A.kidx + ((int64_t)blockIdx.y * A.na + row) * A.N : nullptr;
    const double rs = A.row_scale ? A.row_scale[row] : 1.0;
    const int64_t omax = A.na - 1;
    const int m = (int)cl.m, N = (int)A.N;
    const unsigned span = (unsigned)max(0, min((int)cl.V, N - blk * (int)cl.V));
#pragma unroll
    for (int it = 0; it < NB; ++it) {
        const int idx = tid + it * NT64, g = idx % G, u = idx / G;
#pragma unroll
        for (int k = 0; k < RL; ++k) {
            const int dcol = (u + k * STR) * Rp + g + (c0 - m);
            if ((unsigned)dcol >= span) continue;
            const int j = dcol + blk * (int)cl.V;
            const c64 W = zw[it * RL + k], D = NOD ? c64{0.0, 0.0} : zd[it * RL + k];
            const double c = W.x * rs, d = W.y * rs, a = D.x * rs, b = D.y * rs;
            Wo[j] = make_double2(c, d);
            if constexpr (NOD) continue;
            if (Do) Do[j] = make_double2(a, b);
            if (wo) wo[j] = mag_lt(c, d, A.gamma) ? (double)INFINITY : fabs(phase_ratio(a, b, c, d));
            if (ko) {
                unsigned short kk = 0xFFFFu;
                if (mag_gt(c, d, A.gamma)) {
                    const int64_t kb = bin_of_point(a, b, c, d, false, 0.0, sp, omax);
                    kk = (unsigned short)(sp.flipud ? omax - kb : kb);
                }
                ko[j] = kk;
            }
        }
    }
}

template <int L, int G, int R1, int R2, int R3>
static int launch_zoom64(const BlockArgs64& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    if (A.n_items == 0) return 0;
    if (!A.dWx && !A.w && !A.kidx)                 // Wx alone
        hipLaunchKernelGGL((blockzoom_f64_kernel<L, G, R1, R2, R3, true>), dim3((unsigned)A.n_items, (unsigned)nsig),
                           dim3(FftGeom<double>::NT), 0, stream, A, sp);
    else
    hipLaunchKernelGGL((blockzoom_f64_kernel<L, G, R1, R2, R3>), dim3((unsigned)A.n_items, (unsigned)nsig),
                       dim3(FftGeom<double>::NT), 0, stream, A, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}

int BlockPlan::run64(int sig, int nsig, double* Wx, double* dWx, double* w, unsigned short* kidx,
                     const double* row_scale, double dt, const SsqParams& sp, hipStream_t stream) {
    BlockArgs64 A;
    A.rows = rows; A.classes = classes; A.pbank = (const double*)pbank; A.pxi = (const double*)pxi;
    A.ctw = (const c64*)ctw; A.xb = (const c64*)xb; A.row_scale = row_scale;
    A.Wx = Wx; A.dWx = dWx; A.w = w; A.kidx = kidx;
    A.M = M; A.N = N; A.na = na;
    A.inv_dt = 1.0 / dt; A.gamma = sp.gamma; A.sig = sig;
    int rc = 0;
#define ZOOM64(slot, L, G, R1, R2, R3)                                                            \
    A.items = (const int4*)items[slot]; A.n_items = n_items[slot];                                 \
    A.ftw = (const c64*)ftw + ftw_off[slot];                                                       \
    if ((rc = launch_zoom64<L, G, R1, R2, R3>(A, sp, nsig, stream))) return rc;
    ZOOM64(0, 128, 16, 16, 8, 1)
    ZOOM64(1, 256, 8, 16, 16, 1)
    ZOOM64(2, 512, 4, 8, 8, 8)
    ZOOM64(3, 1024, 2, 16, 8, 8)
    ZOOM64(4, 2048, 1, 16, 16, 8)
#undef ZOOM64
    return 0;
}

// ======================================================================= exact rows
// Rows whose impulse response is not compact (pass-band cut by the Nyquist frequency)
// get the reference's full-length transform, as a four-step FFT over M = A x B with
// one intermediate array Z in HBM (the only one on the fast path; ~4 MB per row):
//   pass 1: for every k1 < A, B-point iFFT over k2 of X[k1 + A k2] = psih[k] xh[k]
//           (and of the derivative spectrum), times e^{2 pi i k1 n2 / M} / M. The
//           twiddle is the product of two small per-workgroup LDS tables
//           (n2 = 32 n2_hi + n2_lo), not a gather from the M-entry table per point.
//   pass 2: for every n2 < B, A-point iFFT over k1 of Z[k1][n2] -> y[n2 + B n1],
//           fused with the same unpad / phase / bin-map epilogue as the block kernel.
// Z is stored blocked for pass 2: [n2 / G2][k1][n2 % G2] with G2 the number of n2
// columns a pass-2 workgroup owns, so that a pass-2 workgroup reads one contiguous
// A*G2*8-byte slab and pass 1 (which transposes through LDS) writes G1*G2*8-byte runs.
struct ExactArgs {
    const float* bank; const int64_t* band_off; const int32_t* band_lo;
    const int32_t* rows; const c32* xh; c32* Z; const c32* twM; const c32* ftw;
    const float* row_scale;
    float* Wx; float* dWx; float* w; unsigned short* kidx;
    int64_t M, N, na; int A, B, n1pad, sig, n_rows;   // sig: first signal (blockIdx.z adds)
    int G2;                                           // n2 columns per pass-2 workgroup
    double h; float inv_dt; double gamma;
};

template <int L, int G, int R1, int R2, int R3>
__global__ __launch_bounds__(NT) void exact_pass1_kernel(ExactArgs E) {
    __shared__ c32 buf[D_POINTS + 64];
    constexpr int RL = (R3 > 1) ? R3 : R2;
    const int tid = threadIdx.x, r = blockIdx.y, row = E.rows[r];
    const int c0 = blockIdx.x * G;                          // first k1 of this workgroup
    const int lo = E.band_lo[row];
    const int64_t off0 = E.band_off[row];
    const int len = (int)(E.band_off[row + 1] - off0);
    const float* psi = E.bank + off0;
    const c32* xh = E.xh + (int64_t)(E.sig + (int)blockIdx.z) * (E.M / 2 + 1);
    c32 zw[PPT], zd[PPT];
    {
        constexpr int NB = PPT / R1, STR = L / R1;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int kk = (c0 + g) + E.A * (u + k * STR);       // DFT bin
                const int off = kk - lo;
                c32 z = {0.f, 0.f}, dz = {0.f, 0.f};
                if (off >= 0 && off < len) {
                    const float p = psi[off];
                    const c32 X = xh[kk];
                    z = {p * X.x, p * X.y};
                    const float mm = (float)((double)kk * E.h) * E.inv_dt;
                    dz = {-(z.y * mm), z.x * mm};
                }
                zw[it * R1 + k] = z; zd[it * R1 + k] = dz;
            }
        }
    }
    // twiddle tables of this workgroup's G values of k1: e^{2 pi i k1 n2 / M} / M =
    // thi[g][n2 >> 5] * tlo[g][n2 & 31]
    constexpr int NH = (L + 31) / 32;
    __shared__ c32 tlo[G * 32], thi[G * NH];
    {
        const float fM = (float)E.M;                          // power of two: exact
        for (int i = tid; i < G * 32; i += NT)
            tlo[i] = E.twM[(unsigned)(c0 + i / 32) * (unsigned)(i % 32)];
        for (int i = tid; i < G * NH; i += NT) {
            const c32 t = E.twM[(unsigned)(c0 + i / NH) * (unsigned)((i % NH) * 32)];   // < M
            thi[i] = {t.x * fM, t.y * fM};
        }
    }
    constexpr int NBL = PPT / RL, STRL = L / RL;
    const int G2 = E.G2, lg2 = __ffs(G2) - 1;
    constexpr int LG = (G == 1) ? 0 : (G == 2) ? 1 : (G == 4) ? 2 : (G == 8) ? 3 : (G == 16) ? 4 : 5;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr) {
        c32 (&v)[PPT] = tr ? zd : zw;
        lds_ifft<L, G, R1, R2, R3>(v, buf, E.ftw, tid);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NBL; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const int n2 = u + k * STRL;
                const c32 tw = cmul_v(thi[g * NH + (n2 >> 5)], tlo[g * 32 + (n2 & 31)]);
                buf[g * (L + 1) + n2] = cmul_v(v[it * RL + k], tw);
            }
        }
        __syncthreads();
        c32* Zt = E.Z + (((int64_t)blockIdx.z * E.n_rows + r) * 2 + tr) * E.M;
#pragma unroll
        for (int it = 0; it < PPT; ++it) {
            // consecutive lanes: n2 % G2 fastest, then this workgroup's k1 -> G*G2-element runs
            // (G2 and G are powers of two: shifts, not divisions)
            const int idx = tid + it * NT, n2i = idx & (G2 - 1), g = (idx >> lg2) & (G - 1);
            const int n2t = idx >> (lg2 + LG), n2 = (n2t << lg2) + n2i;
            Zt[((int64_t)n2t * E.A + (c0 + g)) * G2 + n2i] = buf[g * (L + 1) + n2];
        }
    }
}

template <int L, int G, int R1, int R2, int R3, bool LEAN>
__global__ __launch_bounds__(NT) void exact_pass2_kernel(ExactArgs E, SsqParams sp) {
    __shared__ c32 buf[D_POINTS];
    constexpr int RL = (R3 > 1) ? R3 : R2;
    const int tid = threadIdx.x, r = blockIdx.y, row = E.rows[r];
    // a workgroup writes G consecutive outputs per n1 (G*8-byte pieces of Wx, G*2 of the bin map):
    // give each XCD (workgroup b -> XCD b % 8) a contiguous range of n2 so the pieces of one
    // line meet in one L2
    const int bx = (gridDim.x & 7) ? (int)blockIdx.x
                                   : (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    const int c0 = bx * G;                                  // first n2 of this workgroup
    const c32* ZW = E.Z + (((int64_t)blockIdx.z * E.n_rows + r) * 2) * E.M;
    const c32* ZD = ZW + E.M;
    c32 zw[PPT], zd[PPT];
    {
        constexpr int NB = PPT / R1, STR = L / R1;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int64_t q = (int64_t)bx * E.A * G + (u + k * STR) * G + g;   // blocked Z
                zw[it * R1 + k] = ZW[q]; zd[it * R1 + k] = ZD[q];
            }
        }
    }
    lds_ifft<L, G, R1, R2, R3>(zw, buf, E.ftw, tid);
    lds_ifft<L, G, R1, R2, R3>(zd, buf, E.ftw, tid);
    constexpr int NB = PPT / RL, STR = L / RL;
    const EmitRow er = make_emit_row(E.Wx, E.dWx, E.w, E.kidx, E.row_scale, E.sig + (int)blockIdx.z, (int)blockIdx.z, row, E.na, E.N, E.gamma, sp.flipud);
    const int N = (int)E.N;
    unsigned pend = 0;
    auto emit_all = [&](auto grid_tag) {
        constexpr int GRID = decltype(grid_tag)::value;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const int n = (c0 + g) + E.B * (u + k * STR);
                const int j = n - E.n1pad;
                if ((unsigned)j >= (unsigned)N) continue;
                if (emit_point<LEAN, GRID>(er, j, zw[it * RL + k], zd[it * RL + k], sp))
                    pend |= 1u << (it * RL + k);
            }
        }
    };
    dispatch_grid<LEAN>(sp.grid, emit_all);
    while (__builtin_amdgcn_ballot_w64(pend != 0)) {
        const unsigned low = pend & (0u - pend);
        pend ^= low;
        c32 W = {0.f, 0.f}, D = {0.f, 0.f};
        int j = 0;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < RL; ++k)
                if (low == (1u << (it * RL + k))) {
                    W = zw[it * RL + k]; D = zd[it * RL + k]; j = (c0 + g) + E.B * (u + k * STR) - E.n1pad;
                }
        }
        if (low) emit_point_exact(er, er.k + j, W, D, sp);
    }
}

// gather the overlapping blocks of the (periodic) padded signals, all classes in one
// launch (blockIdx.y = class)
// T: the element gathered -- a real sample of the padded signal (classes over x, into `blocks`)
// or a complex sample of its analytic signal (classes with analytic = 1, straight into their
// slots of `xb`, transformed in place)
template <typename T, bool ANALYTIC>
__global__ __launch_bounds__(256) void gather_blocks_kernel(const T* __restrict__ xp,
                                                            T* __restrict__ dst,
                                                            const BlockClassDev* __restrict__ classes,
                                                            int64_t M, int64_t n1, int64_t batch) {
    const BlockClassDev k = classes[blockIdx.y];
    const int64_t total = batch * k.nb * k.P;
    const int64_t lead = (k.P == M) ? 0 : n1 - k.m;       // the single-block class starts at 0
    T* out = dst + (ANALYTIC ? k.xb_off : k.blk_off);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = t % k.P, bb = t / k.P, b = bb % k.nb, s = bb / k.nb;
        int64_t src = (lead + b * k.V + p) % M;
        if (src < 0) src += M;
        out[t] = xp[s * M + src];
    }
}

// One-sided spectrum of the analytic signal: X_a[k] = xh[k] for k < M / 2, half of it at the
// Nyquist bin (the bank's halving of that bin, wavelets.py:86-95, moved to the signal -- exact),
// zero above. The inverse transform of it follows in place.
template <typename C>
__global__ __launch_bounds__(256) void analytic_spectrum_kernel(const C* __restrict__ xh, C* __restrict__ xa,
                                                                int64_t M, int64_t batch) {
    const int64_t half = M / 2, total = batch * M;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t % M, s = t / M;
        C v; v.x = 0; v.y = 0;
        if (k <= half) {
            v = xh[s * (half + 1) + k];
            if (k == half) { v.x *= 0.5f; v.y *= 0.5f; }      // (exact in either precision)
        }
        xa[t] = v;
    }
}


// ---- block spectra of the P = 4096 classes in ONE launch (float32, round 5) ----------------------------------
// Rounds 1-4: gather kernel per kind -> batched rocFFT (real-to-complex for the blocks of x: two kernels; complex for
// the blocks of the analytic signal) per block length: at config 2 -- where every class has P = 4096 -- five launches
// of 6-30 us per launch group (5 us of 327 per transform at 16 signals, 32 of 430 for a single one). Here a workgroup
// takes TWO consecutive real blocks a, b of a signal as one complex sequence a + ib (or one analytic block), gathers
// them from the periodic padded signal, runs the 4096-point LDS transform of ssq_ldsfft.h (radices 16 x 16 x 16;
// twiddles: the class' own e^{2 pi i q / P} table) and writes the spectra the block kernels read:
//   Z' = sum_p z[p] e^{+2 pi i k p / P} = Z[P - k]  (the transform at hand is the inverse one, unnormalised), so
//   X_a[k] = (Z'[P-k] + conj Z'[k]) / 2,   X_b[k] = (Z'[P-k] - conj Z'[k]) / 2i,   k <= P / 2;   analytic: X[k] = Z'[P-k].
struct BlockSpecArgs {
    const float* xp; const c32* xa; c32* xb;
    const c32* xh;                     // the padded signals' half spectra (block_spectra_multi_kernel: a class with P = M)
    const BlockClassDev* classes; const c32* ctw;
    int64_t M, n1;
    int cls[8]; int first[9];          // classes served, first workgroup of each (first[n]: the grid's x size)
    int ncls;
};
__global__ __launch_bounds__(NT) void block_spectra4096_kernel(BlockSpecArgs A) {
    constexpr int P = 4096;
    __shared__ c32 buf[P];
    const int tid = threadIdx.x;
    int b = (int)blockIdx.x, c = 0;
    while (c + 1 < A.ncls && b >= A.first[c + 1]) ++c;
    b -= A.first[c];
    const BlockClassDev k = A.classes[A.cls[c]];
    const int64_t sig = blockIdx.y, M = A.M;
    const int64_t lead = A.n1 - k.m;
    c32 z[PPT];
    const bool ana = k.analytic != 0;
    const int b0 = ana ? b : 2 * b, b1 = b0 + 1;                 // blocks of this workgroup
    const bool two = !ana && b1 < (int)k.nb;
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        const int p = tid + t * NT;
        int64_t s0 = (lead + (int64_t)b0 * k.V + p) % M; if (s0 < 0) s0 += M;
        if (ana) z[t] = A.xa[sig * M + s0];
        else {
            int64_t s1 = (lead + (int64_t)b1 * k.V + p) % M; if (s1 < 0) s1 += M;
            z[t] = {A.xp[sig * M + s0], two ? A.xp[sig * M + s1] : 0.f};
        }
    }
    lds_ifft<P, 1, 16, 16, 16>(z, buf, A.ctw + k.ctw_off, tid);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < PPT; ++t) buf[tid + t * NT] = z[t];    // natural order: Z'[tid + 256 t]
    __syncthreads();
    c32* out0 = A.xb + k.xb_off + (sig * k.nb + b0) * k.xb_stride;
    if (ana) {
#pragma unroll
        for (int t = 0; t < PPT; ++t) { const int f = tid + t * NT; out0[f] = buf[(P - f) & (P - 1)]; }
    } else {
        c32* out1 = out0 + k.xb_stride;
        for (int f = tid; f <= P / 2; f += NT) {
            const c32 Pk = buf[f], Qk = buf[(P - f) & (P - 1)];
            out0[f] = {0.5f * (Qk.x + Pk.x), 0.5f * (Qk.y - Pk.y)};
            if (two) out1[f] = {0.5f * (Qk.y + Pk.y), 0.5f * (Pk.x - Qk.x)};
        }
    }
}

// ---- block spectra of the P = 4096 / 8192 / 16384 classes in ONE launch (float32; small calls, round 6) ---------
// A short signal's plan (config 1: N = 10 000, M = 16 384) has classes of P = 8192 and 16 384 next to the P = 4096
// ones, and their spectra went gather kernel -> rocFFT's two-kernel transforms -> real-to-complex post-processing, six
// launches of 3-7 us in a row on a call of 0.1 ms. Here one launch serves every class. A workgroup has 256 QMAX
// threads (QMAX = the longest block / 4096 = 2 or 4) and runs CW = 4 QMAX columns of 1024 points through lds_ifft at
// once (16 points per thread as everywhere; measured no faster than QMAX columns of 4096 points, whose passes' LDS
// accesses conflict far more -- "r7c" / "r7d" of profiles/r6_ab_history.txt: the banks are not what bounds a launch of a
// few workgroups): a block of P = 1024 C points is its C interleaved sub-sequences y[j C + r]
// (the class' e^{2 pi i q / P} table read at every C-th entry), so a workgroup holds CW / C blocks ("slots": pairs of
// real blocks as one complex sequence, or one analytic block, as above), and the last radix-C step
//   Y'[f + 1024 s] = sum_r e^{2 pi i r f / P} e^{2 pi i r s / C} Z'_r[f]
// goes through LDS (the C terms of an f sit in C different lanes). LDS: 32 QMAX KB (static; gfx950: 160 KB).
// Latency, not throughput, is what this kernel is for: BlockPlan::spectra uses it when the launch is at most a couple
// of workgroups per CU and leaves long batches of P > 4096 blocks to gather + rocFFT.
constexpr int WIDE_L = 1024;
// the columns' transform: radices 16 x 8 x 8; entry z[k] = column g's point u + 64 k (g = tid % CW, u = tid / CW);
// exit z[8 it + k] = its output u' + 128 k, u' = (tid + it NTH) / CW. Its twiddles e^{2 pi i q / 1024} come from an
// 8 KB table in LDS (wide_stage_tw: every C-th entry of the class' table, one coalesced read at the kernel's start,
// beside the signal's loads, instead of two dependent table reads per transform: "r7e" -- worth 14 of 47 us to a kernel
// that ran two transforms in a row in one workgroup, nothing measurable to this one).
template <int CW, int C>
__device__ __forceinline__ void wide_stage_tw(c32* __restrict__ stw, const c32* __restrict__ tw, int tid) {
    for (int i = tid; i < WIDE_L; i += 64 * CW) stw[i] = tw[i * C];
}
template <int CW>
__device__ __forceinline__ void wide_ifft(c32 (&z)[PPT], c32* __restrict__ buf, const c32* __restrict__ stw, int tid) {
    lds_ifft<WIDE_L, CW, 16, 8, 8, false, false, 1, 64 * CW>(z, buf, stw, tid);   // (its first barrier: stw is in place)
}
// the last step's twiddles e^{2 pi i r f / P} of a thread's (slot, f) pairs: asked for at the kernel's start
template <int CW, int C>
__device__ __forceinline__ void wide_load_w(c32 (&w)[PPT / C][C], const c32* __restrict__ tw, int tid) {
#pragma unroll
    for (int i = 0; i < PPT / C; ++i) {
        const int f = (tid + i * 64 * CW) % WIDE_L;
        w[i][0] = {1.f, 0.f};
#pragma unroll
        for (int r = 1; r < C; ++r) w[i][r] = tw[r * f];
    }
}
// wide_natural: the transforms' outputs -> buf[slot P + n] = Y'_slot[n] in natural order, behind a barrier.
// In between, Z'_r[n] sits at [slot][r][(n + 4 r) mod 1024]: the rotation keeps both the writes (a wavefront's lanes
// differ in r first) and the reads (consecutive n) off each other's banks.
template <int CW, int C>
__device__ __forceinline__ void wide_natural(const c32 (&z)[PPT], c32* __restrict__ buf, const c32 (&w)[PPT / C][C], int tid) {
    constexpr int L = WIDE_L, P = L * C, NTH = 64 * CW;
    constexpr int NI = PPT / C;                                  // (slot, f) pairs per thread
    __syncthreads();                                             // (the last pass' reads)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * NTH, g = idx % CW, u = idx / CW, slot = g / C, r = g % C;
#pragma unroll
        for (int k = 0; k < 8; ++k) buf[slot * P + r * L + ((u + k * (L / 8) + 4 * r) & (L - 1))] = z[it * 8 + k];
    }
    __syncthreads();
    c32 y[NI][C];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * NTH, sl = e / L, f = e % L;
#pragma unroll
        for (int r = 0; r < C; ++r) {
            const c32 v = buf[sl * P + r * L + ((f + 4 * r) & (L - 1))];
            y[i][r] = r ? cmul_v(v, w[i][r]) : v;
        }
        Dft<C>::run(y[i]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * NTH, sl = e / L, f = e % L;
#pragma unroll
        for (int s2 = 0; s2 < C; ++s2) buf[sl * P + f + s2 * L] = y[i][s2];
    }
    __syncthreads();
}
template <int CW, int C>
__device__ __forceinline__ void block_spectra_wide(const BlockSpecArgs& A, const BlockClassDev& k, int b, c32* __restrict__ buf,
                                                   c32* __restrict__ stw) {
    constexpr int L = WIDE_L, P = L * C, NTH = 64 * CW, SL = CW / C;
    const int tid = threadIdx.x;
    const int g = tid % CW, u = tid / CW, slot = g / C, r = g % C;
    const int64_t sig = blockIdx.y, M = A.M;
    const bool ana = k.analytic != 0;
    const int nunits = ana ? (int)k.nb : ((int)k.nb + 1) / 2;     // a unit: one analytic block or a pair of real ones
    const c32* __restrict__ tw = A.ctw + k.ctw_off;              // e^{2 pi i q / P}
    const c32* __restrict__ xa = A.xa + sig * M;
    const float* __restrict__ xp = A.xp + sig * M;
    const int64_t lead = A.n1 - k.m;
    const int m32 = (int)M;
    wide_stage_tw<CW, C>(stw, tw, tid);
    c32 w[PPT / C][C];
    wide_load_w<CW, C>(w, tw, tid);
    c32 z[PPT];
    {
        const int unit = b * SL + slot;
        const bool live = unit < nunits;
        const int b0 = ana ? unit : 2 * unit, b1 = b0 + 1;
        const bool two = !ana && b1 < (int)k.nb;
        // first sample of each block in the periodic padded signal, reduced once (P <= M: one wrap at most per point)
        int64_t o0 = (lead + (int64_t)b0 * k.V) % M; if (o0 < 0) o0 += M;
        int64_t o1 = (lead + (int64_t)b1 * k.V) % M; if (o1 < 0) o1 += M;
        const int a0 = (int)o0, a1 = (int)o1;
#pragma unroll
        for (int t = 0; t < PPT; ++t) {
            const int p = (u + t * (L / 16)) * C + r;
            int s0 = a0 + p; if (s0 >= m32) s0 -= m32;
            int s1 = a1 + p; if (s1 >= m32) s1 -= m32;
            c32 v = {0.f, 0.f};
            if (live) {
                if (ana) v = xa[s0];
                else v = {xp[s0], two ? xp[s1] : 0.f};
            }
            z[t] = v;
        }
    }
    wide_ifft<CW>(z, buf, stw, tid);
    wide_natural<CW, C>(z, buf, w, tid);
    for (int sl = 0; sl < SL; ++sl) {
        const int unit = b * SL + sl;
        if (unit >= nunits) break;
        const int b0 = ana ? unit : 2 * unit;
        const bool two = !ana && b0 + 1 < (int)k.nb;
        const c32* __restrict__ Y = buf + sl * P;
        c32* out0 = A.xb + k.xb_off + (sig * k.nb + b0) * k.xb_stride;
        if (ana) {
            for (int f = tid; f < P; f += NTH) out0[f] = Y[(P - f) & (P - 1)];
        } else {
            c32* out1 = out0 + k.xb_stride;
            for (int f = tid; f <= P / 2; f += NTH) {
                const c32 Pk = Y[f], Qk = Y[(P - f) & (P - 1)];
                out0[f] = {0.5f * (Qk.x + Pk.x), 0.5f * (Qk.y - Pk.y)};
                if (two) out1[f] = {0.5f * (Qk.y + Pk.y), 0.5f * (Pk.x - Qk.x)};
            }
        }
    }
}
template <int QMAX>
__global__ __launch_bounds__(NT * QMAX) void block_spectra_multi_kernel(BlockSpecArgs A) {
    __shared__ c32 buf[4096 * QMAX];               // 64 / 128 KB
    __shared__ c32 stw[WIDE_L];
    int b = (int)blockIdx.x, c = 0;
    while (c + 1 < A.ncls && b >= A.first[c + 1]) ++c;
    b -= A.first[c];
    const BlockClassDev k = A.classes[A.cls[c]];
    if (k.P == A.M && k.nb == 1 && !k.analytic && A.xh) {
        // one block = the whole periodic signal from sample o on: its spectrum is the signal's own, already there,
        // turned: X_b[k] = xh[k] e^{2 pi i k o / M} -- no transform (config 1's widest class; such a class has one block,
        // and a 16 384-point transform in one workgroup was the launch's longest chain)
        const int64_t M = A.M, sig = blockIdx.y;
        int64_t o = (A.n1 - k.m) % M; if (o < 0) o += M;
        const c32* __restrict__ tw = A.ctw + k.ctw_off;          // e^{2 pi i q / M}
        const c32* __restrict__ xh = A.xh + sig * (M / 2 + 1);
        c32* __restrict__ out0 = A.xb + k.xb_off + sig * k.nb * k.xb_stride;
        const unsigned mask = (unsigned)M - 1u, ou = (unsigned)o;
        for (int f = (int)threadIdx.x; f <= (int)(M / 2); f += NT * QMAX)
            out0[f] = cmul_v(xh[f], tw[((unsigned)f * ou) & mask]);
        return;
    }
    if (k.P == 4096) block_spectra_wide<4 * QMAX, 4>(A, k, b, buf, stw);
    else if (k.P == 8192) block_spectra_wide<4 * QMAX, 8>(A, k, b, buf, stw);
    else if constexpr (QMAX >= 4) block_spectra_wide<4 * QMAX, 16>(A, k, b, buf, stw);
}

template <int L, int G, int R1, int R2, int R3>
static int launch_zoom(const BlockArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    if (A.n_items == 0) return 0;
    const dim3 grid((unsigned)A.n_items, (unsigned)nsig);
    if (A.kidx && !A.dWx && !A.w && !A.row_scale)
        hipLaunchKernelGGL((blockzoom_kernel<L, G, R1, R2, R3, true>), grid, dim3(NT), 0, stream, A, sp);
    else if (!A.kidx && !A.dWx && !A.w)            // Wx alone
        hipLaunchKernelGGL((blockzoom_kernel<L, G, R1, R2, R3, false, true>), grid, dim3(NT), 0, stream, A, sp);
    else
        hipLaunchKernelGGL((blockzoom_kernel<L, G, R1, R2, R3, false>), grid, dim3(NT), 0, stream, A, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}

int BlockPlan::create(const ssq_cwt_blocks_desc& d, int dtype_, int64_t M_, int64_t N_, int64_t n1_,
                      int64_t na_, int64_t max_batch_, int64_t& bytes) {
    M = M_; N = N_; n1 = n1_; na = na_; max_batch = max_batch_; dtype = dtype_;
    const size_t rs = dtype == SSQ_F32 ? 4 : 8;
    nc = d.n_classes;
    hcls.resize(nc);
    int64_t xb_total = 0, blk_max = 0;
    for (int c = 0; c < nc; ++c) {
        BlockClassDev& k = hcls[c];
        k.P = d.classes[5 * c]; k.m = d.classes[5 * c + 1]; k.V = d.classes[5 * c + 2];
        k.nb = d.classes[5 * c + 3];
        k.analytic = d.classes[5 * c + 4] ? 1 : 0;
        SSQ_REQUIRE(c == 0 || hcls[c - 1].analytic <= k.analytic, "block classes: analytic classes must come last");
        k.ctw_off = d.ctw_off[c];
        k.xb_stride = k.analytic ? k.P : k.P / 2 + 1;
        k.xb_off = xb_total;
        xb_total += max_batch * k.nb * k.xb_stride;
        k.blk_off = blk_max;
        if (!k.analytic) blk_max += max_batch * k.nb * k.P;
        n_analytic += (int)k.analytic;
    }
    auto up = [&](void** dst, const void* src, size_t nbytes) -> int {
        SSQ_CHECK_HIP(hipMalloc(dst, nbytes ? nbytes : 1));
        if (nbytes) SSQ_CHECK_HIP(hipMemcpy(*dst, src, nbytes, hipMemcpyHostToDevice));
        bytes += (int64_t)nbytes;
        return 0;
    };
    int rc;
    if ((rc = up((void**)&classes, hcls.data(), sizeof(BlockClassDev) * nc))) return rc;
    static_assert(sizeof(BlockRowDev) == 6 * sizeof(int32_t), "row table layout");
    if ((rc = up((void**)&rows, d.rows, sizeof(BlockRowDev) * na))) return rc;
    if ((rc = up((void**)&pbank, d.pbank, rs * d.n_pbank))) return rc;
    if ((rc = up((void**)&pxi, d.pxi, rs * d.n_pbank))) return rc;
    if ((rc = up((void**)&ctw, d.ctw, 2 * rs * (size_t)d.ctw_off[nc]))) return rc;
    if ((rc = up((void**)&ftw, d.ftw, 2 * rs * (size_t)d.n_ftw))) return rc;
    for (int s = 0; s < 5; ++s) {
        n_items[s] = d.n_items[s];
        ftw_off[s] = d.ftw_off[s];
        if ((rc = up((void**)&items[s], d.items[s], sizeof(int4) * (size_t)d.n_items[s]))) return rc;
        if (d.n_items[s]) h_items[s].assign(d.items[s], d.items[s] + 4 * d.n_items[s]);
    }
    SSQ_CHECK_HIP(hipMalloc((void**)&xb, 2 * rs * (size_t)xb_total)); bytes += 2 * rs * xb_total;
    SSQ_CHECK_HIP(hipMalloc((void**)&blocks, rs * (size_t)blk_max)); bytes += rs * blk_max;
    // classes of equal block length and kind are contiguous in `blocks` and `xb`: one batched
    // real-to-complex transform per length (complex, in place in `xb`, for the analytic classes)
    for (int c = 0; c < nc;) {
        int e = c;
        int64_t nblocks = 0;
        while (e < nc && hcls[e].P == hcls[c].P && hcls[e].analytic == hcls[c].analytic)
            nblocks += max_batch * hcls[e++].nb;
        FftPlan fp;
        rc = fp.create(hcls[c].analytic ? 2 : 0, dtype, (size_t)hcls[c].P, (size_t)nblocks, 1.0);
        if (rc) return rc;
        bytes += (int64_t)fp.work_bytes;
        ffts.push_back(fp); fft_first.push_back(c);
        c = e;
    }
    if (n_analytic) {
        SSQ_CHECK_HIP(hipMalloc(&xa, 2 * rs * (size_t)(max_batch * M))); bytes += 2 * rs * max_batch * M;
        if (AnalyticFft::supports(dtype, M)) {
            ana = new AnalyticFft();
            if ((rc = ana->create(M, max_batch, bytes))) return rc;
        } else {
            rc = inv_m.create(1, dtype, (size_t)M, (size_t)max_batch, 1.0 / (double)M);
            if (rc) return rc;
            bytes += (int64_t)inv_m.work_bytes;
        }
    }
    n_generic = d.n_generic;
    // (SSQ_DEBUG_BLOCK_SPECTRA=rocfft: every class through gather + rocFFT, as in rounds 1-4)
    own4096 = dtype == SSQ_F32 && M > 4096 && !(getenv("SSQ_DEBUG_BLOCK_SPECTRA") && !strcmp(getenv("SSQ_DEBUG_BLOCK_SPECTRA"), "rocfft"));
    // (SSQ_DEBUG_BLOCK_SPECTRA=4096: the one-launch kernel for the P = 4096 classes only, as until round 6's second half)
    own_big = own4096 && !(getenv("SSQ_DEBUG_BLOCK_SPECTRA") && !strcmp(getenv("SSQ_DEBUG_BLOCK_SPECTRA"), "4096"));
    {   // the CU count of the device the plan lives on (the multi-class launch asks how full a launch is)
        int dev = 0; hipDeviceProp_t pr;
        bool lds128 = false;                       // (block_spectra_multi_kernel<4>: 128 KB of LDS; gfx950 has 160 KB)
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) {
            ncu = pr.multiProcessorCount;
            lds128 = (size_t)pr.maxSharedMemoryPerMultiProcessor >= 128 * 1024;
        }
        own_big = own_big && lds128;
    }
    return 0;
}

void BlockPlan::destroy() {
    for (auto& f : ffts) f.destroy();
    inv_m.destroy();
    if (ana) { ana->destroy(); delete ana; ana = nullptr; }
    void* ptrs[] = {xa, twM, zbuf, classes, rows, pbank, pxi, ctw, ftw, xb, blocks, items[0], items[1], items[2],
                    items[3], items[4]};
    for (void* p : ptrs) if (p) (void)hipFree(p);
}

int BlockPlan::spectra(const void* xp, const void* xh, int64_t batch, hipStream_t stream,
                       const unsigned char* need) {
    (void)batch;                                   // planned batch: stale rows are ignored later
    // classes are gathered and transformed in runs of equal block length and kind; with `need`,
    // only the runs that hold a needed class (class indices are contiguous per run). The real
    // classes come first, the analytic ones last: one gather launch per kind. The P = 4096 classes of a
    // float32 plan -- the first ones of each kind -- take block_spectra4096_kernel instead (own4096).
    std::vector<char> run_on(ffts.size(), 1);
    int lo[2] = {nc, nc}, hi[2] = {-1, -1};        // needed class range per kind (rocFFT route)
    bool any_analytic = false;
    BlockSpecArgs S;
    S.ncls = 0; S.first[0] = 0;
    int qmax = 1;                                  // longest block the own kernel serves, in units of 4096
    bool prefix[2] = {true, true};
    for (size_t f = 0; f < ffts.size(); ++f) {
        const int a = fft_first[f], b = f + 1 < ffts.size() ? fft_first[f + 1] : nc;
        bool on = !need;
        for (int c = a; c < b && !on; ++c) on = need[c] != 0;
        const int kind = (int)hcls[a].analytic;
        any_analytic = any_analytic || (on && kind);
        int wanted = 0;                            // (classes of this group the kernel's 8 slots would have to take)
        int64_t group_wgs = 0;
        for (int c = a; c < b; ++c) {
            if (need && !need[c]) continue;
            ++wanted;
            group_wgs += kind ? hcls[c].nb : (hcls[c].nb + 1) / 2;
        }
        // P = 4096: always the library's own kernel; P = 8192 / 16384: the one-launch kernel when the launch is small
        // (a call bound by its launches), gather + rocFFT for long batches of such blocks
        const int64_t Pg = hcls[a].P;
        bool whole = Pg == M && !kind && xh;       // blocks as long as the signal, one per signal: no transform at all
        for (int c = a; c < b && whole; ++c) whole = hcls[c].nb == 1;
        const bool small = own_big && M < ((int64_t)1 << 30) && group_wgs * max_batch <= 2 * (int64_t)ncu &&
                           (whole || ((Pg == 8192 || Pg == 16384) && Pg <= M));
        // (the classes the kernel takes are a prefix of their kind: the gather launch below covers ONE class range per kind,
        // and for the analytic kind it writes into the spectra's own storage)
        if (on && own4096 && prefix[kind] && (Pg == 4096 || small) && S.ncls + wanted <= 8) {     // (a ninth: gather + rocFFT, as for other P)
            for (int c = a; c < b; ++c) {
                if (need && !need[c]) continue;
                S.cls[S.ncls++] = c;
                // (a single block as long as the signal takes no transform -- the kernel turns the signal's own spectrum --
                // and does not ask for the wider workgroup)
                const bool twist = Pg == M && hcls[c].nb == 1 && !kind && xh;
                qmax = std::max(qmax, twist ? 2 : (int)(Pg / 4096));      // (the P = 4096-only kernel knows neither)
            }
            on = false;                            // (not through gather + rocFFT)
        }
        run_on[f] = on;
        if (on) prefix[kind] = false;
        if (on) { lo[kind] = std::min(lo[kind], a); hi[kind] = std::max(hi[kind], b - 1); }
    }
    const size_t rs = dtype == SSQ_F32 ? 4 : 8;
    if (any_analytic) {
        // the analytic signal of every padded signal first
        SSQ_REQUIRE(xh && xa, "analytic block classes need the half spectrum");
        if (ana) {
            int rc = ana->run(xh, xa, max_batch, stream);
            if (rc) return rc;
        } else {
            dim3 ga((unsigned)std::min<int64_t>((max_batch * M + 255) / 256, 4096));
            if (dtype == SSQ_F32)
                hipLaunchKernelGGL(analytic_spectrum_kernel<c32>, ga, dim3(256), 0, stream, (const c32*)xh, (c32*)xa, M, max_batch);
            else
                hipLaunchKernelGGL(analytic_spectrum_kernel<c64>, ga, dim3(256), 0, stream, (const c64*)xh, (c64*)xa, M, max_batch);
            SSQ_LAUNCH_CHECK();
            int rc = inv_m.execute(xa, nullptr, stream);
            if (rc) return rc;
        }
    }
    if (S.ncls) {
        S.xp = (const float*)xp; S.xa = (const c32*)xa; S.xb = (c32*)xb; S.xh = (const c32*)xh;
        S.classes = classes; S.ctw = (const c32*)ctw; S.M = M; S.n1 = n1;
        // a workgroup of the wide kernel holds qmax / (P / 4096) units of a class (a unit: a pair of real blocks or one
        // analytic block); the P = 4096-only kernel one
        for (int i = 0; i < S.ncls; ++i) {
            const BlockClassDev& k = hcls[S.cls[i]];
            const int64_t units = k.analytic ? k.nb : (k.nb + 1) / 2;
            const int64_t per = std::max<int64_t>(1, qmax / (k.P / 4096));
            S.first[i + 1] = S.first[i] + (int)((units + per - 1) / per);
        }
        const dim3 sg((unsigned)S.first[S.ncls], (unsigned)max_batch);
        if (qmax == 1)
            hipLaunchKernelGGL(block_spectra4096_kernel, sg, dim3(NT), 0, stream, S);
        else if (qmax == 2)
            hipLaunchKernelGGL(block_spectra_multi_kernel<2>, sg, dim3(NT * 2), 0, stream, S);
        else
            hipLaunchKernelGGL(block_spectra_multi_kernel<4>, sg, dim3(NT * 4), 0, stream, S);
        SSQ_LAUNCH_CHECK();
    }
    for (int kind = 0; kind < 2; ++kind) {
        if (hi[kind] < lo[kind]) continue;
        int64_t most = 0;
        for (int c = lo[kind]; c <= hi[kind]; ++c) most = std::max<int64_t>(most, max_batch * hcls[c].nb * hcls[c].P);
        dim3 grid((unsigned)std::min<int64_t>((most + 255) / 256, 2048), (unsigned)(hi[kind] - lo[kind] + 1));
        if (kind == 0) {
            if (dtype == SSQ_F32)
                hipLaunchKernelGGL((gather_blocks_kernel<float, false>), grid, dim3(256), 0, stream, (const float*)xp,
                                   (float*)blocks, classes + lo[0], M, n1, max_batch);
            else
                hipLaunchKernelGGL((gather_blocks_kernel<double, false>), grid, dim3(256), 0, stream, (const double*)xp,
                                   (double*)blocks, classes + lo[0], M, n1, max_batch);
            SSQ_LAUNCH_CHECK();
        } else {
            if (dtype == SSQ_F32)
                hipLaunchKernelGGL((gather_blocks_kernel<c32, true>), grid, dim3(256), 0, stream, (const c32*)xa,
                                   (c32*)xb, classes + lo[1], M, n1, max_batch);
            else
                hipLaunchKernelGGL((gather_blocks_kernel<c64, true>), grid, dim3(256), 0, stream, (const c64*)xa,
                                   (c64*)xb, classes + lo[1], M, n1, max_batch);
            SSQ_LAUNCH_CHECK();
        }
    }
    for (size_t f = 0; f < ffts.size(); ++f) {
        if (!run_on[f]) continue;
        const BlockClassDev& k = hcls[fft_first[f]];
        int rc = k.analytic ? ffts[f].execute((char*)xb + (size_t)k.xb_off * 2 * rs, nullptr, stream)
                            : ffts[f].execute((char*)blocks + (size_t)k.blk_off * rs,
                                              (char*)xb + (size_t)k.xb_off * 2 * rs, stream);
        if (rc) return rc;
    }
    return 0;
}

int BlockPlan::run(int sig, int nsig, float* Wx, float* dWx, float* w, unsigned short* kidx,
                   const float* row_scale, double dt, const SsqParams& sp, hipStream_t stream,
                   const int64_t* limit) {
    BlockArgs A;
    A.rows = rows; A.classes = classes; A.pbank = (const float*)pbank; A.pxi = (const float*)pxi;
    A.ctw = (const c32*)ctw;
    A.xb = (const c32*)xb; A.row_scale = row_scale;
    A.Wx = Wx; A.dWx = dWx; A.w = w; A.kidx = kidx;
    A.M = M; A.N = N; A.na = na;
    A.h = (2.0 * 3.141592653589793) / (double)M;
    A.inv_dt = 1.0f / (float)dt;
    A.gamma = sp.gamma; A.sig = sig;
    int rc = 0;
    {   // A transform too small to fill the GPU class by class (C1: five launches of 10-18 us each): one launch.
        // (SSQ_DEBUG_CWT_BLOCKS_MULTI=0/1 forces; default: when no class has more than two workgroups per CU)
        const char* fe = getenv("SSQ_DEBUG_CWT_BLOCKS_MULTI");                 // (read per call: tests switch it)
        const int force = fe ? atoi(fe) : -1;
        int64_t total = 0, biggest = 0; int used = 0;
        for (int s = 0; s < 5; ++s) {
            const int64_t n = limit ? limit[s] : n_items[s];
            total += n; biggest = std::max(biggest, n * nsig); used += n > 0;
        }
        // (... or when the whole call is only a few rounds of workgroups -- three of them share a CU: a single signal
        // at config 2 is 1512 + 1344 workgroups, two rounds per class launched one after the other, 3.7 rounds
        // together: block stage 107 -> 89 us, round 5)
        // (... and the fused form's lean kernels always: at config 2, 16 signals per launch, the two large classes in
        // one launch take 58.5 us per transform against 64.0 one after the other -- one tail instead of two -- round 5)
        const bool lean = A.kidx && !A.dWx && !A.w && !A.row_scale;
        const bool multi = force >= 0 ? force != 0
                                      : (used > 1 && (lean || biggest <= 2 * (int64_t)ncu || total * nsig <= 18 * (int64_t)ncu));
        if (multi && total > 0) {
            BlockMultiArgs Mx;
            Mx.A = A; Mx.A.items = nullptr; Mx.A.n_items = total; Mx.A.ftw = nullptr;
            int acc = 0;
            for (int s = 0; s < 5; ++s) {
                Mx.items[s] = (const int4*)items[s]; Mx.ftw[s] = (const c32*)ftw + ftw_off[s];
                Mx.first[s] = acc; acc += (int)(limit ? limit[s] : n_items[s]);
            }
            Mx.first[5] = acc;
            const dim3 grid((unsigned)total, (unsigned)nsig);
            if (lean)
                hipLaunchKernelGGL((blockzoom_multi_kernel<true>), grid, dim3(NT), 0, stream, Mx, sp);
            else if (!Mx.A.kidx && !Mx.A.dWx && !Mx.A.w)   // Wx alone
                hipLaunchKernelGGL((blockzoom_multi_kernel<false, true>), grid, dim3(NT), 0, stream, Mx, sp);
            else
                hipLaunchKernelGGL((blockzoom_multi_kernel<false>), grid, dim3(NT), 0, stream, Mx, sp);
            SSQ_LAUNCH_CHECK();
            return 0;
        }
    }
#define ZOOM(slot, L, G, R1, R2, R3)                                                              \
    A.items = (const int4*)items[slot]; A.n_items = limit ? limit[slot] : n_items[slot];                                 \
    A.ftw = (const c32*)ftw + ftw_off[slot];                                                       \
    if ((rc = launch_zoom<L, G, R1, R2, R3>(A, sp, nsig, stream))) return rc;
    ZOOM(0, 128, 32, 16, 8, 1)
    ZOOM(1, 256, 16, 16, 16, 1)
    ZOOM(2, 512, 8, 8, 8, 8)
    ZOOM(3, 1024, 4, 16, 8, 8)
    ZOOM(4, 2048, 2, 16, 16, 8)
#undef ZOOM
    return 0;
}


// ---- exact rows: host side
static int log2i(int64_t v) { int l = 0; while ((1ll << l) < v) ++l; return l; }

int BlockPlan::setup_exact(const float* bank_dev, const int64_t* band_off_dev, const int32_t* band_lo_dev,
                           const int32_t* gen_rows_dev, const std::vector<int64_t>& h_off,
                           const std::vector<int32_t>& h_lo, const std::vector<int32_t>& h_gen,
                           int64_t& bytes) {
    exact_ok = false;
    if (h_gen.empty()) return 0;
    const int lm = log2i(M);
    if ((1ll << lm) != M || lm < 14 || lm > 22) return 0;
    for (int32_t i : h_gen) {                       // analytic rows only
        int64_t len = h_off[i + 1] - h_off[i];
        if (h_lo[i] + len > M / 2 + 1) return 0;
    }
    exA = 1 << ((lm + 1) / 2); exB = (int)(M / exA);
    e_bank = bank_dev; e_off = band_off_dev; e_lo = band_lo_dev; e_rows = gen_rows_dev;
    n_exact = (int)h_gen.size();
    std::vector<float> tw((size_t)2 * M);
    for (int64_t q = 0; q < M; ++q) {
        double ang = 2.0 * 3.14159265358979323846 * (double)q / (double)M;
        tw[2 * q] = (float)(cos(ang) / (double)M); tw[2 * q + 1] = (float)(sin(ang) / (double)M);
    }
    SSQ_CHECK_HIP(hipMalloc(&twM, 8 * (size_t)M)); bytes += 8 * M;
    SSQ_CHECK_HIP(hipMemcpy(twM, tw.data(), 8 * (size_t)M, hipMemcpyHostToDevice));
    SSQ_CHECK_HIP(hipMalloc(&zbuf, (size_t)group * n_exact * 2 * M * 8)); bytes += (int64_t)group * n_exact * 2 * M * 8;
    exact_ok = true;
    return 0;
}

template <int L, int G, int R1, int R2, int R3>
static int launch_exact1(const ExactArgs& E, int n_rows, int nsig, hipStream_t stream) {
    hipLaunchKernelGGL((exact_pass1_kernel<L, G, R1, R2, R3>), dim3((unsigned)(E.A / G), (unsigned)n_rows, (unsigned)nsig),
                       dim3(NT), 0, stream, E);
    SSQ_LAUNCH_CHECK();
    return 0;
}
template <int L, int G, int R1, int R2, int R3>
static int launch_exact2(const ExactArgs& E, const SsqParams& sp, int n_rows, int nsig, hipStream_t stream) {
    const dim3 grid((unsigned)(E.B / G), (unsigned)n_rows, (unsigned)nsig);
    if (E.kidx && !E.dWx && !E.w && !E.row_scale)
        hipLaunchKernelGGL((exact_pass2_kernel<L, G, R1, R2, R3, true>), grid, dim3(NT), 0, stream, E, sp);
    else
        hipLaunchKernelGGL((exact_pass2_kernel<L, G, R1, R2, R3, false>), grid, dim3(NT), 0, stream, E, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}

int BlockPlan::run_exact(int sig, int nsig, const void* xh_all, float* Wx, float* dWx, float* w,
                         unsigned short* kidx, const float* row_scale, double dt, const SsqParams& sp,
                         hipStream_t stream) {
    ExactArgs E;
    E.bank = e_bank; E.band_off = e_off; E.band_lo = e_lo; E.rows = e_rows;
    E.xh = (const c32*)xh_all; E.Z = (c32*)zbuf; E.twM = (const c32*)twM; E.row_scale = row_scale;
    E.Wx = Wx; E.dWx = dWx; E.w = w; E.kidx = kidx;
    E.M = M; E.N = N; E.na = na; E.A = exA; E.B = exB; E.n1pad = (int)n1; E.sig = sig; E.n_rows = n_exact;
    E.G2 = D_POINTS / exA;
    E.h = (2.0 * 3.141592653589793) / (double)M; E.inv_dt = 1.0f / (float)dt; E.gamma = sp.gamma;
    auto slot_of = [](int L) { return L == 128 ? 0 : L == 256 ? 1 : L == 512 ? 2 : L == 1024 ? 3 : 4; };
    int rc = 0;
    E.ftw = (const c32*)ftw + ftw_off[slot_of(exB)];
    switch (exB) {
        case 128: rc = launch_exact1<128, 32, 16, 8, 1>(E, n_exact, nsig, stream); break;
        case 256: rc = launch_exact1<256, 16, 16, 16, 1>(E, n_exact, nsig, stream); break;
        case 512: rc = launch_exact1<512, 8, 8, 8, 8>(E, n_exact, nsig, stream); break;
        case 1024: rc = launch_exact1<1024, 4, 16, 8, 8>(E, n_exact, nsig, stream); break;
        default: rc = launch_exact1<2048, 2, 16, 16, 8>(E, n_exact, nsig, stream); break;
    }
    if (rc) return rc;
    E.ftw = (const c32*)ftw + ftw_off[slot_of(exA)];
    switch (exA) {
        case 128: return launch_exact2<128, 32, 16, 8, 1>(E, sp, n_exact, nsig, stream);
        case 256: return launch_exact2<256, 16, 16, 16, 1>(E, sp, n_exact, nsig, stream);
        case 512: return launch_exact2<512, 8, 8, 8, 8>(E, sp, n_exact, nsig, stream);
        case 1024: return launch_exact2<1024, 4, 16, 8, 8>(E, sp, n_exact, nsig, stream);
        default: return launch_exact2<2048, 2, 16, 16, 8>(E, sp, n_exact, nsig, stream);
    }
}

}  // namespace ssq
