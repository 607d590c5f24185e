// ssq_cwt_tiles.hip -- the column-tile path of the fused ssq_cwt form (float32, gfx950).
//
// Math and planning: ssqueezepy_amd/_tiles.py. The kernels:
//
//   tilefft_*             the intermediates: band of row i (K bins around bin kc of the M-grid) x
//                         spectrum of the padded signal x compensated bank value, inverse FFT of
//                         length L = M / R (R: the row's decimation) -> the samples u_i[q]
//                         (plan-owned, ~34 MB per signal at N=160k). Classes of 64 .. 4096 entries:
//                         `tilefft_small_kernel` (one LDS transform per row, G rows per
//                         workgroup); classes of 2^13 .. 2^22 entries: `tilefft_four_kernel<1 / 2>`,
//                         a four-step transform whose first pass forms the band on the fly (the
//                         zero-padded spectrum never exists in memory). Three launches per launch
//                         group for all classes. `tile_spectra_kernel` + a batched rocFFT inverse
//                         per class is the older route (SSQ_TILE_FFT=rocfft).
//
//   tile_kernel           one persistent workgroup per CU (12 wavefronts, 3 per SIMD) walks
//                         64-column tiles of one signal after the other. The 64 columns x na bins
//                         of the tile's Tx live in LDS (which is why there is one workgroup per
//                         CU); lane = column. The steps (4 consecutive rows) of all tiles are
//                         dealt to the wavefronts round-robin. Per step and lane:
//                           interpolated rows: ONE 8-byte load of u_i per row (the lanes hold a
//                             window of consecutive samples, taps come from the neighbours with
//                             ds_bpermute), 8 taps x (phi, phi') as packed FMAs, modulation by
//                             hardware sin / cos of the exact phase kc n mod M, Wx stored
//                             (512-byte runs), phase transform + bin exactly as the other fused
//                             kernels do (ssq_point_math.inl);
//                           rows read back: Wx and the 2-byte bin the block / exact kernels left.
//                         The arithmetic of different steps runs concurrently; only the
//                         reassignment T[bin] += Wx * const is ordered, by a ticket in LDS (step
//                         S may update the tile once `turn` says so): every cell receives its
//                         contributions in ascending row order, the reference's
//                         (algos.py:859-953), so the float sums are bit-identical to the CPU loop
//                         on the same Wx / dWx. Inside a step the four rows' cells are read
//                         together and chained in registers when they coincide (same lane = same
//                         column: no cross-lane traffic), then written in row order. Between two
//                         tiles all wavefronts write their share of the finished tile to Tx.
//                         No workgroup barrier after the prologue, no atomics on data.
//
//                         Round 3 tried the other split -- producer wavefronts that only compute,
//                         one (or four) updater wavefronts that only reassign, hand-over through an
//                         L2-resident ring -- and measured it slower (400-550 us per transform
//                         against 320): a single wavefront issues a dependent instruction every
//                         8-10 cycles, so a serial stream of ~100 instructions per step cannot keep
//                         up with fifteen producers. What that round kept from it: loads issued
//                         unconditionally so that the compiler's wait counts stay static, the
//                         float64-weight fold, one step per ticket. Everything else that was
//                         measured and dropped: DESIGN.md section 4.8, profiles/r3_ab_history.txt.
//
// Compiled with -ffp-contract=off (bin indices); multiply-adds that may fuse are written as
// explicit fmaf so every instantiation rounds identically.
#include "ssq_common.h"
#include "ssq_tiles.h"
#include "ssq_ldsfft.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace ssq {

#include "ssq_point_math.inl"

constexpr int TILE_COLS = 64;     // columns per workgroup (one per lane)
#ifndef SSQ_TILE_G
#define SSQ_TILE_G 4
#endif
constexpr int TILE_G = SSQ_TILE_G;   // rows per step (the host's RSUB, _tiles.py)
constexpr int TILE_W = 8;         // taps
constexpr int TILE_NOBIN = 0xFFFF;
// tuning experiments (A/B builds, tools/ab_build.sh; WRONG RESULTS): 1 = no reassignment (tickets
// only), 4 = no arithmetic, 8 = no priorities, 32 = taps without the cross-lane gather, 64 = no
// modulation, 128 = no bin arithmetic, 256 = no Wx store (round 4: none of 32 .. 128 changes the
// kernel's time, 256 takes 25 us off -- profiles/r4_ab_history.txt); tile2_kernel: 8, 256, 512 = no bin
// arithmetic, 1024 = no gather
#ifndef SSQ_TILE_EXP
#define SSQ_TILE_EXP 0
#endif

struct TileArgs {
    const int4* pstep;                               // one packed record per step (see TilePlan::create)
    const int2* prow;                                // 4 packed row records per step (see TilePlan::create)
    const float4* wtab; const float2* U;
    const void* cst;
    float2* Wx; float2* dWx; float2* Tx; const unsigned short* kidx;
    int64_t N, na;
    int nsteps, n1, mmask, sig0, nsig;
    float inv_m;         // 1 / M
    float theta_scale;   // 2 pi / (M dt): theta of a row = kc * theta_scale
    float cst0;          // the reassignment weight when it is the same for every row
    unsigned long long* counters;   // [0] += tiles finished (what actually ran)
    unsigned long long* trace;      // tuning aid (SSQ_TILE_TRACE): shader-clock stamps of one workgroup
    double gamma;
};

__global__ __launch_bounds__(256) void tile_spectra_kernel(const float2* __restrict__ xh_all,
                                                           int64_t xh_stride, int sig0,
                                                           const TileIRow* __restrict__ irows,
                                                           const float* __restrict__ tbank,
                                                           float2* __restrict__ U) {
    const TileIRow r = irows[blockIdx.y];
    const int s = blockIdx.z;
    const float2* xh = xh_all + (int64_t)(sig0 + s) * xh_stride;
    float2* u = U + r.ubase + (int64_t)s * r.sig_stride;
    const int half = r.L >> 1;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < r.L; p += gridDim.x * blockDim.x) {
        const int kk = p < half ? p : p - r.L;            // signed baseband bin
        const int t = r.kc + kk - r.lo;
        float2 z = make_float2(0.f, 0.f);
        if (t >= 0 && t < r.K) {
            const float2 x = xh[r.lo + t];
            const float b = tbank[r.tb_off + t];
            z = make_float2(x.x * b, x.y * b);
        }
        u[p] = z;
    }
}

// ---- the long classes (L >= 2^14) of the intermediates: a four-step inverse FFT of our own.
// rocFFT took 48 us per transform for them at config 2 (a single-kernel 16 384-point transform
// at 0.5 TB/s, three passes for 65 536 points) plus the spectra kernel's write of the
// zero-padded band; they are not hidden behind the block kernels (measured: side stream or
// not, the same time), so they sit on the critical path. Here: L = A B, bin k = A k2 + k1,
// sample q = B q1 + q2,
//   pass 1  for every k1: B-point inverse FFT over k2 of the band -- formed on the fly from
//           the signal's spectrum and the compensated bank values, zeros never touch memory --
//           times e^{2 pi i k1 q2 / L} (hardware sin / cos of an exact phase, as the tile
//           kernel's modulation), transposed through LDS into Y, blocked for pass 2;
//   pass 2  for every q2: A-point inverse FFT over k1 -> u[B q1 + q2].
// Both passes are the LDS Stockham transform of the block kernels (ssq_ldsfft.h): 4096 points
// per 256-thread workgroup, 8 + 8 + 8 bytes per sample of HBM / L2 traffic.
struct TileFftArgs {
    const c32* xh; int64_t xh_stride; int sig0;
    const TileIRow* irows;             // the rows of this class
    const float* tbank;
    c32* Y; c32* U;
    const c32* ftw1; const c32* ftw2;  // e^{2 pi i q / B}, e^{2 pi i q / A}
    int A, B, L, G2, nrows;
    int nyq;                           // 1: bin L / 2 counts as +L / 2 (a one-sided spectrum up to Nyquist)
    float inv_l;
};

// (bx, r, z): workgroup inside the class -- k1 group (pass 1) or q2 group (pass 2), row, signal
template <int LB, int G, int R1, int R2, int R3>
__device__ __forceinline__ void tilefft_pass1_body(const TileFftArgs& E, int bx, int r, int z_sig, c32* buf) {
    constexpr int RL = (R3 > 1) ? R3 : R2;
    const int tid = threadIdx.x;
    const TileIRow row = E.irows[r];
    const int c0 = bx * G;                                  // first k1 of this workgroup
    const c32* xh = E.xh + (int64_t)(E.sig0 + z_sig) * E.xh_stride;
    const float* tb = E.tbank + row.tb_off;
    const int half = E.L >> 1;
    c32 z[PPT];
    {
        constexpr int NB = PPT / R1, STR = LB / R1;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int p = (c0 + g) + E.A * (u + k * STR);       // baseband bin, as tile_spectra_kernel
                const int kk = (p < half || (E.nyq && p == half)) ? p : p - E.L;
                const int t = row.kc + kk - row.lo;
                c32 v = {0.f, 0.f};
                if (t >= 0 && t < row.K) {
                    const c32 X = xh[row.lo + t];
                    const float b = tb[t];
                    v = {X.x * b, X.y * b};
                }
                z[it * R1 + k] = v;
            }
        }
    }
    lds_ifft<LB, G, R1, R2, R3>(z, buf, E.ftw1, tid);
    __syncthreads();
    constexpr int NBL = PPT / RL, STRL = LB / RL;
#pragma unroll
    for (int it = 0; it < NBL; ++it) {
        const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
        for (int k = 0; k < RL; ++k) {
            const int q2 = u + k * STRL;
            // k1 q2 < A B = L <= 2^22: the phase is exact in integers and in float
            const float rev = (float)((c0 + g) * q2) * E.inv_l;
            const c32 tw = {__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)};
            buf[g * (LB + 1) + q2] = cmul_v(z[it * RL + k], tw);
        }
    }
    __syncthreads();
    const int G2 = E.G2, lg2 = __ffs(G2) - 1;
    constexpr int LG = (G == 1) ? 0 : (G == 2) ? 1 : (G == 4) ? 2 : (G == 8) ? 3 : (G == 16) ? 4 : (G == 32) ? 5 : 6;
    c32* Yt = E.Y + ((int64_t)z_sig * E.nrows + r) * E.L;
#pragma unroll
    for (int it = 0; it < PPT; ++it) {
        // consecutive lanes: q2 % G2 fastest, then this workgroup's k1 -> runs of G * G2 entries
        const int idx = tid + it * NT, q2i = idx & (G2 - 1), g = (idx >> lg2) & (G - 1);
        const int q2t = idx >> (lg2 + LG), q2 = (q2t << lg2) + q2i;
        Yt[((int64_t)q2t * E.A + (c0 + g)) * G2 + q2i] = buf[g * (LB + 1) + q2];
    }
}

template <int LA, int G, int R1, int R2, int R3>
__device__ __forceinline__ void tilefft_pass2_body(const TileFftArgs& E, int bx, int r, int z_sig, c32* buf) {
    constexpr int RL = (R3 > 1) ? R3 : R2;
    const int tid = threadIdx.x;
    const TileIRow row = E.irows[r];
    const c32* Yr = E.Y + ((int64_t)z_sig * E.nrows + r) * E.L;
    c32 z[PPT];
    {
        constexpr int NB = PPT / R1, STR = LA / R1;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < R1; ++k)
                z[it * R1 + k] = Yr[(int64_t)bx * LA * G + (u + k * STR) * G + g];     // blocked Y
        }
    }
    lds_ifft<LA, G, R1, R2, R3>(z, buf, E.ftw2, tid);
    c32* u_out = E.U + row.ubase + (int64_t)z_sig * row.sig_stride;
    constexpr int NB = PPT / RL, STR = LA / RL;
#pragma unroll
    for (int it = 0; it < NB; ++it) {
        const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
        for (int k = 0; k < RL; ++k)
            u_out[(bx * G + g) + E.B * (u + k * STR)] = z[it * RL + k];
    }
}

// all four-step classes of a launch group in one launch per pass (the shorter classes alone do not
// fill the chip, and every launch has a tail): workgroup -> class by ranges, then (group, row, signal)
struct TileFourArgs {
    TileFftArgs E[10];
    int first_block[11];     // workgroups before class c
    int nx[10], slot[10];    // k1 / q2 groups per (row, signal); transform length = 64 << slot
    int ncls;
};
template <int PASS>
__global__ __launch_bounds__(NT) void tilefft_four_kernel(TileFourArgs A) {
    __shared__ c32 buf[D_POINTS + 64];
    int b = (int)blockIdx.x, c = 0;
    while (c + 1 < A.ncls && b >= A.first_block[c + 1]) ++c;
    b -= A.first_block[c];
    const TileFftArgs& E = A.E[c];
    const int nx = A.nx[c];
    const int bx = b % nx, rz = b / nx, r = rz % E.nrows, z_sig = rz / E.nrows;
    if (PASS == 1) {
        switch (A.slot[c]) {
            case 1: tilefft_pass1_body<128, 32, 16, 8, 1>(E, bx, r, z_sig, buf); break;
            case 2: tilefft_pass1_body<256, 16, 16, 16, 1>(E, bx, r, z_sig, buf); break;
            case 3: tilefft_pass1_body<512, 8, 8, 8, 8>(E, bx, r, z_sig, buf); break;
            case 4: tilefft_pass1_body<1024, 4, 16, 8, 8>(E, bx, r, z_sig, buf); break;
            default: tilefft_pass1_body<2048, 2, 16, 16, 8>(E, bx, r, z_sig, buf); break;
        }
    } else {
        switch (A.slot[c]) {
            case 0: tilefft_pass2_body<64, 64, 8, 8, 1>(E, bx, r, z_sig, buf); break;
            case 1: tilefft_pass2_body<128, 32, 16, 8, 1>(E, bx, r, z_sig, buf); break;
            case 2: tilefft_pass2_body<256, 16, 16, 16, 1>(E, bx, r, z_sig, buf); break;
            case 3: tilefft_pass2_body<512, 8, 8, 8, 8>(E, bx, r, z_sig, buf); break;
            case 4: tilefft_pass2_body<1024, 4, 16, 8, 8>(E, bx, r, z_sig, buf); break;
            default: tilefft_pass2_body<2048, 2, 16, 16, 8>(E, bx, r, z_sig, buf); break;
        }
    }
}

// The short classes (64 .. 4096 entries per row) in one kernel: G (row, signal) pairs of a class
// per workgroup, band -> LDS transform -> samples, transposed through LDS so that every row is
// written as a run. Replaces the spectra kernel (a write of the zero-padded band) + a rocFFT launch
// per class.
template <int L, int G, int R1, int R2, int R3>
__device__ __forceinline__ void tilefft_small_body(const TileFftArgs& E, int npairs, int block, c32* buf) {
    constexpr int RL = (R3 > 1) ? R3 : R2;
    constexpr int LGL = (L == 64) ? 6 : (L == 128) ? 7 : (L == 256) ? 8 : (L == 512) ? 9 : (L == 1024) ? 10 : (L == 2048) ? 11 : 12;
    const int tid = threadIdx.x;
    const int half = L >> 1;
    c32 z[PPT];
    {
        constexpr int NB = PPT / R1, STR = L / R1;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + it * NT, g = idx % G, u = idx / G;
            const int j = block * G + g;             // (row, signal) pair: row fastest
            const bool live = j < npairs;
            const int jr = live ? j % E.nrows : 0, js = live ? j / E.nrows : 0;
            const TileIRow row = E.irows[jr];
            const c32* xh = E.xh + (int64_t)(E.sig0 + js) * E.xh_stride;
            const float* tb = E.tbank + row.tb_off;
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int p = u + k * STR;
                const int kk = p < half ? p : p - L;
                const int t = row.kc + kk - row.lo;
                c32 v = {0.f, 0.f};
                if (live && t >= 0 && t < row.K) {
                    const c32 X = xh[row.lo + t];
                    const float b = tb[t];
                    v = {X.x * b, X.y * b};
                }
                z[it * R1 + k] = v;
            }
        }
    }
    lds_ifft<L, G, R1, R2, R3>(z, buf, E.ftw1, tid);
    __syncthreads();
    constexpr int NBL = PPT / RL, STRL = L / RL;
#pragma unroll
    for (int it = 0; it < NBL; ++it) {
        const int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
        for (int k = 0; k < RL; ++k) buf[g * (L + 1) + u + k * STRL] = z[it * RL + k];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < PPT; ++it) {
        const int idx = tid + it * NT, q = idx & (L - 1), g = idx >> LGL;
        const int j = block * G + g;
        if (j < npairs) {
            const int jr = j % E.nrows, js = j / E.nrows;
            const TileIRow row = E.irows[jr];
            E.U[row.ubase + (int64_t)js * row.sig_stride + q] = buf[g * (L + 1) + q];
        }
    }
}

// all short classes of a launch group in ONE launch: each class alone is a few dozen workgroups
// (32 rows x 16 signals / G), far too few to fill 256 CUs -- launched one after the other they
// cost ~190 us per group, side by side what the longest of them takes
struct TileSmallArgs {
    TileFftArgs E[7];
    int first_block[8];      // workgroups before class c
    int npairs[7], slot[7];  // (row, signal) pairs of the class; L = 64 << slot
    int ncls;
};
__global__ __launch_bounds__(NT) void tilefft_small_kernel(TileSmallArgs A) {
    __shared__ c32 buf[D_POINTS + 64];
    int b = (int)blockIdx.x, c = 0;
    while (c + 1 < A.ncls && b >= A.first_block[c + 1]) ++c;
    b -= A.first_block[c];
    switch (A.slot[c]) {
        case 0: tilefft_small_body<64, 64, 8, 8, 1>(A.E[c], A.npairs[c], b, buf); break;
        case 1: tilefft_small_body<128, 32, 16, 8, 1>(A.E[c], A.npairs[c], b, buf); break;
        case 2: tilefft_small_body<256, 16, 16, 16, 1>(A.E[c], A.npairs[c], b, buf); break;
        case 3: tilefft_small_body<512, 8, 8, 8, 8>(A.E[c], A.npairs[c], b, buf); break;
        case 4: tilefft_small_body<1024, 4, 16, 8, 8>(A.E[c], A.npairs[c], b, buf); break;
        case 5: tilefft_small_body<2048, 2, 16, 16, 8>(A.E[c], A.npairs[c], b, buf); break;
        default: tilefft_small_body<4096, 1, 16, 16, 16>(A.E[c], A.npairs[c], b, buf); break;
    }
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
#ifdef SSQ_NO_CMUL_PK
    return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
#else
    ssq_f2 av, bv, dv;
    av.x = a.x; av.y = a.y; bv.x = b.x; bv.y = b.y;
    SSQ_CMUL_PK(dv, av, bv);
    return make_float2(dv.x, dv.y);
#endif
}

// workgroup-scope synchronisation through LDS words (all wavefronts of a workgroup share the
// CU's L1, so workgroup scope costs waits only, no cache maintenance)
__device__ __forceinline__ int lds_load_acquire(const int* p) {
    return __scoped_atomic_load_n(p, __ATOMIC_ACQUIRE, __MEMORY_SCOPE_WRKGRP);
}
__device__ __forceinline__ void lds_store_release(int* p, int v) {
    __scoped_atomic_store_n(p, v, __ATOMIC_RELEASE, __MEMORY_SCOPE_WRKGRP);
}
// The ticket itself: LDS serves the operations of one wavefront in program order and those of the
// CU's wavefronts from one queue, so a tile access issued before the ticket store is performed
// before an access another wavefront issues after it has read the new ticket -- no wait for
// completion is needed on either side (the compiler is kept from moving LDS accesses across).
__device__ __forceinline__ int ticket_peek(const int* p) {
    const int v = __scoped_atomic_load_n(p, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
    asm volatile("" ::: "memory");
    return v;
}
// Earliest deadline first: the wavefronts of a SIMD compete for its issue slots (the oldest wins by
// default, so the youngest would always be late for its turn and everybody would wait for it); a
// wavefront raises its priority as its turn comes closer.
__device__ __forceinline__ void ticket_priority(const int* turn, int ticket) {
#if !(SSQ_TILE_EXP & 8)
    const int d = __builtin_amdgcn_readfirstlane(ticket - ticket_peek(turn));
    if (d <= 3) __builtin_amdgcn_s_setprio(3);
    else if (d <= 6) __builtin_amdgcn_s_setprio(2);
    else if (d <= 9) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ __forceinline__ void ticket_pass(int* turn, int next, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __scoped_atomic_store_n(turn, next, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
    asm volatile("" ::: "memory");
}

// bin of a point the float32 screens could not decide (flipped as Tx wants it), or -1 when it
// does not contribute: the exact double sequence of the CPU path (~0.05 % of the points). Inline:
// a call would put the parameters on the stack and make the compiler wait for every load in
// flight at the join.
__device__ __forceinline__ int exact_bin(float2 W, float2 D, const SsqParams& sp, int omax, double gamma) {
    if (!(mag_of(W.x, W.y) > gamma)) return -1;
    const int ke = (int)bin_of_point_exact(D.x, D.y, W.x, W.y, sp, (int64_t)omax);
    return sp.flipud ? omax - ke : ke;
}

// trace (tuning aid): [wavefront][step slot < 128][4 stamps], then 64 extra words
constexpr int TRACE_STEPS = 128, TRACE_K = 8, TRACE_WORDS = 16 * TRACE_STEPS * TRACE_K + 64;
// (compiled in with -DSSQ_TILE_TRACE_BUILD, tools/ab_build.sh: the stamps cost registers)
#ifdef SSQ_TILE_TRACE_BUILD
#define TILE_STAMP(on, wave, j, k)                                                               \
    do { if (tr && (on) && (j) < TRACE_STEPS && c == 0)                                        \
             tr[((size_t)(wave) * TRACE_STEPS + (j)) * TRACE_K + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TILE_STAMP(on, wave, j, k) do { (void)(on); } while (0)
#endif
constexpr int TRACE_TILE = 2;

// the additive term of one point and how it is folded into a cell, in the CPU path's arithmetic:
// float32 data with a float64 weight vector accumulates through double (algos.py:66-79)
template <bool CST64> struct TileTerm {
    using type = float;
    using wtype = float;
    static __device__ __forceinline__ float make(float z, float w) { return z * w; }
    static __device__ __forceinline__ float fold(float o, float t) { return o + t; }
};
template <> struct TileTerm<true> {
    using type = double;
    using wtype = double;
    static __device__ __forceinline__ double make(float z, double w) { return (double)z * w; }
    static __device__ __forceinline__ float fold(float o, double t) { return (float)((double)o + t); }
};

// ---- the reassignment of one step (4 rows) into the tile, in row order; lane = column. Rows of
// a step that hit the same cell are chained in registers: the cells are read together, a row
// that hits the cell of an earlier row of the step starts from that row's result, the cells are
// written back in row order. `cell` of a point without contribution is the lane's scratch cell.
// Everything that does not depend on the tile is prepared BEFORE the step's turn (the ticket
// section is the serial part of a tile): the cells' addresses, the terms, and the "same cell"
// tests as bit masks, so that inside the turn a select is one v_bfi_b32 per word.
struct Update4Prep {
    int off[TILE_G];                   // byte offset of the cell in the tile
    int same[TILE_G][TILE_G];          // [r][q], q < r: all ones if row q hits the cell of row r
};
__device__ __forceinline__ void update4_prepare(const int (&cell)[TILE_G], Update4Prep& u) {
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) {
        u.off[r] = cell[r] * 8;
        SSQ_OPAQUE_V(u.off[r]);        // (materialised here, not behind the ticket)
#pragma unroll
        for (int q = 0; q < r; ++q) { u.same[r][q] = cell[q] == cell[r] ? -1 : 0; SSQ_OPAQUE_V(u.same[r][q]); }
    }
}
__device__ __forceinline__ float bit_select(int m, float a, float b) {      // m ? a : b, per bit
    int d;
    SSQ_BFI(d, m, __float_as_int(a), __float_as_int(b));
    return __int_as_float(d);
}
// The wait for the step's turn and the read of its cells in ONE LDS round trip: the wavefront
// next in line issues the ticket read and, right behind it, the reads of its four cells; LDS
// serves them in that order, so when the ticket read shows the step's turn the cell reads were
// served after the predecessor's writes (otherwise the batch is thrown away and issued again).
// Wavefronts further from their turn only look at the ticket, with a pause in between.
__device__ __forceinline__ void ticket_wait_read4(const int* turn, int ticket, unsigned char* tile,
                                                  const Update4Prep& u, float2 (&t)[TILE_G]) {
    for (;;) {
        const int d = ticket - ticket_peek(turn);
        if (d <= 1) break;
        __builtin_amdgcn_s_sleep(1);
    }
    for (;;) {
        const int v = __scoped_atomic_load_n(turn, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) {
            // (a relaxed atomic read: a plain one could be hoisted out of the loop, a volatile one
            // loses the LDS address space)
            const unsigned long long bits = __scoped_atomic_load_n(reinterpret_cast<unsigned long long*>(tile + u.off[r]),
                                                                   __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
            t[r].x = __int_as_float((int)(unsigned)bits); t[r].y = __int_as_float((int)(unsigned)(bits >> 32));
        }
        asm volatile("" ::: "memory");
        if (v == ticket) break;
        // (s_sleep 0 is the shortest pause there is; under the CPU emulation it is where the
        // other wavefronts get to run)
        __builtin_amdgcn_s_sleep(0);
    }
}
template <typename TM>
__device__ __forceinline__ void update4_finish(unsigned char* tile, const Update4Prep& u, float2 (&t)[TILE_G],
                                               const typename TM::type (&vx)[TILE_G], const typename TM::type (&vy)[TILE_G]) {
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) {
#pragma unroll
        for (int q = 0; q < r; ++q) {
            t[r].x = bit_select(u.same[r][q], t[q].x, t[r].x);
            t[r].y = bit_select(u.same[r][q], t[q].y, t[r].y);
        }
        t[r].x = TM::fold(t[r].x, vx[r]); t[r].y = TM::fold(t[r].y, vy[r]);
    }
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) *reinterpret_cast<float2*>(tile + u.off[r]) = t[r];
}
// (pins a term in a register before the ticket)
__device__ __forceinline__ void keep_term(float& x) { SSQ_OPAQUE_V(x); }
__device__ __forceinline__ void keep_term(double& x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    SSQ_OPAQUE_V(lo); SSQ_OPAQUE_V(hi);
    x = __hiloint2double(hi, lo);
}

__host__ __device__ inline size_t tile_lds_bytes(int64_t na) {
    return (size_t)(na + 1) * TILE_COLS * 8 + 16;
}

// CSTK: reassignment weights -- 0 one float (cst0), 1 a float per row, 2 a double per row
template <int GRID, bool STORE_D, int NW, int CSTK>
__global__ __launch_bounds__(64 * NW) void tile_kernel(TileArgs A, SsqParams sp) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int c = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t N = A.N;
    const unsigned nN = (unsigned)N;
    const int na = (int)A.na, omax = na - 1;
    float2* T = reinterpret_cast<float2*>(lds_raw);           // (na + 1) x 64 cells, the last row: scratch
    int* turn = reinterpret_cast<int*>(lds_raw + (size_t)(na + 1) * TILE_COLS * 8);
    int* wdone = turn + 1;
    for (int k = wv; k <= na; k += NW) T[k * TILE_COLS + c] = make_float2(0.f, 0.f);
    if (threadIdx.x == 0) { *turn = 0; *wdone = 0; }
    __syncthreads();
    const int scratch = na * TILE_COLS + c;

    // The workgroup is persistent: it walks the tiles blockIdx.x, + gridDim.x, ... of the launch
    // group (tile = 64 columns of one signal). All steps of all its tiles form one sequence
    // S = 0, 1, ...: wavefront w takes S = w, w + NW, ... (positions advance monotonically, so
    // divisions are replaced by repeated subtraction). Ticket of step S of tile itl: S + itl --
    // one extra ticket per tile, during which the finished tile is written out.
    const int ntx = (int)((N + TILE_COLS - 1) / TILE_COLS);
    const int ntot = ntx * A.nsig;
    const int ntl = ntot > (int)blockIdx.x ? (ntot - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int nst = A.nsteps;
    const int total = nst * ntl;
    struct Pos { int S, itl, st, tx, sg; };                   // a step: sequence number, tile, step in tile, tile position
    auto advance = [&](Pos& q, int by) {
        q.S += by; q.st += by;
        while (q.st >= nst && q.itl < ntl) {
            q.st -= nst; ++q.itl; q.tx += (int)gridDim.x;
            while (q.tx >= ntx) { q.tx -= ntx; ++q.sg; }
        }
    };
    unsigned long long* tr = (A.trace && (int)blockIdx.x == (100 < (int)gridDim.x ? 100 : (int)gridDim.x - 1)) ? A.trace : nullptr;

    // rows k = wv, wv + NW, ... of the finished tile go to Tx and are cleared; the last
    // wavefront to finish opens the next tile's tickets. (Round 3 also measured the write-out by
    // ONE wavefront, inside the turn of the tile's last step, so that the others never meet: a
    // single wavefront stores 300 x 512 bytes in ~17 k cycles -- 345 us per transform against 275.)
    auto write_out = [&](int itl, int tx, int sg) {
        const int boundary = (itl + 1) * nst + itl;           // the ticket after the tile's last step
        while (ticket_peek(turn) != boundary) __builtin_amdgcn_s_sleep(1);
        const unsigned col = (unsigned)(tx * TILE_COLS + c);
        const bool ok = col < nN;
        float2* Tx = A.Tx + (int64_t)(A.sig0 + sg) * na * N;
        for (int k0 = wv; k0 < na; k0 += 4 * NW) {
            float2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int k = k0 + q * NW; v[q] = T[(k < na ? k : na) * TILE_COLS + c]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + q * NW;
                if (k < na) {
                    T[k * TILE_COLS + c] = make_float2(0.f, 0.f);
                    if (ok) Tx[(unsigned)k * nN + col] = v[q];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (c == 0) {
            const int before = __scoped_atomic_fetch_add(wdone, 1, __ATOMIC_ACQ_REL, __MEMORY_SCOPE_WRKGRP);
            if (before + 1 == NW * (itl + 1)) {
                lds_store_release(turn, boundary + 1);
                if (A.counters) __scoped_atomic_fetch_add(A.counters, 1ull, __ATOMIC_RELAXED, __MEMORY_SCOPE_DEVICE);
            }
        }
    };

    const float g2 = (float)(A.gamma * A.gamma);
    const float m2hi = g2 * 1.000004f, m2lo = g2 * 0.999996f;
    const int fx = sp.flipud ? -1 : 0, fa = sp.flipud ? na : 0;
    using TM = TileTerm<CSTK == 2>;
    using term_t = typename TM::type;
    using w_t = typename TM::wtype;
    const w_t* cstv = (const w_t*)A.cst;

    // Step and row records are the same for every lane. They are fetched with vector loads from a
    // lane-independent address (one request per wavefront) rather than scalar loads: scalar and
    // LDS operations share one counter (lgkmcnt) and scalar loads return out of order, so a
    // scalar load in flight turns every wait for a ds_bpermute result into a full drain.
    int vz = 0;
    SSQ_OPAQUE_V(vz);
    const int2* rows2 = reinterpret_cast<const int2*>(A.prow) + vz;   // per row: row | pad << 9 | kc << 10, offset of its samples
    const int4* steps4 = reinterpret_cast<const int4*>(A.pstep) + vz;   // per step: kind | lgR << 1 | weight offset << 8, L - 1, stride, base
    const w_t* cstu = cstv + vz;

    // Software pipeline over this wavefront's steps: the records of a step are fetched while the
    // step before it is computed, its samples (or Wx and bins) half a step ahead, the
    // interpolation weights when the last taps of the step before are done. ALL loads are issued
    // unconditionally (past the last step: the last step again, results unused): the compiler
    // counts the loads in flight per path, and a path that skips some turns every wait into a
    // full drain.
    int4 sa; int2 rec[TILE_G];                // next step: its packed record, its rows
    auto load_rec = [&](const Pos& q) {
        const int g = q.st;
        sa = steps4[g];
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) rec[r] = rows2[g * TILE_G + r];
    };
    float2 xu[2][TILE_G];                     // samples of the interpolated rows / Wx of the rows read back
    // per row: the packed record (interpolated rows) or the bin (rows read back) -- one register
    // either way (16 wavefronts need the step pipeline under 128 registers)
    int xq[2][TILE_G];
    int xnv[2];                               // rows of the step that are not padding
    w_t xc[2][CSTK == 0 ? 1 : TILE_G];        // per-row weights
    int xkind[2], xbaddr[2], xwoff[2], xmask[2];
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    auto load = [&](auto BB, const Pos& q) {   // data of the step whose records are in (sa, sb, rec)
        constexpr int b = decltype(BB)::value;
        const int sax = __builtin_amdgcn_readfirstlane(sa.x);
        const int kind = sax & 1;
        xkind[b] = kind;
        const int col0 = q.tx * TILE_COLS, col = col0 + c;
        const int colc = col < (int)N ? col : (int)N - 1;         // loads stay in range
        const int nabs = A.n1 + colc, nabs0 = A.n1 + col0;
        const int lgR = (sax >> 1) & 31;
        xwoff[b] = sax >> 8; xmask[b] = (1 << lgR) - 1;
        const int q0 = nabs >> lgR, qb = (nabs0 >> lgR) - (TILE_W / 2 - 1);
        // interpolated rows: the sample this lane holds (lanes past the widest window any lane
        // needs repeat the last one); rows read back: the lane's own point
        const int wlast = (63 >> lgR) + TILE_W;
        const unsigned uidx = (unsigned)((qb + (c < wlast ? c : wlast)) & __builtin_amdgcn_readfirstlane(sa.y));
        xbaddr[b] = (q0 - (TILE_W / 2 - 1) - qb) * 4;            // lane that holds tap 0
        // Addresses: a wave-uniform 64-bit base (scalar arithmetic on the records, which arrived half
        // a step ago) + one 32-bit byte offset per lane that is the same for the four rows.
        const char* Ub8 = reinterpret_cast<const char*>(A.U + __builtin_amdgcn_readfirstlane(sa.w)
                                                        + (int64_t)q.sg * __builtin_amdgcn_readfirstlane(sa.z));
        const char* Wx8 = reinterpret_cast<const char*>(A.Wx + (int64_t)(A.sig0 + q.sg) * na * N);
        const char* kx8 = reinterpret_cast<const char*>(A.kidx + (int64_t)q.sg * na * N);
        const unsigned vo = kind ? uidx * 8u : (unsigned)colc * 8u;
        int nv = 0;
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) {
            const int2 d = rec[r];
            const int dx = __builtin_amdgcn_readfirstlane(d.x);
            const unsigned row = (unsigned)dx & 0x1FFu;
            nv += ((dx >> 9) & 1) ^ 1;
            // one 8-byte load either way: a sample of u (interpolated) or Wx (read back)
            const char* base = kind ? Ub8 + (size_t)(unsigned)__builtin_amdgcn_readfirstlane(d.y) * 8u
                                    : Wx8 + (size_t)row * (nN * 8u);
            xu[b][r] = *reinterpret_cast<const float2*>(base + vo);
            if (kind) xq[b][r] = d.x;
            else xq[b][r] = *reinterpret_cast<const unsigned short*>(kx8 + (size_t)row * (nN * 2u) + (unsigned)colc * 2u);
            if (CSTK != 0) xc[b][r] = cstu[row];
        }
        xnv[b] = nv;
    };
    ssq_f2 wt[TILE_W];                        // (phi_t, phi'_t / (R dt)) of the step in hand
    auto load_wt = [&](auto BB, const Pos& q) {
        constexpr int b = decltype(BB)::value;
        const int col = q.tx * TILE_COLS + c;
        const int nabs = A.n1 + (col < (int)N ? col : (int)N - 1);
        // (the table is stored tap pair by tap pair, [4][R] float4 per class: the 64 lanes of a load
        // read one run of consecutive phases, not 64 separate 64-byte rows)
        const float4* wp = A.wtab + (int64_t)xwoff[b] * 4 + (nabs & xmask[b]);
        const int wstride = xmask[b] + 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = wp[t * wstride];
            wt[2 * t].x = v.x; wt[2 * t].y = v.y; wt[2 * t + 1].x = v.z; wt[2 * t + 1].y = v.w;
        }
    };

    Pos pc; pc.S = 0; pc.itl = 0; pc.st = 0;
    pc.sg = (int)blockIdx.x / ntx; pc.tx = (int)blockIdx.x - pc.sg * ntx;
    int w_itl = 0, w_tx = pc.tx, w_sg = pc.sg;                // next tile to write out
    advance(pc, wv);                          // the step computed
    auto write_outs_before = [&](int itl) {   // every finished tile before tile `itl`, in order
        while (w_itl < itl) {
            write_out(w_itl, w_tx, w_sg);
            ++w_itl; w_tx += (int)gridDim.x;
            while (w_tx >= ntx) { w_tx -= ntx; ++w_sg; }
        }
    };
    Pos pl = pc;                              // a valid step for the loads past the end
    auto clampp = [&](const Pos& q) { return q.S < total ? q : pl; };
    if (pc.S < total) {
        load_rec(pc); load(B0{}, pc); load_wt(B0{}, pc);
        Pos pn = pc; advance(pn, NW);         // the step whose data are loaded next
        load_rec(clampp(pn));
        auto step = [&](auto BB, auto BN) {
            constexpr int b = decltype(BB)::value;
            const bool trk = tr && pc.itl == TRACE_TILE;
            TILE_STAMP(trk, wv, pc.st, 0);
            ticket_priority(turn, pc.S + pc.itl);
            const int col0 = pc.tx * TILE_COLS, col = col0 + c;
            const bool colok = col < (int)N;
            const int colc = colok ? col : (int)N - 1;
            const int nabs = A.n1 + colc;
            int cell[TILE_G]; term_t vx[TILE_G], vy[TILE_G];
            Pos pnn = pn;
            if (__builtin_amdgcn_readfirstlane(xkind[b]) == 0) {
                // rows read back: Wx and the bin are there
                load(BN, clampp(pn)); load_wt(BN, clampp(pn)); advance(pnn, NW); load_rec(clampp(pnn));
#pragma unroll
                for (int r = 0; r < TILE_G; ++r) {
                    const int kk = xq[b][r] & 0xFFFF;
                    const bool act = r < xnv[b] && colok && kk != TILE_NOBIN;
                    cell[r] = act ? kk * TILE_COLS + c : scratch;
                    const w_t cs = CSTK == 0 ? (w_t)A.cst0 : xc[b][CSTK == 0 ? 0 : r];
                    vx[r] = TM::make(xu[b][r].x, cs); vy[r] = TM::make(xu[b][r].y, cs);
                }
            } else {
                char* Wx8 = reinterpret_cast<char*>(A.Wx + (int64_t)(A.sig0 + pc.sg) * na * N);
                char* dWx8 = STORE_D ? reinterpret_cast<char*>(A.dWx + (int64_t)(A.sig0 + pc.sg) * na * N) : nullptr;
                const unsigned colc8 = (unsigned)colc * 8u;
                const int baddr = xbaddr[b];
#pragma unroll
                for (int r = 0; r < ((SSQ_TILE_EXP & 4) ? 0 : TILE_G); ++r) {
                    if (r == TILE_G / 2) {
                        // the next step: its data now (its records came in at the end of the step before)
                        load(BN, clampp(pn));
                        TILE_STAMP(trk, wv, pc.st, 1);
                        ticket_priority(turn, pc.S + pc.itl);
                    }
                    // (a, a') = sum_t (phi_t, phi'_t) u[q0 - 3 + t]  (baseband): real and imaginary
                    // parts as two packed accumulators (a_re, a'_re), (a_im, a'_im)
                    ssq_f2 are2, aim2;
                    if (r == 0) TILE_STAMP(trk, wv, pc.st, 4);
                    {
                        int fr[TILE_W], fi[TILE_W];
                        const int ur = __float_as_int(xu[b][r].x), ui = __float_as_int(xu[b][r].y);
#if SSQ_TILE_EXP & 32
#pragma unroll
                        for (int t = 0; t < TILE_W; ++t) { fr[t] = ur + t * baddr; fi[t] = ui ^ (t * baddr); }
#else
                        SSQ_BPERMUTE_OFF(fr[0], baddr, ur, 0);  SSQ_BPERMUTE_OFF(fi[0], baddr, ui, 0);
                        SSQ_BPERMUTE_OFF(fr[1], baddr, ur, 4);  SSQ_BPERMUTE_OFF(fi[1], baddr, ui, 4);
                        SSQ_BPERMUTE_OFF(fr[2], baddr, ur, 8);  SSQ_BPERMUTE_OFF(fi[2], baddr, ui, 8);
                        SSQ_BPERMUTE_OFF(fr[3], baddr, ur, 12); SSQ_BPERMUTE_OFF(fi[3], baddr, ui, 12);
                        SSQ_BPERMUTE_OFF(fr[4], baddr, ur, 16); SSQ_BPERMUTE_OFF(fi[4], baddr, ui, 16);
                        SSQ_BPERMUTE_OFF(fr[5], baddr, ur, 20); SSQ_BPERMUTE_OFF(fi[5], baddr, ui, 20);
                        SSQ_BPERMUTE_OFF(fr[6], baddr, ur, 24); SSQ_BPERMUTE_OFF(fi[6], baddr, ui, 24);
                        SSQ_BPERMUTE_OFF(fr[7], baddr, ur, 28); SSQ_BPERMUTE_OFF(fi[7], baddr, ui, 28);
                        SSQ_LDS_WAIT();
#endif
                        if (r == 0) TILE_STAMP(trk, wv, pc.st, 5);
#pragma unroll
                        for (int t = 0; t < TILE_W; ++t) {
                            ssq_f2 sv; sv.x = __int_as_float(fr[t]); sv.y = __int_as_float(fi[t]);
                            if (t == 0) { SSQ_PK_MUL_LO(are2, wt[0], sv); SSQ_PK_MUL_HI(aim2, wt[0], sv); }
                            else { SSQ_PK_FMA_LO(are2, wt[t], sv); SSQ_PK_FMA_HI(aim2, wt[t], sv); }
                        }
                    }
                    if (r == TILE_G - 1) {
                        // the taps of this step are done: the next step's weights, then the records of
                        // the step after it. Loads return in issue order: what the next step needs first
                        // (samples, weights) must not queue behind loads it needs later.
                        load_wt(BN, clampp(pn)); advance(pnn, NW); load_rec(clampp(pnn));
                    }
                    const float are = are2.x, aim = aim2.x;
                    float dre = are2.y, dim = aim2.y;
                    // d/dt of e^{i theta n} a(n):  e^{i theta n} (i theta a + a'),  theta = 2 pi kc / (M dt)
                    const int xrow = __builtin_amdgcn_readfirstlane(xq[b][r]);
                    const int kcs = (int)((unsigned)xrow >> 10);               // centre bin (wave-uniform)
                    const float theta = (float)kcs * A.theta_scale;
                    dre = __builtin_fmaf(-theta, aim, dre);
                    dim = __builtin_fmaf(theta, are, dim);
                    // e^{2 i pi kc n / M}: the phase kc n mod M is exact in integers and in float
                    // (M <= 2^24, checked by the host), v_sin_f32 / v_cos_f32 take revolutions (measured on
                    // the M = 2^18 circle: max abs error 1.2e-7, as good as a float table)
                    const float rev = (float)(__umul24((unsigned)kcs, (unsigned)nabs) & (unsigned)A.mmask) * A.inv_m;   // (both < 2^24: full-rate multiply)
#if SSQ_TILE_EXP & 64
                    const float2 Wv = make_float2(are + rev, aim), Dv = make_float2(dre, dim);
#else
                    const float2 tw = make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev));
                    const float2 Wv = cmulf(tw, make_float2(are, aim));
                    const float2 Dv = cmulf(tw, make_float2(dre, dim));
#endif
                    // (rows that only pad a step repeat the previous row -- same address, same value --
                    // and lanes past the last column repeat its point; neither contributes below)
                    const bool pad = (xrow >> 9) & 1;
                    const size_t rowoff = (size_t)((unsigned)xrow & 0x1FFu) * (nN * 8u);   // wave-uniform
                    if (!(SSQ_TILE_EXP & 256) || Wv.x == 123.456f) *reinterpret_cast<float2*>(Wx8 + rowoff + colc8) = Wv;
                    if (STORE_D) *reinterpret_cast<float2*>(dWx8 + rowoff + colc8) = Dv;
                    // phase transform and bin: as emit_point<LEAN> of the block kernels
#if SSQ_TILE_EXP & 128
                    int kout = (int)((unsigned)xrow & 0x1FFu);
                    if (Dv.x == 123.456f) kout = 0;
#else
                    const float cc = Wv.x, dd = Wv.y, aa = Dv.x, bb = Dv.y;
                    const float m2 = cc * cc + dd * dd, num = bb * cc - aa * dd;
                    const bool above = m2 > m2hi, below = m2 < m2lo;
                    const float w32 = fabsf(num * __builtin_amdgcn_rcpf(m2 * 6.2831855f));
                    bool ok;
                    const int kb = bin_screen_cwt<GRID>(w32, sp, omax, ok);
                    const int kf = (kb ^ fx) + fa;
                    const bool live = colok && !pad;
                    int kout = (above && live) ? kf : -1;
                    // undecided by the float32 screens (~0.05 % of the points, one row in 30): the exact
                    // double path
                    const bool pend = live && !(below | (above & ok));
                    if (__builtin_amdgcn_ballot_w64(pend)) {
                        if (pend) kout = exact_bin(Wv, Dv, sp, omax, A.gamma);
                    }
#endif
                    // (a point without contribution adds to the lane's scratch cell)
                    cell[r] = kout >= 0 ? kout * TILE_COLS + c : scratch;
                    const w_t cs = CSTK == 0 ? (w_t)A.cst0 : xc[b][CSTK == 0 ? 0 : r];
                    vx[r] = TM::make(Wv.x, cs); vy[r] = TM::make(Wv.y, cs);
                    if (r == 0) TILE_STAMP(trk, wv, pc.st, 6);
                    if (r == 1) TILE_STAMP(trk, wv, pc.st, 7);
                }
                if (SSQ_TILE_EXP & 4) {
                    load(BN, clampp(pn)); load_wt(BN, clampp(pn)); advance(pnn, NW); load_rec(clampp(pnn));
#pragma unroll
                    for (int r = 0; r < TILE_G; ++r) { cell[r] = scratch; vx[r] = term_t(0); vy[r] = term_t(0); }
                }
            }
            TILE_STAMP(trk, wv, pc.st, 2);
            // the step's update, in ticket order (tiles finished before it are written out first)
            Update4Prep up4;
            update4_prepare(cell, up4);
#pragma unroll
            for (int r = 0; r < TILE_G; ++r) { keep_term(vx[r]); keep_term(vy[r]); }
            write_outs_before(pc.itl);
            const int ticket = pc.S + pc.itl;
            float2 tcell[TILE_G];
            ticket_wait_read4(turn, ticket, lds_raw, up4, tcell);
            __builtin_amdgcn_wave_barrier();
            if (!(SSQ_TILE_EXP & 1)) update4_finish<TM>(lds_raw, up4, tcell, vx, vy);
            __builtin_amdgcn_wave_barrier();
            ticket_pass(turn, ticket + 1, c);
            TILE_STAMP(trk, wv, pc.st, 3);
            pl = pc; pc = pn; pn = pnn;
        };
        for (;;) {
            if (pc.S >= total) break;
            step(B0{}, B1{});
            if (pc.S >= total) break;
            step(B1{}, B0{});
        }
    }
    write_outs_before(ntl);
}


// =====================================================================================
// tile2_kernel (round 4): the same work without the ticket chain.
//
// What round 4 measured on the MI355X (profiles/r4_ab_history.txt): the ticketed kernel above
// spends a quarter of every tile at its boundary and is otherwise paced by the hand-overs (177 us
// of chain alone, 230 us of arithmetic alone, 256 us together); its time does not change when
// the gather, the modulation or the bin arithmetic are taken out, it is the same on 64 and on 256
// CUs (per tile), and LDS *float32* atomics, the obvious way around the tickets, take 193 cycles
// per wavefront instruction -- while ds_add_f64 takes 13.6 and ds_add_u64 10.8
// (tools/probes/lds_atomic_probe.hip).
//
// So the tile is kept in float64 and every wavefront adds its terms as soon as it has them
// (ds_add_f64, no return value): 16 bytes per cell, hence COLS = 32 columns per tile (16 when
// na > 318) and 64 / COLS consecutive rows per wavefront instruction (lane = sub-row h x column).
// Nothing orders the wavefronts inside a tile, so
//   * a wavefront owns a CONTIGUOUS block of the tile's rows (cost-balanced by the host), the
//     same block for every tile: consecutive rows share their decimation class, and the
//     interpolation weights of a class depend on the column only through n mod R -- the same
//     for every tile of a persistent workgroup whose tile stride (gridDim x COLS columns) is a
//     multiple of R: weights are re-read at class changes only, not per step;
//   * an item (= one wavefront instruction's rows) carries 16 bytes of state (one packed
//     record), the pipeline is: record two items ahead, samples one item ahead;
//   * a tile ends with two hardware barriers (all terms in / tile written out and cleared)
//     instead of 76 hand-overs.
// The sum of a cell is the float64 sum of its float32 (or float64) terms, rounded once: it
// differs from the reference's running float32 sum (algos.py:912-924) by that sum's own
// rounding, ~1e-7 of the largest cell (tests bound it at 1e-6); the bins are the same integers.
// float64 addition is not associative either, but with 300 terms of 24-bit mantissas the
// double sum's own rounding error is ~1e-16 relative: the float32 result differs between two
// arrival orders only when the exact sum lies within that of a float32 rounding boundary.
template <int COLS> struct Tile2Geo {
    static constexpr int RPI = 64 / COLS;              // rows per wavefront instruction
    static constexpr int LGC = COLS == 32 ? 5 : 4;
};
__host__ __device__ inline size_t tile2_lds_bytes(int64_t na, int cols) {
    return (size_t)(na + 1) * cols * 16;
}

struct Tile2Args {
    const int* items;        // [n_items][8]: row0 | npad << 9 | kind << 12 | lgR << 13, samples' offset of sub-row 0
                             // (class + row), row0 * N * 8, entries between two signals' rows of the class, kc of the sub-rows
    const int4* waves;       // [NW]: first item, end, first item of the wavefront's second class (= end: none), 0
    const float4* wtab; const float2* U;
    const void* cst;
    float2* Wx; float2* dWx; float2* Tx; const unsigned short* kidx;
    unsigned short* kdump;   // STORE_K builds: the bin of every point as it is consumed, (signal, row, column); else null
    int64_t N, na;
    int n_items, n1, mmask, lgM, sig0, nsig, group;
    int carry;                                       // the walk b, b + G, ... runs through the signals' boundaries
    float inv_m, theta_scale, cst0;
    unsigned long long* counters;
    double gamma;
};

// tuning aid (-DSSQ_TILE2_PROF, A/B builds): shader-clock time one workgroup's wavefronts spend in the
// phases of an item, summed over the launch -> counters[8 + 8 * wavefront + phase] (dumped by TilePlan::run
// with SSQ_TILE2_PROF_DUMP=1 in the environment). Each stamp waits for the LDS / scalar queue: perturbs.
#ifdef SSQ_TILE2_PROF
#define T2_STAMP(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         prof[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define T2_STAMP(k) do { } while (0)
#endif

// Both tile kernels are bound by the instructions they issue, of every kind (round 4,
// profiles/r4_ab_history.txt: one instruction per cycle and CU; 210 per 64 points in the ticketed
// kernel). This one is built to issue few:
//   * an item (64 / COLS consecutive rows x COLS columns) has ONE scalar record (s_load through the
//     constant address space: the index is wavefront-uniform): the sub-rows are consecutive rows of
//     one class, so a lane's addresses are scalar bases + per-lane constants;
//   * one load of samples (or Wx + bin for rows read back) and one store of Wx per item, one
//     16-byte-per-lane store of Tx per 4 (8) rows x COLS columns of a finished tile;
//   * the interpolation weights stay in registers for the whole launch: a wavefront's block of
//     rows spans at most two decimation classes (the host cuts the blocks that way), and a lane's
//     weights depend on its column only through n mod R, the same for every tile of a workgroup
//     whose tile stride is a multiple of R (the launcher picks the grid that way).
// STORE_K (diagnostic builds, ssq_cwt_plan_set_bin_dump): every point's bin index goes to A.kdump as the
// reassignment consumes it -- what pins the kernel's index work as integers against the oracle's map.
template <int GRID, bool STORE_D, int NW, int CSTK, int COLS, bool STORE_K = false>
__global__ __launch_bounds__(64 * NW) void tile2_kernel(Tile2Args A, SsqParams sp) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int RPI = Tile2Geo<COLS>::RPI, LGC = Tile2Geo<COLS>::LGC;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & (COLS - 1), h = lane >> LGC, hb4 = (lane & ~(COLS - 1)) * 4;
    const int64_t N = A.N;
    const unsigned nN = (unsigned)N;
    const int na = (int)A.na, omax = na - 1;
    double2* T = reinterpret_cast<double2*>(lds_raw);          // (na + 1) x COLS cells, the last row: scratch
    for (int k = threadIdx.x; k < (na + 1) * COLS; k += 64 * NW) T[k] = make_double2(0.0, 0.0);
    __syncthreads();
    // (LDS byte addresses of the lane's column in row 0 and in the scratch row)
    const int c16 = c * 16 + (int)SSQ_LDS_ADDR(lds_raw);
    const int scratch16 = na * COLS * 16 + c16;
    const int full_rounds = na / (NW * RPI);                   // write-out rounds (NW * RPI rows each) that are complete

    const int ntx = (int)((N + COLS - 1) / COLS);
    const int G = (int)gridDim.x;
    // Workgroup b walks tiles b, b + G, ... -- of each signal (then a signal's last round is short for the
    // workgroups past ntx mod G, launch after launch: 304 against 320 tiles at config 2), or, A.carry, of the
    // signals laid end to end (the launcher allows it when the lanes' weights survive the boundary).
    const int per_sig = (int)blockIdx.x < ntx ? (ntx - (int)blockIdx.x + G - 1) / G : 0;
    const int ntl = A.carry ? (int)(((int64_t)A.nsig * ntx - (int)blockIdx.x + G - 1) / G)
                            : per_sig * A.nsig;                // tiles of this workgroup
    const auto* waves = SSQ_CONST_PTR(int4, A.waves);
    const int i0 = waves[wv].x, i1 = waves[wv].y, isp = waves[wv].z, ni = i1 - i0;
    // The wavefronts of a SIMD compete for its issue slots and the oldest wins: left alone, the four
    // youngest wavefronts of the workgroup finish their rows of every tile last and the others wait
    // for them at the barrier (measured: 14.7 k of 37 k cycles per tile); fixed priorities against the
    // age only turn the order around. So the priorities rotate: the four wavefronts of a SIMD (w, w + 4,
    // w + 8, w + 12) alternate between two levels, two high and two low at any time, swapped with every
    // item (+2 %; four rotating levels measured the same and cost three more branches per item).
    int prio = (wv >> 2) & 1;
    auto rotate_priority = [&]() {
#if !(SSQ_TILE_EXP & 8)
        SSQ_PRIO_TOGGLE(prio);                                 // (two levels, swapped with every item)
#endif
    };
    const float g2 = (float)(A.gamma * A.gamma);
    const float m2hi = g2 * 1.000004f, m2lo = g2 * 0.999996f;
    const int fx = sp.flipud ? -1 : 0, fa = sp.flipud ? na : 0;
    using TM = TileTerm<CSTK == 2>;
    using w_t = typename TM::wtype;
    const auto* cstv = SSQ_CONST_PTR(w_t, A.cst);

    // ---- a tile's end: all terms in (barrier), every wavefront writes its share of the rows to
    // Tx and clears them, tile free again (barrier). A lane takes two neighbouring columns of a row:
    // one 16-byte store, a wavefront instruction = 128 / COLS rows (N even; otherwise column by column).
    auto finish_tile = [&](int tx, int sg) {
        SSQ_WG_BARRIER();
        float2* Tx = A.Tx + (int64_t)(A.sig0 + sg) * na * N;
        constexpr int NA_CAP = COLS == 32 ? 320 : 512;
#ifndef SSQ_TILE2_NO16
#define SSQ_TILE2_NO16 1
#endif
        if (!SSQ_TILE2_NO16 && (N & 1) == 0) {
            constexpr int HC = COLS / 2, RW = 64 / HC;         // column pairs per row, rows per instruction
            const int j2 = (lane & (HC - 1)) * 2, rr = lane / HC;
            const unsigned col = (unsigned)(tx * COLS + j2);
            const bool ok = col < nN;                          // (N even: the pair is inside or outside together)
            constexpr int ROUNDS = (NA_CAP + NW * RW - 1) / (NW * RW);
#pragma unroll
            for (int m = 0; m < ROUNDS; ++m) {
                if (m * NW * RW < na) {                        // (wave-uniform: rounds past the last row fall away)
                    const int k = (wv + m * NW) * RW + rr;
                    const int kc_ = k < na ? k : na;
                    const double2 v0 = T[kc_ * COLS + j2], v1 = T[kc_ * COLS + j2 + 1];
                    T[kc_ * COLS + j2] = make_double2(0.0, 0.0);       // (the scratch row is cleared along the way)
                    T[kc_ * COLS + j2 + 1] = make_double2(0.0, 0.0);
                    if (ok && k < na)
                        *reinterpret_cast<float4*>(Tx + (unsigned)k * nN + col) =
                            make_float4((float)v0.x, (float)v0.y, (float)v1.x, (float)v1.y);
                    asm volatile("" ::: "memory");             // (keeps the rounds from being batched into registers)
                }
            }
        } else {
            constexpr int ROUNDS = (NA_CAP + NW * RPI - 1) / (NW * RPI), RR = NW * RPI;    // RR rows per round
            const int k0 = wv * RPI + h;                       // the lane's row in round 0
            if ((tx + 1) * COLS <= (int)nN) {
                // every column of the tile exists (all but a signal's last tile when COLS does not divide N):
                // the rounds below the last need no masks -- scalar base per round + a per-lane constant
                char* tb = reinterpret_cast<char*>(Tx) + (size_t)tx * (COLS * 8);
                const unsigned voff = ((unsigned)k0 * nN + (unsigned)c) * 8u;
                // (the round count and the rounds' distance are re-read as scalars at every use: hoisted out
                // of the item loop, the compiler keeps ten lane masks and ten 64-bit offsets in spilled registers)
                int fr = full_rounds;
                size_t step = (size_t)RR * (size_t)N * 8;
#pragma unroll
                for (int m = 0; m < ROUNDS - 1; ++m) {
                    SSQ_OPAQUE_S(fr); SSQ_OPAQUE_S(step);
                    if (m < fr) {                              // (wave-uniform)
                        const int k = k0 + m * RR;
                        const double2 v = T[k * COLS + c];
                        T[k * COLS + c] = make_double2(0.0, 0.0);
                        if (!(SSQ_TILE_EXP & 4096) || v.x == 123.0)
                            *reinterpret_cast<float2*>(tb + (size_t)voff) = make_float2((float)v.x, (float)v.y);
                        tb += step;
                        asm volatile("" ::: "memory");         // (keeps the rounds from being batched into registers)
                    }
                }
                {   // the last round: the rows left, and the scratch row cleared by the lanes past them
                    const int k = k0 + fr * RR;
                    const int kc_ = k < na ? k : na;
                    const double2 v = T[kc_ * COLS + c];
                    T[kc_ * COLS + c] = make_double2(0.0, 0.0);
                    if (k < na && (!(SSQ_TILE_EXP & 4096) || v.x == 123.0))
                        *reinterpret_cast<float2*>(tb + (size_t)voff) = make_float2((float)v.x, (float)v.y);
                }
            } else {
                const unsigned col = (unsigned)(tx * COLS + c);
                const bool ok = col < nN;
#pragma unroll 1
                for (int m = 0; m * RR < na + 1; ++m) {
                    const int k = k0 + m * RR;
                    const int kc_ = k < na ? k : na;
                    const double2 v = T[kc_ * COLS + c];
                    T[kc_ * COLS + c] = make_double2(0.0, 0.0);
                    if (ok && k < na) Tx[(unsigned)k * nN + col] = make_float2((float)v.x, (float)v.y);
                }
            }
        }
        if (threadIdx.x == 0 && A.counters)
            __scoped_atomic_fetch_add(A.counters, 1ull, __ATOMIC_RELAXED, __MEMORY_SCOPE_DEVICE);
        SSQ_WG_BARRIER();
    };

    if (ni <= 0) {                                             // more wavefronts than items: write-outs only
        int tx = (int)blockIdx.x, sg = 0;
        for (int j = 0; j < ntl; ++j) {
            finish_tile(tx, sg);
            tx += G;
            if (tx >= ntx) { tx = A.carry ? tx - ntx : (int)blockIdx.x; ++sg; }
        }
        return;
    }

    // ---- the wavefront's sequence of (tile, item) positions, software-pipelined over a ring of three
    // data slots (the loop is unrolled three times, the slots are compile-time): while position p is
    // computed, the data of p + 2 go out. Two cursors walk the same sequence, the loads' two positions
    // ahead of the arithmetic's; each is an item index and the tile as the kernel uses it: n of the
    // tile's first column (n1 + first column), the signal, and the byte offset of (signal, row 0, first
    // column) in Wx -- moved by constants when the cursor's item index wraps (no 64-bit products, and no
    // position records copied around per item). Past the last tile the loads' cursor stays on it (all
    // loads unconditional, see the note in tile_kernel: what they fetch there is valid and unused).
    struct Pos { int nabs0, sg; int64_t off8; };
    const int nabs_step = G * COLS, nabs_first = A.n1 + (int)blockIdx.x * COLS, nabs_last = A.n1 + (ntx - 1) * COLS;
    const int64_t off8_step = (int64_t)G * COLS * 8;
    // (a signal's end: back to the workgroup's first tile, or -- carry -- on by the same stride into the next signal)
    const int64_t off8_wrap = A.carry ? ((int64_t)na * N + (int64_t)(G - ntx) * COLS) * 8
                                      : ((int64_t)na * N - (int64_t)(per_sig - 1) * G * COLS) * 8;
    const int nabs_back = ntx * COLS;
    auto next_tile = [&](Pos q) {
        Pos r = q;
        r.nabs0 += nabs_step;
        const bool wrap = r.nabs0 > nabs_last;
        r.off8 += wrap ? off8_wrap : off8_step;
        if (wrap) { r.nabs0 = A.carry ? r.nabs0 - nabs_back : nabs_first; ++r.sg; }
        return (wrap && r.sg >= A.nsig) ? q : r;               // (the tile after the last: the last)
    };
    const int total = ntl * ni;                                // positions of this wavefront
    typedef int int8v __attribute__((ext_vector_type(8)));
    const auto* items = SSQ_CONST_PTR(int8v, A.items);
    // per-lane constants of the addresses: the lane's place inside an item's rows
    const unsigned lane_row8 = (unsigned)h * nN * 8u + (unsigned)c * 8u;       // bytes: sub-row h, column c
    const unsigned lane_col8 = (unsigned)c * 8u;

    // data of a position: (interpolated) the lane's sample of its sub-row's window, or (rows read
    // back) Wx and the bin of the lane's point
    struct Data { float2 u; int kq; };
    const char* const U8 = reinterpret_cast<const char*>(A.U);
    const char* const WX8 = reinterpret_cast<const char*>(A.Wx) + (size_t)((int64_t)A.sig0 * na * N) * 8u;
    const char* const KX8 = reinterpret_cast<const char*>(A.kidx);
    // (ANY0: the wavefront's block holds rows read back; a wavefront of interpolated rows only -- most
    // are -- runs a loop without the bin load and the kind tests: one vector-memory instruction less per
    // item, and the CU's vector-memory path takes one wavefront instruction per ~20 cycles)
    auto load_data = [&](auto any0, const int8v R, const Pos& q) {
        constexpr bool ANY0 = decltype(any0)::value;
        Data d;
        const int w0 = R[0];
        const int kind = ANY0 ? (w0 >> 12) & 1 : 1;
        const char* base; unsigned voff;
        const char* kbase = reinterpret_cast<const char*>(A.items); unsigned koff = (unsigned)lane * 2u;
        if (kind) {                                            // (wave-uniform; the loads themselves stay outside)
            // sample (qb + min(c, wlast)) mod L of row h of the item, h * L entries on
            const int lgR = (w0 >> 13) & 31;
            const int qb = (q.nabs0 >> lgR) - (TILE_W / 2 - 1);
            const int wlast = ((COLS - 1) >> lgR) + TILE_W;
            const int lmask = A.mmask >> lgR;                  // L - 1, L = M / R
            voff = (((unsigned)((qb + (c < wlast ? c : wlast)) & lmask)) + ((unsigned)h << (A.lgM - lgR))) * 8u;
            base = U8 + ((size_t)(unsigned)R[1] + (size_t)((unsigned)q.sg * (unsigned)R[3])) * 8u;
        } else {
            // point (row0 + h, column) -- the last column's for lanes past it, the last real row's for
            // padded sub-rows -- and its bin
            const int npad = (w0 >> 9) & 7;
            unsigned lr8 = lane_row8;
            if (RPI == 2) { if (npad) lr8 = lane_col8; }
            else if (npad) lr8 = (unsigned)min(h, RPI - 1 - npad) * nN * 8u + lane_col8;
            if (q.nabs0 == nabs_last) {                        // (the last tile may be partial)
                const int col = q.nabs0 - A.n1 + c;
                if (col >= (int)N) lr8 -= (unsigned)(col - ((int)N - 1)) * 8u;
            }
            voff = lr8;
            base = WX8 + ((size_t)q.off8 + (unsigned)R[2]);
            kbase = KX8 + (((size_t)q.off8 + (unsigned)R[2]) >> 2);
            koff = lr8 >> 2;
        }
        d.u = *reinterpret_cast<const float2*>(base + (size_t)voff);
        // (the bin: a load either way, from a harmless address for interpolated rows -- a conditional
        // load costs the compiler its count of loads in flight)
        if constexpr (ANY0) d.kq = (int)*reinterpret_cast<const unsigned short*>(kbase + (size_t)koff);
        else d.kq = 0;
        return d;
    };
    // the weights of the wavefront's (up to) two classes, for the lane's column phase: once
    ssq_f2 wta[TILE_W], wtb[TILE_W];
    auto load_wt = [&](ssq_f2 (&wt)[TILE_W], int it, int woff) {
        const int lgR = (items[it][0] >> 13) & 31;
        const int nabs = A.n1 + (int)blockIdx.x * COLS + c;    // (every tile of this workgroup: the same n mod R)
        const int R = 1 << lgR;
        const float4* wp = A.wtab + (int64_t)woff * 4 + (nabs & (R - 1));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = wp[t * R];
            wt[2 * t].x = v.x; wt[2 * t].y = v.y; wt[2 * t + 1].x = v.z; wt[2 * t + 1].y = v.w;
        }
    };
    load_wt(wta, i0, waves[wv].w & 0xFFFF);
    load_wt(wtb, isp < i1 ? isp : i0, (int)((unsigned)waves[wv].w >> 16));

#ifdef SSQ_TILE2_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    using Yes = std::true_type; using No = std::false_type;
    auto run = [&](auto any0) {
    constexpr bool ANY0 = decltype(any0)::value;
    Data D[3];
    Pos tc, tl;                                                // the tile of the arithmetic's cursor, of the loads'
    tc.nabs0 = nabs_first; tc.sg = 0; tc.off8 = (int64_t)blockIdx.x * COLS * 8;
    tl = tc;
    if (total <= 0) return;
    int it_c = i0, it_l = i0;
    bool tc_last = tc.nabs0 == nabs_last && (N & (COLS - 1)) != 0;   // the arithmetic's tile is a signal's last, partial one
    int left = total;                                          // positions not yet finished
    auto step_loads = [&]() { if (++it_l >= i1) { it_l = i0; tl = next_tile(tl); } };
    // the records of the position in hand and of the one whose data go out next: asked for (through
    // the scalar cache) at the end of the position before, so that they are there when it starts
    int8v Rc = items[i0];
    D[0] = load_data(any0, Rc, tl);
    step_loads();
    D[1] = load_data(any0, items[it_l], tl);
    step_loads();
    // (the third slot: position 0 again -- a load like the loop's, so that the compiler's count of the loads in
    // flight at the loop's head is the loop's own; a plain copy made the first body wait for one load too many)
    D[2] = load_data(any0, Rc, tc);
    int8v Rn = items[it_l];
    // the per-row reassignment weights of the position in hand (scalar loads, asked for with its records)
    w_t csn[RPI];
    auto load_cs = [&](int row0) {
        if (CSTK != 0) {
#pragma unroll
            for (int k = 0; k < RPI; ++k) csn[k] = cstv[min(row0 + k, omax)];
        }
    };
    load_cs(Rc[0] & 0x1FF);
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    bool more = true;
    auto body = [&](auto KK) {
        constexpr int k0 = decltype(KK)::value, k1 = (k0 + 1) % 3, k2 = (k0 + 2) % 3;
        const Pos pc = tc;
        T2_STAMP(0);                                           // (loop overhead, the previous item's tail)
        rotate_priority();
        if (!(SSQ_TILE_EXP & 32768)) D[k2] = load_data(any0, Rn, tl);         // the data of p + 2
        step_loads();
#ifdef SSQ_TILE2_STRICT                                        // (A/B builds: every body waits like the loop's first)
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
#endif
        T2_STAMP(1);                                           // loads of p + 2 issued
        const Data dc = D[k0];
        const int w0 = Rc[0];
        const int npad = (w0 >> 9) & 7, kind = ANY0 ? (w0 >> 12) & 1 : 1;
        const int nabs = pc.nabs0 + c;                         // (lanes past the last column: results unused)
        // (every lane's point counts, except in a class's last item -- padded sub-rows -- and in the last
        // tile of a signal when N is not a multiple of the tile: a wave-uniform test keeps the rest free)
        bool livept = true;
        if ((w0 & 0xE00) != 0 || tc_last) livept = h < RPI - npad && nabs - A.n1 < (int)N;
        int cell16; float tvx, tvy;
        if (kind == 0) {
            const int kk = dc.kq & 0xFFFF;
            cell16 = (livept && kk != TILE_NOBIN) ? kk * (COLS * 16) + c16 : scratch16;
            tvx = dc.u.x; tvy = dc.u.y;
            if constexpr (STORE_K) {
                char* kd8 = reinterpret_cast<char*>(A.kdump) + (((size_t)((int64_t)A.sig0 * na * N) * 8u + (size_t)pc.off8 + (unsigned)Rc[2]) >> 2);
                if (livept) *reinterpret_cast<unsigned short*>(kd8 + (size_t)(lane_row8 >> 2)) = (unsigned short)kk;
            }
        } else {
            const int lgR = (w0 >> 13) & 31;
            const int qb3 = pc.nabs0 >> lgR;                   // window start + 3: tap 0 of sample q0 sits in lane q0 - qb3
            const int baddr = (((nabs >> lgR) - qb3) << 2) + hb4;
            ssq_f2 A2, D2;
            {
                int fr[TILE_W], fi[TILE_W];
                int ur = __float_as_int(dc.u.x), ui = __float_as_int(dc.u.y);
#ifdef SSQ_TILE2_PROF
                SSQ_OPAQUE_V(ur); SSQ_OPAQUE_V(ui);
                T2_STAMP(2);                                   // the item's samples are there
#endif
#if SSQ_TILE_EXP & 1024
                for (int t = 0; t < TILE_W; ++t) { fr[t] = ur + t; fi[t] = ui - t; }
#else
                SSQ_BPERMUTE_OFF(fr[0], baddr, ur, 0);  SSQ_BPERMUTE_OFF(fi[0], baddr, ui, 0);
                SSQ_BPERMUTE_OFF(fr[1], baddr, ur, 4);  SSQ_BPERMUTE_OFF(fi[1], baddr, ui, 4);
                SSQ_BPERMUTE_OFF(fr[2], baddr, ur, 8);  SSQ_BPERMUTE_OFF(fi[2], baddr, ui, 8);
                SSQ_BPERMUTE_OFF(fr[3], baddr, ur, 12); SSQ_BPERMUTE_OFF(fi[3], baddr, ui, 12);
                SSQ_BPERMUTE_OFF(fr[4], baddr, ur, 16); SSQ_BPERMUTE_OFF(fi[4], baddr, ui, 16);
                SSQ_BPERMUTE_OFF(fr[5], baddr, ur, 20); SSQ_BPERMUTE_OFF(fi[5], baddr, ui, 20);
                SSQ_BPERMUTE_OFF(fr[6], baddr, ur, 24); SSQ_BPERMUTE_OFF(fi[6], baddr, ui, 24);
                SSQ_BPERMUTE_OFF(fr[7], baddr, ur, 28); SSQ_BPERMUTE_OFF(fi[7], baddr, ui, 28);
                SSQ_LDS_WAIT();
#endif
                T2_STAMP(3);                                   // taps gathered
#if SSQ_TILE_EXP & 16384
                A2.x = __int_as_float(fr[0] ^ fr[7]); A2.y = __int_as_float(fi[0] ^ fi[7]); D2.x = __int_as_float(fr[1] ^ fr[6]); D2.y = __int_as_float(fi[2] ^ fi[5] ^ fr[3] ^ fr[4] ^ fi[1]);
                if (fr[2] == 77 && fr[5] == 78 && fi[3] == 7 && fi[4] == 7 && fi[6] == 1 && fi[7] == 0) A2.x = wta[0].x + wtb[3].y;
#endif
                // (A2 = (a_re, a_im), D2 = (a'_re, a'_im): the pairs the modulation multiplies)
                ssq_f2 sv[TILE_W];
#pragma unroll
                for (int t = 0; t < TILE_W; ++t) { sv[t].x = __int_as_float(fr[t]); sv[t].y = __int_as_float(fi[t]); }
                if ((SSQ_TILE_EXP & 16384) == 0 && it_c < isp) {   // (wave-uniform: the wavefront's first or second class)
                    SSQ_TAPS8(A2, D2, wta, sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]);
                } else if ((SSQ_TILE_EXP & 16384) == 0) {
                    SSQ_TAPS8(A2, D2, wtb, sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]);
                }
            }
            int kcs = Rc[4];                                    // centre bin of the lane's row
#pragma unroll
            for (int k = 1; k < RPI; ++k) if (h == k) kcs = Rc[4 + k];
            const float theta = (float)kcs * A.theta_scale;
            // d/dt of e^{i theta n} a(n):  e^{i theta n} (i theta a + a')
            D2.x = __builtin_fmaf(-theta, A2.y, D2.x);
            D2.y = __builtin_fmaf(theta, A2.x, D2.y);
            const float rev = (float)(__umul24((unsigned)kcs, (unsigned)nabs) & (unsigned)A.mmask) * A.inv_m;
            ssq_f2 tw2, W2, V2;
            tw2.x = __builtin_amdgcn_cosf(rev); tw2.y = __builtin_amdgcn_sinf(rev);
            SSQ_CMUL_PK(W2, tw2, A2);
            SSQ_CMUL_PK(V2, tw2, D2);
            const float2 Wv = make_float2(W2.x, W2.y);
            const float2 Dv = make_float2(V2.x, V2.y);
            // (lanes past the last column hold another column's weights, padded sub-rows another row's
            // samples: their values go nowhere)
            char* wx8 = const_cast<char*>(WX8) + ((size_t)pc.off8 + (unsigned)Rc[2]);
            if (livept && (!(SSQ_TILE_EXP & 256) || Wv.x == 123.456f)) *reinterpret_cast<float2*>(wx8 + (size_t)lane_row8) = Wv;
            if (STORE_D) {
                char* dwx8 = reinterpret_cast<char*>(A.dWx) + ((size_t)((int64_t)A.sig0 * na * N) * 8u + (size_t)pc.off8 + (unsigned)Rc[2]);
                if (livept) *reinterpret_cast<float2*>(dwx8 + (size_t)lane_row8) = Dv;
            }
            // phase transform and bin: as emit_point<LEAN> of the block kernels
            const float cc = Wv.x, dd = Wv.y, aa = Dv.x, bb = Dv.y;
            const float m2 = cc * cc + dd * dd, num = bb * cc - aa * dd;
            const bool above = m2 > m2hi, below = m2 < m2lo;
            const float w32 = fabsf(num * __builtin_amdgcn_rcpf(m2 * 6.2831855f));
#if SSQ_TILE_EXP & 512
            int kout = (livept && w32 != 123.f) ? (kcs & 255) : -1;
#else
            bool ok;
            const int kb = bin_screen_cwt<GRID>(w32, sp, omax, ok);
            const int kf = (kb ^ fx) + fa;
            int kout = (above && livept) ? kf : -1;
            const bool pend = livept && !(below | (above & ok));
            if (pend) kout = exact_bin(Wv, Dv, sp, omax, A.gamma);
#endif
            cell16 = kout >= 0 ? kout * (COLS * 16) + c16 : scratch16;
            tvx = Wv.x; tvy = Wv.y;
            if constexpr (STORE_K) {
                char* kd8 = reinterpret_cast<char*>(A.kdump) + (((size_t)((int64_t)A.sig0 * na * N) * 8u + (size_t)pc.off8 + (unsigned)Rc[2]) >> 2);
                if (livept) *reinterpret_cast<unsigned short*>(kd8 + (size_t)(lane_row8 >> 2)) = (unsigned short)(kout >= 0 ? kout : TILE_NOBIN);
            }
            T2_STAMP(4);                                       // arithmetic, store, bin
        }
        {
            w_t cs = (w_t)A.cst0;
            if (CSTK != 0) {
                cs = csn[0];
#pragma unroll
                for (int k = 1; k < RPI; ++k) if (h == k) cs = csn[k];
            }
            const double ax = (double)TM::make(tvx, cs), ay = (double)TM::make(tvy, cs);
#if SSQ_TILE_EXP & 2048
            if (ax == 123.0 && cell16 == 7) { SSQ_LDS_ADD_F64_AT(cell16, 0, ax); SSQ_LDS_ADD_F64_AT(cell16, 8, ay); }
#else
            SSQ_LDS_ADD_F64_AT(cell16, 0, ax);
            SSQ_LDS_ADD_F64_AT(cell16, 8, ay);
#endif
        }
        T2_STAMP(5);                                           // terms added
        more = --left > 0;
        const bool tile_end = ++it_c >= i1;                    // (the block's last item: the tile is complete)
        if (tile_end) it_c = i0;
        Rc = items[it_c];                                      // the next position's records (see above)
        Rn = items[it_l];
        load_cs(Rc[0] & 0x1FF);                                // ... and its rows' weights, when there is one per row
        if (tile_end) {
            if (!(SSQ_TILE_EXP & 8192) || !more) { finish_tile((pc.nabs0 - A.n1) >> LGC, pc.sg); T2_STAMP(6); }
            tc = next_tile(tc);
            tc_last = tc.nabs0 == nabs_last && (N & (COLS - 1)) != 0;
        }
    };
#ifdef SSQ_TILE2_PEEL
    body(K0{});
    if (more) for (;;) {
        body(K1{}); if (!more) break;
        body(K2{}); if (!more) break;
        body(K0{}); if (!more) break;
    }
#else
    for (;;) {
        body(K0{}); if (!more) break;
        body(K1{}); if (!more) break;
        body(K2{}); if (!more) break;
    }
#endif
    };
    // (a block spans at most two classes: its first item and the first of its second class tell)
    const bool has0 = !((items[i0][0] >> 12) & 1) || (isp < i1 && !((items[isp][0] >> 12) & 1));
    // (measured: a second loop without the bin load and the kind tests for the wavefronts of interpolated
    // rows only -- one vector-memory instruction and ten scalar ones less per item -- is SLOWER, 230 vs 220 us)
#ifdef SSQ_TILE2_TWOLOOPS
    if (has0) run(Yes{}); else run(No{});
#else
    (void)has0;
    run(Yes{});
#endif
#ifdef SSQ_TILE2_PROF
    if (blockIdx.x == 100 % gridDim.x && lane == 0 && A.counters)
        for (int k = 0; k < 8; ++k) A.counters[8 + 8 * wv + k] = prof[k];
#endif
}

// ---------------------------------------------------------------------------- host side
int tile_rows_per_step() { return TILE_G; }
int TilePlan::create(const ssq_cwt_tiles_desc& d, int64_t M_, int64_t N_, int64_t n1_, int64_t na_, int group_,
                     double dt_, int64_t& bytes) {
    M = M_; N = N_; n1 = n1_; na = na_; group = group_; dt = dt_;
    nsegs = d.n_segs; nsteps = d.n_steps; n_irows = d.n_irows; u_total = d.u_total;
    SSQ_REQUIRE(nsegs >= 1 && nsteps >= 1 && n_irows >= 1 && d.n_classes >= 1, "empty tile tables");
    SSQ_REQUIRE((d.reserved ? d.reserved : 4) == TILE_G, "tile tables hold %d rows per step, this build walks %d",
                d.reserved ? d.reserved : 4, TILE_G);
    {
        int dev = 0; hipDeviceProp_t pr;
        SSQ_CHECK_HIP(hipGetDevice(&dev));
        SSQ_CHECK_HIP(hipGetDeviceProperties(&pr, dev));
        ncu = pr.multiProcessorCount;
        if (const char* e = getenv("SSQ_TILE_GRID")) if (atoi(e) > 0) ncu = atoi(e);
        // both tile kernels keep a tile of up to 160 KB in a workgroup's LDS (gfx950); a device with
        // less refuses the tile path here instead of failing at the first launch
        SSQ_REQUIRE((size_t)pr.maxSharedMemoryPerMultiProcessor >= 160 * 1024,
                    "the tile path needs 160 KB of LDS per workgroup, the device has %zu",
                    (size_t)pr.maxSharedMemoryPerMultiProcessor);
    }
    SSQ_REQUIRE(na * N < ((int64_t)1 << 29) && na < 512, "na = %lld, N = %lld: outside the tile path's 32-bit offsets",
                (long long)na, (long long)N);
    // the modulation phase kc * n mod M is formed with a 24-bit multiply and carried in a float
    // (and the centre bin, < M / 2, shares a word with the row: 22 bits)
    SSQ_REQUIRE(M <= ((int64_t)1 << 23), "the tile path needs a padded length <= 2^23");
    SSQ_REQUIRE((int64_t)group * u_total < ((int64_t)1 << 31), "tile intermediates exceed 2^31 entries");
    auto up = [&](void** dst, const void* src, size_t nbytes) -> int {
        SSQ_CHECK_HIP(hipMalloc(dst, nbytes ? nbytes : 1));
        if (nbytes) SSQ_CHECK_HIP(hipMemcpy(*dst, src, nbytes, hipMemcpyHostToDevice));
        bytes += (int64_t)nbytes;
        return 0;
    };
    int rc;
    static_assert(sizeof(TileSeg) == 32 && sizeof(TileRow) == 16 && sizeof(TileIRow) == 32, "table layout");
    {   // one packed record per step (a wavefront's consecutive steps are usually of different
        // segments): kind | log2 R << 1 | weight-table offset << 8, L - 1, entries between two
        // signals' rows of the class, entries before the class
        std::vector<int32_t> hs((size_t)nsteps * 4);
        const TileSeg* sg = reinterpret_cast<const TileSeg*>(d.segs);
        int64_t covered = 0;
        for (int i = 0; i < nsegs; ++i) {
            SSQ_REQUIRE(sg[i].first == covered && sg[i].nsteps >= 1 && sg[i].first + sg[i].nsteps <= nsteps,
                        "tile segment %d does not continue the step list", i);
            SSQ_REQUIRE(sg[i].kind == 0 || sg[i].kind == 1, "tile segment %d: bad kind", i);
            SSQ_REQUIRE(sg[i].lgR >= 0 && sg[i].lgR < 32 && sg[i].wtab_off >= 0 && sg[i].wtab_off < (1 << 23),
                        "tile segment %d does not fit its packed record", i);
            for (int t = 0; t < sg[i].nsteps; ++t) {
                int32_t* q = &hs[((size_t)sg[i].first + t) * 4];
                q[0] = sg[i].kind | (sg[i].lgR << 1) | (sg[i].wtab_off << 8);
                q[1] = sg[i].lmask; q[2] = sg[i].sig_stride; q[3] = sg[i].cls_base;
            }
            covered += sg[i].nsteps;
        }
        SSQ_REQUIRE(covered == nsteps, "tile segments cover %lld of %d steps", (long long)covered, nsteps);
        if ((rc = up((void**)&steps, hs.data(), hs.size() * 4))) return rc;
    }
    {   // row records as the kernel reads them: row | padding << 9 | centre bin << 10, offset of the
        // row's samples inside its class
        const TileRow* rw = reinterpret_cast<const TileRow*>(d.rows);
        std::vector<int32_t> hp((size_t)nsteps * TILE_G * 2);
        for (size_t i = 0; i < (size_t)nsteps * TILE_G; ++i) {
            const int32_t row = rw[i].row & 0xFFFF;
            SSQ_REQUIRE(row < 512 && rw[i].kc >= 0 && rw[i].kc < (1 << 22), "tile row %zu does not fit its packed record", i);
            hp[2 * i] = row | (rw[i].row < 0 ? 0x200 : 0) | (int32_t)((uint32_t)rw[i].kc << 10);
            hp[2 * i + 1] = rw[i].ubase;
        }
        if ((rc = up((void**)&rows, hp.data(), hp.size() * 4))) return rc;
    }
    {   // tile2_kernel's items: 64 / COLS consecutive rows of a step, one record of 8 words per item:
        // row0 | padded sub-rows << 9 | kind << 12 | lgR << 13, samples of sub-row 0 (class offset + row
        // in class * L), row0 * N * 8, entries between two signals' rows of the class, centre bins
        const TileRow* rw = reinterpret_cast<const TileRow*>(d.rows);
        const TileSeg* sg = reinterpret_cast<const TileSeg*>(d.segs);
        cols2 = tile2_lds_bytes(na, 32) <= 160 * 1024 ? 32 : 16;
        if (const char* e = getenv("SSQ_TILE2_COLS")) if (atoi(e) == 16) cols2 = 16;     // (tuning aid)
        SSQ_REQUIRE(tile2_lds_bytes(na, cols2) <= 160 * 1024, "na = %lld: the Tx tile exceeds the LDS", (long long)na);
        const int rpi = 64 / cols2;
        SSQ_REQUIRE(TILE_G % rpi == 0, "tile tables: %d rows per step, %d per item", TILE_G, rpi);
        n_items2 = nsteps * TILE_G / rpi;
        std::vector<int32_t> hi8((size_t)n_items2 * 8, 0);
        std::vector<float> cost((size_t)n_items2, 0.f);
        std::vector<int32_t> icls((size_t)n_items2, 0), woff((size_t)n_items2, 0);
        const float rb_cost = getenv("SSQ_TILE2_RB_COST") ? (float)atof(getenv("SSQ_TILE2_RB_COST")) : 0.7f;
        tile2_ok = true;
        lgr_max2 = 0;
        for (int i = 0; i < nsegs; ++i) {
            const int64_t L = (int64_t)M >> sg[i].lgR;
            if (sg[i].kind && (sg[i].wtab_off >= 65536 || sg[i].sig_stride % L)) tile2_ok = false;
            if (sg[i].kind) lgr_max2 = std::max(lgr_max2, (int)sg[i].lgR);
            for (int t = 0; t < sg[i].nsteps * TILE_G / rpi; ++t) {
                const size_t it = (size_t)sg[i].first * TILE_G / rpi + t;
                const TileRow* r = rw + it * rpi;
                int npad = 0;
                for (int k = 0; k < rpi; ++k) {
                    if (r[k].row < 0) ++npad;
                    else if (npad) tile2_ok = false;                  // padding trails
                    // the sub-rows are consecutive rows of the class (the padding repeats the last one)
                    if (r[k].row >= 0 && ((r[k].row & 0xFFFF) != (r[0].row & 0xFFFF) + k
                                          || (sg[i].kind && r[k].ubase != r[0].ubase + k * L))) tile2_ok = false;
                    hi8[8 * it + 4 + k] = r[k].kc;
                }
                const int32_t row0 = r[0].row & 0xFFFF;
                hi8[8 * it] = row0 | (npad << 9) | (sg[i].kind << 12) | (sg[i].lgR << 13);
                hi8[8 * it + 1] = sg[i].kind ? sg[i].cls_base + r[0].ubase : 0;
                hi8[8 * it + 2] = (int32_t)(uint32_t)((int64_t)row0 * N * 8);
                hi8[8 * it + 3] = sg[i].kind ? sg[i].sig_stride : 0;
                cost[it] = sg[i].kind ? 1.0f : rb_cost;               // what rows read back cost next to interpolated ones
                icls[it] = sg[i].kind ? 1 + sg[i].lgR : 0;
                woff[it] = sg[i].kind ? sg[i].wtab_off : 0;
            }
        }
        if ((rc = up((void**)&items2, hi8.data(), hi8.size() * 4))) return rc;
        // Contiguous, cost-balanced blocks of items per wavefront, each spanning at most TWO classes
        // (kind / decimation): the kernel keeps the weights of two classes in registers. Tables for
        // 12 and 16 wavefronts, [nw][4] = first item, end, first item of the second class, the
        // weights' table offsets of the two classes (16 bits each).
        std::vector<int> run_start;                        // maximal runs of one class
        for (int it = 0; it < n_items2; ++it)
            if (it == 0 || icls[it] != icls[it - 1]) run_start.push_back(it);
        const int nruns = (int)run_start.size();
        run_start.push_back(n_items2);
        std::vector<double> pre((size_t)n_items2 + 1, 0.0);
        for (int it = 0; it < n_items2; ++it) pre[it + 1] = pre[it] + cost[it];
        std::vector<int32_t> wt_;
        // (SSQ_TILE2_WAVE_SPEED="a,b,c,d": relative speeds of the wavefront groups 0-3 / 4-7 / 8-11 / 12-15 to size the
        // blocks by. The -DSSQ_TILE2_PROF stamps show the older wavefronts of a SIMD -- w, w + 4, w + 8, w + 12 share one --
        // finishing the same work 10-19 % sooner and waiting at the tile's end; sizing the blocks by that changes
        // nothing (221 +- 3 us for every setting tried): a SIMD's total is what counts, and it is the same.)
        float speed[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (const char* e = getenv("SSQ_TILE2_WAVE_SPEED")) {
            float v[4];
            if (sscanf(e, "%f,%f,%f,%f", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0)
                for (int k = 0; k < 4; ++k) speed[k] = v[k];
        }
        for (int nw : {12, 16}) {
            int cur = 0;
            double stot = 0, sacc = 0;
            for (int w = 0; w < nw; ++w) stot += speed[std::min(w / 4, 3)];
            for (int w = 0; w < nw; ++w) {
                sacc += speed[std::min(w / 4, 3)];
                if (cur >= n_items2) { wt_.insert(wt_.end(), {n_items2, n_items2, n_items2, 0}); continue; }
                int r0 = 0;
                while (run_start[r0 + 1] <= cur) ++r0;
                const int maxe = run_start[std::min(r0 + 2, nruns)];          // at most the rest of this run and the next
                const int rem = nw - w - 1;
                const int rmin = std::max(r0, nruns - 2 * rem);               // the rest must fit the remaining wavefronts
                int mine = rem == 0 ? n_items2 : run_start[std::min(rmin, nruns)];
                mine = std::max(mine, cur + 1);
                int e = cur;
                const double want = pre[n_items2] * sacc / stot;
                while (e < n_items2 && pre[e + 1] <= want + 1e-9) ++e;
                e = std::min(std::max(e, mine), maxe);
                if (rem == 0) { e = n_items2; if (e > maxe) tile2_ok = false; }
                const int isp = run_start[r0 + 1] < e ? run_start[r0 + 1] : e;
                wt_.insert(wt_.end(), {cur, e, isp, woff[cur] | (woff[std::min(isp, n_items2 - 1)] << 16)});
                cur = e;
            }
            if (cur < n_items2) tile2_ok = false;
        }
        if ((rc = up((void**)&wave_first2, wt_.data(), wt_.size() * 4))) return rc;
    }
    {   // weights, per class (R phases from wtab_off on): [phase][4 tap pairs] -> [tap pair][phase]
        const TileSeg* sg = reinterpret_cast<const TileSeg*>(d.segs);
        const float* src = reinterpret_cast<const float*>(d.wtab);
        std::vector<float> w((size_t)16 * d.n_phases, 0.f);
        std::vector<char> done((size_t)d.n_phases, 0);
        for (int i = 0; i < nsegs; ++i) {
            if (sg[i].kind != 1) continue;
            const int64_t R = (int64_t)1 << sg[i].lgR, o = sg[i].wtab_off;
            SSQ_REQUIRE(o >= 0 && o + R <= d.n_phases, "tile segment %d: weights outside the table", i);
            if (done[(size_t)o]) continue;
            done[(size_t)o] = 1;
            for (int64_t ph = 0; ph < R; ++ph)
                for (int t = 0; t < 4; ++t)
                    for (int k = 0; k < 4; ++k)
                        w[(size_t)(16 * o + (t * R + ph) * 4 + k)] = src[(size_t)(16 * (o + ph) + 4 * t + k)];
        }
        if ((rc = up(&wtab, w.data(), (size_t)64 * d.n_phases))) return rc;
    }
    if ((rc = up(&tbank, d.tbank, (size_t)4 * d.n_tbank))) return rc;
    cls.resize(d.n_classes);
    // classes of 2^14 entries and more: four-step kernels, L = A B with A <= B, both 128 .. 2048
    // (SSQ_TILE_FFT=rocfft keeps every class on rocFFT)
    const bool own_fft = !(getenv("SSQ_TILE_FFT") && !strcmp(getenv("SSQ_TILE_FFT"), "rocfft"));
    int64_t y_entries = 0;
    for (int c = 0; c < d.n_classes; ++c) {
        cls[c] = {d.classes[4 * c], d.classes[4 * c + 1], d.classes[4 * c + 2], 0, 0, 0};
        SSQ_REQUIRE(cls[c].L >= 2 && (cls[c].L & (cls[c].L - 1)) == 0 && cls[c].nrows >= 1, "bad tile class %d", c);
        lmax = std::max(lmax, cls[c].L);
        int lg = 0;
        while (((int64_t)1 << lg) < cls[c].L) ++lg;
        if (own_fft && lg >= 13 && lg <= 22) {
            cls[c].B = 1 << ((lg + 1) / 2); cls[c].A = 1 << (lg / 2);
            y_entries += (int64_t)group * cls[c].nrows * cls[c].L;
        } else if (own_fft && lg >= 6 && lg <= 12) {
            cls[c].B = 1;                              // one-pass kernel
        }
    }
    std::vector<TileIRow> hi;
    hi.reserve((size_t)n_irows);
    for (int pass = 0; pass < 2; ++pass)            // rows sorted by class, the four-step classes first
        for (int c = 0; c < d.n_classes; ++c) {
            if ((cls[c].A > 0 || cls[c].B > 0) != (pass == 0)) continue;
            if (pass == 1 && n_irows_fft == 0) first_irow_fft = (int)hi.size();
            cls[c].first = (int)hi.size();
            for (int r = 0; r < n_irows; ++r) {
                const int64_t* q = d.irows + 8 * r;
                if ((int)q[6] != c) continue;
                SSQ_REQUIRE(q[4] == cls[c].L && q[2] >= 1 && 2 * q[2] <= q[4]
                            && q[1] >= 0 && q[1] + q[2] <= M / 2 + 1 && q[5] >= 0 && q[5] + q[2] <= d.n_tbank
                            && q[7] >= 0 && q[7] < cls[c].nrows, "bad tile row %d", r);
                hi.push_back({(int32_t)q[0], (int32_t)q[1], (int32_t)q[2], (int32_t)q[3], (int32_t)q[4], (int32_t)q[5],
                              (int32_t)((int64_t)group * cls[c].upre + q[7] * cls[c].L), (int32_t)(cls[c].nrows * cls[c].L)});
            }
            SSQ_REQUIRE((int64_t)hi.size() - cls[c].first == cls[c].nrows, "tile class %d: %lld rows listed, %lld declared",
                        c, (long long)((int64_t)hi.size() - cls[c].first), (long long)cls[c].nrows);
            if (pass == 1) n_irows_fft += (int)cls[c].nrows;
        }
    SSQ_REQUIRE((int)hi.size() == n_irows, "tile rows of unknown classes");
    if ((rc = up((void**)&irows, hi.data(), sizeof(TileIRow) * n_irows))) return rc;
    if (own_fft) {
        SSQ_CHECK_HIP(hipMalloc(&Y, (size_t)8 * std::max<int64_t>(y_entries, 1))); bytes += 8 * y_entries;
        std::vector<float> tw;
        for (int s = 0; s < 7; ++s) {
            const int Lp = 64 << s;
            ftw_off[s] = (int64_t)tw.size() / 2;
            for (int q = 0; q < Lp; ++q) {
                const double a = 6.283185307179586 * (double)q / (double)Lp;
                tw.push_back((float)std::cos(a)); tw.push_back((float)std::sin(a));
            }
        }
        if ((rc = up(&ftw, tw.data(), tw.size() * 4))) return rc;
    }
    // (+ 4 rows of slack: tile2_kernel's padded sub-rows read, and discard, the rows behind a class's last)
    SSQ_CHECK_HIP(hipMalloc(&U, (size_t)8 * (group * u_total + 4 * lmax))); bytes += 8 * (group * u_total + 4 * lmax);
    SSQ_CHECK_HIP(hipMemset(U, 0, (size_t)8 * (group * u_total + 4 * lmax)));
    SSQ_CHECK_HIP(hipMalloc((void**)&counters, 4096));       // [0]: tiles done; [8 ..]: tuning aid (SSQ_TILE2_PROF)
    SSQ_CHECK_HIP(hipMemset(counters, 0, 4096));
    for (size_t c = 0; c < cls.size(); ++c) {
        FftPlan fp;
        if (!cls[c].A && !cls[c].B) {
            rc = fp.create(1, SSQ_F32, (size_t)cls[c].L, (size_t)(group * cls[c].nrows), 1.0);
            if (rc) return rc;
            bytes += (int64_t)fp.work_bytes;
        }
        ffts.push_back(fp);
    }
    for (int t = 0; t < 5; ++t) n_items_tile[t] = d.n_items_tile[t];
    if (!getenv("SSQ_TILE_SERIAL")) {
        SSQ_CHECK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        SSQ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        SSQ_CHECK_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    return 0;
}

void TilePlan::destroy() {
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); side = nullptr; }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    ev_fork = ev_join = nullptr;
    for (auto& f : ffts) f.destroy();
    ffts.clear();
    void* ptrs[] = {steps, rows, irows, wtab, tbank, U, counters, Y, ftw, items2, wave_first2};
    items2 = nullptr; wave_first2 = nullptr;
    for (void* p : ptrs) if (p) (void)hipFree(p);
    steps = nullptr; rows = nullptr; irows = nullptr; wtab = tbank = U = Y = ftw = nullptr;
    counters = nullptr;
}

int TilePlan::spectra(int sig, int nsig, const void* xh_all, hipStream_t stream) {
    // short classes: band -> samples, all of them in one launch (the longest rows first)
    {
        TileSmallArgs S;
        S.ncls = 0; S.first_block[0] = 0;
        for (int want = 6; want >= 0; --want)
            for (size_t c = 0; c < cls.size() && S.ncls < 7; ++c) {
                if (cls[c].A || !cls[c].B) continue;
                int sl = 0;
                while ((64 << sl) < cls[c].L) ++sl;
                if (sl != want) continue;
                TileFftArgs& E = S.E[S.ncls];
                E.xh = (const c32*)xh_all; E.xh_stride = M / 2 + 1; E.sig0 = sig;
                E.irows = irows + cls[c].first; E.tbank = (const float*)tbank;
                E.Y = nullptr; E.U = (c32*)U;
                E.A = 0; E.B = 0; E.L = (int)cls[c].L; E.nrows = (int)cls[c].nrows; E.G2 = 0; E.inv_l = 0.f; E.nyq = 0;
                E.ftw1 = (const c32*)ftw + ftw_off[sl]; E.ftw2 = nullptr;
                const int G = D_POINTS / (int)cls[c].L, npairs = E.nrows * nsig;
                S.npairs[S.ncls] = npairs; S.slot[S.ncls] = sl;
                S.first_block[S.ncls + 1] = S.first_block[S.ncls] + (npairs + G - 1) / G;
                ++S.ncls;
            }
        if (S.ncls) {
            hipLaunchKernelGGL(tilefft_small_kernel, dim3((unsigned)S.first_block[S.ncls]), dim3(NT), 0, stream, S);
            SSQ_LAUNCH_CHECK();
        }
    }
    // four-step classes: band -> samples, one launch per pass for all of them (the longest first)
    {
        TileFourArgs F1, F2;
        F1.ncls = 0; F1.first_block[0] = 0; F2.first_block[0] = 0;
        int64_t y_off = 0;
        for (size_t c = 0; c < cls.size() && F1.ncls < 10; ++c) {       // (classes come longest first)
            if (!cls[c].A) continue;
            const int k = F1.ncls;
            TileFftArgs E;
            E.xh = (const c32*)xh_all; E.xh_stride = M / 2 + 1; E.sig0 = sig;
            E.irows = irows + cls[c].first; E.tbank = (const float*)tbank;
            E.Y = (c32*)Y + y_off; E.U = (c32*)U;
            y_off += (int64_t)group * cls[c].nrows * cls[c].L;
            E.A = cls[c].A; E.B = cls[c].B; E.L = (int)cls[c].L; E.nrows = (int)cls[c].nrows;
            E.G2 = D_POINTS / E.A;                         // q2 columns per pass-2 workgroup
            E.inv_l = 1.0f / (float)cls[c].L; E.nyq = 0;
            int sa = 0, sb = 0;                            // table slots: L' = 64 << slot
            while ((64 << sa) < E.A) ++sa;
            while ((64 << sb) < E.B) ++sb;
            E.ftw1 = (const c32*)ftw + ftw_off[sb]; E.ftw2 = (const c32*)ftw + ftw_off[sa];
            F1.E[k] = E; F2.E[k] = E;
            F1.slot[k] = sb; F2.slot[k] = sa;
            F1.nx[k] = E.A / (D_POINTS / E.B);             // k1 groups: G = 4096 / B columns each
            F2.nx[k] = E.B / E.G2;
            F1.first_block[k + 1] = F1.first_block[k] + F1.nx[k] * E.nrows * nsig;
            F2.first_block[k + 1] = F2.first_block[k] + F2.nx[k] * E.nrows * nsig;
            ++F1.ncls;
        }
        F2.ncls = F1.ncls;
        if (F1.ncls) {
            hipLaunchKernelGGL(tilefft_four_kernel<1>, dim3((unsigned)F1.first_block[F1.ncls]), dim3(NT), 0, stream, F1);
            SSQ_LAUNCH_CHECK();
            hipLaunchKernelGGL(tilefft_four_kernel<2>, dim3((unsigned)F2.first_block[F2.ncls]), dim3(NT), 0, stream, F2);
            SSQ_LAUNCH_CHECK();
        }
    }
    if (!n_irows_fft) return 0;
    int64_t lmax_fft = 0;
    for (size_t c = 0; c < cls.size(); ++c) if (!cls[c].A && !cls[c].B) lmax_fft = std::max(lmax_fft, cls[c].L);
    const dim3 grid((unsigned)std::min<int64_t>((lmax_fft + 255) / 256, 64), (unsigned)n_irows_fft, (unsigned)nsig);
    hipLaunchKernelGGL(tile_spectra_kernel, grid, dim3(256), 0, stream, (const float2*)xh_all, M / 2 + 1, sig,
                       irows + first_irow_fft, (const float*)tbank, (float2*)U);
    SSQ_LAUNCH_CHECK();
    for (size_t c = 0; c < cls.size(); ++c) {
        if (cls[c].A || cls[c].B) continue;
        // the planned batch covers `group` signals; slots past nsig hold stale finite data
        int rc = ffts[c].execute((float2*)U + (size_t)group * cls[c].upre, nullptr, stream);
        if (rc) return rc;
    }
    return 0;
}

// ---- the analytic signal through the four-step kernels (ssq_tiles.h)
bool AnalyticFft::supports(int dtype, int64_t M) {
    if (dtype != SSQ_F32 || (M & (M - 1))) return false;
    if (getenv("SSQ_TILE_FFT") && !strcmp(getenv("SSQ_TILE_FFT"), "rocfft")) return false;
    return M >= ((int64_t)1 << 13) && M <= ((int64_t)1 << 22);
}
int AnalyticFft::create(int64_t M_, int64_t max_batch_, int64_t& bytes) {
    M = M_; max_batch = max_batch_;
    int lg = 0;
    while (((int64_t)1 << lg) < M) ++lg;
    B = 1 << ((lg + 1) / 2); A = 1 << (lg / 2);
    auto up = [&](void** dst, const void* src, size_t nbytes) -> int {
        SSQ_CHECK_HIP(hipMalloc(dst, nbytes));
        SSQ_CHECK_HIP(hipMemcpy(*dst, src, nbytes, hipMemcpyHostToDevice));
        bytes += (int64_t)nbytes;
        return 0;
    };
    int rc;
    std::vector<float> tw;
    for (int which = 0; which < 2; ++which) {
        const int Lp = which ? B : A;
        (which ? off_b : off_a) = (int64_t)tw.size() / 2;
        for (int q = 0; q < Lp; ++q) {
            const double a = 6.283185307179586 * (double)q / (double)Lp;
            tw.push_back((float)std::cos(a)); tw.push_back((float)std::sin(a));
        }
    }
    if ((rc = up(&ftw, tw.data(), tw.size() * 4))) return rc;
    // weights: 1 / M on bins [0, M / 2), half of it at the Nyquist bin (both exact: M is a power of two)
    std::vector<float> w((size_t)(M / 2 + 1), 1.0f / (float)M);
    w[(size_t)(M / 2)] = 0.5f / (float)M;
    if ((rc = up(&tb, w.data(), w.size() * 4))) return rc;
    const TileIRow r = {0, 0, (int32_t)(M / 2 + 1), 0, (int32_t)M, 0, 0, (int32_t)M};
    if ((rc = up((void**)&irow, &r, sizeof(r)))) return rc;
    SSQ_CHECK_HIP(hipMalloc(&Y, (size_t)8 * max_batch * M)); bytes += 8 * max_batch * M;
    return 0;
}
void AnalyticFft::destroy() {
    void* ptrs[] = {Y, ftw, tb, irow};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    Y = ftw = tb = nullptr; irow = nullptr;
}
int AnalyticFft::run(const void* xh_all, void* xa, int64_t batch, hipStream_t stream) {
    TileFourArgs F;
    F.ncls = 1; F.first_block[0] = 0;
    TileFftArgs& E = F.E[0];
    E.xh = (const c32*)xh_all; E.xh_stride = M / 2 + 1; E.sig0 = 0;
    E.irows = irow; E.tbank = (const float*)tb;
    E.Y = (c32*)Y; E.U = (c32*)xa;
    E.A = A; E.B = B; E.L = (int)M; E.nrows = 1; E.G2 = D_POINTS / A; E.nyq = 1;
    E.inv_l = 1.0f / (float)M;
    E.ftw1 = (const c32*)ftw + off_b; E.ftw2 = (const c32*)ftw + off_a;
    int sa = 0, sb = 0;
    while ((64 << sa) < A) ++sa;
    while ((64 << sb) < B) ++sb;
    TileFourArgs F2 = F;
    F.slot[0] = sb; F.nx[0] = A / (D_POINTS / B);
    F.first_block[1] = F.nx[0] * (int)batch;
    F2.slot[0] = sa; F2.nx[0] = B / E.G2;
    F2.first_block[1] = F2.nx[0] * (int)batch;
    hipLaunchKernelGGL(tilefft_four_kernel<1>, dim3((unsigned)F.first_block[1]), dim3(NT), 0, stream, F);
    SSQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(tilefft_four_kernel<2>, dim3((unsigned)F2.first_block[1]), dim3(NT), 0, stream, F2);
    SSQ_LAUNCH_CHECK();
    return 0;
}

// wavefronts per workgroup: 12 by default = 3 per SIMD (168 VGPRs: the step pipeline needs ~160;
// at 16 wavefronts / 128 registers it spills and measured slower); SSQ_TILE_NW = 8 | 12 | 16
// selects another build of the kernel (tuning aid)
template <int GRID, bool STORE_D, int NW, int CSTK>
static int launch_tile_c(const TilePlan& P, const TileArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    auto kern = tile_kernel<GRID, STORE_D, NW, CSTK>;
    const size_t lds = tile_lds_bytes(P.na);
    static bool attr_set = false;            // per instantiation
    if (!attr_set) {
        SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    // persistent workgroups, one per CU (the tile fills the LDS)
    const int64_t ntot = ((P.N + TILE_COLS - 1) / TILE_COLS) * nsig;
    const dim3 grid((unsigned)std::min<int64_t>(ntot, P.ncu));
    TileArgs B = A; B.nsig = nsig;
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, B, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}
template <int GRID, bool STORE_D, int NW>
static int launch_tile_nw(const TilePlan& P, const TileArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    const int cstk = sp.cst_f64 ? 2 : (sp.cst_uniform ? 0 : 1);
    if (cstk == 0) return launch_tile_c<GRID, STORE_D, NW, 0>(P, A, sp, nsig, stream);
    if (cstk == 1) return launch_tile_c<GRID, STORE_D, NW, 1>(P, A, sp, nsig, stream);
    return launch_tile_c<GRID, STORE_D, NW, 2>(P, A, sp, nsig, stream);
}
template <int GRID, bool STORE_D>
static int launch_tile(const TilePlan& P, const TileArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    return launch_tile_nw<GRID, STORE_D, 12>(P, A, sp, nsig, stream);
}

// ---- tile2_kernel launch
template <int GRID, bool STORE_D, int NW, int CSTK, int COLS, bool STORE_K = false>
static int launch_tile2_c(const TilePlan& P, const Tile2Args& A, const SsqParams& sp, hipStream_t stream) {
    auto kern = tile2_kernel<GRID, STORE_D, NW, CSTK, COLS, STORE_K>;
    const size_t lds = tile2_lds_bytes(P.na, COLS);
    static bool attr_set = false;            // per instantiation
    if (!attr_set) {
        SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int64_t ntx = (P.N + COLS - 1) / COLS;
    // Persistent workgroups: as many as fit a CU's LDS side by side, workgroup b walks tiles b, b + G,
    // ... of every signal. The kernel keeps a lane's interpolation weights for the whole launch, so
    // the columns of a workgroup's tiles must agree mod R for every class: G * COLS a multiple of the
    // largest R (or a single tile per signal and workgroup).
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, (size_t)(32 / NW)));
    const int64_t cap = (int64_t)P.ncu * per_cu;
    const int64_t q = std::max<int64_t>(1, ((int64_t)1 << P.lgr_max2) / COLS);
    const int64_t G = ntx <= cap ? ntx : std::max<int64_t>(q, cap / q * q);
    // ... and through the signals' boundaries when a signal's tile count keeps that phase too
    // (SSQ_TILE2_CARRY=0: every signal's walk starts at the workgroup's own tile)
    const char* ce = getenv("SSQ_TILE2_CARRY");              // (read per launch: tests switch it)
    const bool carry_on = !(ce && atoi(ce) == 0);
    Tile2Args B = A;
    B.carry = (carry_on && ntx > G && ntx % q == 0) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(64 * NW), lds, stream, B, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}
template <int GRID, bool STORE_D, int NW, int COLS>
static int launch_tile2_k(const TilePlan& P, const Tile2Args& A, const SsqParams& sp, hipStream_t stream) {
    const int cstk = sp.cst_f64 ? 2 : (sp.cst_uniform ? 0 : 1);
    if (A.kdump) {
        // the diagnostic builds exist for the default wavefront count and one weight per transform (the bins do
        // not depend on the weights: 'log' scales, what the full-size index test runs)
        if constexpr (NW == 16) {
            SSQ_REQUIRE(cstk == 0, "bin dump: built for uniform reassignment weights ('log' scales)");
            return launch_tile2_c<GRID, STORE_D, NW, 0, COLS, true>(P, A, sp, stream);
        } else {
            SSQ_REQUIRE(false, "bin dump: built for 16 wavefronts per workgroup (unset SSQ_TILE_NW)");
        }
    }
    if (cstk == 0) return launch_tile2_c<GRID, STORE_D, NW, 0, COLS>(P, A, sp, stream);
    if (cstk == 1) return launch_tile2_c<GRID, STORE_D, NW, 1, COLS>(P, A, sp, stream);
    return launch_tile2_c<GRID, STORE_D, NW, 2, COLS>(P, A, sp, stream);
}
template <int GRID, bool STORE_D>
static int launch_tile2(const TilePlan& P, Tile2Args& A, const SsqParams& sp, hipStream_t stream) {
    static const int nw = [] { const char* e = getenv("SSQ_TILE_NW"); int v = e ? atoi(e) : 16; return v == 12 ? 12 : 16; }();
    // wave_first2 holds the blocks for 12 and for 16 wavefronts one after the other
    // (measured, round 4: 16-column tiles with 16 wavefronts 245 us, with 8 wavefronts and two workgroups per CU
    // 260-275 us, against 220 us for the 32-column tile -- profiles/r4_ab_history.txt)
    A.waves = reinterpret_cast<const int4*>(P.wave_first2) + (nw == 12 ? 0 : 12);
    if (P.cols2 == 32) {
        if (nw == 12) return launch_tile2_k<GRID, STORE_D, 12, 32>(P, A, sp, stream);
        return launch_tile2_k<GRID, STORE_D, 16, 32>(P, A, sp, stream);
    }
    if (nw == 12) return launch_tile2_k<GRID, STORE_D, 12, 16>(P, A, sp, stream);
    return launch_tile2_k<GRID, STORE_D, 16, 16>(P, A, sp, stream);
}

// SSQ_TILE_ORDER = ordered: the ticketed kernel (float32 sums in the reference's order, bit for bit; na <=
// 318); default: tile2_kernel (float64 tile, unordered adds: the same bins, sums rounded once)
bool tile_ordered() { return reassign_ordered(); }
bool TilePlan::usable() const {
    if (!tile_ordered() && tile2_ok) return true;
    return tile_lds_bytes(na) <= 160 * 1024;
}
int TilePlan::tile_cols() const { return !usable() ? 0 : (tile_ordered() || !tile2_ok) ? TILE_COLS : cols2; }

int TilePlan::run(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                  const void* cst, float cst0, const SsqParams& sp, hipStream_t stream, unsigned short* kdump) {
    SSQ_REQUIRE(!kdump || (!tile_ordered() && tile2_ok), "bin dump: the default tile kernel only (unset SSQ_TILE_ORDER)");
    if (!tile_ordered() && tile2_ok) {
        Tile2Args B;
        B.kdump = kdump;
        B.items = reinterpret_cast<const int*>(items2); B.waves = nullptr;
        B.wtab = (const float4*)wtab; B.U = (const float2*)U; B.cst = cst;
        B.Wx = (float2*)Wx; B.dWx = (float2*)dWx; B.Tx = (float2*)Tx; B.kidx = kidx;
        B.N = N; B.na = na; B.n_items = n_items2; B.n1 = (int)n1; B.mmask = (int)(M - 1);
        B.lgM = 0; while (((int64_t)1 << B.lgM) < M) ++B.lgM;
        B.sig0 = sig; B.nsig = nsig; B.group = group; B.inv_m = 1.0f / (float)M;
        B.theta_scale = (float)(6.283185307179586 / ((double)M * dt)); B.cst0 = cst0;
        B.counters = counters; B.gamma = sp.gamma; B.carry = 0;
#define TILE2_LAUNCH(G)                                                                     \
        return dWx ? launch_tile2<G, true>(*this, B, sp, stream) : launch_tile2<G, false>(*this, B, sp, stream);
        auto launch2 = [&]() -> int {
            if (sp.grid == SSQ_GRID_LOG) { TILE2_LAUNCH(SSQ_GRID_LOG) }
            if (sp.grid == SSQ_GRID_LOG_PIECEWISE) { TILE2_LAUNCH(SSQ_GRID_LOG_PIECEWISE) }
            TILE2_LAUNCH(SSQ_GRID_LIN)
        };
        const int rc2 = launch2();
        if (!rc2 && getenv("SSQ_TILE2_PROF_DUMP")) {         // tuning aid, see T2_STAMP
            SSQ_CHECK_HIP(hipStreamSynchronize(stream));
            unsigned long long h[8 + 8 * 16];
            SSQ_CHECK_HIP(hipMemcpy(h, counters, sizeof h, hipMemcpyDeviceToHost));
            for (int w = 0; w < 16; ++w) {
                fprintf(stderr, "tile2 prof w%2d:", w);
                for (int k = 0; k < 8; ++k) fprintf(stderr, " %10llu", h[8 + 8 * w + k]);
                fprintf(stderr, "\n");
            }
        }
        return rc2;
#undef TILE2_LAUNCH
    }
    SSQ_REQUIRE(tile_lds_bytes(na) <= 160 * 1024, "na = %lld: the ordered tile kernel's Tx tile exceeds the LDS", (long long)na);
    TileArgs A;
    A.pstep = reinterpret_cast<const int4*>(steps); A.prow = reinterpret_cast<const int2*>(rows);
    A.wtab = (const float4*)wtab; A.U = (const float2*)U; A.cst = cst;
    A.Wx = (float2*)Wx; A.dWx = (float2*)dWx; A.Tx = (float2*)Tx; A.kidx = kidx;
    A.N = N; A.na = na; A.nsteps = nsteps; A.n1 = (int)n1; A.mmask = (int)(M - 1);
    A.sig0 = sig; A.nsig = nsig; A.inv_m = 1.0f / (float)M;
    A.theta_scale = (float)(6.283185307179586 / ((double)M * dt)); A.cst0 = cst0;
    A.counters = counters;
    A.gamma = sp.gamma;
    static unsigned long long* trace_buf = nullptr;
    const char* trace_path = getenv("SSQ_TILE_TRACE");
    if (trace_path && !trace_buf) SSQ_CHECK_HIP(hipMalloc((void**)&trace_buf, 8 * TRACE_WORDS));
    if (trace_buf) SSQ_CHECK_HIP(hipMemsetAsync(trace_buf, 0, 8 * TRACE_WORDS, stream));
    A.trace = trace_buf;
    auto dump_trace = [&]() -> int {
        if (!trace_buf) return 0;
        SSQ_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h(TRACE_WORDS);
        SSQ_CHECK_HIP(hipMemcpy(h.data(), trace_buf, 8 * h.size(), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(trace_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        return 0;
    };
#define TILE_LAUNCH(G)                                                                      \
    { int rc_ = dWx ? launch_tile<G, true>(*this, A, sp, nsig, stream)                     \
                    : launch_tile<G, false>(*this, A, sp, nsig, stream);                   \
      return rc_ ? rc_ : dump_trace(); }
    if (sp.grid == SSQ_GRID_LOG) { TILE_LAUNCH(SSQ_GRID_LOG) }
    if (sp.grid == SSQ_GRID_LOG_PIECEWISE) { TILE_LAUNCH(SSQ_GRID_LOG_PIECEWISE) }
    TILE_LAUNCH(SSQ_GRID_LIN)
#undef TILE_LAUNCH
}

int64_t TilePlan::tiles_done(hipStream_t stream) {
    unsigned long long v = 0;
    if (!counters) return 0;
    if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, counters, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

}  // namespace ssq
