// ssq_tile_fft.hip -- the intermediates of the column-tile path (float32, gfx950): the decimated baseband
// samples u_i[q] of the rows the tile kernels interpolate, and the analytic signal of the padded batch.
//
//   band of row i (K bins around bin kc of the M-grid) x spectrum of the padded signal x compensated bank value,
//   inverse FFT of length L = M / R (R: the row's decimation) -> u_i[q] (plan-owned, ~34 MB per signal at N=160k).
//   Classes of 64 .. 4096 entries: `tilefft_small_kernel` (one LDS transform per row, G rows per workgroup);
//   classes of 2^13 .. 2^22 entries: `tilefft_four_kernel<1 / 2>`, a four-step transform whose first pass forms
//   the band on the fly (the zero-padded spectrum never exists in memory). Three launches per launch group for
//   all classes. `tile_spectra_kernel` + a batched rocFFT inverse per class is the older route
//   (SSQ_DEBUG_TILE_FFT=rocfft). Math and planning: ssqueezepy_amd/_tiles.py; the reference evaluates the same rows as
//   full-length inverse FFTs (ssqueezepy/_cwt.py:167-177).
#include "ssq_common.h"
#include "ssq_tiles.h"
#include "ssq_ldsfft.h"
#include <algorithm>
#include <cmath>

namespace ssq {

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

// ---------------------------------------------------------------------------- host side
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
    if (getenv("SSQ_DEBUG_TILE_FFT") && !strcmp(getenv("SSQ_DEBUG_TILE_FFT"), "rocfft")) return false;
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

}  // namespace ssq
