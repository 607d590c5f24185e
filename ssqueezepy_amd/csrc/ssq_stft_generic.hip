// ssq_stft_generic.hip -- the fused STFT kernel for the window lengths that are not powers of two (float32, gfx950).
// Plan and the other routes: ssq_stft.hip. Reference: ssqueezepy/_stft.py:127-147 (framing, window, rFFT),
// utils/stft_utils.py:69-98 (buffer), algos.py:956-984 (the bin rule of the fused ssq_stft form).
// Compiled with -ffp-contract=off (bin indices; see ssq_kernels.hip).
#include "ssq_stft.h"
#include <algorithm>
#include "ssq_dft_tables.h"
#include <cmath>

namespace ssq {

#include "ssq_point_math.inl"

// ---- fused framing + window + FFT for ANY n_fft whose prime factors are <= 31 (float32) ---------------------
// The reference's published ssq_stft benchmark is n_fft = 598 = 2 x 13 x 23, hop 1 (examples/benchmarks.py:78-79);
// through framing kernel -> rocFFT -> epilogue it moved ~4 GB for 0.77 GB of results. Here: the same fused form as
// stft_fused_kernel -- the two real frames a = frame * window, b = frame * diff_window as ONE complex transform of
// z = a - ib, separated by Hermitian symmetry -- with a mixed-radix Stockham transform in LDS:
//   * G consecutive frames of one signal per workgroup, points [q][g] in LDS (frame index fastest: conflict-free, and
//     the results leave as runs of G x 8 bytes); two buffers of n_fft * G * 8 bytes, together <= 78 KB, so that two
//     workgroups share a CU (n_fft = 598: G = 8);
//   * one pass per factor, from one buffer to the other, a butterfly per work-item and turn -- radices 16 / 8 / 4 / 2
//     with the butterflies of ssq_ldsfft.h, the odd primes 3 .. 31 as direct sums in their symmetric form (pairs
//     x_j +- x_{R-j}: R^2 real multiply-adds per butterfly instead of 4 R^2, roots as instruction literals,
//     ssq_dft_tables.h), outputs stored pair by pair. (First version, measured on the MI355X: in place, every
//     butterfly of a pass held in registers across the barrier -- 512 registers, 1.3 KB of scratch, 185 KB of code:
//     2.9 ms at n_fft = 598, hop 1.)
//   * twiddles e^{2 pi i q / n_fft} from an n_fft-entry table (cached).
// The epilogue is stft_fused_kernel's (Sx; dSx or the 2-byte bin of the fused ssq_stft form).
struct StftGenArgs {
    StftFusedArgs F;
    const c32* tw;                                  // e^{+2 pi i q / n}, q < n
    int n, G, lgG, npass;
    int radix[GEN_MAX_PASSES];
    unsigned magic[GEN_MAX_PASSES];                 // ceil(2^32 / Ns) of the pass: j / Ns = umulhi(j, magic), j < n < 2^16
};

// The butterfly of an odd prime radix, from registers straight to LDS: V[k] = sum_t v[t] e^{+2 pi i t k / R} in its
// symmetric form -- with s_j = v_j + v_{R-j}, d_j = v_j - v_{R-j} (j = 1 .. H = (R - 1) / 2):
//   V[k], V[R-k] = (v_0 + sum_j s_j cos(2 pi j k / R)) +- i (sum_j d_j sin(2 pi j k / R))
// -- 4 H^2 ~ R^2 real multiply-adds instead of 4 R^2; the roots are instruction literals. Each pair of outputs is
// stored as soon as it is formed, so only the sums and differences stay live (2 (R - 1) registers).
template <int R> struct GenDft {
    template <typename Store>
    static __device__ __forceinline__ void run(c32 (&v)[R], Store&& store) {
        constexpr int H = (R - 1) / 2;
        c32 sm[H], df[H];
#pragma unroll
        for (int j = 1; j <= H; ++j) { sm[j - 1] = cadd(v[j], v[R - j]); df[j - 1] = csub(v[j], v[R - j]); }
        const c32 x0 = v[0];
        c32 acc0 = x0;
#pragma unroll
        for (int j = 0; j < H; ++j) acc0 = cadd(acc0, sm[j]);
        store(0, acc0);
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            c32 a = x0, b = {0.f, 0.f};
#pragma unroll
            for (int j = 1; j <= H; ++j) {
                const float cc = DftRoots<R>::c[(j * k) % R], ss = DftRoots<R>::s[(j * k) % R];
                a.x = fma_(sm[j - 1].x, cc, a.x); a.y = fma_(sm[j - 1].y, cc, a.y);
                b.x = fma_(df[j - 1].x, ss, b.x); b.y = fma_(df[j - 1].y, ss, b.y);
            }
            store(k, c32{a.x - b.y, a.y + b.x});          // a + i b
            store(R - k, c32{a.x + b.y, a.y - b.x});      // a - i b
        }
    }
};
template <int R> struct GenDftPow2 {
    template <typename Store>
    static __device__ __forceinline__ void run(c32 (&v)[R], Store&& store) {
        if constexpr (R == 2) dft2(v[0], v[1]); else Dft<R>::run(v);
#pragma unroll
        for (int t = 0; t < R; ++t) store(t, v[t]);
    }
};
template <> struct GenDft<2> : GenDftPow2<2> {};
template <> struct GenDft<4> : GenDftPow2<4> {};
template <> struct GenDft<8> : GenDftPow2<8> {};
template <> struct GenDft<16> : GenDftPow2<16> {};

// one Stockham pass of radix R over the G columns ([q][g], n points each), from `in` to `out` (two LDS buffers);
// Ns = product of the radices before it. Butterfly j of a column: inputs q = j + t n / R, twiddle index k = j mod
// Ns, outputs (j / Ns) Ns R + k + t Ns.
template <int R>
__device__ __forceinline__ void gen_pass(const c32* __restrict__ in, c32* __restrict__ out, const c32* __restrict__ tw,
                                         int n, int G, int lgG, int Ns, unsigned magic, int tid) {
    const int nbf = n / R, nb = nbf << lgG;         // butterflies per column, in the workgroup
    const int tstep = nbf / Ns;                     // n / (Ns R)
    for (int i = tid; i < nb; i += GEN_NT) {
        const int g = i & (G - 1), j = i >> lgG;
        const int qd = Ns > 1 ? (int)__umulhi((unsigned)j, magic) : j, k = j - qd * Ns;     // j / Ns, j mod Ns
        c32 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = in[((j + t * nbf) << lgG) + g];
        if (Ns > 1) {
            const int ts = k * tstep;
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = cmul_v(v[t], tw[t * ts]);
        }
        c32* o = out + (((j - k) * R + k) << lgG) + g;     // (j / Ns) Ns R + k
        const int ostr = Ns << lgG;
        GenDft<R>::run(v, [&](int t, c32 val) { o[t * ostr] = val; });
    }
    __syncthreads();
}

__global__ __launch_bounds__(GEN_NT) void stft_generic_kernel(StftGenArgs B, SsqParams sp) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const StftFusedArgs& A = B.F;
    const int n = B.n, G = B.G, lgG = B.lgG;
    c32* cur = reinterpret_cast<c32*>(lds_raw);
    c32* oth = cur + (n << lgG);
    int bx = blockIdx.x;
    if (A.xcd) bx = (bx & 7) * (int)(gridDim.x >> 3) + (bx >> 3);
    const int tid = threadIdx.x, c0 = bx * G;
    if (c0 >= A.n_hops) return;
    const float* xp = A.xp + (int64_t)blockIdx.y * A.padlen;
    const bool deriv = A.dSx != nullptr || A.kidx != nullptr || A.Tx != nullptr;
    for (int i = tid; i < (n << lgG); i += GEN_NT) {
        const int g = i & (G - 1), r = i >> lgG, c = c0 + g;
        // modulated: the frame is rotated by ceil(n_fft/2) (utils/stft_utils.py:76-82)
        const int s = !A.modulated ? r : (r < A.s20 ? A.s21 + r : r - A.s20);
        float a = 0.f, b = 0.f;
        if (c < A.n_hops) {
            const float v = xp[(int64_t)A.hop * c + s];
            a = v * A.window[r];
            if (deriv) b = v * A.diff_window[r];
        }
        cur[i] = {a, -b};
    }
    __syncthreads();
    // (the fused ssq_stft form: the rows' frequencies of the epilogue's points are asked for a transform ahead)
    float sfq[GEN_MAX_EPI];
    if (A.Tx) {
#pragma unroll
        for (int it = 0; it < GEN_MAX_EPI; ++it) sfq[it] = A.Sfs[min((tid + it * GEN_NT) >> lgG, n >> 1)];
    }
    int Ns = 1;
    for (int p = 0; p < B.npass; ++p) {
        const int R = B.radix[p];
        const unsigned mg = B.magic[p];
        switch (R) {
            case 2: gen_pass<2>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 3: gen_pass<3>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 4: gen_pass<4>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 5: gen_pass<5>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 7: gen_pass<7>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 8: gen_pass<8>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 11: gen_pass<11>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 13: gen_pass<13>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 16: gen_pass<16>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 17: gen_pass<17>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 19: gen_pass<19>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 23: gen_pass<23>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            case 29: gen_pass<29>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
            default: gen_pass<31>(cur, oth, B.tw, n, G, lgG, Ns, mg, tid); break;
        }
        Ns *= R;
        c32* t_ = cur; cur = oth; oth = t_;
    }
    const c32* buf = cur;
    // Z' = IDFT(a - ib) = conj(FFT(a + ib)); A[f] = (Z[f] + conj(Z[n-f])) / 2, B[f] = (Z[f] - conj(Z[n-f])) / 2i
    const int64_t base = (int64_t)blockIdx.y * A.rows * A.n_hops;
    const int nrow = (n >> 1) + 1;
    if (A.Tx) {
        // The fused ssq_stft form sums Tx of the workgroup's frames itself, as stft_fused_kernel does (round 6): every
        // point leaves the buffer for registers (its Sx on the way to HBM, its bin beside it), then both buffers
        // become a real and an imaginary plane of nrow x G float64 cells ((n + 2) G 8 bytes of the 2 n G 8 there are).
        constexpr int NI = GEN_MAX_EPI;                 // (n / 2 + 1) G / 256 <= 10: n G * 16 <= GEN_LDS_BYTES, G <= 16
        float vr[NI], vi[NI]; unsigned kq[NI];
        const int omax = (int)A.rows - 1, cells = nrow << lgG;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * GEN_NT, f = min(i >> lgG, nrow - 1), g = i & (G - 1), c = c0 + g;
            const bool valid = i < cells && c < A.n_hops;
            const c32 P = buf[(f << lgG) + g], Q = buf[((f ? n - f : 0) << lgG) + g];
            const float sr = 0.5f * (P.x + Q.x), si = 0.5f * (Q.y - P.y);
            const float dr = -0.5f * (P.y + Q.y), di = 0.5f * (Q.x - P.x);
            if (valid) A.Sx[base + (int64_t)f * A.n_hops + c] = make_float2(sr, si);
            const bool on = valid && mag_gt(sr, si, A.gamma);
            const int kb = bin_of_point_stft(dr, di, sr, si, sfq[it], sp, omax, on);
            kq[it] = on ? (unsigned)(sp.flipud ? omax - kb : kb) : 0xFFFFu;
            vr[it] = sr; vi[it] = si;
        }
        __syncthreads();                                // the buffer's last read
        double* const txt = reinterpret_cast<double*>(lds_raw);
        for (int i = tid; i < 2 * cells; i += GEN_NT) txt[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            if (kq[it] == 0xFFFFu) continue;
            const int i = tid + it * GEN_NT, f = i >> lgG, g = i & (G - 1);
            // the term in the CPU path's arithmetic (float32 product; float64 with a float64 weight vector)
            double tr, ti;
            if (sp.cst_f64) {
                const double w = ((const double*)A.cst)[A.cst_uniform ? 0 : f];
                tr = (double)vr[it] * w; ti = (double)vi[it] * w;
            } else {
                const float w = ((const float*)A.cst)[A.cst_uniform ? 0 : f];
                tr = (double)(vr[it] * w); ti = (double)(vi[it] * w);
            }
            const unsigned off = ((kq[it] << lgG) + (unsigned)g) * 8u;
            SSQ_LDS_ADD_F64(txt, off, tr);
            SSQ_LDS_ADD_F64(txt, off + (unsigned)cells * 8u, ti);
        }
        __syncthreads();
        for (int i = tid; i < cells; i += GEN_NT) {
            const int f = i >> lgG, g = i & (G - 1), c = c0 + g;
            if (c >= A.n_hops) continue;
            A.Tx[base + (int64_t)f * A.n_hops + c] = make_float2((float)txt[i], (float)txt[cells + i]);
        }
        return;
    }
    for (int i = tid; i < (nrow << lgG); i += GEN_NT) {
        const int f = i >> lgG, g = i & (G - 1), c = c0 + g;
        if (c >= A.n_hops) continue;
        const c32 P = buf[(f << lgG) + g], Q = buf[((f ? n - f : 0) << lgG) + g];
        const int64_t q = base + (int64_t)f * A.n_hops + c;
        const float sr = 0.5f * (P.x + Q.x), si = 0.5f * (Q.y - P.y);
        A.Sx[q] = make_float2(sr, si);
        if (!deriv) continue;
        const float dr = -0.5f * (P.y + Q.y), di = 0.5f * (Q.x - P.x);
        if (A.dSx) A.dSx[q] = make_float2(dr, di);
        if (A.kidx) {                 // the fused kernels' rule (algos.py:956-984): |Sx| > gamma, then the bin
            const int64_t omax = A.rows - 1;
            unsigned short kk = 0xFFFFu;
            if (mag_gt(sr, si, A.gamma)) {
                const int64_t kb = bin_of_point(dr, di, sr, si, true, A.Sfs[f], sp, omax);
                kk = (unsigned short)(sp.flipud ? omax - kb : kb);
            }
            A.kidx[q] = kk;
        }
    }
}

// factors of n for the mixed-radix kernel (16s, 8s, 4s, a 2, then the odd primes up to 31); false when n has a
// larger prime factor, too many factors, or does not fit the LDS
bool stft_generic_plan(int64_t n, int* radix, int* npass, int* G) {
    if (n < 2) return false;
    int np = 0;
    int64_t m = n;
    auto push = [&](int r) { if (np < GEN_MAX_PASSES) radix[np] = r; ++np; };
    while (m % 16 == 0) { push(16); m /= 16; }
    while (m % 8 == 0) { push(8); m /= 8; }
    while (m % 4 == 0) { push(4); m /= 4; }
    while (m % 2 == 0) { push(2); m /= 2; }
    for (int r : {3, 5, 7, 11, 13, 17, 19, 23, 29, 31}) while (m % r == 0) { push(r); m /= r; }
    if (m != 1 || np > GEN_MAX_PASSES) return false;
    if (n * 16 > GEN_LDS_BYTES || n >= 65536) return false;     // (two buffers of n x G x 8 bytes; 16-bit butterfly indices)
    int g = 16;
    // Four workgroups per CU (39 KB each) where at least four frames still fit, two (78 KB) for the longer transforms:
    // n_fft = 598 at G = 4 instead of 8 measured 0.87 -> 0.78 ms at hop 1 ("r6y15"; G = 2, eight workgroups: 1.21 ms).
    // (SSQ_DEBUG_STFT_GEN_LDS=<KB> moves the first limit -- A/B aid)
    static const int64_t soft_cap = [] { const char* e = getenv("SSQ_DEBUG_STFT_GEN_LDS"); return e && atoi(e) > 0 ? (int64_t)atoi(e) * 1024 : (int64_t)GEN_LDS_BYTES / 2; }();
    while (g > 4 && n * g * 16 > std::min<int64_t>(soft_cap, GEN_LDS_BYTES)) g >>= 1;
    while (g > 1 && n * g * 16 > GEN_LDS_BYTES) g >>= 1;
    *npass = np; *G = g;
    return true;
}

int launch_stft_generic(const StftFusedArgs& A, const SsqParams& sp, const c32* tw, int n, const int* radix,
                               int npass, int G, int64_t batch, hipStream_t stream) {
    static const bool remap = [] { const char* e = getenv("SSQ_DEBUG_STFT_XCD"); return !e || atoi(e) != 0; }();
    StftGenArgs B;
    B.F = A; B.tw = tw; B.n = n; B.G = G; B.npass = npass;
    B.lgG = 0; while ((1 << B.lgG) < G) ++B.lgG;
    {
        int64_t Ns = 1;
        for (int p = 0; p < GEN_MAX_PASSES; ++p) {
            B.radix[p] = p < npass ? radix[p] : 1;
            // j / Ns = umulhi(j, ceil(2^32 / Ns)), exact for j Ns < 2^32 (j < n < 2^16, Ns <= n); Ns = 1 is not divided
            B.magic[p] = Ns > 1 ? (unsigned)((((uint64_t)1 << 32) + (uint64_t)Ns - 1) / (uint64_t)Ns) : 0u;
            Ns *= B.radix[p];
        }
    }
    unsigned nb = (unsigned)((A.n_hops + G - 1) / G);
    B.F.xcd = remap && nb >= 64;
    if (B.F.xcd) nb = (nb + 7u) & ~7u;
    const size_t lds = (size_t)2 * n * G * sizeof(c32);
    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stft_generic_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_BYTES));
    // (the batch rides on grid.y: slices of at most 65535 signals)
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb_sig = std::min<int64_t>(65535, batch - b0);
        StftGenArgs C = B;
        const int64_t pts = b0 * A.rows * A.n_hops;
        C.F.xp = A.xp + b0 * A.padlen;
        C.F.Sx = A.Sx + pts;
        C.F.dSx = A.dSx ? A.dSx + pts : nullptr;
        C.F.kidx = A.kidx ? A.kidx + pts : nullptr;
        C.F.Tx = A.Tx ? A.Tx + pts : nullptr;
        hipLaunchKernelGGL(stft_generic_kernel, dim3(nb, (unsigned)nb_sig), dim3(GEN_NT), lds, stream, C, sp);
        SSQ_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace ssq
