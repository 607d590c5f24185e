// ssq_ldsfft.h -- complex FFT building blocks shared by the fused kernels, for float and
// double: in-register radix-4/8/16 butterflies and `lds_ifft`, a Stockham inverse FFT of G
// columns of L points through LDS, 16 points per thread (float: 4096 complex points per
// 256-thread workgroup; double: 2048 points per 128-thread workgroup -- both 32 KiB of LDS).
// Used by the CWT block / four-step kernels (ssq_cwt_blocks.hip) and the fused STFT
// kernel (ssq_stft.hip). Device code only; include inside a .hip translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include "ssq_common.h"

namespace ssq {

template <typename R> struct cx { R x, y; };
using c32 = cx<float>;
using c64 = cx<double>;

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename R> __device__ __forceinline__ cx<R> cmul(cx<R> a, cx<R> b) {
    return {fma_(a.x, b.x, -(a.y * b.y)), fma_(a.x, b.y, a.y * b.x)};
}
// the same product for operands that live in registers (twiddles from tables, data): float32 in two
// packed instructions (SSQ_CMUL_PK; rounds the a.x products first, where `cmul` rounds the a.y ones)
__device__ __forceinline__ cx<float> cmul_v(cx<float> a, cx<float> b) {
#ifdef SSQ_NO_CMUL_PK
    return cmul(a, b);
#else
    ssq_f2 av, bv, dv;
    av.x = a.x; av.y = a.y; bv.x = b.x; bv.y = b.y;
    SSQ_CMUL_PK(dv, av, bv);
    return {dv.x, dv.y};
#endif
}
__device__ __forceinline__ cx<double> cmul_v(cx<double> a, cx<double> b) { return cmul(a, b); }
template <typename R> __device__ __forceinline__ cx<R> cadd(cx<R> a, cx<R> b) { return {a.x + b.x, a.y + b.y}; }
template <typename R> __device__ __forceinline__ cx<R> csub(cx<R> a, cx<R> b) { return {a.x - b.x, a.y - b.y}; }
// multiply by +i (inverse-transform rotation)
template <typename R> __device__ __forceinline__ cx<R> mul_i(cx<R> a) { return {-a.y, a.x}; }

// in-place inverse DFTs: V[k] = sum_t v[t] e^{+2 pi i t k / R}
template <typename R> __device__ __forceinline__ void dft2(cx<R>& a, cx<R>& b) { cx<R> t = a; a = cadd(t, b); b = csub(t, b); }

template <typename R>
__device__ __forceinline__ void dft4(cx<R>& v0, cx<R>& v1, cx<R>& v2, cx<R>& v3) {
    cx<R> a = cadd(v0, v2), b = csub(v0, v2), c = cadd(v1, v3), d = mul_i(csub(v1, v3));
    v0 = cadd(a, c); v1 = cadd(b, d); v2 = csub(a, c); v3 = csub(b, d);
}

template <int R> struct Dft;
template <> struct Dft<4> {
    template <typename R> static __device__ __forceinline__ void run(cx<R> (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }
};
template <> struct Dft<8> {
    template <typename R> static __device__ __forceinline__ void run(cx<R> (&v)[8]) {
        // 8 = 2 x 4, decimation in time: even/odd 4-point DFTs, twiddle W8^k
        cx<R> e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
        dft4(e[0], e[1], e[2], e[3]);
        dft4(o[0], o[1], o[2], o[3]);
        const R h = (R)0.70710678118654752440;
        o[1] = {h * (o[1].x - o[1].y), h * (o[1].x + o[1].y)};     // * e^{+i pi/4}
        o[2] = mul_i(o[2]);                                        // * e^{+i pi/2}
        o[3] = {-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y)};    // * e^{+i 3pi/4}
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = cadd(e[k], o[k]); v[k + 4] = csub(e[k], o[k]); }
    }
};
template <> struct Dft<16> {
    template <typename R> static __device__ __forceinline__ void run(cx<R> (&v)[16]) {
        // 16 = 4 x 4: t = t1 + 4 t2, k = 4 k1 + k2 ... columns t1, DFT over t2, twiddle, DFT over t1
        const R c1 = (R)0.92387953251128675613, s1 = (R)0.38268343236508977173;   // cos/sin(pi/8)
        const R h = (R)0.70710678118654752440;
        cx<R> a[4][4];
#pragma unroll
        for (int t1 = 0; t1 < 4; ++t1) {
            a[t1][0] = v[t1]; a[t1][1] = v[t1 + 4]; a[t1][2] = v[t1 + 8]; a[t1][3] = v[t1 + 12];
            dft4(a[t1][0], a[t1][1], a[t1][2], a[t1][3]);      // index k2
        }
        // twiddle W16^{t1*k2} = e^{+2 pi i t1 k2 / 16}
        const cx<R> w[10] = {{1, 0}, {c1, s1}, {h, h}, {s1, c1}, {0, 1}, {-s1, c1}, {-h, h}, {-c1, s1},
                           {-1, 0}, {-c1, -s1}};
#pragma unroll
        for (int t1 = 1; t1 < 4; ++t1)
#pragma unroll
            for (int k2 = 1; k2 < 4; ++k2) a[t1][k2] = cmul(a[t1][k2], w[t1 * k2]);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            cx<R> b0 = a[0][k2], b1 = a[1][k2], b2 = a[2][k2], b3 = a[3][k2];
            dft4(b0, b1, b2, b3);                               // index k1
            v[k2] = b0; v[k2 + 4] = b1; v[k2 + 8] = b2; v[k2 + 12] = b3;
        }
    }
};

constexpr int PPT = 16;            // points per thread
template <typename R> struct FftGeom {
    static constexpr int NT = sizeof(R) == 4 ? 256 : 128;   // threads per workgroup
    static constexpr int D = NT * PPT;                      // complex points per workgroup
};
constexpr int D_POINTS = FftGeom<float>::D;     // float32 kernels: 4096 points,
constexpr int NT = FftGeom<float>::NT;          //                  256 threads

// One L-point inverse FFT per column for G columns, Stockham autosort, LDS [q][g].
// `v` holds the pass-1 inputs on entry (butterfly u = idx / G, column g = idx % G,
// idx = tid + it*NT, input t at q = u + t*L/R1) and the final-pass outputs on exit
// (n_hi = u + t*L/RL with RL the last radix, same idx -> (u, g) mapping).
// EARLY_TW (float32): the twiddles of passes 2 and 3 are requested before the barrier in front of the pass instead of
// after its LDS reads -- the table reads' latency runs under the barrier (block rows 58.9 -> 56.7 us at config 2; no
// change for the intermediates' kernels, which wait on HBM). FRESH: `buf` holds nothing a wavefront may still be
// reading (a kernel's first transform): no barrier before pass 1. TWS: the twiddle table is e^{2 pi i q / (L TWS)} --
// a longer transform's table read at every TWS-th entry (the sub-transforms of block_spectra_multi_kernel). NTH: the
// workgroup's thread count when it is not the geometry's 256 / 128 (16 points per thread all the same).
template <int L, int G, int R1, int R2, int R3, bool EARLY_TW = false, bool FRESH = false, int TWS = 1, int NTH = 0, typename R>
__device__ __forceinline__ void lds_ifft(cx<R> (&v)[PPT], cx<R>* __restrict__ buf,
                                         const cx<R>* __restrict__ ftw, int tid) {
    constexpr int NT = NTH ? NTH : FftGeom<R>::NT;
    static_assert(L * G == NT * PPT, "L * G must fill the workgroup");
    constexpr bool three = (R3 > 1);
    constexpr bool ETW = EARLY_TW && sizeof(R) == 4;
    if constexpr (!FRESH) __syncthreads();         // LDS free (previous transform's reads done)
    // ---- pass 1 (Ns = 1): inputs already in registers
    {
        constexpr int NB = PPT / R1;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            cx<R> t[R1];
#pragma unroll
            for (int k = 0; k < R1; ++k) t[k] = v[it * R1 + k];
            Dft<R1>::run(t);
            int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < R1; ++k) buf[(u * R1 + k) * G + g] = t[k];
        }
    }
    cx<R> tw2[ETW ? PPT / R2 : 1][ETW ? R2 : 1];
    if constexpr (ETW) {
        constexpr int NB = PPT / R2, Ns = R1, TW = L / (Ns * R2);
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int kk = ((tid + it * NT) / G) % Ns;
#pragma unroll
            for (int k = 1; k < R2; ++k) tw2[it][k] = ftw[kk * k * TW * TWS];
        }
    }
    __syncthreads();
    // ---- pass 2 (Ns = R1)
    {
        constexpr int NB = PPT / R2, Ns = R1, STR = L / R2;
        cx<R> t[NB][R2];
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            int idx = tid + it * NT, g = idx % G, u = idx / G;
#pragma unroll
            for (int k = 0; k < R2; ++k) t[it][k] = buf[(u + k * STR) * G + g];
        }
        if (three) __syncthreads();                // in-place: all reads before any write
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            int idx = tid + it * NT, g = idx % G, u = idx / G;
            int kk = u % Ns;
            constexpr int TW = L / (Ns * R2);
#pragma unroll
            for (int k = 1; k < R2; ++k) t[it][k] = cmul_v(t[it][k], ETW ? tw2[ETW ? it : 0][ETW ? k : 0] : ftw[kk * k * TW * TWS]);
            Dft<R2>::run(t[it]);
            if (three) {
                int j0 = (u / Ns) * Ns * R2 + kk;
#pragma unroll
                for (int k = 0; k < R2; ++k) buf[(j0 + k * Ns) * G + g] = t[it][k];
            } else {
#pragma unroll
                for (int k = 0; k < R2; ++k) v[it * R2 + k] = t[it][k];
            }
        }
    }
    if constexpr (three) {
        cx<R> tw3[ETW ? PPT / R3 : 1][ETW ? R3 : 1];
        if constexpr (ETW) {
#pragma unroll
            for (int it = 0; it < PPT / R3; ++it) {
                const int kk = ((tid + it * NT) / G) % (R1 * R2);
#pragma unroll
                for (int k = 1; k < R3; ++k) tw3[it][k] = ftw[kk * k * TWS];
            }
        }
        __syncthreads();
        constexpr int NB = PPT / R3, Ns = R1 * R2, STR = L / R3;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            int idx = tid + it * NT, g = idx % G, u = idx / G;
            cx<R> t[R3];
#pragma unroll
            for (int k = 0; k < R3; ++k) t[k] = buf[(u + k * STR) * G + g];
            int kk = u % Ns;                       // == u (Ns*R3 == L)
#pragma unroll
            for (int k = 1; k < R3; ++k) t[k] = cmul_v(t[k], ETW ? tw3[ETW ? it : 0][ETW ? k : 0] : ftw[kk * k * TWS]);
            Dft<R3>::run(t);
#pragma unroll
            for (int k = 0; k < R3; ++k) v[it * R3 + k] = t[k];
        }
    }
}

}  // namespace ssq
