// ssq_ridge.hip -- time-frequency ridge tracking on the device.
//
// Replaces the loop nests of ssqueezepy/ridge_extraction.py:113-232 for transforms that
// already live in HBM:
//   ssq_ridge_energy   np.abs(Tf)**2                                    (:129)
//   ssq_ridge_neglog   -log(energy / energy.max(axis=0) + eps)          (:138-139)
//   ssq_ridge_track    forward pass  pe[f,t] += min_g(pe[g,t-1] + P[f,g])   (:163-176),
//                      argmin per column (:157-158), backward pass (:202-214)
//   ssq_ridge_clear    ridge energy + zeroing of the +-bw band          (:145-150)
// The recurrence is sequential in time (a min-plus matrix-vector product per step whose
// sums are rounded, so steps cannot be re-associated without changing results): one
// workgroup owns it. Per step the na^2 candidates are spread over 1024 threads; the
// penalty P[f,g] = penalty * (s_f - s_g)^2 is re-formed in registers exactly as NumPy
// forms the matrix (difference, square, times penalty, each rounded in the penalty
// dtype) because the matrix itself (na^2 values) fits neither LDS nor the register
// file; E / pe move through LDS in (na x 32)-column tiles so that HBM sees 128-byte runs.
// Everything is bit-identical to the reference's loops for identical E.
#include "ssq_common.h"
#include <cmath>
#include <cstdlib>
#include <limits>

namespace ssq {

// -------------------------------------------------------------------- elementwise
// correctly rounded (clang's default for HIP: v_sqrt_f32 plus the +-1 ulp fix-up; the
// `__fsqrt_rn` intrinsic lowers to the bare 1-ulp instruction)
__device__ __forceinline__ float sqrt_rn(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_rn(double x) { return sqrt(x); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return fma(a, b, c); }

template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void ridge_energy_kernel(const T* __restrict__ Tf, T* __restrict__ en,
                                                           int64_t total) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        T a;
        if (CPLX) {
            // |z| as NumPy's complex `absolute` loop forms it: m * sqrt(fma(r, r, 1)) with
            // m = max(|re|, |im|), r = min / max (verified bit for bit against np.abs on 10^6
            // complex64 / complex128 values). Energies the reference finds exactly equal stay
            // exactly equal here, which is what its `< eps` tie tests in the tracking see.
            const T re = fabs(Tf[2 * q]), im = fabs(Tf[2 * q + 1]);
            const T m = re > im ? re : im, n = re > im ? im : re;
            a = m;
            if (m > T(0) && !isinf(m)) {
                const T r = n / m;
                a = m * sqrt_rn(fma_(r, r, T(1)));
            }
        } else {
            a = fabs(Tf[q]);
        }
        en[q] = a * a;
    }
}

// float32 `log` as NumPy's SIMD loop evaluates it (the reference's `np.log` on a float32
// array): frexp, fold the mantissa into [sqrt(1/2), sqrt(2)), a 5/5 rational minimax in
// x - 1 by Horner with fused multiply-adds, one division, and fma(exponent, ln 2, .).
// Verified bit for bit against np.log on 10^6 values; with NumPy's |z| above it makes the
// float32 negative-log energy -- and with it every tie the tracking sees -- identical to the
// reference's. Arguments here lie in [eps, 1 + eps]; anything but a positive normal number
// goes to logf.
__device__ __forceinline__ float np_logf(float x) {
    if (!(x >= 1.17549435e-38f) || isinf(x)) return logf(x);
    int ei;
    float m = frexpf(x, &ei);                      // [0.5, 1)
    float e = (float)ei;
    if (m < 0.70710678118654752440f) { m = m + m; e = e - 1.0f; }
    const float t = m - 1.0f;
    float num = fmaf(2.589979117907922693523e-002f, t, 3.808837741388407920751e-001f);
    num = fmaf(num, t, 1.480000633576506585156e+000f);
    num = fmaf(num, t, 2.112677543073053063722e+000f);
    num = fmaf(num, t, 9.999999999999998702752e-001f);
    num = fmaf(num, t, 0.0f);
    float den = fmaf(5.875095403124574342950e-003f, t, 1.546476374983906719538e-001f);
    den = fmaf(den, t, 9.864942958519418960339e-001f);
    den = fmaf(den, t, 2.453006071784736363091e+000f);
    den = fmaf(den, t, 2.612677543073109236779e+000f);
    den = fmaf(den, t, 1.0f);
    return fmaf(e, 6.931471805599453094172e-001f, num / den);
}
__device__ __forceinline__ float neg_log(float v) { return -np_logf(v); }
__device__ __forceinline__ double neg_log(double v) { return -log(v); }

template <typename T>
__global__ __launch_bounds__(64) void ridge_neglog_kernel(const T* __restrict__ en, T* __restrict__ E,
                                                          T eps, int64_t na, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= n) return;
    en += (int64_t)blockIdx.y * na * n; E += (int64_t)blockIdx.y * na * n;      // (blockIdx.y: transform of the batch)
    T mx = en[j];
    for (int64_t i = 1; i < na; ++i) { const T v = en[i * n + j]; mx = v > mx ? v : mx; }
    for (int64_t i = 0; i < na; ++i) E[i * n + j] = neg_log(en[i * n + j] / mx + eps);
}

template <typename T>
__global__ __launch_bounds__(64) void ridge_argmin_kernel(const T* __restrict__ pe, int64_t* __restrict__ ridge,
                                                          int64_t na, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= n) return;
    pe += (int64_t)blockIdx.y * na * n; ridge += (int64_t)blockIdx.y * n;
    T best = pe[j]; int64_t bi = 0;
    for (int64_t i = 1; i < na; ++i) { const T v = pe[i * n + j]; if (v < best) { best = v; bi = i; } }
    // np.unravel_index(argmin, (na, n))[1]: the column coordinate of the flat index
    ridge[j] = bi % n;
}

template <typename T>
__global__ __launch_bounds__(64) void ridge_clear_kernel(T* __restrict__ en, const int64_t* __restrict__ ridge,
                                                         double bw, T* __restrict__ ridge_e,
                                                         int64_t na, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= n) return;
    en += (int64_t)blockIdx.y * na * n; ridge += (int64_t)blockIdx.y * n;
    if (ridge_e) ridge_e += (int64_t)blockIdx.y * n;
    const int64_t r = ridge[j];
    if (r < 0 || r >= na) {                // (the reference raises IndexError; the tracking passes never produce it)
        if (ridge_e) ridge_e[j] = (T)0;    // ... but the caller's buffer is reused across ridges: never leave it stale
        return;
    }
    if (ridge_e) ridge_e[j] = en[r * n + j];
    // energy[int(r - bw):int(r + bw), j] = 0 with Python's slice rules
    int64_t lo = (int64_t)trunc((double)r - bw), hi = (int64_t)trunc((double)r + bw);
    if (lo < 0) { lo += na; if (lo < 0) lo = 0; }
    if (hi < 0) { hi += na; if (hi < 0) hi = 0; }
    if (lo > na) lo = na;
    if (hi > na) hi = na;
    for (int64_t i = lo; i < hi; ++i) en[i * n + j] = (T)0;
}

// -------------------------------------------------------------------- forward pass
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<double> { typedef double4 type; };

template <typename T>
__device__ __forceinline__ void lds_load4(const T* p, T (&v)[4]) {
    const typename Vec4<T>::type q = *reinterpret_cast<const typename Vec4<T>::type*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}

// v_min / v_min3 (NaN-free data: the reference's np.amin would propagate a NaN)
__device__ __forceinline__ float min_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double min_(double a, double b) { return __builtin_fmin(a, b); }

// four candidates g of one row f: m <- min(m, pe[g] + penalty * (s_f - s_g)^2), every
// operation rounded separately in its array's type, as NumPy forms the same values
template <typename T, typename TP>
struct Cand4 {
    static __device__ __forceinline__ void run(T& m, TP sf, TP pen, const T (&p4)[4], const TP (&g4)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const TP d = sf + g4[e];                             // s_f - s_g
            const TP p = pen * (d * d);
            m = min_(m, p4[e] + (T)p);
        }
    }
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <>
struct Cand4<float, float> {                                     // packed fp32: 2.5 instructions / candidate
    static __device__ __forceinline__ void run(float& m, float sf, float pen, const float (&p4)[4],
                                               const float (&g4)[4]) {
        const f32x2 s2 = {sf, sf}, pn = {pen, pen};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 g = {g4[2 * h], g4[2 * h + 1]}, pv = {p4[2 * h], p4[2 * h + 1]};
            f32x2 d = s2 + g;
            d = d * d;
            d = pn * d;
            d = pv + d;
            m = min_(min_(m, d.x), d.y);
        }
    }
};


// Forward-pass geometry. Work item (s, q), q < nf = ceil(na / F), s < S: the F rows q, q + nf,
// ..., q + (F - 1) nf (they share the candidates read from LDS) against the candidates g in
// [s C, (s + 1) C).
//   generic  : 1024 threads, C at run time, the penalty re-formed every step (2.5 VALU
//              instructions per candidate);
//   register : 512 threads (256 VGPRs each), (F, C) template constants, the thread's F x C
//              penalties formed once and kept in registers (1 instruction per candidate);
//              float32 penalties with nf * ceil(na / C) <= 512: up to 320 rows in float32.
struct RidgeGeom {
    int F, nf, S, C, TT, creg, nt;
    size_t lds;   // dynamic LDS bytes
};

template <typename T, typename TP>
static RidgeGeom ridge_geometry(int64_t na) {
    RidgeGeom g;
    g.F = 4; g.creg = 0; g.nt = 1024;
    static const bool no_reg = getenv("SSQ_DEBUG_RIDGE_GENERIC") != nullptr;
    if (sizeof(TP) == 4 && !no_reg) {
        const int fc[3][2] = {{4, 16}, {4, 32}, {5, 40}};      // instantiated below
        for (int i = 0; i < (sizeof(T) == 4 ? 3 : 1); ++i)
            if (((na + fc[i][0] - 1) / fc[i][0]) * ((na + fc[i][1] - 1) / fc[i][1]) <= 512) {
                g.F = fc[i][0]; g.creg = fc[i][1];
                break;
            }
    }
    g.nf = (int)((na + g.F - 1) / g.F);
    if (g.creg) {
        g.nt = 512; g.C = g.creg; g.S = (int)((na + g.C - 1) / g.C);
    } else {
        g.S = g.nt / g.nf;
        if (g.S < 1) g.S = 1;
        if (g.S > 32) g.S = 32;
        g.C = (int)(((na + g.S - 1) / g.S + 3) / 4 * 4);
    }
    g.TT = 32;
    for (;;) {
        g.lds = (size_t)3 * g.S * g.C * sizeof(T) + (size_t)g.S * na * sizeof(T) +
                (size_t)na * (g.TT + 1) * sizeof(T) + 64;
        if (g.lds <= 150 * 1024 || g.TT == 1) break;
        g.TT /= 2;
    }
    return g;
}

template <typename T, typename TP, int NT, int RIDGE_F, int CREG>
__global__ __launch_bounds__(NT) void ridge_fw_kernel(const T* __restrict__ E, T* __restrict__ pe,
                                                      const TP* __restrict__ sc, TP pen, int na,
                                                      int64_t n, int nf, int S, int Crt, int TT) {
    extern __shared__ __attribute__((aligned(32))) unsigned char smem[];
    E += (int64_t)blockIdx.x * na * n; pe += (int64_t)blockIdx.x * na * n;      // one workgroup per transform of the batch
    const int C = CREG ? CREG : Crt;
    const int SC = S * C, W = TT + 1;
    T* prev0 = reinterpret_cast<T*>(smem);                      // pe[:, t-1], +inf beyond na
    T* prev1 = prev0 + SC;
    TP* negs = reinterpret_cast<TP*>(prev1 + SC);               // -sc[g] (a T-sized slot each)
    T* part = prev1 + 2 * SC;                                   // (S, na) partial minima
    T* tile = part + (size_t)S * na;                            // (na, TT + 1) E in, pe out
    const int tid = threadIdx.x;
    const T inf = std::numeric_limits<T>::infinity();
    for (int i = tid; i < SC; i += NT) {
        prev0[i] = inf; prev1[i] = inf;
        negs[i] = i < na ? -sc[i] : (TP)0;
    }
    // work item w = s * nf + q: the lanes of a wavefront share the slice s (at most two slices
    // per wavefront), so the candidate reads below are LDS broadcasts
    const int nwork = nf * S, s0 = tid / nf, q0 = tid - s0 * nf;
    const int ntile = (int)((n + TT - 1) / TT);
    const int lt = 31 - __builtin_clz((unsigned)TT);           // TT is a power of two
    T* cur = prev0;
    T* nxt = prev1;
    // register variant: this thread's penalties P[f_u, g], g in its slice (nwork <= NT)
    constexpr int NP = CREG ? CREG : 1;
    TP Preg[RIDGE_F][NP];
    if constexpr (CREG != 0) {
        __syncthreads();
        if (tid < nwork) {
#pragma unroll
            for (int u = 0; u < RIDGE_F; ++u) {
                const int f = q0 + u * nf;
                const TP sf = f < na ? -negs[f] : (TP)0;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const TP d = sf + negs[s0 * C + k];          // s_f - s_g
                    Preg[u][k] = pen * (d * d);
                }
            }
        }
    }
    for (int b = 0; b < ntile; ++b) {
        const int64_t t0 = (int64_t)b * TT;
        const int len = (int)((n - t0) < TT ? (n - t0) : TT);
        __syncthreads();
        for (int i0 = tid; i0 < na * TT; i0 += NT * 8) {       // 8 loads in flight per thread
            T v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * NT, f = i >> lt, tt = i & (TT - 1);
                v8[j] = (T)0;
                if (f < na && tt < len) v8[j] = E[(int64_t)f * n + t0 + tt];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * NT, f = i >> lt, tt = i & (TT - 1);
                if (f < na && tt < len) tile[f * W + tt] = v8[j];
            }
        }
        __syncthreads();
        int tl = 0;
        if (b == 0) {                                           // pe[:, 0] = E[:, 0]
            for (int f = tid; f < na; f += NT) cur[f] = tile[f * W];
            __syncthreads();
            tl = 1;
        }
        for (; tl < len; ++tl) {
            if constexpr (CREG != 0) {
                if (tid < nwork) {
                    T m[RIDGE_F];
#pragma unroll
                    for (int u = 0; u < RIDGE_F; ++u) m[u] = inf;
                    const T* pv = cur + s0 * C;
#pragma unroll
                    for (int k = 0; k < NP; k += 4) {
                        T p4[4];
                        lds_load4(pv + k, p4);
#pragma unroll
                        for (int u = 0; u < RIDGE_F; ++u) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) m[u] = min_(m[u], p4[e] + (T)Preg[u][(k + e) % NP]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RIDGE_F; ++u) {
                        const int f = q0 + u * nf;
                        if (f < na) part[s0 * na + f] = m[u];
                    }
                }
            } else {
                for (int w = tid; w < nwork; w += NT) {
                    int s = s0, q = q0;
                    if (w != tid) { s = w / nf; q = w - s * nf; }
                    TP sf[RIDGE_F]; T m[RIDGE_F];
#pragma unroll
                    for (int u = 0; u < RIDGE_F; ++u) {
                        const int f = q + u * nf;
                        sf[u] = f < na ? -negs[f] : (TP)0;
                        m[u] = inf;
                    }
                    const T* pv = cur + s * C;
                    const TP* ng = negs + s * C;
                    for (int k = 0; k < C; k += 4) {
                        T p4[4]; TP g4[4];
                        lds_load4(pv + k, p4);
                        lds_load4(ng + k, g4);
#pragma unroll
                        for (int u = 0; u < RIDGE_F; ++u) Cand4<T, TP>::run(m[u], sf[u], pen, p4, g4);
                    }
#pragma unroll
                    for (int u = 0; u < RIDGE_F; ++u) {
                        const int f = q + u * nf;
                        if (f < na) part[s * na + f] = m[u];
                    }
                }
            }
            __syncthreads();
            for (int f = tid; f < na; f += NT) {
                T m = part[f];
                for (int k = 1; k < S; ++k) m = min_(m, part[k * na + f]);
                const T v = tile[f * W + tl] + m;
                nxt[f] = v;
                tile[f * W + tl] = v;
            }
            __syncthreads();
            T* sw = cur; cur = nxt; nxt = sw;
        }
        for (int i = tid; i < na * TT; i += NT) {
            const int f = i >> lt, tt = i & (TT - 1);
            if (tt < len) pe[(int64_t)f * n + t0 + tt] = tile[f * W + tt];
        }
    }
}

// -------------------------------------------------------------------- backward pass
// One wavefront walks t = n-2 .. 0 (every step depends on the index chosen at t+1); the
// workgroup's eight wavefronts stage the (na x TT+1)-column tiles of pe and E in LDS.
constexpr int RIDGE_BW_NT = 512;
template <typename T, typename TP>
__global__ __launch_bounds__(RIDGE_BW_NT) void ridge_bw_kernel(const T* __restrict__ E, const T* __restrict__ pe,
                                                       const TP* __restrict__ sc, TP pen, T eps, int na,
                                                       int64_t n, int64_t* __restrict__ ridge, int TT) {
    extern __shared__ __attribute__((aligned(32))) unsigned char smem[];
    E += (int64_t)blockIdx.x * na * n; pe += (int64_t)blockIdx.x * na * n; ridge += (int64_t)blockIdx.x * n;
    const int W = TT + 3;                                       // TT + 1 columns, odd stride
    T* peT = reinterpret_cast<T*>(smem);
    T* eT = peT + (size_t)na * W;
    TP* scs = reinterpret_cast<TP*>(eT + (size_t)na * W);
    int* idx = reinterpret_cast<int*>(scs + na + (na & 1));
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < na; i += RIDGE_BW_NT) scs[i] = sc[i];
    const int lt = 31 - __builtin_clz((unsigned)TT);           // TT is a power of two
    if (n < 2) return;
    int r = (int)ridge[n - 1];
    const int K = (na + 63) / 64;
    const int ntile = (int)((n - 1 + TT - 1) / TT);             // steps t = 0 .. n-2
    for (int b = ntile - 1; b >= 0; --b) {
        const int64_t t0 = (int64_t)b * TT;
        const int len = (int)((n - 1 - t0) < TT ? (n - 1 - t0) : TT);   // steps in this tile
        __syncthreads();
        for (int i0 = tid; i0 < na * TT; i0 += RIDGE_BW_NT * 4) {   // columns t0 .. t0 + len - 1
            T a4[4], b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * RIDGE_BW_NT, f = i >> lt, tt = i & (TT - 1);
                a4[j] = b4[j] = (T)0;
                if (f < na && tt < len) {
                    a4[j] = pe[(int64_t)f * n + t0 + tt];
                    b4[j] = E[(int64_t)f * n + t0 + tt];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * RIDGE_BW_NT, f = i >> lt, tt = i & (TT - 1);
                if (f < na && tt < len) { peT[f * W + tt] = a4[j]; eT[f * W + tt] = b4[j]; }
            }
        }
        for (int f = tid; f < na; f += RIDGE_BW_NT) {              // column t0 + len (<= n - 1)
            peT[f * W + len] = pe[(int64_t)f * n + t0 + len];
            eT[f * W + len] = E[(int64_t)f * n + t0 + len];
        }
        for (int i = tid; i < len; i += RIDGE_BW_NT) idx[i] = (int)ridge[t0 + i];
        __syncthreads();
        if (tid < 64) {
            for (int tl = len - 1; tl >= 0; --tl) {
                const T val = peT[r * W + tl + 1] - eT[r * W + tl + 1];
                const TP sr = scs[r];
                const int fwd = idx[tl];                        // the forward argmin of column tl
                int best = -1;
                for (int k1 = K; k1 > 0 && best < 0; k1 -= 4) {  // rows [64 (k1 - 4), 64 k1), top first
                    T c4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {               // independent LDS reads
                        const int f = (k1 - 1 - j) * 64 + lane;
                        c4[j] = (T)0;
                        if (f >= 0 && f < na) {
                            const TP d = sr - scs[f];
                            c4[j] = peT[f * W + tl] + (T)(pen * (d * d));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = (k1 - 1 - j) * 64 + lane;
                        const bool hit = f >= 0 && f < na && fabs(val - c4[j]) < eps;
                        const unsigned long long bal = __ballot(hit);
                        if (bal && best < 0) best = (k1 - 1 - j) * 64 + 63 - __builtin_clzll(bal);
                    }
                }
                r = best >= 0 ? best : fwd;
                if (lane == 0) idx[tl] = r;
            }
        }
        __syncthreads();
        for (int i = tid; i < len; i += RIDGE_BW_NT) ridge[t0 + i] = idx[i];
    }
}

template <typename T, typename TP>
static int ridge_track_t(const T* E, T* pe, const TP* sc, double penalty, double eps, int64_t na,
                         int64_t n, int64_t* ridge, int64_t batch, hipStream_t stream) {
    const RidgeGeom g = ridge_geometry<T, TP>(na);
    SSQ_REQUIRE(g.lds <= 160 * 1024, "ssq_ridge_track: %lld rows do not fit the workgroup's LDS",
                (long long)na);
#define FW_LAUNCH(NT, F, CREG)                                                                     \
    do {                                                                                           \
        auto fw = ridge_fw_kernel<T, TP, NT, F, CREG>;                                             \
        SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fw),                       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds)); \
        hipLaunchKernelGGL(fw, dim3((unsigned)batch), dim3(NT), g.lds, stream, E, pe, sc, (TP)penalty, (int)na, n, \
                           g.nf, g.S, g.C, g.TT);                                                  \
    } while (0)
    constexpr bool PF = sizeof(TP) == 4, TF = sizeof(T) == 4;   // register variants: see ridge_geometry
    if (PF && g.creg == 16) FW_LAUNCH(512, 4, (PF ? 16 : 0));
    else if (PF && TF && g.creg == 32) FW_LAUNCH(512, 4, (PF && TF ? 32 : 0));
    else if (PF && TF && g.creg == 40) FW_LAUNCH(512, 5, (PF && TF ? 40 : 0));
    else FW_LAUNCH(1024, 4, 0);
#undef FW_LAUNCH
    SSQ_LAUNCH_CHECK();
    hipLaunchKernelGGL((ridge_argmin_kernel<T>), dim3((unsigned)((n + 63) / 64), (unsigned)batch), dim3(64), 0, stream,
                       (const T*)pe, ridge, na, n);
    SSQ_LAUNCH_CHECK();
    int TT = 32;
    size_t lds;
    for (;;) {
        lds = (size_t)2 * na * (TT + 3) * sizeof(T) + (size_t)(na + 2) * sizeof(TP) + (size_t)TT * 4 + 64;
        if (lds <= 150 * 1024 || TT == 1) break;
        TT /= 2;
    }
    SSQ_REQUIRE(lds <= 160 * 1024, "ssq_ridge_track: %lld rows do not fit the workgroup's LDS",
                (long long)na);
    auto bwk = ridge_bw_kernel<T, TP>;
    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bwk),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(bwk, dim3((unsigned)batch), dim3(RIDGE_BW_NT), lds, stream, E, (const T*)pe, sc, (TP)penalty, (T)eps,
                       (int)na, n, ridge, TT);
    SSQ_LAUNCH_CHECK();
    return 0;
}

}  // namespace ssq

using namespace ssq;

extern "C" {

int ssq_ridge_energy(int dtype, int is_complex, const void* Tf, void* energy, int64_t na, int64_t n,
                     void* stream) {
    SSQ_REQUIRE(Tf && energy, "ssq_ridge_energy: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(na >= 1 && n >= 1, "ssq_ridge_energy: bad shape (%lld, %lld)", (long long)na, (long long)n);
    const int64_t total = na * n;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t s = as_stream(stream);
#define LAUNCH(T, C) hipLaunchKernelGGL((ridge_energy_kernel<T, C>), dim3(blocks), dim3(256), 0, s, \
                                        (const T*)Tf, (T*)energy, total)
    if (dtype == SSQ_F32) { if (is_complex) LAUNCH(float, true); else LAUNCH(float, false); }
    else { if (is_complex) LAUNCH(double, true); else LAUNCH(double, false); }
#undef LAUNCH
    SSQ_LAUNCH_CHECK();
    return 0;
}

int ssq_ridge_neglog(int dtype, const void* energy, void* E, double eps, int64_t na, int64_t n,
                     void* stream) {
    return ssq_ridge_neglog_batch(dtype, energy, E, eps, na, n, 1, stream);
}

int ssq_ridge_neglog_batch(int dtype, const void* energy, void* E, double eps, int64_t na, int64_t n,
                           int64_t batch, void* stream) {
    SSQ_REQUIRE(energy && E, "ssq_ridge_neglog: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(na >= 1 && n >= 1 && batch >= 1 && batch <= 65535, "ssq_ridge_neglog: bad shape (%lld, %lld, %lld)",
                (long long)batch, (long long)na, (long long)n);
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)batch);
    hipStream_t s = as_stream(stream);
    if (dtype == SSQ_F32)
        hipLaunchKernelGGL((ridge_neglog_kernel<float>), grid, dim3(64), 0, s, (const float*)energy,
                           (float*)E, (float)eps, na, n);
    else
        hipLaunchKernelGGL((ridge_neglog_kernel<double>), grid, dim3(64), 0, s, (const double*)energy,
                           (double*)E, eps, na, n);
    SSQ_LAUNCH_CHECK();
    return 0;
}

int ssq_ridge_track(int dtype, int penalty_f32, const void* E, void* pe, const void* sc, double penalty,
                    double eps, int64_t na, int64_t n, int64_t* ridge, void* stream) {
    return ssq_ridge_track_batch(dtype, penalty_f32, E, pe, sc, penalty, eps, na, n, ridge, 1, stream);
}

int ssq_ridge_track_batch(int dtype, int penalty_f32, const void* E, void* pe, const void* sc, double penalty,
                          double eps, int64_t na, int64_t n, int64_t* ridge, int64_t batch, void* stream) {
    SSQ_REQUIRE(E && pe && sc && ridge, "ssq_ridge_track: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(na >= 1 && n >= 1 && na <= 16384 && batch >= 1 && batch <= 65535,
                "ssq_ridge_track: bad shape (%lld, %lld, %lld)", (long long)batch, (long long)na, (long long)n);
    SSQ_REQUIRE(dtype == SSQ_F64 || penalty_f32, "ssq_ridge_track: float32 data take a float32 penalty");
    hipStream_t s = as_stream(stream);
    if (dtype == SSQ_F32)
        return ridge_track_t<float, float>((const float*)E, (float*)pe, (const float*)sc, penalty, eps, na, n,
                                           ridge, batch, s);
    if (penalty_f32)
        return ridge_track_t<double, float>((const double*)E, (double*)pe, (const float*)sc, penalty, eps,
                                            na, n, ridge, batch, s);
    return ridge_track_t<double, double>((const double*)E, (double*)pe, (const double*)sc, penalty, eps, na,
                                         n, ridge, batch, s);
}

int ssq_ridge_clear(int dtype, void* energy, const int64_t* ridge, double bw, void* ridge_e, int64_t na,
                    int64_t n, void* stream) {
    return ssq_ridge_clear_batch(dtype, energy, ridge, bw, ridge_e, na, n, 1, stream);
}

int ssq_ridge_clear_batch(int dtype, void* energy, const int64_t* ridge, double bw, void* ridge_e, int64_t na,
                          int64_t n, int64_t batch, void* stream) {
    SSQ_REQUIRE(energy && ridge, "ssq_ridge_clear: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(batch >= 1 && batch <= 65535, "ssq_ridge_clear: bad batch %lld", (long long)batch);
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)batch);
    hipStream_t s = as_stream(stream);
    if (dtype == SSQ_F32)
        hipLaunchKernelGGL((ridge_clear_kernel<float>), grid, dim3(64), 0, s, (float*)energy, ridge, bw,
                           (float*)ridge_e, na, n);
    else
        hipLaunchKernelGGL((ridge_clear_kernel<double>), grid, dim3(64), 0, s, (double*)energy, ridge, bw,
                           (double*)ridge_e, na, n);
    SSQ_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
