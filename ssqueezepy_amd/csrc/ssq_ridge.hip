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
#include <limits>

namespace ssq {

// -------------------------------------------------------------------- elementwise
template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void ridge_energy_kernel(const T* __restrict__ Tf, T* __restrict__ en,
                                                           int64_t total) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        T a;
        if (CPLX) {
            const T re = Tf[2 * q], im = Tf[2 * q + 1];
            // |z| as glibc's hypotf forms it (double sqrt of the exact sum, rounded once more)
            if (sizeof(T) == 4) a = (T)sqrt((double)re * (double)re + (double)im * (double)im);
            else a = (T)hypot((double)re, (double)im);
        } else {
            a = fabs(Tf[q]);
        }
        en[q] = a * a;
    }
}

__device__ __forceinline__ float neg_log(float v) { return -logf(v); }
__device__ __forceinline__ double neg_log(double v) { return -log(v); }

template <typename T>
__global__ __launch_bounds__(64) void ridge_neglog_kernel(const T* __restrict__ en, T* __restrict__ E,
                                                          T eps, int64_t na, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= n) return;
    T mx = en[j];
    for (int64_t i = 1; i < na; ++i) { const T v = en[i * n + j]; mx = v > mx ? v : mx; }
    for (int64_t i = 0; i < na; ++i) E[i * n + j] = neg_log(en[i * n + j] / mx + eps);
}

template <typename T>
__global__ __launch_bounds__(64) void ridge_argmin_kernel(const T* __restrict__ pe, int64_t* __restrict__ ridge,
                                                          int64_t na, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= n) return;
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
    const int64_t r = ridge[j];
    if (r < 0 || r >= na) return;          // (the reference raises IndexError)
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

constexpr int RIDGE_NT = 1024;     // forward-pass workgroup
constexpr int RIDGE_F = 4;         // rows f per thread (they share the candidates read from LDS)

struct RidgeGeom {
    int nf;       // groups of RIDGE_F rows: group q owns rows q, q + nf, q + 2 nf, q + 3 nf
    int S;        // threads per group; thread s scans candidates g in [s C, (s + 1) C)
    int C;        // candidates per thread, a multiple of 4
    int TT;       // tile width in time steps (power of two)
    size_t lds;   // dynamic LDS bytes
};

template <typename T, typename TP>
static RidgeGeom ridge_geometry(int64_t na) {
    RidgeGeom g;
    g.nf = (int)((na + RIDGE_F - 1) / RIDGE_F);
    g.S = RIDGE_NT / g.nf;
    if (g.S < 1) g.S = 1;
    if (g.S > 32) g.S = 32;
    g.C = (int)(((na + g.S - 1) / g.S + 3) / 4 * 4);
    g.TT = 32;
    for (;;) {
        g.lds = (size_t)3 * g.S * g.C * sizeof(T) + (size_t)g.S * na * sizeof(T) +
                (size_t)na * (g.TT + 1) * sizeof(T) + 64;
        if (g.lds <= 150 * 1024 || g.TT == 1) break;
        g.TT /= 2;
    }
    return g;
}

template <typename T, typename TP>
__global__ __launch_bounds__(RIDGE_NT) void ridge_fw_kernel(const T* __restrict__ E, T* __restrict__ pe,
                                                            const TP* __restrict__ sc, TP pen, int na,
                                                            int64_t n, int nf, int S, int C, int TT) {
    extern __shared__ __attribute__((aligned(32))) unsigned char smem[];
    const int SC = S * C, W = TT + 1;
    T* prev0 = reinterpret_cast<T*>(smem);                      // pe[:, t-1], +inf beyond na
    T* prev1 = prev0 + SC;
    TP* negs = reinterpret_cast<TP*>(prev1 + SC);               // -sc[g] (a T-sized slot each)
    T* part = prev1 + 2 * SC;                                   // (S, na) partial minima
    T* tile = part + (size_t)S * na;                            // (na, TT + 1) E in, pe out
    const int tid = threadIdx.x;
    const T inf = std::numeric_limits<T>::infinity();
    for (int i = tid; i < SC; i += RIDGE_NT) {
        prev0[i] = inf; prev1[i] = inf;
        negs[i] = i < na ? -sc[i] : (TP)0;
    }
    const int q0 = tid / S, s = tid - q0 * S, qstride = RIDGE_NT / S;
    const int ntile = (int)((n + TT - 1) / TT);
    T* cur = prev0;
    T* nxt = prev1;
    for (int b = 0; b < ntile; ++b) {
        const int64_t t0 = (int64_t)b * TT;
        const int len = (int)((n - t0) < TT ? (n - t0) : TT);
        __syncthreads();
        for (int i = tid; i < na * TT; i += RIDGE_NT) {
            const int f = i / TT, tt = i - f * TT;
            if (tt < len) tile[f * W + tt] = E[(int64_t)f * n + t0 + tt];
        }
        __syncthreads();
        int tl = 0;
        if (b == 0) {                                           // pe[:, 0] = E[:, 0]
            for (int f = tid; f < na; f += RIDGE_NT) cur[f] = tile[f * W];
            __syncthreads();
            tl = 1;
        }
        for (; tl < len; ++tl) {
            for (int q = q0 < qstride ? q0 : nf; q < nf; q += qstride) {
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
            __syncthreads();
            for (int f = tid; f < na; f += RIDGE_NT) {
                T m = part[f];
                for (int k = 1; k < S; ++k) m = min_(m, part[k * na + f]);
                const T v = tile[f * W + tl] + m;
                nxt[f] = v;
                tile[f * W + tl] = v;
            }
            __syncthreads();
            T* sw = cur; cur = nxt; nxt = sw;
        }
        for (int i = tid; i < na * TT; i += RIDGE_NT) {
            const int f = i / TT, tt = i - f * TT;
            if (tt < len) pe[(int64_t)f * n + t0 + tt] = tile[f * W + tt];
        }
    }
}

// -------------------------------------------------------------------- backward pass
// One wavefront walks t = n-2 .. 0 (every step depends on the index chosen at t+1); the
// workgroup's four wavefronts stage the (na x TT+1)-column tiles of pe and E in LDS.
template <typename T, typename TP>
__global__ __launch_bounds__(256) void ridge_bw_kernel(const T* __restrict__ E, const T* __restrict__ pe,
                                                       const TP* __restrict__ sc, TP pen, T eps, int na,
                                                       int64_t n, int64_t* __restrict__ ridge, int TT) {
    extern __shared__ __attribute__((aligned(32))) unsigned char smem[];
    const int W = TT + 3;                                       // TT + 1 columns, odd stride
    T* peT = reinterpret_cast<T*>(smem);
    T* eT = peT + (size_t)na * W;
    TP* scs = reinterpret_cast<TP*>(eT + (size_t)na * W);
    int* idx = reinterpret_cast<int*>(scs + na + (na & 1));
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < na; i += 256) scs[i] = sc[i];
    if (n < 2) return;
    int r = (int)ridge[n - 1];
    const int K = (na + 63) / 64;
    const int ntile = (int)((n - 1 + TT - 1) / TT);             // steps t = 0 .. n-2
    for (int b = ntile - 1; b >= 0; --b) {
        const int64_t t0 = (int64_t)b * TT;
        const int len = (int)((n - 1 - t0) < TT ? (n - 1 - t0) : TT);   // steps in this tile
        __syncthreads();
        for (int i = tid; i < na * (len + 1); i += 256) {
            const int f = i / (len + 1), tt = i - f * (len + 1);
            peT[f * W + tt] = pe[(int64_t)f * n + t0 + tt];
            eT[f * W + tt] = E[(int64_t)f * n + t0 + tt];
        }
        for (int i = tid; i < len; i += 256) idx[i] = (int)ridge[t0 + i];
        __syncthreads();
        if (tid < 64) {
            for (int tl = len - 1; tl >= 0; --tl) {
                const T val = peT[r * W + tl + 1] - eT[r * W + tl + 1];
                const TP sr = scs[r];
                int best = -1;
                for (int k = K - 1; k >= 0; --k) {
                    const int f = k * 64 + lane;
                    bool hit = false;
                    if (f < na) {
                        const TP d = sr - scs[f];
                        const TP p = pen * (d * d);
                        const T c = peT[f * W + tl] + (T)p;
                        hit = fabs(val - c) < eps;
                    }
                    const unsigned long long bal = __ballot(hit);
                    if (bal) { best = k * 64 + 63 - __builtin_clzll(bal); break; }
                }
                r = best >= 0 ? best : idx[tl];
                if (lane == 0) idx[tl] = r;
            }
        }
        __syncthreads();
        for (int i = tid; i < len; i += 256) ridge[t0 + i] = idx[i];
    }
}

template <typename T, typename TP>
static int ridge_track_t(const T* E, T* pe, const TP* sc, double penalty, double eps, int64_t na,
                         int64_t n, int64_t* ridge, hipStream_t stream) {
    const RidgeGeom g = ridge_geometry<T, TP>(na);
    SSQ_REQUIRE(g.lds <= 160 * 1024, "ssq_ridge_track: %lld rows do not fit the workgroup's LDS",
                (long long)na);
    auto fw = ridge_fw_kernel<T, TP>;
    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fw),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
    hipLaunchKernelGGL(fw, dim3(1), dim3(RIDGE_NT), g.lds, stream, E, pe, sc, (TP)penalty, (int)na, n,
                       g.nf, g.S, g.C, g.TT);
    SSQ_LAUNCH_CHECK();
    hipLaunchKernelGGL((ridge_argmin_kernel<T>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream,
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
    hipLaunchKernelGGL(bwk, dim3(1), dim3(256), lds, stream, E, (const T*)pe, sc, (TP)penalty, (T)eps,
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
    SSQ_REQUIRE(energy && E, "ssq_ridge_neglog: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(na >= 1 && n >= 1, "ssq_ridge_neglog: bad shape (%lld, %lld)", (long long)na, (long long)n);
    const dim3 grid((unsigned)((n + 63) / 64));
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
    SSQ_REQUIRE(E && pe && sc && ridge, "ssq_ridge_track: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(na >= 1 && n >= 1 && na <= 16384, "ssq_ridge_track: bad shape (%lld, %lld)",
                (long long)na, (long long)n);
    SSQ_REQUIRE(dtype == SSQ_F64 || penalty_f32, "ssq_ridge_track: float32 data take a float32 penalty");
    hipStream_t s = as_stream(stream);
    if (dtype == SSQ_F32)
        return ridge_track_t<float, float>((const float*)E, (float*)pe, (const float*)sc, penalty, eps, na, n,
                                           ridge, s);
    if (penalty_f32)
        return ridge_track_t<double, float>((const double*)E, (double*)pe, (const float*)sc, penalty, eps,
                                            na, n, ridge, s);
    return ridge_track_t<double, double>((const double*)E, (double*)pe, (const double*)sc, penalty, eps, na,
                                         n, ridge, s);
}

int ssq_ridge_clear(int dtype, void* energy, const int64_t* ridge, double bw, void* ridge_e, int64_t na,
                    int64_t n, void* stream) {
    SSQ_REQUIRE(energy && ridge, "ssq_ridge_clear: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    const dim3 grid((unsigned)((n + 63) / 64));
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
