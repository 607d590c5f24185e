// ssq_tile_dev.h -- device-side pieces shared by the two column-tile kernels of the fused ssq_cwt form
// (ssq_tile_ordered.hip: the ticketed float32 tile; ssq_tile_f64.hip: the float64 tile with unordered adds).
// Include inside namespace ssq, after ssq_point_math.inl.
#pragma once

constexpr int TILE_W = 8;         // taps
constexpr int TILE_NOBIN = 0xFFFF;

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

// bin of a point the float32 screens could not decide (flipped as Tx wants it), or -1 when it
// does not contribute: the exact double sequence of the CPU path (~0.05 % of the points). Inline:
// a call would put the parameters on the stack and make the compiler wait for every load in
// flight at the join.
__device__ __forceinline__ int exact_bin(float2 W, float2 D, const SsqParams& sp, int omax, double gamma) {
    if (!(mag_of(W.x, W.y) > gamma)) return -1;
    const int ke = (int)bin_of_point_exact(D.x, D.y, W.x, W.y, sp, (int64_t)omax);
    return sp.flipud ? omax - ke : ke;
}

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

