// ssq_point_math.inl -- per-point arithmetic of the phase transform and the bin map.
// Included inside `namespace ssq` by every translation unit that computes bins, all
// of which are compiled with -ffp-contract=off: the operations below are specified
// one by one (oracle/ssq_oracle.c, "numba typing") so bin indices reproduce the CPU
// path bit for bit.
// ------------------------------------------------------------ device math
#define SSQ_TWO_PI 6.283185307179586

// |z| as the CPU path sees it: abs(complex64) is a float32 (hypotf), abs(complex128)
// a float64 (hypot). sqrt of the exact double sum of squares, rounded to float,
// reproduces a correctly rounded hypotf except for ~1e-9 of inputs.
__device__ __forceinline__ double mag_of(float c, float d) {
    return (double)(float)sqrt((double)c * (double)c + (double)d * (double)d);
}
__device__ __forceinline__ double mag_of(double c, double d) { return hypot(c, d); }

// Im(dWx / Wx) / 2pi with the reference CPU path's types (numba promotion rules):
// float32 numerator and |Wx|^2, float64 from the 2pi literal onward.
__device__ __forceinline__ double phase_ratio(float a, float b, float c, float d) {
    float num = b * c - a * d;
    float m2 = c * c + d * d;
    return (double)num / ((double)m2 * SSQ_TWO_PI);
}
__device__ __forceinline__ double phase_ratio(double a, double b, double c, double d) {
    return (b * c - a * d) / ((c * c + d * d) * SSQ_TWO_PI);
}

__device__ __forceinline__ int64_t clamp_round(double t, int64_t omax) {
    if (!(t > 0.0)) return 0;            // negatives, -inf, NaN -> bin 0
    if (t >= (double)omax) return omax;  // also +inf
    int64_t k = (int64_t)rint(t);        // round half to even
    return k > omax ? omax : k;
}

// closed-form nearest-bin map (find_closest_{log,lin}; algos.py:356-449)
__device__ __forceinline__ int64_t bin_from_wl(double wl, const SsqParams& sp, int64_t omax) {
    if (sp.grid == SSQ_GRID_LOG) return clamp_round((wl - sp.p[0]) / sp.p[1], omax);
    if (wl > sp.p[1]) {
        double t = (wl - sp.p[1]) / sp.p[3];
        if (!(t < 4.0e18)) return omax;
        int64_t k = (int64_t)rint(t) + (int64_t)sp.p[4];
        return k > omax ? omax : (k < 0 ? 0 : k);
    }
    return clamp_round((wl - sp.p[0]) / sp.p[2], omax);
}

__device__ __forceinline__ int64_t bin_from_w(double w, const SsqParams& sp, int64_t omax) {
    if (sp.grid == SSQ_GRID_LIN) return clamp_round((w - sp.p[0]) / sp.p[1], omax);
    return bin_from_wl(log2(w), sp, omax);
}

// bins from a stored phase transform: numba types np.log2(float32) as float32
__device__ __forceinline__ int64_t bin_from_stored_w(float w, const SsqParams& sp, int64_t omax) {
    if (sp.grid == SSQ_GRID_LIN) return clamp_round(((double)w - sp.p[0]) / sp.p[1], omax);
    return bin_from_wl((double)log2f(w), sp, omax);
}
__device__ __forceinline__ int64_t bin_from_stored_w(double w, const SsqParams& sp, int64_t omax) {
    return bin_from_w(w, sp, omax);
}

