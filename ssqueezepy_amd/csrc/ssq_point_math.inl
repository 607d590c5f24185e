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


// index of the internal bin map: row-major like Wx. (A tile-major layout, contiguous
// per 16-column tile of the reassignment kernel, was measured: it turns the
// producers' 32-byte runs into scattered writes and costs far more than it saves.)
__device__ __forceinline__ int64_t kidx_index(int64_t i, int64_t j, int64_t na, int64_t n) {
    (void)na;
    return i * n + j;
}

// |z| > gamma (fused form) and |z| < gamma (two-step form) for float32 data, screened in
// float32: |z|^2 is compared with gamma^2 scaled by (1 +- 2e-6); only values inside
// that sliver need the exact float-rounded hypot. `g2lo`/`g2hi` come from the host.
__device__ __forceinline__ bool mag_gt(float c, float d, double gamma) {
    float m2 = c * c + d * d;
    float g2 = (float)(gamma * gamma);
    if (m2 > g2 * 1.000004f) return true;
    if (m2 < g2 * 0.999996f) return false;
    return mag_of(c, d) > gamma;
}
__device__ __forceinline__ bool mag_gt(double c, double d, double gamma) { return mag_of(c, d) > gamma; }
// the float32 screen of mag_gt on its own: 1 = above, 0 = below, -1 = inside the sliver
__device__ __forceinline__ int mag_gt_screen(float c, float d, double gamma) {
    float m2 = c * c + d * d;
    float g2 = (float)(gamma * gamma);
    if (m2 > g2 * 1.000004f) return 1;
    if (m2 < g2 * 0.999996f) return 0;
    return -1;
}
__device__ __forceinline__ bool mag_lt(float c, float d, float gamma) {
    float m2 = c * c + d * d;
    float g2 = gamma * gamma;
    if (m2 > g2 * 1.000004f) return false;
    if (m2 < g2 * 0.999996f && g2 > 1e-36f) return true;
    return mag_of(c, d) < (double)gamma;
}
__device__ __forceinline__ bool mag_lt(double c, double d, double gamma) { return mag_of(c, d) < gamma; }

// ---- float32 screening -------------------------------------------------------
// Most of the cost of a point is the double division + double log2 of the exact bin
// map. For float32 data the same bin can be obtained from a float32 estimate
// t32 ~ (log2 w - vlmin)/dvl whenever t32 is not within `guard` bins of a rounding
// boundary (k + 1/2): `guard` bounds the estimate's total error (v_log_f32 <= 1 ulp,
// reciprocal-multiply division, float-rounded grid constants; finalize_params). Points
// inside the guard band -- a fraction ~2*guard, well under 1 % -- and anything
// non-finite take the exact path, so the result is identical to it everywhere.
// Returns the pre-flip bin, or -2 when the point needs the exact path.
// `werr` bounds the absolute error of the float32 `w` itself, `lerr` that of log2(w)
// (1.4427 * werr / w: a constant for the CWT, where the error of w is relative).
__device__ __forceinline__ int bin_screen_f32(float w, float werr, float lerr, const SsqParams& sp,
                                              int omax) {
    if (!(w > 1e-30f && w < 1e30f) || sp.guard >= 0.25f) return -2;
    float t, g = sp.guard;
    if (sp.grid == SSQ_GRID_LIN) {
        t = (w - sp.pf[0]) * sp.pf[1];
        g = g + (werr + (fabsf(w) + fabsf(sp.pf[0])) * 2e-7f) * sp.pf[1];
    } else {
        float wl = __log2f(w);
        if (sp.grid == SSQ_GRID_LOG) {
            t = (wl - sp.pf[0]) * sp.pf[1];
            g = g + lerr * sp.pf[1];
        } else {
            float dv = wl - sp.pf[1];
            if (fabsf(dv) < 2e-5f + lerr) return -2;   // on the segment boundary
            if (dv > 0.f) {
                t = dv * sp.pf[3];
                g = g + lerr * sp.pf[3] + fabsf(t) * 4e-7f;
                if (!(t < 1e9f) || !(g < 0.25f)) return -2;
                if (fabsf((t - floorf(t)) - 0.5f) < g) return -2;
                int k = (int)rintf(t) + (int)sp.pf[4];
                return k > omax ? omax : (k < 0 ? 0 : k);
            }
            t = (wl - sp.pf[0]) * sp.pf[2];
            g = g + lerr * sp.pf[2];
        }
    }
    g = g + fabsf(t) * 4e-7f;
    if (!(t < 1e9f) || !(t > -1e9f) || !(g < 0.25f)) return -2;
    if (t < g) return 0;                 // exact map gives 0 for t <= 0 and for 0 < t < 1/2
    if (fabsf((t - floorf(t)) - 0.5f) < g) return -2;
    int k = (int)rintf(t);
    return k > omax ? omax : k;
}

// bin (pre-flip) of a point from (dWx, Wx) = (a + ib, c + id), optional STFT row
// frequency; float32 data goes through the screen, double data straight to the exact map
// Branch-free form of `bin_screen_f32` for the CWT (error of w relative: werr = 5e-7 w,
// lerr = 1.4428 * 5e-7), grid kind as a template parameter: the same quantities and
// the same decisions, as selects. `ok` = the screen decided; the return value is the
// pre-flip bin when it did. Used inside the unrolled epilogues of the fused kernels,
// where a dozen divergent early-outs per point cost more than the arithmetic.
template <int GRID>
__device__ __forceinline__ int bin_screen_cwt(float w, const SsqParams& sp, int omax, bool& ok) {
    constexpr float REL = 5e-7f, LERR = 1.4428f * 5e-7f;
    float t, g = sp.guard;
    bool valid = true;
    int kofs = 0;
    bool seg1 = false;
    if (GRID == SSQ_GRID_LIN) {
        t = (w - sp.pf[0]) * sp.pf[1];
        g = g + (w * REL + (fabsf(w) + fabsf(sp.pf[0])) * 2e-7f) * sp.pf[1];
    } else {
        const float wl = __log2f(w);
        if (GRID == SSQ_GRID_LOG) {
            t = (wl - sp.pf[0]) * sp.pf[1];
            g = g + LERR * sp.pf[1];
        } else {
            const float dv = wl - sp.pf[1];
            valid = !(fabsf(dv) < 2e-5f + LERR);       // on the segment boundary
            seg1 = dv > 0.f;
            t = seg1 ? dv * sp.pf[3] : (wl - sp.pf[0]) * sp.pf[2];
            g = g + LERR * (seg1 ? sp.pf[3] : sp.pf[2]);
            kofs = seg1 ? (int)sp.pf[4] : 0;
        }
    }
    // w = 0, inf, NaN or out of float range make t (and with it g) infinite or NaN, and
    // |t| >= 1e6 alone pushes g past 1/4: one comparison covers every "cannot decide"
    g = g + fabsf(t) * 4e-7f;
    valid &= g < 0.25f;
    const float rt = rintf(t);
    const bool zero = !seg1 & (t < g);     // exact map: 0 for t <= 0 and for 0 < t < 1/2
    const bool near = (0.5f - fabsf(t - rt)) < g;      // within g of a rounding boundary
    ok = valid & (zero | !near);
    // t < g < 1/4 rounds to <= 0, so the clamp already yields 0 in the `zero` case
    const int k = (int)rt + kofs;
    return min(max(k, 0), omax);
}

// `bin_screen_f32` with its early-outs as selects (the grid kind is a scalar branch): the same estimate and the same
// guard, `ok` = the screen decided, the return value the pre-flip bin when it did. For the unrolled epilogue of the fused
// STFT kernel, where every early-out of a point is a divergent branch of the wavefront.
__device__ __forceinline__ int bin_screen_sel(float w, float werr, float lerr, const SsqParams& sp, int omax, bool& ok) {
    float t, g = sp.guard;
    bool valid = true, seg1 = false;
    int kofs = 0;
    if (sp.grid == SSQ_GRID_LIN) {
        t = (w - sp.pf[0]) * sp.pf[1];
        g = g + (werr + (fabsf(w) + fabsf(sp.pf[0])) * 2e-7f) * sp.pf[1];
    } else {
        const float wl = __log2f(w);
        if (sp.grid == SSQ_GRID_LOG) {
            t = (wl - sp.pf[0]) * sp.pf[1];
            g = g + lerr * sp.pf[1];
        } else {
            const float dv = wl - sp.pf[1];
            valid = !(fabsf(dv) < 2e-5f + lerr);       // on the segment boundary
            seg1 = dv > 0.f;
            t = seg1 ? dv * sp.pf[3] : (wl - sp.pf[0]) * sp.pf[2];
            g = g + lerr * (seg1 ? sp.pf[3] : sp.pf[2]);
            kofs = seg1 ? (int)sp.pf[4] : 0;
        }
    }
    // (w = inf or NaN, a log of 0, |t| beyond 1e6: g is infinite, NaN or past 1/4 -- one comparison)
    g = g + fabsf(t) * 4e-7f;
    valid &= g < 0.25f;
    const float rt = rintf(t);
    const bool zero = !seg1 & (t < g);     // exact map: 0 for t <= 0 and for 0 < t < 1/2
    const bool near = (0.5f - fabsf(t - rt)) < g;
    ok = valid & (zero | !near);
    const int k = (int)rt + kofs;
    return min(max(k, 0), omax);
}

// the two halves of the float32 `bin_of_point` below, for kernels that keep the
// exact path out of their unrolled loops: `_screen` returns -2 when undecided
__device__ __forceinline__ int bin_of_point_screen(float a, float b, float c, float d,
                                                   const SsqParams& sp, int omax) {
    float num = b * c - a * d;
    float m2 = c * c + d * d;
    float w32 = fabsf(num * __builtin_amdgcn_rcpf(m2 * 6.2831855f));
    return bin_screen_f32(w32, w32 * 5e-7f, 1.4428f * 5e-7f, sp, omax);
}
__device__ __forceinline__ int64_t bin_of_point_exact(float a, float b, float c, float d,
                                                      const SsqParams& sp, int64_t omax) {
    float num = b * c - a * d;
    float m2 = c * c + d * d;
    return bin_from_w(fabs((double)num / ((double)m2 * SSQ_TWO_PI)), sp, omax);
}

__device__ __forceinline__ int64_t bin_of_point(float a, float b, float c, float d, bool stft,
                                                float sfs, const SsqParams& sp, int64_t omax) {
    float num = b * c - a * d;
    float m2 = c * c + d * d;
    // hardware reciprocal (1 ulp): the estimate's relative error stays under 3e-7
    // (2pi as float 3e-8, product 6e-8, v_rcp_f32 1.2e-7, product 6e-8)
    float r32 = num * __builtin_amdgcn_rcpf(m2 * 6.2831855f);
    float w32, werr, lerr;
    if (stft) {
        w32 = fabsf(sfs - r32); werr = (fabsf(sfs) + fabsf(r32)) * 5e-7f;
        lerr = 1.4428f * werr * __builtin_amdgcn_rcpf(w32);
    } else { w32 = fabsf(r32); werr = w32 * 5e-7f; lerr = 1.4428f * 5e-7f; }
    int k = bin_screen_f32(w32, werr, lerr, sp, (int)omax);
    if (k != -2) return k;
    double r = (double)num / ((double)m2 * SSQ_TWO_PI);
    double w = stft ? fabs((double)sfs - r) : fabs(r);
    return bin_from_w(w, sp, omax);
}
// the STFT form of the float32 `bin_of_point` over `bin_screen_sel`: one rare branch (the exact map) per point, taken
// only where the bin is `wanted` (a point below gamma has none: its zero magnitude must not buy a float64 division)
__device__ __forceinline__ int bin_of_point_stft(float a, float b, float c, float d, float sfs,
                                                 const SsqParams& sp, int omax, bool wanted) {
    const float num = b * c - a * d;
    const float m2 = c * c + d * d;
    const float r32 = num * __builtin_amdgcn_rcpf(m2 * 6.2831855f);
    const float w32 = fabsf(sfs - r32), werr = (fabsf(sfs) + fabsf(r32)) * 5e-7f;
    const float lerr = 1.4428f * werr * __builtin_amdgcn_rcpf(w32);
    bool ok;
    int k = bin_screen_sel(w32, werr, lerr, sp, omax, ok);
    if (!ok && wanted) {
        const double r = (double)num / ((double)m2 * SSQ_TWO_PI);
        k = (int)bin_from_w(fabs((double)sfs - r), sp, omax);
    }
    return k;
}
// float64 data: the exact map costs a double division and a double log2 per point (a third of the float64 block
// kernels' instructions). The float32 screen decides it for all but the points near a rounding boundary as well, given
// an honest bound on the estimate's error: the inputs rounded to float32 (6e-8 each), the two products and their
// difference (cancellation: the bound is absolute, on the products' magnitudes), the reciprocal, 2 pi as a float.
__device__ __forceinline__ int64_t bin_of_point(double a, double b, double c, double d, bool stft,
                                                double sfs, const SsqParams& sp, int64_t omax) {
#ifndef SSQ_F64_NO_SCREEN
    {
        const float af = (float)a, bf = (float)b, cf = (float)c, df = (float)d;
        const float p1 = bf * cf, p2 = af * df, num = p1 - p2;
        const float m2 = cf * cf + df * df;
        const float inv = __builtin_amdgcn_rcpf(m2 * 6.2831855f);
        const float r32 = num * inv;
        // |num - exact| <= (|p1| + |p2|) * 2.5e-7 (inputs 2 x 6e-8, product 6e-8, difference 6e-8): 4e-7 budgeted; the
        // scale 1 / (2 pi m2) carries 4e-7 more (m2's inputs and sum, 2 pi, the reciprocal, the product)
        const float rerr = (fabsf(p1) + fabsf(p2)) * 4e-7f * inv + fabsf(r32) * 4e-7f;
        float w32, werr;
        if (stft) {
            const float sf = (float)sfs;
            w32 = fabsf(sf - r32); werr = rerr + (fabsf(sf) + fabsf(r32)) * 2e-7f;
        } else { w32 = fabsf(r32); werr = rerr; }
        // (magnitudes outside float32's comfortable range: m2 under- or overflows, the estimate is void)
        if (m2 > 1e-30f && m2 < 1e30f) {
            const float lerr = 1.4428f * werr * __builtin_amdgcn_rcpf(w32);
            const int k = bin_screen_f32(w32, werr, lerr, sp, (int)omax);
            if (k != -2) return k;
        }
    }
#endif
    double r = phase_ratio(a, b, c, d);
    double w = stft ? fabs(sfs - r) : fabs(r);
    return bin_from_w(w, sp, omax);
}
