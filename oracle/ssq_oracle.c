/*
 * ssq_oracle.c -- CPU restatement of the reference's synchrosqueezing loop nests.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is imported, linked or executed
 * by the product path (ssqueezepy_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * What it restates (reference = OverLordGoldDragon/ssqueezepy v0.6.6, all paths
 * relative to /root/reference/ssqueezepy/):
 *   orc_phase_cwt_*      algos.py:720-740   (_phase_cwt / _phase_cwt_par)
 *   orc_phase_stft_*     algos.py:794-816   (_phase_stft / _phase_stft_par)
 *   orc_ssq_cwt_*        algos.py:859-953   (_ssq_cwt_log_piecewise, _ssq_cwt_log,
 *                                            _ssq_cwt_lin and their _par twins)
 *   orc_ssq_stft_*       algos.py:956-984   (_ssq_stft / _ssq_stft_par)
 *   orc_indexed_sum_*    algos.py:172-250   (_indexed_sum_{log,log_piecewise,lin})
 *   orc_replace_under_abs_*  algos.py:545-557
 *   orc_buffer_*         utils/stft_utils.py:69-98 (_buffer / _buffer_par)
 *   orc_ridge_fw_*, orc_ridge_bw_*   ridge_extraction.py:163-232 (forward / backward
 *                        penalised-energy passes of extract_ridges)
 *
 * Arithmetic types. The reference's loop nests are Python source compiled by numba;
 * for float32 inputs the *type* of three sub-expressions depends on who evaluates
 * the source, so the oracle takes a `typing` argument:
 *   ORC_TYPING_NUMBA (0): numba's promotion rules -- float32 op float64-literal is
 *       float64, `float32 ** int` stays float32 (numba BinOpPower: "Ensure that
 *       float32 ** int doesn't go through DP computations"). So
 *       num = B*C - A*D and |Wx|^2 = C*C + D*D are float32, and everything from
 *       `* 6.283185307179586` onward (division, log2, bin arithmetic) is float64.
 *       This is what the reference's CPU path computes when run as shipped, and it
 *       is the semantics the HIP kernels implement.
 *   ORC_TYPING_NUMPY (1): NumPy-2 scalar rules (Python floats are weak) -- the whole
 *       chain stays float32. This is what the same source computes with numba
 *       disabled, which is how the reference's own test-suite runs it for coverage
 *       (tests/z_all_test.py:8-20) and the only way it can be executed in the build
 *       container (numba is not installable). Golden vectors generated that way
 *       (oracle/gen_golden.py) pin this mode bit-for-bit; the two modes share every
 *       line of control flow and differ only in the declared type of w/wl/t.
 * For float64 inputs the two modes differ only in how `x**2` is formed (see sq64).
 * `round` is round-half-to-even in both (Python 3 `round`, numba lowers it to
 * llvm.rint). Degenerate w == 0 (log2 -> -inf) maps to bin 0 before the optional
 * flip, matching `max(..., 0)` at algos.py:907 (SURVEY.md section 8(a'), item 7).
 *
 * `out[k, j] += Wx[i, j] * const[i]`: when the caller's `const` vector is float64
 * while the data is complex64 (the 'log-piecewise' case, where const = ln2 / nv with
 * nv a float64 array: ssqueezing.py:126, utils/cwt_utils.py:397-409) the product
 * and the sum are formed in double and rounded to float32 on store, as both numba
 * and NumPy do for complex64 * float64; otherwise they are float32 operations.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define ORC_TYPING_NUMBA 0
#define ORC_TYPING_NUMPY 1

#define ORC_GRID_LOG 0
#define ORC_GRID_LOG_PIECEWISE 1
#define ORC_GRID_LIN 2

#define TWO_PI 6.283185307179586

/* `x**2`: numba lowers a literal integer power to a multiplication; un-jitted, a
 * NumPy scalar `**` goes through libm pow()/powf(), which glibc does not always
 * round correctly (observed: 3 of 9600 random doubles differ from x*x by 1 ulp) */
static volatile double k_two = 2.0;     /* volatile: keep gcc from folding pow(x,2) */
static inline double sq64(double x, int typing) {
    return typing == 1 ? pow(x, k_two) : x * x;
}
static inline float sq32(float x, int typing) {
    return typing == 1 ? powf(x, (float)k_two) : x * x;
}

/* round-half-even of a finite-or-not double, clamped into [0, omax] */
static inline int64_t clamp_round(double t, int64_t omax) {
    if (!(t > 0.0)) return 0;            /* negatives, -inf, NaN */
    if (t >= (double)omax) return omax;  /* also +inf */
    int64_t k = (int64_t)rint(t);
    return k > omax ? omax : k;
}

/* grid params: p[0..4]
 *   LOG:           vlmin, dvl
 *   LOG_PIECEWISE: vlmin0, vlmin1, dvl0, dvl1, idx1
 *   LIN:           vmin, dv
 * (reference: _get_params_find_closest_log, algos.py:356-374; lin: algos.py:84-86)
 */

/* ------------------------------------------------------------------ float64 */
static inline int64_t bin_from_w_f64(double w, int grid, const double* p,
                                     int64_t omax) {
    if (grid == ORC_GRID_LIN)
        return clamp_round((w - p[0]) / p[1], omax);
    double wl = log2(w);
    if (grid == ORC_GRID_LOG)
        return clamp_round((wl - p[0]) / p[1], omax);
    /* log-piecewise: algos.py:871-875 */
    if (wl > p[1]) {
        double t = (wl - p[1]) / p[3];
        if (!(t < 4.0e18)) return omax;
        int64_t k = (int64_t)rint(t) + (int64_t)p[4];
        return k > omax ? omax : (k < 0 ? 0 : k);
    }
    return clamp_round((wl - p[0]) / p[2], omax);
}

/* ------------------------------------------------- float32, NumPy-typed chain */
static inline int64_t clamp_round_f32(float t, int64_t omax) {
    if (!(t > 0.0f)) return 0;
    if (t >= (float)omax) return omax;
    int64_t k = (int64_t)rintf(t);
    return k > omax ? omax : k;
}

static inline int64_t bin_from_w_f32np(float w, int grid, const double* p,
                                       int64_t omax) {
    if (grid == ORC_GRID_LIN)
        return clamp_round_f32((w - (float)p[0]) / (float)p[1], omax);
    float wl = log2f(w);
    if (grid == ORC_GRID_LOG)
        return clamp_round_f32((wl - (float)p[0]) / (float)p[1], omax);
    if (wl > (float)p[1]) {
        float t = (wl - (float)p[1]) / (float)p[3];
        if (!(t < 4.0e18f)) return omax;
        int64_t k = (int64_t)rintf(t) + (int64_t)p[4];
        return k > omax ? omax : (k < 0 ? 0 : k);
    }
    return clamp_round_f32((wl - (float)p[0]) / (float)p[2], omax);
}

/* =========================================================== phase transforms */
void orc_phase_cwt_f64(const double* Wx, const double* dWx, double* out,
                       int64_t na, int64_t n, double gamma, int typing) {
    for (int64_t q = 0; q < na * n; ++q) {
        double C = Wx[2*q], D = Wx[2*q+1], A = dWx[2*q], B = dWx[2*q+1];
        if (hypot(C, D) < gamma) { out[q] = INFINITY; continue; }
        out[q] = fabs((B*C - A*D) / ((sq64(C, typing) + sq64(D, typing)) * TWO_PI));
    }
}

void orc_phase_cwt_f32(const float* Wx, const float* dWx, float* out,
                       int64_t na, int64_t n, float gamma, int typing) {
    for (int64_t q = 0; q < na * n; ++q) {
        float C = Wx[2*q], D = Wx[2*q+1], A = dWx[2*q], B = dWx[2*q+1];
        if (hypotf(C, D) < gamma) { out[q] = INFINITY; continue; }
        float num = B*C - A*D, m2 = sq32(C, typing) + sq32(D, typing);
        if (typing == ORC_TYPING_NUMPY)
            out[q] = fabsf(num / (m2 * (float)TWO_PI));
        else
            out[q] = (float)fabs((double)num / ((double)m2 * TWO_PI));
    }
}

void orc_phase_stft_f64(const double* Sx, const double* dSx, const double* Sfs,
                        double* out, int64_t na, int64_t n, double gamma,
                        int typing) {
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < n; ++j) {
            int64_t q = i*n + j;
            double C = Sx[2*q], D = Sx[2*q+1], A = dSx[2*q], B = dSx[2*q+1];
            if (hypot(C, D) < gamma) { out[q] = INFINITY; continue; }
            out[q] = fabs(Sfs[i] - (B*C - A*D)
                          / ((sq64(C, typing) + sq64(D, typing)) * TWO_PI));
        }
}

void orc_phase_stft_f32(const float* Sx, const float* dSx, const float* Sfs,
                        float* out, int64_t na, int64_t n, float gamma,
                        int typing) {
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < n; ++j) {
            int64_t q = i*n + j;
            float C = Sx[2*q], D = Sx[2*q+1], A = dSx[2*q], B = dSx[2*q+1];
            if (hypotf(C, D) < gamma) { out[q] = INFINITY; continue; }
            float num = B*C - A*D, m2 = sq32(C, typing) + sq32(D, typing);
            if (typing == ORC_TYPING_NUMPY)
                out[q] = fabsf(Sfs[i] - num / (m2 * (float)TWO_PI));
            else
                out[q] = (float)fabs((double)Sfs[i]
                                     - (double)num / ((double)m2 * TWO_PI));
        }
}

/* ======================================================= fused reassignment
 * One routine per dtype covers the four reference kernels: `Sfs == NULL` selects the
 * CWT form  w = |Im(dWx/Wx)| / 2pi, otherwise the STFT form w = |Sfs[i] - ...|.
 * `cst` is the per-row weight vector; `cst_f64` tells whether it is double (see the
 * header). `kout`, if non-NULL, also receives the bin of every point (-1 where the
 * point is below threshold) so index parity can be tested exactly.
 * Loop order = the reference's `_par` form (columns outer, rows inner ascending);
 * the serial form visits the same (k, j) cells in the same row order, so the float
 * sums are identical.
 */
void orc_ssq_f64(const double* Wx, const double* dWx, const double* Sfs,
                 double* out, const double* cst, int64_t na, int64_t n,
                 double gamma, int grid, const double* p, int flipud,
                 int typing, int32_t* kout, int parallel) {
    int64_t omax = na - 1;
    #pragma omp parallel for schedule(static) if (parallel)
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < na; ++i) {
            int64_t q = i*n + j;
            double C = Wx[2*q], D = Wx[2*q+1];
            if (!(hypot(C, D) > gamma)) { if (kout) kout[q] = -1; continue; }
            double A = dWx[2*q], B = dWx[2*q+1];
            double r = (B*C - A*D) / ((sq64(C, typing) + sq64(D, typing)) * TWO_PI);
            double w = Sfs ? fabs(Sfs[i] - r) : fabs(r);
            int64_t k = bin_from_w_f64(w, grid, p, omax);
            if (flipud) k = omax - k;
            if (kout) kout[q] = (int32_t)k;
            out[2*(k*n + j)]     += C * cst[i];
            out[2*(k*n + j) + 1] += D * cst[i];
        }
}

void orc_ssq_f32(const float* Wx, const float* dWx, const float* Sfs,
                 float* out, const void* cst, int cst_f64, int64_t na, int64_t n,
                 double gamma, int grid, const double* p, int flipud,
                 int typing, int32_t* kout, int parallel) {
    int64_t omax = na - 1;
    const float*  c32 = (const float*)cst;
    const double* c64 = (const double*)cst;
    #pragma omp parallel for schedule(static) if (parallel)
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < na; ++i) {
            int64_t q = i*n + j;
            float C = Wx[2*q], D = Wx[2*q+1];
            /* abs(complex64) is float32; gamma is a Python float: float64 under
             * numba, weak (-> float32) under NumPy-2 scalar rules */
            float mag = hypotf(C, D);
            int keep = (typing == ORC_TYPING_NUMPY) ? (mag > (float)gamma)
                                                    : ((double)mag > gamma);
            if (!keep) { if (kout) kout[q] = -1; continue; }
            float A = dWx[2*q], B = dWx[2*q+1];
            float num = B*C - A*D, m2 = sq32(C, typing) + sq32(D, typing);
            int64_t k;
            if (typing == ORC_TYPING_NUMPY) {
                float r = num / (m2 * (float)TWO_PI);
                float w = Sfs ? fabsf(Sfs[i] - r) : fabsf(r);
                k = bin_from_w_f32np(w, grid, p, omax);
            } else {
                double r = (double)num / ((double)m2 * TWO_PI);
                double w = Sfs ? fabs((double)Sfs[i] - r) : fabs(r);
                k = bin_from_w_f64(w, grid, p, omax);
            }
            if (flipud) k = omax - k;
            if (kout) kout[q] = (int32_t)k;
            float* o = out + 2*(k*n + j);
            if (cst_f64) {
                o[0] = (float)((double)o[0] + (double)C * c64[i]);
                o[1] = (float)((double)o[1] + (double)D * c64[i]);
            } else {
                o[0] += C * c32[i];
                o[1] += D * c32[i];
            }
        }
}

/* =========================================================== indexed sum
 * Two-step form used with get_w=True: `w` is given (inf where below threshold).
 * numba: np.log2(float32) is float32, then `- vlmin` promotes to float64;
 * NumPy : float32 throughout.
 */
void orc_indexed_sum_f64(const double* Wx, const double* w, double* out,
                         const double* cst, int64_t na, int64_t n, int grid,
                         const double* p, int flipud, int parallel) {
    int64_t omax = na - 1;
    #pragma omp parallel for schedule(static) if (parallel)
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < na; ++i) {
            int64_t q = i*n + j;
            if (isinf(w[q])) continue;
            int64_t k = bin_from_w_f64(w[q], grid, p, omax);
            if (flipud) k = omax - k;
            out[2*(k*n + j)]     += Wx[2*q]   * cst[i];
            out[2*(k*n + j) + 1] += Wx[2*q+1] * cst[i];
        }
}

static inline int64_t bin_from_w_f32nb(float w, int grid, const double* p,
                                       int64_t omax) {
    /* numba typing of _indexed_sum_*: log2 in float32, the rest in float64 */
    if (grid == ORC_GRID_LIN)
        return clamp_round(((double)w - p[0]) / p[1], omax);
    double wl = (double)log2f(w);
    if (grid == ORC_GRID_LOG)
        return clamp_round((wl - p[0]) / p[1], omax);
    if (wl > p[1]) {
        double t = (wl - p[1]) / p[3];
        if (!(t < 4.0e18)) return omax;
        int64_t k = (int64_t)rint(t) + (int64_t)p[4];
        return k > omax ? omax : (k < 0 ? 0 : k);
    }
    return clamp_round((wl - p[0]) / p[2], omax);
}

void orc_indexed_sum_f32(const float* Wx, const float* w, float* out,
                         const void* cst, int cst_f64, int64_t na, int64_t n,
                         int grid, const double* p, int flipud, int typing,
                         int parallel) {
    int64_t omax = na - 1;
    const float*  c32 = (const float*)cst;
    const double* c64 = (const double*)cst;
    #pragma omp parallel for schedule(static) if (parallel)
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < na; ++i) {
            int64_t q = i*n + j;
            if (isinf(w[q])) continue;
            int64_t k = (typing == ORC_TYPING_NUMPY)
                        ? bin_from_w_f32np(w[q], grid, p, omax)
                        : bin_from_w_f32nb(w[q], grid, p, omax);
            if (flipud) k = omax - k;
            float* o = out + 2*(k*n + j);
            if (cst_f64) {
                o[0] = (float)((double)o[0] + (double)Wx[2*q]   * c64[i]);
                o[1] = (float)((double)o[1] + (double)Wx[2*q+1] * c64[i]);
            } else {
                o[0] += Wx[2*q]   * c32[i];
                o[1] += Wx[2*q+1] * c32[i];
            }
        }
}

/* ====================================================== replace_under_abs */
void orc_replace_under_abs_f64(double* w, const double* ref, int64_t cnt,
                               double value, double replacement) {
    for (int64_t q = 0; q < cnt; ++q)
        if (hypot(ref[2*q], ref[2*q+1]) < value) w[q] = replacement;
}

void orc_replace_under_abs_f32(float* w, const float* ref, int64_t cnt,
                               double value, float replacement) {
    for (int64_t q = 0; q < cnt; ++q)
        if ((double)hypotf(ref[2*q], ref[2*q+1]) < value) w[q] = replacement;
}

/* ================================================================ buffer
 * STFT framing, out is (seg_len, n_segs) C-contiguous here (the reference fills a
 * Fortran-ordered array; values per (row, col) are what matters).
 * modulated: frame rotated so its second half comes first
 * (utils/stft_utils.py:76-82: out[:s20] = x[start+s21 : start+s21+s20],
 *  out[s20:] = x[start : start+s21]).
 */
#define DEF_BUFFER(NAME, T)                                                        \
void NAME(const T* x, T* out, int64_t n_x, int64_t seg_len, int64_t n_overlap,    \
          int modulated) {                                                        \
    int64_t hop = seg_len - n_overlap;                                            \
    int64_t n_segs = (n_x - seg_len) / hop + 1;                                   \
    int64_t s20 = (seg_len + 1) / 2;                                              \
    int64_t s21 = (seg_len % 2 == 1) ? s20 - 1 : s20;                             \
    for (int64_t c = 0; c < n_segs; ++c) {                                        \
        int64_t start = hop * c;                                                  \
        for (int64_t r = 0; r < seg_len; ++r) {                                   \
            int64_t src;                                                          \
            if (!modulated)      src = start + r;                                 \
            else if (r < s20)    src = start + s21 + r;                           \
            else                 src = start + (r - s20);                         \
            out[r * n_segs + c] = x[src];                                         \
        }                                                                         \
    }                                                                             \
}
DEF_BUFFER(orc_buffer_f32, float)
DEF_BUFFER(orc_buffer_f64, double)

/* ======================================================== ridge tracking
 * ridge_extraction.py:163-232. Forward pass (`__accumulated_penalty_energy_fw`,
 * :163-176): for t >= 1, pe[f, t] += min_g(pe[g, t-1] + P[f, g]), every sum formed
 * in the array type (the elementwise sum of two rows, then amin, then `+=`).
 * Backward pass (`__accumulated_penalty_energy_bw`, :202-214): for t = n-2 .. 0 with
 * r = ridge[t+1]: val = pe[r, t+1] - e[r, t+1]; every f with
 * |val - (pe[f, t] + P[r, f])| < eps overwrites ridge[t] in ascending f (the last
 * one stays); none -> the forward argmin ridge[t] stays.
 * pe, e: (na, n) row-major; P: (na, na) row-major in the same type (a float32
 * penalty matrix promoted to double is exact). */
#define DEF_RIDGE(SFX, T, ABS)                                                     \
void orc_ridge_fw_##SFX(T* pe, const T* P, int64_t na, int64_t n) {               \
    T* prev = (T*)__builtin_malloc(sizeof(T) * (size_t)na);                       \
    T* cur = (T*)__builtin_malloc(sizeof(T) * (size_t)na);                        \
    for (int64_t f = 0; f < na; ++f) prev[f] = pe[f * n];                         \
    for (int64_t t = 1; t < n; ++t) {                                             \
        for (int64_t f = 0; f < na; ++f) {                                        \
            const T* Pf = P + f * na;                                             \
            T m = prev[0] + Pf[0];                                                \
            for (int64_t g = 1; g < na; ++g) {                                    \
                T v = prev[g] + Pf[g];                                            \
                if (v < m) m = v;                                                 \
            }                                                                     \
            cur[f] = pe[f * n + t] + m;                                           \
        }                                                                         \
        for (int64_t f = 0; f < na; ++f) pe[f * n + t] = cur[f];                  \
        T* sw = prev; prev = cur; cur = sw;                                       \
    }                                                                             \
    __builtin_free(prev); __builtin_free(cur);                                    \
}                                                                                 \
void orc_ridge_bw_##SFX(const T* e, const T* P, const T* pe, int64_t* ridge,      \
                        double eps, int64_t na, int64_t n) {                      \
    const T epsT = (T)eps;                                                        \
    for (int64_t t = n - 2; t >= 0; --t) {                                        \
        const int64_t r = ridge[t + 1];                                           \
        const T val = pe[r * n + t + 1] - e[r * n + t + 1];                       \
        for (int64_t f = 0; f < na; ++f) {                                        \
            const T c = pe[f * n + t] + P[r * na + f];                            \
            if (ABS(val - c) < epsT) ridge[t] = f;                                \
        }                                                                         \
    }                                                                             \
}
DEF_RIDGE(f32, float, fabsf)
DEF_RIDGE(f64, double, fabs)
