/*
 * ssq_hip.h -- C ABI of libssq_hip.so, the MI355X (gfx950) engine behind
 * ssqueezepy_amd's cwt() / stft() / ssq_cwt() / ssq_stft().
 *
 * The reference (OverLordGoldDragon/ssqueezepy v0.6.6) has no FFI: its GPU seam is
 * Python-level -- CuPy RawModule kernels launched with raw device pointers on
 * torch's current stream (ssqueezepy/utils/gpu_utils.py:10-21, algos.py:100-105).
 * This header is that seam restated as a C ABI: plain pointers and sizes, no torch
 * or HIP types in the signatures (`stream` is a hipStream_t passed as void*, NULL =
 * the default stream), every entry point asynchronous on its stream, int status
 * returns (0 = ok, <0 = error, text via ssq_last_error()), no exceptions across the
 * boundary. Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   dtype      SSQ_F32 / SSQ_F64: the *real* type; complex arrays are interleaved
 *              (re, im) pairs of it (what torch.view_as_real gives, algos.py:61-64).
 *   layouts    2-D arrays are row-major (rows = scales / frequency bins,
 *              cols = time); batched arrays put the signal index first.
 *   ownership  the caller owns every I/O buffer (device memory); plans own their
 *              FFT plans, workspace and the device copy of the filter bank.
 *   threading  one plan per host thread / stream at a time; plans are immutable
 *              after creation except through the documented setters.
 */
#ifndef SSQ_HIP_H
#define SSQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSQ_F32 0
#define SSQ_F64 1

#define SSQ_GRID_LOG 0            /* params: vlmin, dvl                         */
#define SSQ_GRID_LOG_PIECEWISE 1  /* params: vlmin0, vlmin1, dvl0, dvl1, idx1   */
#define SSQ_GRID_LIN 2            /* params: vmin, dv                           */

#define SSQ_PAD_NONE (-1)
#define SSQ_PAD_ZERO 0
#define SSQ_PAD_REFLECT 1
#define SSQ_PAD_SYMMETRIC 2
#define SSQ_PAD_REPLICATE 3
#define SSQ_PAD_WRAP 4

/* ------------------------------------------------------------------ runtime */
int         ssq_version(void);          /* 105 (104: without ssq_cwt_plan_tile_kernel; 103: without ssq_build_sha / ssq_cwt_plan_set_bin_dump; 102: without ssq_ridge_*_batch; 101: without ssq_cwt_plan_tile_cols; 100: block classes without the `analytic` column) */
/* The git commit of the device code this library was built from: the last commit that touched
 * ssqueezepy_amd/csrc or include/ ("<sha>-dirty" when the build tree had uncommitted changes there,
 * "unknown" when built outside a git checkout). Measurement records carry it (bench.py, profiles/):
 * evidence and library can be matched without trusting a file name. */
const char* ssq_build_sha(void);
const char* ssq_last_error(void);
int         ssq_device_count(int* count);
int         ssq_set_device(int device);
/* name[0..len) <- gcnArchName of `device`; *cus <- compute-unit count */
int         ssq_device_info(int device, char* name, int len, int* cus,
                            int64_t* hbm_bytes);

/* Raw device-memory helpers for hosts that do not bring their own allocator
 * (the Python layer normally passes torch-owned device pointers instead). */
int ssq_malloc(void** ptr, int64_t bytes);
int ssq_free(void* ptr);
int ssq_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream);
int ssq_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream);
int ssq_memset(void* dst, int value, int64_t bytes, void* stream);
int ssq_stream_synchronize(void* stream);

/* ---------------------------------------------------- kernel-level seam
 * These are the reference's L2 kernels. All arrays are
 * (batch, na, n) device arrays; `batch` >= 1.
 */

/* w = |Wx| < gamma ? inf : |Im(dWx / Wx)| / 2pi
 * replaces phase_cwt_cpu / phase_cwt_gpu (algos.py:706-781). */
int ssq_phase_cwt(int dtype, const void* Wx, const void* dWx, void* w,
                  int64_t batch, int64_t na, int64_t n, double gamma,
                  void* stream);

/* w = |Sx| < gamma ? inf : |Sfs[i] - Im(dSx / Sx) / 2pi|
 * replaces phase_stft_cpu / phase_stft_gpu (algos.py:784-856). Sfs: (na,) real. */
int ssq_phase_stft(int dtype, const void* Sx, const void* dSx, const void* Sfs,
                   void* w, int64_t batch, int64_t na, int64_t n, double gamma,
                   void* stream);

/* Fused phase transform + bin search + accumulate:
 *   for every (i, j) with |Wx[i,j]| > gamma:  Tx[k(i,j), j] += Wx[i,j] * cst[i]
 * with k from `grid`/`params` (and mirrored, k -> na-1-k, if `flipud`).
 * `Sfs` NULL selects the CWT form of w, non-NULL the STFT form.
 * `cst` is a device vector of na weights: real `dtype`, or float64 when
 * `cst_f64` != 0 (complex64 data weighted by a float64 vector accumulates in
 * double, as the reference does for 'log-piecewise' scales).
 * Tx is OVERWRITTEN (zero-filled and accumulated by the kernel).
 * `kmap` (optional, int32 (batch, na, n)) receives the bin of every point, -1 where
 * the point is below threshold.
 * replaces ssqueeze_fast -> _ssq_cwt_{log,log_piecewise,lin}[_par], _ssq_stft[_par]
 * and the CUDA strings ssq_cwt_* / ssq_stft (algos.py:126-150, 859-984, 1008-1167).
 * Summation order per output cell is ascending i, as in the reference.
 */
int ssq_ssqueeze(int dtype, const void* Wx, const void* dWx, const void* Sfs,
                 void* Tx, const void* cst, int cst_f64, int64_t batch,
                 int64_t na, int64_t n, double gamma, int grid,
                 const double* params, int flipud, int32_t* kmap, void* stream);

/* Same accumulate, bins taken from a precomputed phase transform `w` (inf = skip).
 * replaces indexed_sum_onfly -> _indexed_sum_{log,log_piecewise,lin}[_par] and the
 * CUDA strings indexed_sum_* (algos.py:153-250, 1169-1268). */
int ssq_indexed_sum(int dtype, const void* Wx, const void* w, void* Tx,
                    const void* cst, int cst_f64, int64_t batch, int64_t na,
                    int64_t n, int grid, const double* params, int flipud,
                    void* stream);

/* w[q] = replacement where |ref[q]| < value.
 * replaces replace_under_abs (algos.py:498-579). */
int ssq_replace_under_abs(int dtype, void* w, const void* ref, int64_t count,
                          double value, double replacement, void* stream);

/* STFT framing: out (batch, seg_len, n_segs) <- x (batch, n_x);
 * n_segs = (n_x - seg_len) / (seg_len - n_overlap) + 1; `modulated` rotates every
 * frame by ceil(seg_len/2). replaces buffer / _buffer_gpu
 * (utils/stft_utils.py:20-138). */
int ssq_buffer(int dtype, const void* x, void* out, int64_t batch, int64_t n_x,
               int64_t seg_len, int64_t n_overlap, int modulated, void* stream);

/* Signal extension: out (batch, n1 + n + n2) <- x (batch, n).
 * replaces padsignal (utils/common.py:54-158). */
int ssq_pad_signal(int dtype, const void* x, void* out, int64_t batch, int64_t n,
                   int64_t n1, int64_t n2, int padtype, void* stream);

/* ------------------------------------------------------------------ inverses
 * Column reductions of arrays that already live on the device; sums run in the
 * reference's order (ascending row, the array's precision), results are bit-identical
 * to the NumPy expressions they replace.
 *
 * out (batch, n) <- sum_i Re(Z[b, i, :]) [/ divisor[i]].   replaces
 *   `(Wx.real / norm(scales)).sum(axis=-2)` of _icwt_1int (_cwt.py:472-476) and
 *   `Tx.real.sum(axis=0)` of issq_cwt / issq_stft (_ssq_cwt.py:369-371, _ssq_stft.py:191).
 * `divisor`: real (na,) in the data dtype, or NULL. */
int ssq_colsum(int dtype, const void* Z, const void* divisor, void* out, int64_t batch,
               int64_t na, int64_t n, void* stream);

/* Component inversion around curves: out (ncomp + 1, n) float64; row k < ncomp sums
 * Re(Z[lo[k][j] .. hi[k][j], j]) in double, row ncomp sums the rows no band covers in
 * the data dtype. lo/hi: int32 (ncomp, n), inclusive, empty when lo > hi.
 * replaces _invert_components (_ssq_cwt.py:381-403). */
int ssq_band_colsum(int dtype, const void* Z, const int32_t* lo, const int32_t* hi,
                    int64_t ncomp, double* out, int64_t na, int64_t n, void* stream);

/* Double-integral inverse CWT core: out (n_up) real <- Re ifft( sum_a fft(Wp[a]) * psih[a] ).
 * `Wp` (na, n_up) complex: the padded transform, overwritten (used as FFT workspace);
 * `psih` (na, n_up) real: wavelet samples already divided by the scale normalisation.
 * replaces the loop of _icwt_2int (_cwt.py:448-469; its (-1)^k factor and ifftshift cancel
 * for the even padded lengths it uses, and the sum over scales commutes with the iFFT). */
int ssq_icwt2(int dtype, void* Wp, const void* psih, void* out, int64_t na, int64_t n_up,
              void* stream);

/* Frequency-domain differentiation of the rows of a CWT-like array (trigdiff,
 * utils/common.py:161-245): out (rows, N) complex <- ifft(fft(Ap) * 1j * xi * fs)[:, n1 : n1 + N].
 * `Ap` (rows, n_up) complex: the (padded) rows, overwritten (FFT workspace); `xi` (n_up) real:
 * the frequency grid `_xifn(1, n_up)` in the data dtype. */
int ssq_trigdiff(int dtype, void* Ap, const void* xi, double fs, void* out, int64_t rows,
                 int64_t n_up, int64_t n1, int64_t N, void* stream);

/* Inverse STFT: x (N) <- Sx (n_fft/2 + 1, n_hops) complex. irfft of every column
 * (rocFFT), fftshift of the frame if `modulated`, overlap-add with win_a = window^a,
 * division by the overlap-added win_a1 = window^(a+1), trim of n_fft/2 leading samples.
 * replaces the body of istft (_stft.py:238-256: irfft, unbuffer, window_norm). */
int ssq_istft(int dtype, const void* Sx, const void* win_a, const void* win_a1, void* x,
              int64_t n_fft, int64_t n_hops, int64_t hop_len, int64_t N, int modulated,
              void* stream);

/* ------------------------------------------------------------ ridge extraction
 * The loop nests of extract_ridges (ridge_extraction.py:113-232) on arrays that are
 * already on the device; (na, n) arrays are row-major, real, in `dtype`.
 *
 * energy <- |Tf|^2, `np.abs(Tf)**2` (:129). Tf: (na, n) complex (interleaved) when
 * `is_complex`, else real. */
int ssq_ridge_energy(int dtype, int is_complex, const void* Tf, void* energy, int64_t na,
                     int64_t n, void* stream);

/* E <- -log(energy / max_i energy[i, j] + eps)   (:138-139). */
int ssq_ridge_neglog(int dtype, const void* energy, void* E, double eps, int64_t na, int64_t n,
                     void* stream);

/* Forward-backward tracking (fw_bw_ridge_tracking, :91-111 -> :143-232):
 *   pe[f, t] = E[f, t] + min_g(pe[g, t-1] + P[f, g]),  P[f, g] = penalty * (sc[f] - sc[g])^2
 *   ridge[t] = argmin_f pe[f, t], then for t = n-2 .. 0 the last f with
 *   |pe[r, t+1] - E[r, t+1] - (pe[f, t] + P[r, f])| < eps, r = ridge[t+1], if any.
 * `sc` (na,): the (log) scales in the penalty dtype -- float32 when `penalty_f32`
 * (every input but complex128, :113-117), else float64; `pe` (na, n) and `ridge` (n,)
 * int64 are outputs. Bit-identical to the reference's loops for identical E. */
int ssq_ridge_track(int dtype, int penalty_f32, const void* E, void* pe, const void* sc,
                    double penalty, double eps, int64_t na, int64_t n, int64_t* ridge,
                    void* stream);

/* ridge_e[j] <- energy[ridge[j], j] (NULL: skipped), then
 * energy[int(ridge[j] - bw) : int(ridge[j] + bw), j] = 0 with Python's slice rules (:142-150). */
int ssq_ridge_clear(int dtype, void* energy, const int64_t* ridge, double bw, void* ridge_e,
                    int64_t na, int64_t n, void* stream);

/* The three steps above over `batch` transforms at once (ABI 103): contiguous (batch, na, n) arrays, `ridge`
 * and `ridge_e` (batch, n); one workgroup per transform in the tracking passes -- a pass is one workgroup's
 * walk over time, so a batch costs about what one transform does until the CUs run out. No counterpart in the
 * reference (its extract_ridges takes one transform); results per transform are those of the calls above. */
int ssq_ridge_neglog_batch(int dtype, const void* energy, void* E, double eps, int64_t na, int64_t n,
                           int64_t batch, void* stream);
int ssq_ridge_track_batch(int dtype, int penalty_f32, const void* E, void* pe, const void* sc,
                          double penalty, double eps, int64_t na, int64_t n, int64_t* ridge,
                          int64_t batch, void* stream);
int ssq_ridge_clear_batch(int dtype, void* energy, const int64_t* ridge, double bw, void* ridge_e,
                          int64_t na, int64_t n, int64_t batch, void* stream);

/* ----------------------------------------------------------------- CWT plan
 * Replaces the body of cwt() (_cwt.py:255-306: pad -> fft -> Psih*xh -> ifft
 * [-> *1j*xi/dt -> ifft] -> unpad) and, when ssq parameters are set, the
 * ssq_cwt() tail (_ssq_cwt.py:266-289 -> ssqueezing.py:122-146).
 *
 * The filter bank is passed in *banded* form: row i is non-negligible only on DFT
 * bins [band_lo[i], band_lo[i] + band_len[i]) of the M-point grid (bins above M/2
 * are negative frequencies); its values are bank[band_off[i] .. band_off[i+1]).
 * The Nyquist bin must already be halved (wavelets.py:86-95).
 */
typedef struct ssq_cwt_plan ssq_cwt_plan;

typedef struct {
    int      dtype;        /* SSQ_F32 | SSQ_F64 (the wavelet's dtype)             */
    int      padtype;      /* SSQ_PAD_*                                           */
    int64_t  n;            /* signal length N                                     */
    int64_t  m;            /* padded length M (== n when padtype NONE)            */
    int64_t  n1;           /* left pad                                            */
    int64_t  na;           /* number of scales                                    */
    const void*    bank;       /* host, real dtype, concatenated band values     */
    const int64_t* band_off;   /* host, na + 1                                    */
    const int32_t* band_lo;    /* host, na                                        */
    double   dt;           /* sampling period (derivative multiplier 1j*xi/dt)    */
    const void*    row_scale;  /* host, na reals or NULL: per-row output scaling  */
                               /* (sqrt(scale) when l1_norm=False, _cwt.py:307)   */
    int64_t  max_batch;    /* largest batch the plan will be executed with        */
    int      algo;         /* 0 = auto, 1 = force generic (rocFFT) path           */
} ssq_cwt_desc;

int  ssq_cwt_plan_create(ssq_cwt_plan** plan, const ssq_cwt_desc* desc);
void ssq_cwt_plan_destroy(ssq_cwt_plan* plan);

/* Synchrosqueezing parameters for subsequent executes (host pointers, copied). */
int  ssq_cwt_plan_set_ssq(ssq_cwt_plan* plan, int grid, const double* params,
                          const void* cst, int cst_f64, int flipud, double gamma);

/* x: (batch, n) real `dtype` on the device. Outputs (any may be NULL):
 *   Wx, dWx, Tx : (batch, na, n) complex;   w : (batch, na, n) real.
 * `rpadded` != 0 returns Wx/dWx of padded width m instead (Tx/w must be NULL).
 * Tx requires ssq_cwt_plan_set_ssq(); bins come from dWx (fused form) unless `w` is
 * requested, in which case they come from the rounded `w` (two-step form,
 * get_w=True in the reference). */
int  ssq_cwt_execute(ssq_cwt_plan* plan, const void* x, int64_t batch, void* Wx,
                     void* dWx, void* Tx, void* w, int rpadded, void* stream);

/* Optional fast path ("overlap-save zoom" iFFT, float32 or float64, power-of-two m,
 * analytic bank): tables planned on the host (ssqueezepy_amd/_blocks.py documents the
 * decomposition; all pointers are host arrays, copied). Rows with class -1 in `rows`
 * -- listed in `generic_rows` -- keep using the exact full-length path.
 * Must be called before the first execute. No reference counterpart: the reference
 * evaluates every row as one length-m inverse FFT (_cwt.py:167-177); this is the
 * same filter bank applied block-wise. */
typedef struct {
    int            n_classes;
    const int64_t* classes;      /* n_classes x 5: P, margin, valid, blocks/signal, analytic   */
                                 /* (1: blocks of the analytic signal, for rows continued past */
                                 /* the Nyquist bin; such classes come last)                   */
    const int32_t* rows;         /* na x 6: class, kappa_lo, K_P, L', G, pbank_off  */
    const void*    pbank;        /* P-grid band values of the block rows (plan dtype) */
    const void*    pxi;          /* xi at the same bins (same indexing, plan dtype)   */
    int64_t        n_pbank;
    const void*    ctw;          /* complex column twiddles exp(2i pi q/P) (plan dtype) */
    const int64_t* ctw_off;      /* n_classes + 1                                   */
    const void*    ftw;          /* complex FFT twiddles exp(2i pi q/L') (plan dtype) */
    int64_t        n_ftw;
    int64_t        ftw_off[5];   /* per L' = 128, 256, 512, 1024, 2048              */
    const int32_t* items[5];     /* per L': n_items x 4: row, block, c0, class      */
    int64_t        n_items[5];
    const int32_t* generic_rows; /* rows left on the exact path                     */
    int64_t        n_generic;
} ssq_cwt_blocks_desc;

int  ssq_cwt_plan_set_blocks(ssq_cwt_plan* plan, const ssq_cwt_blocks_desc* desc);

/* Optional column-tile path of the fused ssq_cwt form (float32, power-of-two m, block
 * tables set, Tx requested without w): rows that are at least 2x oversampled after a
 * decimation by R >= 4 are not transformed at full length at all. Their band is
 * inverse-transformed at length m / R into a plan-owned intermediate, and one kernel per
 * 64-column tile interpolates Wx / dWx from it (8-tap Kaiser-Bessel kernel and its
 * derivative, modulation by hardware sin / cos of an exact phase), writes Wx, and reassigns
 * into an LDS-resident tile of Tx in ascending row order -- Wx is never read back and no bin map is written for those rows. The other
 * rows keep the block / exact kernels; the tile kernel reads their Wx and bin map back.
 * ssqueezepy_amd/_tiles.py documents the decomposition and builds the tables (host
 * arrays, copied). Must follow ssq_cwt_plan_set_blocks, before the first execute.
 * Replaces, for those rows, _cwt.py:167-177 (ifft of Psih*xh, and of its 1j*xi multiple)
 * together with ssqueezing.py:122-146 -> algos.py:859-953 (the reassignment loop). */
typedef struct {
    int32_t        n_segs;
    const int32_t* segs;         /* n_segs x 8: kind (0 read back, 1 interpolate), first step, */
                                 /* steps, log2 R, weight-table offset (phases), intermediate  */
                                 /* stride per signal, L - 1, class offset (complex entries)   */
    int32_t        n_steps;
    const int32_t* rows;         /* 4 n_steps x 4: row (sign bit: repeats the previous row to  */
                                 /* pad a step), offset in class, kc, theta = 2 pi kc / (m dt) */
                                 /* (float bits). The rows of a step are consecutive.           */
    const void*    wtab;         /* float32 [n_phases][16]: per tap, phi and phi' / (R dt)     */
    int64_t        n_phases;
    const void*    tbank;        /* float32: band values / (phi_hat m) of the interpolated rows */
    int64_t        n_tbank;
    int32_t        n_irows;
    const int64_t* irows;        /* n_irows x 8: row, lo, K, kc, L, tbank offset, class, index in class */
    int32_t        n_classes;
    const int64_t* classes;      /* n_classes x 4: L, rows, entries per signal before it, log2 R */
    int64_t        u_total;      /* complex entries of the intermediate per signal             */
    int64_t        n_items_tile[5]; /* per L': leading block items that remain (rows read back) */
    int32_t        reserved;     /* rows per step of `rows` (0 = 4): must equal ssq_cwt_tile_rows_per_step() */
} ssq_cwt_tiles_desc;

int  ssq_cwt_plan_set_tiles(ssq_cwt_plan* plan, const ssq_cwt_tiles_desc* desc);

/* Per-stage timing with HIP events on the execute stream (measurement aid, off by
 * default; execute synchronises when it is on). `stage_ms[4]` receives the time
 * accumulated since the last reset: 0 = pad + forward FFT + block spectra, 1 = block
 * rows, 2 = exact / generic rows, 3 = reassignment; `*signals` the transforms covered.
 * `enable` = 1 / 0 switches timing on / off and resets the accumulators, -1 only reads. */
int  ssq_cwt_plan_timing(ssq_cwt_plan* plan, int enable, double* stage_ms, int64_t* signals);

/* signals a plan processes per kernel launch (its launch group; the batch is walked in
 * groups of this size) */
int  ssq_cwt_plan_group(const ssq_cwt_plan* plan);

/* What executed: the number of column tiles the column-tile kernel has finished on this plan
 * since its creation (0 without tile tables). Synchronises `stream`. An execute that took the
 * tile path adds batch * ceil(n / ssq_cwt_plan_tile_cols(plan)) (tile kernel 3 with an odd left padding: ceil((n + 1) /
 * 32) -- its tiles start one column early); one that took the block path +
 * separate reassignment adds nothing. (Tests assert on this rather than on the plan's `algo`
 * label.) */
int64_t ssq_cwt_plan_tiles_done(ssq_cwt_plan* plan, void* stream);
/* Columns per tile of the kernel the next execute launches: 32 or 16 (float64 tile in LDS,
 * unordered ds_add_f64 accumulation -- the default), 64 with SSQ_TILE_ORDER=ordered in the
 * environment (float32 tile, terms added in the reference's row order; na <= 318). 0 without
 * tile tables. */
int  ssq_cwt_plan_tile_cols(const ssq_cwt_plan* plan);
/* Which column-tile kernel the next execute launches (ABI 105): 0 none, 1 the ordered float32 tile
 * (SSQ_TILE_ORDER=ordered), 2 the float64 tile with one column per lane (tile2_kernel), 3 the float64 tile with a
 * column pair per lane (tile3_kernel: the default whenever the tile holds 32 columns, i.e. up to 318 rows;
 * SSQ_DEBUG_TILE_PAIR=0 switches it off). Same results in 2 and 3. */
int  ssq_cwt_plan_tile_kernel(const ssq_cwt_plan* plan);
/* Diagnostic (ABI 105): the first `n` (<= 512) 64-bit words of the tile path's counter block -- word 0 = tiles done (as
 * above); words 64.. = per-wavefront shader-clock sums per phase of workgroup 0, filled by profiling builds of the tile
 * kernel only (-DSSQ_T3_PROF=1, tools/r6), zero otherwise. Synchronises `stream`. */
int  ssq_cwt_plan_tile_counters(ssq_cwt_plan* plan, unsigned long long* out, int n, void* stream);

/* Diagnostic (tests; not needed by a caller of the transforms): from now on every fused execute
 * (Tx requested, bins from dWx) also writes the bin index of every point AS THE REASSIGNMENT CONSUMED
 * IT -- 0 .. na-1, 0xFFFF for a point below gamma -- to `kmap`, a caller-owned device array
 * (batch, na, n) of uint16; NULL switches it off. On the column-tile path the default tile kernel
 * (a separate diagnostic build of it: uniform reassignment weights, i.e. 'log' scales, 16 wavefronts)
 * stores the bins it computes for the rows it interpolates and the ones it reads back for the others;
 * on the block path + separate reassignment the plan's own bin map is copied. With
 * SSQ_TILE_ORDER=ordered on the tile path the execute fails (that kernel's Tx is the CPU loop's bit
 * for bit, which pins its bins). Replaces the reference's `ssqueeze_fast(..., get_k)`-style
 * introspection: /root/reference has none (algos.py:859-984 computes k inline), the oracle's
 * `get_k` is the other side of the comparison. */
int  ssq_cwt_plan_set_bin_dump(ssq_cwt_plan* plan, unsigned short* kmap);

/* Rows per step the tile kernel of this build walks (4; the host tables of
 * ssq_cwt_plan_set_tiles must be built for the same number: `rows` holds that many records per
 * step). */
int  ssq_cwt_tile_rows_per_step(void);

/* bytes of device memory held by the plan (bank + workspace) */
int64_t ssq_cwt_plan_bytes(const ssq_cwt_plan* plan);
/* name of the compute path the plan selected ("rocfft", "zoom+rocfft", ...) */
const char* ssq_cwt_plan_algo(const ssq_cwt_plan* plan);

/* ---------------------------------------------------------------- STFT plan
 * Replaces the body of stft() (_stft.py:127-147,166-170: pad -> buffer -> *window
 * -> rfft along the frame axis) and, with ssq parameters, the ssq_stft() tail
 * (_ssq_stft.py:102-122).
 */
typedef struct ssq_stft_plan ssq_stft_plan;

typedef struct {
    int      dtype;
    int      padtype;
    int64_t  n;            /* signal length                                       */
    int64_t  n_fft;        /* frame length                                        */
    int64_t  hop_len;
    int      modulated;
    const void* window;       /* host, n_fft reals (already ifftshift-ed if        */
    const void* diff_window;  /* modulated; diff_window already times fs), or NULL */
    int64_t  max_batch;
} ssq_stft_desc;

int  ssq_stft_plan_create(ssq_stft_plan** plan, const ssq_stft_desc* desc);
void ssq_stft_plan_destroy(ssq_stft_plan* plan);
int  ssq_stft_plan_set_ssq(ssq_stft_plan* plan, const void* Sfs, int grid,
                           const double* params, const void* cst, int cst_f64,
                           int flipud, double gamma);
/* rows = n_fft/2 + 1, n_hops = (n - 1) / hop_len + 1 -> query */
int  ssq_stft_plan_shape(const ssq_stft_plan* plan, int64_t* rows, int64_t* n_hops);
/* name of the route the plan takes for the framing + window + FFT stage: "fused" (float32, n_fft a power of
 * two in [128, 2048]: one launch, LDS transform), "fused-mixed-radix" (float32, any other n_fft whose prime
 * factors are <= 31 and that fits the LDS: one launch, mixed-radix LDS transform -- the reference's benchmark
 * shape n_fft = 598, examples/benchmarks.py:78-79), "rocfft" (everything else, float64: framing kernel + batched
 * rocFFT with strided output). Tests assert on it. */
const char* ssq_stft_plan_algo(const ssq_stft_plan* plan);
/* x: (batch, n). Sx, dSx, Tx: (batch, rows, n_hops) complex; w: real. */
int  ssq_stft_execute(ssq_stft_plan* plan, const void* x, int64_t batch, void* Sx,
                      void* dSx, void* Tx, void* w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSQ_HIP_H */
