// ssq_tiles.h -- device tables and plan object of the column-tile path of the fused
// ssq_cwt form (kernels in ssq_cwt_tiles.hip, host planning in _tiles.py).
#pragma once
#include "ssq_common.h"
#include "ssq_fft.h"
#include <vector>

namespace ssq {

// a run of steps of one kind (and, for interpolated rows, one decimation class)
struct TileSeg {
    int32_t kind;        // 0 = rows read back (Wx + bin map in HBM), 1 = rows interpolated
    int32_t first;       // first step
    int32_t nsteps;
    int32_t lgR;         // log2 of the decimation R
    int32_t wtab_off;    // first phase of the class in the weight table
    int32_t sig_stride;  // complex entries between two signals' rows of the class
    int32_t lmask;       // L - 1 (row length of the class, a power of two)
    int32_t cls_base;    // complex entries before the class in the intermediate buffer
};
// one row of a step (4 per step; row < 0: none)
struct TileRow {
    int32_t row;
    int32_t ubase;       // offset of the row's samples inside its class (signal 0)
    int32_t kc;          // centre bin of the band on the M-point grid
    float   theta;       // 2 pi kc / (M dt) (informative: the kernel forms it as kc * 2 pi / (M dt))
};
// one interpolated row as the spectra kernel sees it
struct TileIRow {
    int32_t row, lo, K, kc, L, tb_off;
    int32_t ubase;       // complex entries: class offset + index in class * L (signal 0)
    int32_t sig_stride;
};

// The analytic signal of the padded batch (block classes with analytic = 1, ssq_blocks.h) through the
// four-step kernels of ssq_cwt_tiles.hip: one-sided spectrum formed on the fly from the half spectrum
// (Nyquist bin halved, 1 / M folded in) -> length-M inverse transform. float32, M = 2^13 .. 2^22;
// replaces a spectrum kernel + rocFFT's two or three kernels.
struct AnalyticFft {
    int64_t M = 0, max_batch = 0;
    int A = 0, B = 0;
    void* Y = nullptr; void* ftw = nullptr; void* tb = nullptr; TileIRow* irow = nullptr;
    int64_t off_a = 0, off_b = 0;           // twiddle tables e^{2 pi i q / A}, e^{2 pi i q / B} inside ftw
    static bool supports(int dtype, int64_t M);
    int create(int64_t M, int64_t max_batch, int64_t& bytes);
    void destroy();
    int run(const void* xh_all, void* xa, int64_t batch, hipStream_t stream);
};

// ---- geometry both tile kernels and the plan agree on
constexpr int TILE_COLS = 64;           // ordered kernel: columns per tile (one per lane)
#ifndef SSQ_TILE_G
#define SSQ_TILE_G 4
#endif
constexpr int TILE_G = SSQ_TILE_G;      // rows per step of the host's tables (_tiles.py: RSUB)
constexpr int TILE2_NW = 16;            // tile2_kernel: wavefronts per workgroup (one workgroup per CU)
constexpr int TILE3_NW = 16;            // tile3_kernel: the same
// LDS of a workgroup: the ordered kernel's (na + 1) x 64 float32 pairs + ticket words; the default kernel's
// (na + 1) x cols float64 pairs
__host__ __device__ inline size_t tile_lds_bytes(int64_t na) { return (size_t)(na + 1) * TILE_COLS * 8 + 16; }
__host__ __device__ inline size_t tile2_lds_bytes(int64_t na, int cols) { return (size_t)(na + 1) * cols * 16; }
bool tile_ordered();                    // SSQ_TILE_ORDER=ordered in the environment (read at every call)

int tile_rows_per_step();               // TILE_G

struct TilePlan {
    int64_t M = 0, N = 0, n1 = 0, na = 0;
    int group = 1;
    double dt = 1.0;
    int nsegs = 0, nsteps = 0, n_irows = 0;
    int64_t u_total = 0, lmax = 0;
    int ncu = 0;                            // persistent workgroups of the tile kernel (one per CU)
    TileSeg* steps = nullptr;               // the segment record of every step
    TileRow* rows = nullptr; TileIRow* irows = nullptr;
    void* wtab = nullptr; void* tbank = nullptr;
    void* U = nullptr;                      // group x u_total complex64
    unsigned long long* counters = nullptr; // [0]: tiles the kernel has finished since plan creation
    // tile2_kernel (float64 tile, unordered adds): packed item records, [steps * 4] int4; the
    // wavefronts' blocks of items ([TILE2_NW][4]); columns per tile
    void* items2 = nullptr; int32_t* wave_first2 = nullptr;
    int n_items2 = 0, cols2 = 32;
    int lgr_max2 = 0;                       // largest decimation (log2) among the interpolated classes
    bool tile2_ok = false;                  // the items could be cut into blocks of at most two classes
    // tile3_kernel (two columns per lane, ssq_tile_pair.hip): items of FOUR rows x 32 columns, permuted so that every
    // wavefront's list is contiguous, and the lists' bounds ([TILE3_NW][4])
    void* items3 = nullptr; int32_t* wave_first3 = nullptr;
    int n_items3 = 0;
    bool tile3_ok = false;
    bool pair_ok() const;                   // tile3_kernel takes this plan (32-column tile, SSQ_DEBUG_TILE_PAIR != 0)
    int tile_kernel() const;                // 0 none, 1 ordered, 2 tile2_kernel, 3 tile3_kernel (what `run` launches now)
    int tile_cols() const;                  // columns per tile of the kernel that `run` launches (0: none can run)
    // Can `run` launch a tile kernel for this plan in the mode selected right now? The default kernel needs the
    // items cut into blocks of at most two classes (tile2_ok), the ordered one (SSQ_TILE_ORDER=ordered, also the
    // stand-in when !tile2_ok) a 64-column float32 tile inside the LDS (na <= 318). Asked by the executor BEFORE it
    // routes any row: a plan that is not usable takes the block kernels + the separate reassignment for every row.
    bool usable() const;
    // A, B > 0: a long class transformed by the four-step kernels of ssq_cwt_tiles.hip (L = A B);
    // A = 0, B = 1: a short class (64 .. 4096 entries) transformed by the one-pass kernel;
    // A = B = 0: rocFFT. first: its rows in `irows` (sorted by class, the classes of our own
    // kernels first)
    struct Cls { int64_t L, nrows, upre; int A, B, first; };
    std::vector<Cls> cls;
    std::vector<FftPlan> ffts;              // one batched inverse per class (unused for four-step classes)
    int n_irows_fft = 0, first_irow_fft = 0;   // rows of the rocFFT classes (the spectra kernel's)
    void* Y = nullptr;                      // four-step intermediates, class after class (group x rows x L each)
    void* ftw = nullptr;                    // e^{2 pi i q / L'}, L' = 64 .. 4096, concatenated
    int64_t ftw_off[7] = {0, 0, 0, 0, 0, 0, 0};
    int64_t n_items_tile[5] = {0, 0, 0, 0, 0};
    std::vector<unsigned char> class_need;   // block classes the remaining block rows use
    // the intermediates (a spectra kernel + small FFTs) run on a side stream, beside the block /
    // exact kernels of the rows that are read back
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    int create(const ssq_cwt_tiles_desc& d, int64_t M, int64_t N, int64_t n1, int64_t na, int group,
               double dt, int64_t& bytes);
    void destroy();
    // intermediates of signals sig .. sig+nsig-1 (nsig <= group) from the spectra of the batch
    int spectra(int sig, int nsig, const void* xh_all, hipStream_t stream);
    // Wx of the interpolated rows, Tx of all rows (the other rows' Wx and bin map must be in place)
    // (kdump: diagnostic -- the bin of every point as the kernel consumed it, (batch signal, row, column), or null)
    int run(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
            const void* cst, float cst0, const SsqParams& sp, hipStream_t stream, unsigned short* kdump = nullptr);
    // ... by the default kernel (ssq_tile_f64.hip) / by the ordered one (ssq_tile_ordered.hip)
    int run_f64(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                const void* cst, float cst0, const SsqParams& sp, hipStream_t stream, unsigned short* kdump);
    int run_ordered(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                    const void* cst, float cst0, const SsqParams& sp, hipStream_t stream);
    int run_pair(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                 const void* cst, float cst0, const SsqParams& sp, hipStream_t stream, unsigned short* kdump);
    // tiles finished by the tile kernel so far (synchronises `stream`): what actually ran
    int64_t tiles_done(hipStream_t stream);
};

}  // namespace ssq
