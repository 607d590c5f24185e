// ssq_blocks.h -- device tables and plan object of the block ("overlap-save zoom")
// CWT fast path (kernels in ssq_cwt_blocks.hip, host planning in _blocks.py).
#pragma once
#include "ssq_common.h"
#include "ssq_fft.h"
#include "ssq_tiles.h"
#include <algorithm>
#include <vector>

namespace ssq {

struct BlockRowDev {        // one per scale row; mirrors _blocks.py `rows` (na, 6) int32
    int32_t cls;            // block class, -1 = row stays on the exact path
    int32_t klo;            // first P-grid bin of the band
    int32_t KP;             // number of P-grid bins in the band
    int32_t L;              // zoom FFT length L'
    int32_t G;              // columns per workgroup (L' * G == 4096)
    int32_t pb_off;         // offset of the row's P-grid band values in `pbank`
};

struct BlockClassDev {
    int64_t P, m, V, nb;    // block length, margin, valid length, blocks per signal
    int64_t ctw_off;        // offset of the class' column twiddles
    int64_t xb_off;         // offset of the class' block spectra
    int64_t blk_off;        // offset of the class' gathered blocks (real classes)
    int64_t xb_stride;      // bins per block spectrum: P / 2 + 1 (real blocks) or P (analytic)
    int64_t analytic;       // 1: blocks of the analytic signal (rows continued past Nyquist, _blocks.py)
};

struct BlockPlan {
    int64_t M = 0, N = 0, n1 = 0, na = 0, max_batch = 1;
    int dtype = SSQ_F32;             // tables, spectra and kernels in this precision
    int group = 1;                   // signals per launch (kernels take them as a grid dimension)
    int ncu = 256;                   // CUs of the plan's device (BlockPlan::create asks the runtime)
    int nc = 0;
    std::vector<BlockClassDev> hcls;
    BlockClassDev* classes = nullptr;
    BlockRowDev* rows = nullptr;
    void* pbank = nullptr;           // real (dtype)
    void* pxi = nullptr;             // real (dtype)
    void* ctw = nullptr; void* ftw = nullptr;
    void* items[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<int32_t> h_items[5];   // host copy (row, block, c0, class)
    int64_t n_items[5] = {0, 0, 0, 0, 0};
    int64_t ftw_off[5] = {0, 0, 0, 0, 0};
    void* blocks = nullptr;          // gathered signal blocks of one class (real)
    void* xb = nullptr;              // block spectra of every class (complex)
    std::vector<FftPlan> ffts;       // one per run of classes with the same block length and kind
    std::vector<int> fft_first;      // first class of each run
    // analytic signal of the padded batch (classes with analytic = 1): one-sided spectrum ->
    // length-M inverse transform, in place
    void* xa = nullptr;
    FftPlan inv_m;                   // rocFFT route (float64, or M outside the four-step kernels' range)
    AnalyticFft* ana = nullptr;      // four-step kernels of ssq_cwt_tiles.hip (float32)
    int n_analytic = 0;
    bool own4096 = false;            // float32: the P = 4096 classes' spectra by block_spectra4096_kernel (one launch)
    bool own_big = false;            // ... and short launches of P = 8192 / 16384 blocks by block_spectra_multi_kernel
    int64_t n_generic = 0;
    // exact (full-length, four-step) path for the rows the blocks cannot take
    bool exact_ok = false;
    int exA = 0, exB = 0, n_exact = 0;
    void* twM = nullptr; void* zbuf = nullptr;
    const float* e_bank = nullptr; const int64_t* e_off = nullptr; const int32_t* e_lo = nullptr;
    const int32_t* e_rows = nullptr;
    int setup_exact(const float* bank_dev, const int64_t* band_off_dev, const int32_t* band_lo_dev,
                    const int32_t* gen_rows_dev, const std::vector<int64_t>& h_off,
                    const std::vector<int32_t>& h_lo, const std::vector<int32_t>& h_gen, int64_t& bytes);
    // exact rows of signals sig .. sig+nsig-1 (nsig <= group); xh_all: spectra of the whole batch
    int run_exact(int sig, int nsig, const void* xh_all, float* Wx, float* dWx, float* w, unsigned short* kidx,
                  const float* row_scale, double dt, const SsqParams& sp, hipStream_t stream);

    int create(const ssq_cwt_blocks_desc& d, int dtype, int64_t M, int64_t N, int64_t n1, int64_t na,
               int64_t max_batch, int64_t& bytes);
    void destroy();
    // block spectra of all classes for the padded batch xp (max_batch x M) and its half
    // spectrum xh (max_batch x (M / 2 + 1), what the analytic classes start from)
    // `need`: per class, 0 = skip (no row of the launch uses it), or nullptr for all
    int spectra(const void* xp, const void* xh, int64_t batch, hipStream_t stream,
                const unsigned char* need = nullptr);
    // all block rows of signals sig .. sig+nsig-1; kidx holds nsig maps
    // `limit`: per L' slot, run only the leading items (the tile path takes the other rows)
    int run(int sig, int nsig, float* Wx, float* dWx, float* w, unsigned short* kidx,
            const float* row_scale, double dt, const SsqParams& sp, hipStream_t stream,
            const int64_t* limit = nullptr);
    // the same for a float64 plan (2048 points per 128-thread workgroup)
    int run64(int sig, int nsig, double* Wx, double* dWx, double* w, unsigned short* kidx,
              const double* row_scale, double dt, const SsqParams& sp, hipStream_t stream);
};

}  // namespace ssq
