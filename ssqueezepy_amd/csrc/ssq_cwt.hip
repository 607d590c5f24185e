// ssq_cwt.hip -- CWT / synchrosqueezed-CWT plan (C ABI: ssq_cwt_*): creation, the tables of the
// fast paths, and the orchestration of one execute.
//
// Per launch group of signals (reference: ssqueezepy/_cwt.py:255-306, _ssq_cwt.py:250-289):
//   x (N) --pad_kernel--> xp (M) --rocFFT R2C--> xh (M/2+1)                       (whole batch)
//   fast paths, when the plan has them (padded power-of-two length, analytic bank):
//     block rows   ssq_cwt_blocks.hip: blockzoom kernels (overlap-save blocks, LDS FFTs) ->
//                  Wx (+ dWx / w / 2-byte bin map)
//     exact rows   ssq_cwt_blocks.hip: four-step full-length FFT for the rows whose band is cut
//                  by the Nyquist bin (float32)
//     tile rows    ssq_cwt_tiles.hip (fused ssq_cwt form only): decimated baseband samples on a
//                  side stream, then the column-tile kernel: Wx of the interpolated rows and Tx
//                  of ALL rows (reads back the Wx + bins the block / exact kernels left)
//   generic path (any length / padtype None / rpadded / non-analytic banks), in row chunks:
//     xh, banded bank --bank_multiply_kernel--> P = psih*xh, dP = P*(1j*xi/dt)
//     --rocFFT C2C inverse, batched, scaled 1/M, in place--> padded Wx, dWx
//     --cwt_epilogue_kernel--> Wx[:, n1:n1+N] (+ dWx / w / bin map)
//   without the tile path: bin map or w --accumulate kernels (ssq_kernels.hip)--> Tx
// Only the returned arrays (Wx, Tx [, dWx, w]) are written at full size; padded intermediates
// live in plan-owned workspaces reused chunk after chunk.
//
// The bank is *banded*: each row keeps only the contiguous run of DFT bins on which the wavelet
// is non-negligible (7 % of na*M at N=160k), so it stays in the 256 MiB Infinity Cache across calls.
//
// Compiled with -ffp-contract=off (the epilogue computes bin indices; see ssq_kernels.hip).
#include "ssq_common.h"
#include "ssq_fft.h"
#include "ssq_blocks.h"
#include "ssq_tiles.h"
#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace ssq {

// from ssq_kernels.hip (same arithmetic, re-declared here as device inlines would
// need a shared header; kept in one place via this include)
#include "ssq_point_math.inl"

template <typename T>
__global__ __launch_bounds__(256) void bank_multiply_kernel(
    const T* __restrict__ xh,        // (M/2+1) complex, spectrum of the real padded signal
    const T* __restrict__ bank, const int64_t* __restrict__ band_off,
    const int32_t* __restrict__ band_lo,
    T* __restrict__ prod,            // [rows][nplanes][M] complex
    int64_t M, const int32_t* __restrict__ rowlist, int64_t row0, int nplanes, double h, T inv_dt) {
    const int64_t r = blockIdx.y;            // row within the chunk
    const int64_t i = rowlist[row0 + r];
    const int64_t lo = band_lo[i];
    const int64_t off = band_off[i];
    const int64_t len = band_off[i + 1] - off;
    T* P = prod + (size_t)r * nplanes * 2 * M;
    T* dP = P + 2 * M;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < M;
         k += (int64_t)gridDim.x * blockDim.x) {
        T re = T(0), im = T(0), dre = T(0), dim = T(0);
        int64_t t = k - lo;
        if (t >= 0 && t < len) {
            T psi = bank[off + t];
            T a, b;
            if (k <= M / 2) { a = xh[2 * k]; b = xh[2 * k + 1]; }
            else { a = xh[2 * (M - k)]; b = -xh[2 * (M - k) + 1]; }
            re = psi * a; im = psi * b;
            if (nplanes > 1) {
                // xi_k = k*h (k <= M/2) or (k-M)*h, formed in double and stored in T
                // (wavelets.py:473-484); multiplier 1j*xi/dt as NumPy forms it: xi * (1/dt)
                int64_t ks = k <= M / 2 ? k : k - M;
                T m = (T)((double)ks * h) * inv_dt;
                dre = -(im * m); dim = re * m;
            }
        }
        P[2 * k] = re; P[2 * k + 1] = im;
        if (nplanes > 1) { dP[2 * k] = dre; dP[2 * k + 1] = dim; }
    }
}

struct EpilogueArgs {
    void* Wx; void* dWx; void* w; unsigned short* kidx;
    int64_t out_cols;     // N, or M when rpadded
    int64_t col0;         // n1, or 0 when rpadded
    int64_t row0;
    const int32_t* rowlist;
    const void* row_scale;
    double gamma;
    int have_ssq;
};

template <typename T>
__global__ __launch_bounds__(256) void cwt_epilogue_kernel(const T* __restrict__ prod, int64_t M,
                                                           int nplanes, int64_t na,
                                                           EpilogueArgs ea, SsqParams sp) {
    const int64_t r = blockIdx.y;
    const int64_t i = ea.rowlist[ea.row0 + r];
    const T* P = prod + (size_t)r * nplanes * 2 * M;
    const T* dP = P + 2 * M;
    const int64_t omax = na - 1;
    T rs = ea.row_scale ? ((const T*)ea.row_scale)[i] : T(1);
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ea.out_cols;
         j += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = ea.col0 + j;
        int64_t q = i * ea.out_cols + j;
        T c = P[2 * p], d = P[2 * p + 1];
        if (ea.row_scale) { c = c * rs; d = d * rs; }
        ((T*)ea.Wx)[2 * q] = c; ((T*)ea.Wx)[2 * q + 1] = d;
        if (nplanes > 1) {
            T a = dP[2 * p], b = dP[2 * p + 1];
            if (ea.row_scale) { a = a * rs; b = b * rs; }
            if (ea.dWx) { ((T*)ea.dWx)[2 * q] = a; ((T*)ea.dWx)[2 * q + 1] = b; }
            if (ea.w) {
                // two-step form: phase_cwt (algos.py:720-740), threshold |Wx| < gamma
                T wv;
                if (mag_lt(c, d, (T)ea.gamma)) wv = (T)INFINITY;
                else wv = (T)fabs(phase_ratio(a, b, c, d));
                ((T*)ea.w)[q] = wv;
            }
            if (ea.kidx) {
                // fused form (algos.py:859-953), threshold |Wx| > gamma
                unsigned short kk = 0xFFFFu;
                if (mag_gt(c, d, ea.gamma)) {
                    int64_t k = bin_of_point(a, b, c, d, false, T(0), sp, omax);
                    kk = (unsigned short)(sp.flipud ? omax - k : k);
                }
                ea.kidx[kidx_index(i, j, na, ea.out_cols)] = kk;
            }
        }
    }
}

}  // namespace ssq

using namespace ssq;

struct ssq_cwt_plan {
    ssq_cwt_desc d;
    int csize() const { return d.dtype == SSQ_F32 ? 8 : 16; }
    int rsize() const { return d.dtype == SSQ_F32 ? 4 : 8; }
    // device copies
    void* bank = nullptr; int64_t* band_off = nullptr; int32_t* band_lo = nullptr;
    void* row_scale = nullptr;
    std::vector<int64_t> h_band_off; std::vector<int32_t> h_band_lo;
    // workspace
    void* xp = nullptr; void* xh = nullptr; void* prod = nullptr; unsigned short* kidx = nullptr;
    int64_t rows_chunk = 0;
    size_t prod_bytes = 0;                // size of `prod` once it is allocated
    int64_t bytes = 0;
    // signals per launch group: the fast-path kernels and the reassignment take `group`
    // signals as a grid dimension (fewer, longer launches; the bin map holds `group` maps)
    int group = 1;
    FftPlan fwd;                          // R2C, batch = max_batch
    std::map<int64_t, FftPlan> inv;       // C2C inverse keyed by transform count
    // ssq
    bool have_ssq = false; SsqParams sp{}; void* cst = nullptr;   // cst: the current entry of `weights`
    WeightVersions weights;
    float cst0 = 0.f;                     // first weight (all of them when sp.cst_uniform)
    PlanOrder order;
    std::string algo = "rocfft";
    // rows evaluated by the exact full-length path (all rows unless a block plan
    // took some over)
    int32_t* gen_rows = nullptr; int64_t n_gen = 0;
    int32_t* all_rows = nullptr;          // identity list (rpadded output bypasses blocks)
    const int32_t* gen_rows_for(bool use_blocks) const { return use_blocks ? gen_rows : all_rows; }
    BlockPlan* blk = nullptr;
    TilePlan* tile = nullptr;             // column-tile path of the fused ssq form (ssq_cwt_tiles.hip)
    bool executed = false;
    unsigned short* kdump = nullptr;      // diagnostic (ssq_cwt_plan_set_bin_dump): caller-owned (batch, na, n) bin map
    // optional per-stage HIP-event timing (bench.py reads it): 0 = pad + forward FFT +
    // block spectra, 1 = block rows, 2 = exact / generic rows, 3 = reassignment
    bool timing = false;
    std::vector<hipEvent_t> tev;          // 5 events per signal slot + 2 per execute
    double stage_ms[4] = {0, 0, 0, 0};
    int64_t timed_signals = 0;
};

static int h2d(void* dst, const void* src, size_t bytes) {
    SSQ_CHECK_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
static int dev_alloc(void** p, size_t bytes, int64_t& acc) {
    SSQ_CHECK_HIP(hipMalloc(p, bytes ? bytes : 1));
    acc += (int64_t)bytes;
    return 0;
}

extern "C" {

int ssq_cwt_plan_create(ssq_cwt_plan** out, const ssq_cwt_desc* desc) {
    SSQ_REQUIRE(out && desc, "ssq_cwt_plan_create: null pointer");
    const ssq_cwt_desc& d = *desc;
    SSQ_REQUIRE(d.dtype == SSQ_F32 || d.dtype == SSQ_F64, "bad dtype %d", d.dtype);
    SSQ_REQUIRE(d.n >= 1 && d.m >= d.n && d.na >= 1, "bad sizes n=%lld m=%lld na=%lld",
                (long long)d.n, (long long)d.m, (long long)d.na);
    SSQ_REQUIRE(d.n1 >= 0 && d.n1 + d.n <= d.m, "bad left pad %lld", (long long)d.n1);
    SSQ_REQUIRE(d.padtype >= SSQ_PAD_NONE && d.padtype <= SSQ_PAD_WRAP, "bad padtype %d", d.padtype);
    SSQ_REQUIRE(d.padtype != SSQ_PAD_NONE || d.m == d.n, "padtype NONE requires m == n");
    SSQ_REQUIRE(d.bank && d.band_off && d.band_lo, "bank arrays must not be null");
    SSQ_REQUIRE(d.dt > 0, "dt must be > 0");
    SSQ_REQUIRE(d.na < 65535, "na must be < 65535");
    for (int64_t i = 0; i < d.na; ++i) {
        int64_t len = d.band_off[i + 1] - d.band_off[i];
        SSQ_REQUIRE(len >= 0 && d.band_lo[i] >= 0 && d.band_lo[i] + len <= d.m,
                    "row %lld: band [%d, +%lld) outside [0, %lld)", (long long)i, d.band_lo[i],
                    (long long)len, (long long)d.m);
    }
    auto* pl = new ssq_cwt_plan();
    pl->d = d;
    if (pl->d.max_batch < 1) pl->d.max_batch = 1;
    pl->h_band_off.assign(d.band_off, d.band_off + d.na + 1);
    pl->h_band_lo.assign(d.band_lo, d.band_lo + d.na);
    pl->d.bank = nullptr; pl->d.band_off = nullptr; pl->d.band_lo = nullptr; pl->d.row_scale = nullptr;
    const int rs = pl->rsize(), cs = pl->csize();
    const int64_t nnz = d.band_off[d.na];
    int rc = 0;
#define TRY(x) do { rc = (x); if (rc) { ssq_cwt_plan_destroy(pl); return rc; } } while (0)
    TRY(dev_alloc(&pl->bank, (size_t)nnz * rs, pl->bytes));
    TRY(dev_alloc((void**)&pl->band_off, (size_t)(d.na + 1) * 8, pl->bytes));
    TRY(dev_alloc((void**)&pl->band_lo, (size_t)d.na * 4, pl->bytes));
    TRY(h2d(pl->bank, d.bank, (size_t)nnz * rs));
    TRY(h2d(pl->band_off, d.band_off, (size_t)(d.na + 1) * 8));
    TRY(h2d(pl->band_lo, d.band_lo, (size_t)d.na * 4));
    if (d.row_scale) {
        TRY(dev_alloc(&pl->row_scale, (size_t)d.na * rs, pl->bytes));
        TRY(h2d(pl->row_scale, d.row_scale, (size_t)d.na * rs));
    }
    TRY(dev_alloc(&pl->xp, (size_t)pl->d.max_batch * d.m * rs, pl->bytes));
    TRY(dev_alloc(&pl->xh, (size_t)pl->d.max_batch * (d.m / 2 + 1) * cs, pl->bytes));
    // product workspace: 2 planes (Wx, dWx) per row, bounded to ~2 GiB
    const size_t per_row = (size_t)2 * d.m * cs;
    const size_t budget = (size_t)2 << 30;
    pl->rows_chunk = std::max<int64_t>(1, std::min<int64_t>(d.na, (int64_t)(budget / per_row)));
    // (allocated by the first execute that sends rows through the generic route: a plan whose rows all run on the
    // block / tile kernels -- the usual case -- never needs these up to 2 GiB)
    pl->prod_bytes = (size_t)pl->rows_chunk * per_row;
    {
        // default: up to 16 signals per launch (measured at config 2: 16 -> +1 % over 8), bin maps
        // bounded to ~2 GiB
        int64_t g = std::min<int64_t>(16, std::max<int64_t>(1, ((int64_t)1 << 30) / (d.na * d.n)));
        if (const char* e = getenv("SSQ_DEBUG_CWT_GROUP")) g = atoi(e);
        pl->group = (int)std::max<int64_t>(1, std::min<int64_t>(g, pl->d.max_batch));
    }
    TRY(dev_alloc((void**)&pl->kidx, ((size_t)pl->group * d.na * d.n + 64) * 2, pl->bytes));
    {
        std::vector<int32_t> all((size_t)d.na);
        for (int64_t i = 0; i < d.na; ++i) all[i] = (int32_t)i;
        TRY(dev_alloc((void**)&pl->gen_rows, (size_t)d.na * 4, pl->bytes));
        TRY(dev_alloc((void**)&pl->all_rows, (size_t)d.na * 4, pl->bytes));
        TRY(h2d(pl->gen_rows, all.data(), (size_t)d.na * 4));
        TRY(h2d(pl->all_rows, all.data(), (size_t)d.na * 4));
        pl->n_gen = d.na;
    }
    TRY(pl->fwd.create(0, d.dtype, (size_t)d.m, (size_t)pl->d.max_batch, 1.0));
    pl->bytes += (int64_t)pl->fwd.work_bytes;
#undef TRY
    *out = pl;
    return 0;
}

void ssq_cwt_plan_destroy(ssq_cwt_plan* pl) {
    if (!pl) return;
    pl->fwd.destroy();
    for (auto& kv : pl->inv) kv.second.destroy();
    if (pl->blk) { pl->blk->destroy(); delete pl->blk; }
    if (pl->tile) { pl->tile->destroy(); delete pl->tile; }
    for (hipEvent_t e : pl->tev) (void)hipEventDestroy(e);
    pl->weights.destroy();
    pl->order.destroy();
    void* ptrs[] = {pl->bank, pl->band_off, pl->band_lo, pl->row_scale, pl->xp, pl->xh, pl->prod,
                    pl->kidx, pl->gen_rows, pl->all_rows};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete pl;
}

int ssq_cwt_plan_set_ssq(ssq_cwt_plan* pl, int grid, const double* params, const void* cst,
                         int cst_f64, int flipud, double gamma) {
    SSQ_REQUIRE(pl && params && cst, "ssq_cwt_plan_set_ssq: null pointer");
    SSQ_REQUIRE(grid >= SSQ_GRID_LOG && grid <= SSQ_GRID_LIN, "unknown grid kind %d", grid);
    // (kernels receive SsqParams by value at launch: an execute already enqueued keeps its own;
    // the lock orders this update against an execute being enqueued by another host thread)
    std::lock_guard<std::mutex> lock(pl->order.mu);
    for (int t = 0; t < 5; ++t) pl->sp.p[t] = params[t];
    pl->sp.grid = grid; pl->sp.flipud = flipud ? 1 : 0; pl->sp.gamma = gamma;
    pl->sp.cst_f64 = (cst_f64 && pl->d.dtype == SSQ_F32) ? 1 : 0;
    {   // the weights come from the host here: note when they are one scalar repeated
        const bool wide = cst_f64 || pl->d.dtype == SSQ_F64;
        bool uni = true;
        for (int64_t i = 1; i < pl->d.na && uni; ++i)
            uni = wide ? ((const double*)cst)[i] == ((const double*)cst)[0]
                       : ((const float*)cst)[i] == ((const float*)cst)[0];
        pl->sp.cst_uniform = uni ? 1 : 0;
    }
    finalize_params(pl->sp);
    size_t bytes = (size_t)pl->d.na * ((cst_f64 || pl->d.dtype == SSQ_F64) ? 8 : 4);
    pl->cst0 = (cst_f64 || pl->d.dtype == SSQ_F64) ? (float)((const double*)cst)[0] : ((const float*)cst)[0];
    int rc = pl->weights.upload(&pl->cst, cst, bytes);
    if (rc) return rc;
    pl->have_ssq = true;
    return 0;
}

int ssq_cwt_plan_set_blocks(ssq_cwt_plan* pl, const ssq_cwt_blocks_desc* bd) {
    SSQ_REQUIRE(pl && bd, "ssq_cwt_plan_set_blocks: null pointer");
    SSQ_REQUIRE(!pl->executed && !pl->blk, "block tables must be set once, before the first execute");
    SSQ_REQUIRE(pl->d.padtype != SSQ_PAD_NONE, "the block path needs a padded (power-of-two) length");
    SSQ_REQUIRE((pl->d.m & (pl->d.m - 1)) == 0, "the block path needs a power-of-two padded length");
    SSQ_REQUIRE(bd->n_classes >= 1 && bd->n_generic >= 0 && bd->n_generic <= pl->d.na, "bad block tables");
    for (int c = 0; c < bd->n_classes; ++c) {
        int64_t P = bd->classes[5 * c];
        SSQ_REQUIRE(P >= 4096 && (P & (P - 1)) == 0 && P <= pl->d.m, "class %d: bad block length %lld", c, (long long)P);
    }
    auto* b = new BlockPlan();
    b->group = pl->group;
    int rc = b->create(*bd, pl->d.dtype, pl->d.m, pl->d.n, pl->d.n1, pl->d.na, pl->d.max_batch, pl->bytes);
    if (rc) { b->destroy(); delete b; return rc; }
    pl->blk = b;
    pl->n_gen = bd->n_generic;
    if (pl->n_gen)
        SSQ_CHECK_HIP(hipMemcpy(pl->gen_rows, bd->generic_rows, (size_t)pl->n_gen * 4, hipMemcpyHostToDevice));
    if (pl->n_gen && pl->d.dtype == SSQ_F32) {      // four-step exact kernels: float32 only
        std::vector<int32_t> hg(bd->generic_rows, bd->generic_rows + bd->n_generic);
        rc = b->setup_exact((const float*)pl->bank, pl->band_off, pl->band_lo, pl->gen_rows,
                            pl->h_band_off, pl->h_band_lo, hg, pl->bytes);
        if (rc) return rc;
    }
    pl->algo = !pl->n_gen ? "blockzoom" : (b->exact_ok ? "blockzoom+fourstep" : "blockzoom+rocfft");
    return 0;
}

int ssq_cwt_plan_set_tiles(ssq_cwt_plan* pl, const ssq_cwt_tiles_desc* td) {
    SSQ_REQUIRE(pl && td, "ssq_cwt_plan_set_tiles: null pointer");
    SSQ_REQUIRE(pl->blk && !pl->executed && !pl->tile,
                "tile tables must be set once, after the block tables, before the first execute");
    SSQ_REQUIRE(pl->d.dtype == SSQ_F32, "the tile path is float32 only");
    for (int t = 0; t < 5; ++t)
        SSQ_REQUIRE(td->n_items_tile[t] >= 0 && td->n_items_tile[t] <= pl->blk->n_items[t],
                    "tile tables: bad block item count in slot %d", t);
    auto* tp = new TilePlan();
    int rc = tp->create(*td, pl->d.m, pl->d.n, pl->d.n1, pl->d.na, pl->group, pl->d.dt, pl->bytes);
    if (rc) { tp->destroy(); delete tp; return rc; }
    tp->class_need.assign((size_t)pl->blk->nc, 0);
    for (int t = 0; t < 5; ++t)
        for (int64_t q = 0; q < td->n_items_tile[t]; ++q) {
            const int c = pl->blk->h_items[t][4 * q + 3];
            if (c >= 0 && c < pl->blk->nc) tp->class_need[(size_t)c] = 1;
        }
    pl->tile = tp;
    pl->algo += "+tiles";
    return 0;
}

int ssq_cwt_plan_timing(ssq_cwt_plan* pl, int enable, double* stage_ms, int64_t* signals) {
    SSQ_REQUIRE(pl, "ssq_cwt_plan_timing: null plan");
    if (stage_ms) for (int t = 0; t < 4; ++t) stage_ms[t] = pl->stage_ms[t];
    if (signals) *signals = pl->timed_signals;
    if (enable >= 0) {
        pl->timing = enable != 0;
        for (int t = 0; t < 4; ++t) pl->stage_ms[t] = 0;
        pl->timed_signals = 0;
    }
    return 0;
}

int ssq_cwt_plan_set_bin_dump(ssq_cwt_plan* pl, unsigned short* kmap) {
    SSQ_REQUIRE(pl, "ssq_cwt_plan_set_bin_dump: null plan");
    std::lock_guard<std::mutex> lock(pl->order.mu);
    pl->kdump = kmap;
    return 0;
}

int ssq_cwt_plan_group(const ssq_cwt_plan* pl) { return pl ? pl->group : 0; }
int ssq_cwt_tile_rows_per_step(void) { return ssq::tile_rows_per_step(); }
int64_t ssq_cwt_plan_tiles_done(ssq_cwt_plan* pl, void* stream) {
    if (!pl || !pl->tile) return 0;
    return pl->tile->tiles_done(as_stream(stream));
}
int ssq_cwt_plan_tile_cols(const ssq_cwt_plan* pl) { return (pl && pl->tile) ? pl->tile->tile_cols() : 0; }
int ssq_cwt_plan_tile_counters(ssq_cwt_plan* pl, unsigned long long* out, int n, void* stream) {
    SSQ_REQUIRE(pl && out && n >= 0 && n <= 512, "ssq_cwt_plan_tile_counters: bad arguments");
    memset(out, 0, (size_t)n * 8);
    if (!pl->tile || !pl->tile->counters) return 0;
    SSQ_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    SSQ_CHECK_HIP(hipMemcpy(out, pl->tile->counters, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}
int ssq_cwt_plan_tile_kernel(const ssq_cwt_plan* pl) { return (pl && pl->tile) ? pl->tile->tile_kernel() : 0; }
int64_t ssq_cwt_plan_bytes(const ssq_cwt_plan* pl) { return pl ? pl->bytes : 0; }
const char* ssq_cwt_plan_algo(const ssq_cwt_plan* pl) { return pl ? pl->algo.c_str() : ""; }

}  // extern "C"

template <typename T>
static int cwt_execute_t(ssq_cwt_plan* pl, const void* x, int64_t batch, void* Wx, void* dWx,
                         void* Tx, void* w, int rpadded, hipStream_t stream) {
    const ssq_cwt_desc& d = pl->d;
    const int64_t M = d.m, N = d.n, na = d.na;
    const int64_t out_cols = rpadded ? M : N;
    const bool deriv = dWx || Tx || w;
    const int nplanes = deriv ? 2 : 1;
    const double h = (2.0 * 3.141592653589793) / (double)M;
    const T inv_dt = T(1) / (T)d.dt;
    if (pl->timing && pl->tev.size() < 2) {
        for (int t = 0; t < 2; ++t) { hipEvent_t e; SSQ_CHECK_HIP(hipEventCreate(&e)); pl->tev.push_back(e); }
    }
    if (pl->timing) (void)hipEventRecord(pl->tev[0], stream);

    // pad (or copy) the whole batch, forward FFT of all signals at once
    const T* xsrc = (const T*)x;
    if (d.padtype != SSQ_PAD_NONE) {
        int rc = ssq_pad_signal(d.dtype, x, pl->xp, batch, N, d.n1, M - N - d.n1, d.padtype, stream);
        if (rc) return rc;
        xsrc = (const T*)pl->xp;
    } else {
        SSQ_CHECK_HIP(hipMemcpyAsync(pl->xp, x, (size_t)batch * M * sizeof(T), hipMemcpyDeviceToDevice, stream));
        xsrc = (const T*)pl->xp;
    }
    if (batch == d.max_batch) {
        int rc = pl->fwd.execute((void*)xsrc, pl->xh, stream);
        if (rc) return rc;
    } else {
        // smaller batch than planned: run the planned batch over the (valid) prefix;
        // rows past `batch` hold stale but finite data and are ignored
        int rc = pl->fwd.execute((void*)xsrc, pl->xh, stream);
        if (rc) return rc;
    }

    pl->executed = true;
    const bool tm = pl->timing;
    if (tm && pl->tev.size() < (size_t)(2 + 6 * batch)) {
        size_t need = (size_t)(2 + 6 * batch);
        while (pl->tev.size() < need) {
            hipEvent_t e; SSQ_CHECK_HIP(hipEventCreate(&e)); pl->tev.push_back(e);
        }
    }
    auto mark = [&](size_t idx) { if (tm) (void)hipEventRecord(pl->tev[idx], stream); };
    const bool use_blocks = pl->blk && !rpadded;
    const int64_t n_gen = use_blocks ? pl->n_gen : na;
    // fused ssq form on the column-tile path: block / exact kernels only for the rows the
    // tile kernel reads back, no separate reassignment launch
    // (a tile plan the selected kernel cannot run -- more rows than the ordered kernel's tile holds -- is decided
    // HERE, before any row is routed: those calls take the block kernels + the separate reassignment)
    const bool use_tiles = use_blocks && pl->tile && pl->tile->usable() && Tx && !w && sizeof(T) == 4;
    // The first launch group's decimated samples need the spectra xh and nothing else: their kernels go to the side stream
    // HERE, beside the analytic signal and the block spectra, not only beside the block rows (a short signal's ssq_cwt:
    // their three launches are the longer branch). SSQ_DEBUG_EARLY_FORK=0: forked behind the block spectra, as before.
    static const bool early_ok = !(getenv("SSQ_DEBUG_EARLY_FORK") && atoi(getenv("SSQ_DEBUG_EARLY_FORK")) == 0);
    const bool early = early_ok && use_tiles && pl->tile->side && !tm;
    if (early) {
        SSQ_CHECK_HIP(hipEventRecord(pl->tile->ev_fork, stream));
        SSQ_CHECK_HIP(hipStreamWaitEvent(pl->tile->side, pl->tile->ev_fork, 0));
        int rc2 = pl->tile->spectra(0, (int)std::min<int64_t>(pl->group, batch), pl->xh, pl->tile->side);
        if (rc2) return rc2;
        SSQ_CHECK_HIP(hipEventRecord(pl->tile->ev_join, pl->tile->side));
    }
    if (use_blocks) {
        int rc = pl->blk->spectra(pl->xp, pl->xh, batch, stream, use_tiles ? pl->tile->class_need.data() : nullptr);
        if (rc) return rc;
    }
    mark(1);
    int64_t slot = 0;                                   // timing slot = launch group
    for (int64_t b0 = 0; b0 < batch; b0 += pl->group, ++slot) {
        const int ng = (int)std::min<int64_t>(pl->group, batch - b0);
        const bool fork = use_tiles && pl->tile->side && !tm;
        if (use_tiles && !(early && b0 == 0)) {   // decimated samples of the interpolated rows: counted with stage 0
            mark(2 + 4 * batch + 2 * slot);
            // (beside the block / exact kernels when not timing stage by stage; the fork also
            // orders this group's samples behind the previous group's tile kernel)
            hipStream_t ss = fork ? pl->tile->side : stream;
            if (fork) {
                SSQ_CHECK_HIP(hipEventRecord(pl->tile->ev_fork, stream));
                SSQ_CHECK_HIP(hipStreamWaitEvent(ss, pl->tile->ev_fork, 0));
            }
            int rc2 = pl->tile->spectra((int)b0, ng, pl->xh, ss);
            if (rc2) return rc2;
            if (fork) SSQ_CHECK_HIP(hipEventRecord(pl->tile->ev_join, ss));
            mark(2 + 4 * batch + 2 * slot + 1);
        }
        mark(2 + 4 * slot);
        unsigned short* kidx = (Tx && !w) ? pl->kidx : nullptr;
        if (use_blocks) {
            if constexpr (sizeof(T) == 4) {
                int rc = pl->blk->run((int)b0, ng, (float*)Wx, (float*)dWx, (float*)w, kidx,
                                      (const float*)pl->row_scale, d.dt, pl->sp, stream,
                                      use_tiles ? pl->tile->n_items_tile : nullptr);
                if (rc) return rc;
            } else {
                int rc = pl->blk->run64((int)b0, ng, (double*)Wx, (double*)dWx, (double*)w, kidx,
                                        (const double*)pl->row_scale, d.dt, pl->sp, stream);
                if (rc) return rc;
            }
        }
        mark(2 + 4 * slot + 1);
        if (use_blocks && pl->blk->exact_ok && n_gen > 0) {
            if constexpr (sizeof(T) == 4) {
                // (sub-groups of n signals, so that the four-step intermediate Z -- 4 MB per row and signal -- could stay
                // in the 256 MiB Infinity Cache between the two passes, were measured on the MI355X at config 2: 16 signals
                // at once 68.9 us per transform, 4: 76.2, 2: 73.7, 1: 85.3 -- the cache does not pay for the smaller launches)
                const int eg = ng;
                for (int s0 = 0; s0 < ng; s0 += eg) {
                    const int ns = std::min(eg, ng - s0);
                    int rc = pl->blk->run_exact((int)b0 + s0, ns, pl->xh, (float*)Wx, (float*)dWx, (float*)w,
                                                kidx ? kidx + (size_t)s0 * na * N : nullptr,
                                                (const float*)pl->row_scale, d.dt, pl->sp, stream);
                    if (rc) return rc;
                }
            }
        } else
        for (int64_t b = b0; b < b0 + ng; ++b) {
        const T* xh = (const T*)pl->xh + (size_t)b * (M / 2 + 1) * 2;
        T* Wx_b = Wx ? (T*)Wx + (size_t)b * na * out_cols * 2 : nullptr;
        T* dWx_b = dWx ? (T*)dWx + (size_t)b * na * out_cols * 2 : nullptr;
        T* w_b = w ? (T*)w + (size_t)b * na * out_cols : nullptr;
        if (n_gen > 0 && !pl->prod) {
            SSQ_CHECK_HIP(hipMalloc(&pl->prod, pl->prod_bytes ? pl->prod_bytes : 1));
            pl->bytes += (int64_t)pl->prod_bytes;
        }
        for (int64_t row0 = 0; row0 < n_gen; row0 += pl->rows_chunk) {
            const int64_t rows = std::min(pl->rows_chunk, n_gen - row0);
            unsigned gx = (unsigned)std::min<int64_t>((M + 255) / 256, 4096);
            hipLaunchKernelGGL((bank_multiply_kernel<T>), dim3(gx, (unsigned)rows), dim3(256), 0, stream,
                               xh, (const T*)pl->bank, pl->band_off, pl->band_lo, (T*)pl->prod, M,
                               pl->gen_rows_for(use_blocks), row0, nplanes, h, inv_dt);
            SSQ_LAUNCH_CHECK();
            const int64_t ntrans = rows * nplanes;
            auto it = pl->inv.find(ntrans);
            if (it == pl->inv.end()) {
                FftPlan fp;
                int rc = fp.create(1, d.dtype, (size_t)M, (size_t)ntrans, 1.0 / (double)M);
                if (rc) return rc;
                pl->bytes += (int64_t)fp.work_bytes;
                it = pl->inv.emplace(ntrans, fp).first;
            }
            int rc = it->second.execute(pl->prod, nullptr, stream);
            if (rc) return rc;
            EpilogueArgs ea;
            ea.Wx = Wx_b; ea.dWx = dWx_b; ea.w = w_b;
            ea.kidx = kidx ? kidx + (size_t)(b - b0) * na * N : nullptr;
            ea.out_cols = out_cols; ea.col0 = rpadded ? 0 : d.n1; ea.row0 = row0;
            ea.rowlist = pl->gen_rows_for(use_blocks);
            ea.row_scale = pl->row_scale; ea.gamma = pl->sp.gamma; ea.have_ssq = pl->have_ssq;
            unsigned ex = (unsigned)std::min<int64_t>((out_cols + 255) / 256, 4096);
            hipLaunchKernelGGL((cwt_epilogue_kernel<T>), dim3(ex, (unsigned)rows), dim3(256), 0, stream,
                               (const T*)pl->prod, M, nplanes, na, ea, pl->sp);
            SSQ_LAUNCH_CHECK();
        }
        }
        mark(2 + 4 * slot + 2);
        if (use_tiles) {
            if (fork) SSQ_CHECK_HIP(hipStreamWaitEvent(stream, pl->tile->ev_join, 0));
            int rc2 = pl->tile->run((int)b0, ng, (float*)Wx, (float*)dWx, (float*)Tx, pl->kidx, pl->cst, pl->cst0, pl->sp, stream,
                                    pl->kdump);
            if (rc2) return rc2;
        } else if (Tx) {
            // (diagnostic: the map the separate reassignment is about to consume)
            if (pl->kdump && !w)
                SSQ_CHECK_HIP(hipMemcpyAsync(pl->kdump + (size_t)b0 * na * N, pl->kidx, (size_t)ng * na * N * 2,
                                             hipMemcpyDeviceToDevice, stream));
            T* Wx_g = (T*)Wx + (size_t)b0 * na * out_cols * 2;
            T* w_g = w ? (T*)w + (size_t)b0 * na * out_cols : nullptr;
            T* Tx_g = (T*)Tx + (size_t)b0 * na * out_cols * 2;
            int rc2 = launch_accumulate(d.dtype, w ? BIN_FROM_W : BIN_FROM_KIDX, Wx_g,
                                        w ? (const void*)w_g : (const void*)pl->kidx, nullptr, Tx_g,
                                        pl->cst, pl->sp, ng, na, N, nullptr, stream);
            if (rc2) return rc2;
        }
        mark(2 + 4 * slot + 3);
    }
    const int64_t nslots = slot;
    if (tm) {
        SSQ_CHECK_HIP(hipEventSynchronize(pl->tev[2 + 4 * (nslots - 1) + 3]));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, pl->tev[0], pl->tev[1]); pl->stage_ms[0] += ms;
        for (int64_t b = 0; use_tiles && b < nslots; ++b) {
            (void)hipEventElapsedTime(&ms, pl->tev[2 + 4 * batch + 2 * b], pl->tev[2 + 4 * batch + 2 * b + 1]);
            pl->stage_ms[0] += ms;
        }
        for (int64_t b = 0; b < nslots; ++b)
            for (int st = 0; st < 3; ++st) {
                (void)hipEventElapsedTime(&ms, pl->tev[2 + 4 * b + st], pl->tev[2 + 4 * b + st + 1]);
                pl->stage_ms[1 + st] += ms;
            }
        pl->timed_signals += batch;
    }
    return 0;
}

extern "C" {

int ssq_cwt_execute(ssq_cwt_plan* pl, const void* x, int64_t batch, void* Wx, void* dWx, void* Tx,
                    void* w, int rpadded, void* stream) {
    SSQ_REQUIRE(pl && x, "ssq_cwt_execute: null pointer");
    SSQ_REQUIRE(batch >= 1 && batch <= pl->d.max_batch, "batch %lld outside [1, %lld]",
                (long long)batch, (long long)pl->d.max_batch);
    SSQ_REQUIRE(Wx || !(dWx || Tx || w), "Wx buffer is required");
    SSQ_REQUIRE(!(rpadded && (Tx || w)), "rpadded output excludes Tx / w");
    SSQ_REQUIRE(!Tx || pl->have_ssq, "Tx requested but ssq parameters were not set");
    SSQ_REQUIRE(!w || pl->have_ssq, "w requested but ssq parameters (gamma) were not set");
    SSQ_REQUIRE(Wx, "Wx buffer is required");
    hipStream_t st = as_stream(stream);
    pl->order.enter(st);
    // (replaying the launches of small transforms from a hipGraph was built in round 2 and
    // measured slower than the eager launches on ROCm 7.2 -- config 1: 0.298 vs 0.126 ms -- so it
    // is gone; what helps small transforms is fewer launches)
    const int rc = pl->d.dtype == SSQ_F32 ? cwt_execute_t<float>(pl, x, batch, Wx, dWx, Tx, w, rpadded, st)
                                          : cwt_execute_t<double>(pl, x, batch, Wx, dWx, Tx, w, rpadded, st);
    pl->order.leave(st);
    return rc;
}

}  // extern "C"
