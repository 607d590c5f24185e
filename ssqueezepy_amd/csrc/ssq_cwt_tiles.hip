// ssq_cwt_tiles.hip -- the column-tile path of the fused ssq_cwt form (float32, gfx950).
//
// Math and planning: ssqueezepy_amd/_tiles.py. Two kernels:
//
//   tile_spectra_kernel   band of row i (K bins around bin kc of the M-grid) x spectrum of the
//                         padded signal x compensated bank value -> baseband spectrum of
//                         length L = M / R, zero outside the band; one batched rocFFT inverse
//                         per decimation class turns it into the samples u_i[q] (plan-owned
//                         intermediate, ~34 MB per signal at N=160k: L2 / Infinity-Cache food).
//
//   tile_kernel           one persistent workgroup per CU walks 64-column tiles of one signal
//                         after the other. The 64 columns x na bins of the tile's Tx live in LDS
//                         (which is why there is one workgroup per CU); lane = column. Two
//                         kinds of wavefronts:
//                           * producers (all but one): take the steps (4 consecutive rows) of
//                             the interpolated rows from a ticket counter; per row and lane ONE
//                             8-byte load of u_i (the lanes hold a window of consecutive
//                             samples, taps come from the neighbours with ds_bpermute), 8 taps
//                             x (phi, phi') as packed FMAs, modulation by hardware sin / cos of
//                             the exact phase kc n mod M, Wx stored (512-byte runs), phase
//                             transform + bin exactly as the other fused kernels do
//                             (ssq_point_math.inl), the 2-byte bin stored to a small ring in
//                             global memory (L2-resident), a flag in LDS. They hold no tile
//                             state and wait for nobody (but the ring's back-pressure).
//                           * the updater (wavefront 0): walks ALL rows of the tile in ascending
//                             order -- the rows the block / exact kernels left in HBM (Wx + bin
//                             map) and, behind the producers' flags, the interpolated rows (Wx
//                             back from L2, bins from the ring) -- and does the reassignment
//                             T[bin] += Wx * const in LDS, several steps of loads in flight.
//                             One wavefront, program order: every cell receives its contributions
//                             in ascending row order, the reference's (algos.py:859-953), so the
//                             float sums are bit-identical to the CPU loop on the same Wx / dWx.
//                             At the end of a tile it writes Tx once and clears the tile.
//                         No workgroup barrier after the prologue, no atomics on data.
//
// Compiled with -ffp-contract=off (bin indices); multiply-adds that may fuse are written as
// explicit fmaf so every instantiation rounds identically.
#include "ssq_common.h"
#include "ssq_tiles.h"
#include <algorithm>
#include <type_traits>

namespace ssq {

#include "ssq_point_math.inl"

constexpr int TILE_COLS = 64;     // columns per workgroup (one per lane)
constexpr int TILE_G = 4;         // rows per step
constexpr int TILE_W = 8;         // taps
constexpr int TILE_RING = 128;    // producer steps whose bins may be in flight (power of two)
#ifndef SSQ_TILE_DEPTH
#define SSQ_TILE_DEPTH 4
#endif
constexpr int TILE_D = SSQ_TILE_DEPTH;   // steps of loads the updater keeps in flight
// tuning experiments (A/B builds, tools/ab_variant.sh): 1 = the updater skips the rows read back,
// 2 = the updater runs at high priority, 4 = the producers skip their arithmetic (WRONG RESULTS with 1, 4)
#ifndef SSQ_TILE_EXP
#define SSQ_TILE_EXP 0
#endif
#ifndef SSQ_TILE_BPSLEEP
#define SSQ_TILE_BPSLEEP 20     // x 64 clocks between two looks at the updaters' progress
#endif
constexpr int TILE_NOBIN = 0xFFFF;
constexpr int TILE_NU = 1;        // updater wavefronts per workgroup

struct TileArgs {
    const TileSeg* steps; const TileRow* rows;       // one TileSeg record per step
    const TileSeg* psegs; const TileRow* prows;      // the same records, producer steps only (dense)
    const int4* usegs; int nusegs;                   // the updaters' view: per segment (first row, rows, kind, producer steps before it)
    const float4* wtab; const float2* U;
    const void* cst;
    float2* Wx; float2* dWx; float2* Tx; const unsigned short* kidx;
    unsigned short* ring;                            // per workgroup: TILE_RING x 4 x 64 bins
    int64_t N, na;
    int nsteps, npsteps, n1, mmask, sig0, nsig;
    int cstk;            // reassignment weights: 0 one float (cst0), 1 float per row, 2 double per row
    float inv_m;         // 1 / M
    float theta_scale;   // 2 pi / (M dt): theta of a row = kc * theta_scale
    float cst0;          // the reassignment weight when it is the same for every row
    unsigned long long* counters;   // [0] += tiles finished by the updater (what actually ran)
    unsigned long long* trace;      // tuning aid (SSQ_TILE_TRACE): shader-clock stamps of one workgroup
    double gamma;
};

__global__ __launch_bounds__(256) void tile_spectra_kernel(const float2* __restrict__ xh_all,
                                                           int64_t xh_stride, int sig0,
                                                           const TileIRow* __restrict__ irows,
                                                           const float* __restrict__ tbank,
                                                           float2* __restrict__ U) {
    const TileIRow r = irows[blockIdx.y];
    const int s = blockIdx.z;
    const float2* xh = xh_all + (int64_t)(sig0 + s) * xh_stride;
    float2* u = U + r.ubase + (int64_t)s * r.sig_stride;
    const int half = r.L >> 1;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < r.L; p += gridDim.x * blockDim.x) {
        const int kk = p < half ? p : p - r.L;            // signed baseband bin
        const int t = r.kc + kk - r.lo;
        float2 z = make_float2(0.f, 0.f);
        if (t >= 0 && t < r.K) {
            const float2 x = xh[r.lo + t];
            const float b = tbank[r.tb_off + t];
            z = make_float2(x.x * b, x.y * b);
        }
        u[p] = z;
    }
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}

// workgroup-scope synchronisation through LDS words (all wavefronts of a workgroup share the
// CU's L1, so workgroup scope costs waits only, no cache maintenance)
__device__ __forceinline__ int lds_load_acquire(const int* p) {
    return __scoped_atomic_load_n(p, __ATOMIC_ACQUIRE, __MEMORY_SCOPE_WRKGRP);
}
__device__ __forceinline__ int lds_load_relaxed(const int* p) {
    return __scoped_atomic_load_n(p, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
}
__device__ __forceinline__ void lds_store_release(int* p, int v) {
    __scoped_atomic_store_n(p, v, __ATOMIC_RELEASE, __MEMORY_SCOPE_WRKGRP);
}
__device__ __forceinline__ void lds_store_relaxed(int* p, int v) {
    __scoped_atomic_store_n(p, v, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
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

// trace (tuning aid): [wavefront][step of the traced tile, < 128][4 stamps], then 64 extra words
constexpr int TRACE_STEPS = 128, TRACE_WORDS = 16 * TRACE_STEPS * 4 + 64;
#define TILE_STAMP(on, wave, j, k)                                                               \
    do { if (tr && (on) && (j) < TRACE_STEPS && c == 0)                                        \
             tr[((size_t)(wave) * TRACE_STEPS + (j)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
constexpr int TRACE_TILE = 2;

// what a step needs to know about its tile (64 columns of one signal of the launch group)
struct TileCtx {
    int tx, sg;              // tile along time, signal of the group
    int colc, nabs, nabs0;   // column of the lane (clamped), its padded index, padded index of column 0
    bool colok;
    int64_t obase;           // element offset of the signal in Wx / dWx / Tx
    int64_t kbase;           // ... in the bin map of the group
};

// the additive term of one point and how it is folded into a cell, in the CPU path's arithmetic:
// float32 data with a float64 weight vector accumulates through double (algos.py:66-79)
template <bool CST64> struct TileTerm {
    using type = float;
    static __device__ __forceinline__ float make(float z, float w) { return z * w; }
    static __device__ __forceinline__ float fold(float o, float t) { return o + t; }
};
template <> struct TileTerm<true> {
    using type = double;
    static __device__ __forceinline__ double make(float z, double w) { return (double)z * w; }
    static __device__ __forceinline__ float fold(float o, double t) { return (float)((double)o + t); }
};
// v_mov_b32_dpp quad_perm: lanes without a source keep `old`
template <int CTRL> __device__ __forceinline__ int tile_dpp(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ float tile_dpp(float old, float v) {
    return __int_as_float(tile_dpp<CTRL>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL> __device__ __forceinline__ double tile_dpp(double old, double v) {
    const int lo = tile_dpp<CTRL>(__double2loint(old), __double2loint(v));
    const int hi = tile_dpp<CTRL>(__double2hiint(old), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ---- the reassignment of one step (4 rows) into the tile, in row order; lane = column. Rows of
// a step that hit the same cell are chained in registers (same lane = same column: no cross-lane
// traffic): the cells are read together, a row that hits the cell of an earlier row of the step
// starts from that row's result, the cells are written back in row order.
template <typename TM>
__device__ __forceinline__ void update4(float2* T, const int (&cell)[TILE_G], const typename TM::type (&vx)[TILE_G],
                                        const typename TM::type (&vy)[TILE_G]) {
    float2 t[TILE_G];
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) t[r] = T[cell[r]];
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) {
#pragma unroll
        for (int q = 0; q < r; ++q) if (cell[q] == cell[r]) t[r] = t[q];
        t[r].x = TM::fold(t[r].x, vx[r]); t[r].y = TM::fold(t[r].y, vy[r]);
    }
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) T[cell[r]] = t[r];
}

// LDS of a workgroup: the tile, then the control words, the flags and the step table
struct TileLds {
    float2* T;          // (na + 1) x 64 cells (the last row: scratch for points that contribute nothing)
    int* next;          // producer ticket counter
    int* upd_done;      // [TILE_NU]: producer steps each updater has consumed
    int* gcnt;          // [TILE_RING / 4]: producer steps finished, per group of 4 consecutive steps
    int4* segtab;       // [nusegs]: first row, rows, kind, producer steps of the tile before the segment
};
__host__ __device__ inline size_t tile_lds_bytes(int64_t na, int nusegs) {
    return (size_t)(na + 1) * TILE_COLS * 8 + 32 + 4 * (TILE_RING / 4) + 16 * (size_t)nusegs;
}

template <int GRID, bool STORE_D, int NW, int CSTK>
__global__ __launch_bounds__(64 * NW) void tile_kernel(TileArgs A, SsqParams sp) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int c = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t N = A.N;
    const unsigned nN = (unsigned)N;
    const int na = (int)A.na, omax = na - 1;
    TileLds L;
    L.T = reinterpret_cast<float2*>(lds_raw);
    L.next = reinterpret_cast<int*>(lds_raw + (size_t)(na + 1) * TILE_COLS * 8);
    L.upd_done = L.next + 4;                                 // 16-byte aligned: read as one int4
    L.gcnt = L.next + 8;
    L.segtab = reinterpret_cast<int4*>(L.gcnt + TILE_RING / 4);
    float2* T = L.T;
    for (int k = wv; k <= na; k += NW) T[k * TILE_COLS + c] = make_float2(0.f, 0.f);
    if (threadIdx.x < 8) L.next[threadIdx.x] = 0;
    for (int k = threadIdx.x; k < TILE_RING / 4; k += 64 * NW) L.gcnt[k] = 0;
    for (int k = threadIdx.x; k < A.nusegs; k += 64 * NW) L.segtab[k] = A.usegs[k];
    __syncthreads();

    // The workgroup is persistent: it walks the tiles blockIdx.x, + gridDim.x, ... of the launch
    // group (tile = 64 columns of one signal); positions advance monotonically, so the
    // divisions are done once, by repeated subtraction
    const int ntx = (int)((N + TILE_COLS - 1) / TILE_COLS);
    const int ntot = ntx * A.nsig;
    const int ntl = ntot > (int)blockIdx.x ? (ntot - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    auto ctx_of = [&](int tx, int sg) {
        TileCtx t;
        t.tx = tx; t.sg = sg;
        const int col0 = tx * TILE_COLS, col = col0 + c;
        t.colok = col < N;
        t.colc = t.colok ? col : (int)N - 1;                 // loads stay in range
        t.nabs = A.n1 + t.colc; t.nabs0 = A.n1 + col0;
        t.obase = (int64_t)(A.sig0 + sg) * na * N;
        t.kbase = (int64_t)sg * na * N;
        return t;
    };
    struct TilePos { int itl, tx, sg; };                     // the workgroup's itl-th tile
    TilePos pos0;
    pos0.itl = 0; pos0.sg = (int)blockIdx.x / ntx; pos0.tx = (int)blockIdx.x - pos0.sg * ntx;
    auto next_tile = [&](TilePos& q) {
        ++q.itl; q.tx += (int)gridDim.x;
        while (q.tx >= ntx) { q.tx -= ntx; ++q.sg; }
    };
    const int nps = A.npsteps;
    unsigned long long* tr = (A.trace && (int)blockIdx.x == (100 < (int)gridDim.x ? 100 : (int)gridDim.x - 1)) ? A.trace : nullptr;
    unsigned short* ring = A.ring + (size_t)blockIdx.x * TILE_RING * TILE_G * TILE_COLS;

    if (wv < TILE_NU) {
        // ------------------------------------------------------------------ the updaters
        // One wavefront, lane = column, walks ALL rows of the tile in ascending order. Its
        // instruction stream is the serial part of the tile, so it is kept lean: a cursor over
        // (tile, segment, row) advanced with a few scalar operations, row addresses as a uniform
        // pointer + the lane's column, no masks (a point that contributes nothing goes to a scratch
        // row of the tile; lanes past the last column work on LDS columns that are never written
        // out), one look at the producers' progress per 4 steps, several steps of loads in flight.
        int* my_done = L.upd_done;
        const int nsg = A.nusegs;
        struct USlot {
            float2 W[TILE_G]; unsigned short kb[TILE_G];      // (bins stay as loaded: a conversion here would wait for the load)
            float cf[TILE_G]; double cd[TILE_G];
            int meta;                                         // wave-uniform: 1 = interpolated rows, 2 = last step of the tile, rows << 2
        };
        const float* cstf = (const float*)A.cst;
        const double* cstd = (const double*)A.cst;
        constexpr int cstk = CSTK;
        auto lane_col = [&](int tx, bool& ok) {               // the lane's column in a tile
            const int col = tx * TILE_COLS + c;
            ok = col < N;
            return ok ? col : (int)N - 1;
        };
        // ---- load side
        TilePos lp = pos0;                                    // tile of the next load
        bool l_ok;
        const unsigned N8 = nN * 8u, maxoff8 = (unsigned)(na - 1) * N8;
        unsigned l_col8 = (unsigned)lane_col(lp.tx, l_ok) * 8u;
        const char* l_Wx8 = reinterpret_cast<const char*>(A.Wx + (int64_t)(A.sig0 + lp.sg) * na * N);
        const char* l_kx8 = reinterpret_cast<const char*>(A.kidx + (int64_t)lp.sg * na * N);
        int l_seg = 0, l_row, l_left, l_kind, l_p;
        {
            const int4 sg0 = L.segtab[0];
            l_row = __builtin_amdgcn_readfirstlane(sg0.x); l_left = __builtin_amdgcn_readfirstlane(sg0.y);
            l_kind = __builtin_amdgcn_readfirstlane(sg0.z); l_p = __builtin_amdgcn_readfirstlane(sg0.w);
        }
        int l_waited = -1;                                    // last group of producer steps known complete
        // Loads are issued unconditionally (past the last step: a repeat of the last addresses, marked
        // empty): the compiler counts the loads in flight per path, and a path that skips some makes
        // every wait a full drain.
        auto uload = [&](USlot& s) {
            const bool real = lp.itl < ntl;
            const int nvalid = real ? (l_left < TILE_G ? l_left : TILE_G) : 0;
            const bool interp = real && l_kind != 0;
            if (!(SSQ_TILE_EXP & 16) && interp && (l_p >> 2) > l_waited) {
                // producer steps are waited for by groups of four (one counter per group)
                const int g = l_p >> 2;
                const int* f = L.gcnt + (g & (TILE_RING / 4 - 1));
                const int want = 4 * ((g >> 5) + 1);          // (TILE_RING / 4 = 32 groups in the ring)
                while (lds_load_acquire(f) < want) __builtin_amdgcn_s_sleep(1);
                l_waited = g;
            }
            s.meta = (interp ? 1 : 0) | ((real && l_left <= TILE_G && l_seg == nsg - 1) ? 2 : 0) | (nvalid << 2);
            // Addresses: a wave-uniform 64-bit base per tile + a 32-bit byte offset (na * N * 8 < 2^32),
            // = (row * N + column) * 8, formed with one scalar and one vector addition per row. Rows
            // past the end of a short step are clamped to the last row of the transform (their
            // points go to the scratch row).
            const unsigned base8 = (unsigned)l_row * N8;
            const char* ringp = reinterpret_cast<const char*>(ring) + (size_t)(l_p & (TILE_RING - 1)) * (TILE_G * TILE_COLS * 2);
            unsigned vo[TILE_G];
#pragma unroll
            for (int r = 0; r < TILE_G; ++r) {
                const unsigned o8 = min(base8 + (unsigned)r * N8, maxoff8);       // wave-uniform
                vo[r] = o8 + l_col8;
                s.W[r] = *reinterpret_cast<const float2*>(l_Wx8 + vo[r]);
                if (cstk == 1) s.cf[r] = cstf[min(l_row + r, na - 1)];
                if (cstk == 2) s.cd[r] = cstd[min(l_row + r, na - 1)];
            }
            // bins: from the ring (interpolated rows; the producers fill all four rows of a slot) or
            // from the bin map (rows read back)
            if (interp) {
#pragma unroll
                for (int r = 0; r < TILE_G; ++r)
                    s.kb[r] = *reinterpret_cast<const unsigned short*>(ringp + (unsigned)(c * 2 + r * TILE_COLS * 2));
            } else {
#pragma unroll
                for (int r = 0; r < TILE_G; ++r)
                    s.kb[r] = *reinterpret_cast<const unsigned short*>(l_kx8 + (vo[r] >> 2));
            }
            if (real) {                                       // advance the cursor
                l_row += TILE_G; l_left -= TILE_G; l_p += l_kind;
                if (l_left <= 0) {
                    if (++l_seg == nsg) {
                        l_seg = 0; next_tile(lp);
                        if (lp.itl < ntl) {
                            l_col8 = (unsigned)lane_col(lp.tx, l_ok) * 8u;
                            l_Wx8 = reinterpret_cast<const char*>(A.Wx + (int64_t)(A.sig0 + lp.sg) * na * N);
                            l_kx8 = reinterpret_cast<const char*>(A.kidx + (int64_t)lp.sg * na * N);
                        }
                    }
                    const int4 sg = L.segtab[l_seg];
                    l_row = __builtin_amdgcn_readfirstlane(sg.x); l_left = __builtin_amdgcn_readfirstlane(sg.y);
                    l_kind = __builtin_amdgcn_readfirstlane(sg.z);
                    l_p = lp.itl * nps + __builtin_amdgcn_readfirstlane(sg.w);
                }
            }
        };
        // ---- update side
        TilePos up = pos0;                                    // the tile being reassigned
        int done = 0;                                         // producer steps consumed
        // the finished tile goes to Tx and is cleared
        auto write_out = [&]() {
            bool ok; const unsigned col8 = (unsigned)lane_col(up.tx, ok) * 8u;
            char* Tx8 = reinterpret_cast<char*>(A.Tx + (int64_t)(A.sig0 + up.sg) * na * N);
            for (int k0 = 0; k0 < na; k0 += 8) {              // 8 bins in flight
                float2 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = T[(k0 + q < na ? k0 + q : na) * TILE_COLS + c];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k0 + q < na) {
                        T[(k0 + q) * TILE_COLS + c] = make_float2(0.f, 0.f);
                        if (ok && !(SSQ_TILE_EXP & 64)) *reinterpret_cast<float2*>(Tx8 + ((unsigned)(k0 + q) * N8 + col8)) = v[q];
                    }
            }
            if (A.counters && c == 0)
                __scoped_atomic_fetch_add(A.counters, 1ull, __ATOMIC_RELAXED, __MEMORY_SCOPE_DEVICE);
            next_tile(up);
        };
        using TM = TileTerm<CSTK == 2>;
        using term_t = typename TM::type;
        auto uprocess = [&](const USlot& s) {
            if (!(SSQ_TILE_EXP & 128)) {
                const int nvalid = s.meta >> 2;
                int cell[TILE_G]; term_t vx[TILE_G], vy[TILE_G];
#pragma unroll
                for (int r = 0; r < TILE_G; ++r) {
                    // (the bin map's "no contribution" mark 0xFFFF and the rows a short step repeats
                    // go to the scratch row)
                    const int kb = s.kb[r];
                    const int bin = r < nvalid ? min(kb, na) : na;
                    cell[r] = bin * TILE_COLS + c;
                    if (cstk == 2) { vx[r] = TM::make(s.W[r].x, s.cd[r]); vy[r] = TM::make(s.W[r].y, s.cd[r]); }
                    else { const float cs = cstk == 1 ? s.cf[r] : A.cst0; vx[r] = TM::make(s.W[r].x, cs); vy[r] = TM::make(s.W[r].y, cs); }
                }
                if (!(SSQ_TILE_EXP & 8)) update4<TM>(T, cell, vx, vy);
            }
            // (the ring slot of the step is free again: its bins were loaded long ago)
            if (s.meta & 1) { ++done; if (c == 0) lds_store_relaxed(my_done, done); }
            if (s.meta & 2) write_out();
        };
        const int total = A.nsteps * ntl;
        USlot sl[TILE_D];
        if (!(SSQ_TILE_EXP & 2)) __builtin_amdgcn_s_setprio(3);   // the serial part of the tile goes first
#pragma unroll
        for (int k = 0; k < TILE_D; ++k) uload(sl[k]);
        for (int g0 = 0; g0 < total; g0 += TILE_D) {
#pragma unroll
            for (int k = 0; k < TILE_D; ++k) { uprocess(sl[k]); uload(sl[k]); }
        }
        return;
    }

    // ---------------------------------------------------------------------- producers
    const int ptotal = nps * ntl;
    const float g2 = (float)(A.gamma * A.gamma);
    const float m2hi = g2 * 1.000004f, m2lo = g2 * 0.999996f;
    const int fx = sp.flipud ? -1 : 0, fa = sp.flipud ? na : 0;
    // a ticket = the next producer step of the workgroup, with the tile it belongs to
    struct Ticket { int p, ps, tx, sg; };                    // ps: index among the tile's producer steps
    TilePos gp = pos0;                                        // tile of the last ticket taken
    auto grab = [&]() {
        int p = 0;
        if (c == 0) p = __scoped_atomic_fetch_add(L.next, 1, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
        Ticket t;
        // (lane 0's value for everybody; as a bpermute so that the CPU emulation of the kernels,
        // where readfirstlane is the identity, sees a real broadcast)
        t.p = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ds_bpermute(0, p));
        // (a ticket past the end repeats the position of the last one: its loads are issued all
        // the same -- valid addresses, results unused -- so that the number of loads in flight
        // does not depend on the path taken)
        const bool real = t.p < ptotal;
        // (the last group of four steps is completed by whoever draws the tickets that do not exist)
        if (!real && t.p < ((ptotal + 3) & ~3) && c == 0)
            __scoped_atomic_fetch_add(&L.gcnt[(t.p >> 2) & (TILE_RING / 4 - 1)], 1, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
        if (real) while (t.p >= (gp.itl + 1) * nps) next_tile(gp);
        t.ps = real ? t.p - gp.itl * nps : 0; t.tx = gp.tx; t.sg = gp.sg;
        return t;
    };

    // Step and row records are the same for every lane. They are fetched with vector loads from a
    // lane-independent address (one request per wavefront) rather than scalar loads: scalar and
    // LDS operations share one counter (lgkmcnt) and scalar loads return out of order, so a
    // scalar load in flight turns every wait for a ds_bpermute result into a full drain.
    int vz = 0;
    SSQ_OPAQUE_V(vz);
    const int4* rows4 = reinterpret_cast<const int4*>(A.prows) + vz;
    const int4* steps4 = reinterpret_cast<const int4*>(A.psegs) + vz;

    // Software pipeline over this wavefront's tickets: the records of a step are fetched while
    // the step before it is computed, its samples half a step ahead.
    int4 sa, sb, rec[TILE_G];                 // next step: (kind, first, nsteps, lgR | wtab, stride, L-1, base), rows
    auto load_rec = [&](const Ticket& k) {
        const int g = k.ps;
        sa = steps4[2 * g]; sb = steps4[2 * g + 1];
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) rec[r] = rows4[g * TILE_G + r];
    };
    float2 xu[2][TILE_G];
    int xr[2][TILE_G], xkc[2][TILE_G];
    int xbaddr[2], xwoff[2], xmask[2];
    auto load = [&](auto BB, const Ticket& k) {              // samples of the step whose records are in (sa, sb, rec)
        constexpr int b = decltype(BB)::value;
        const TileCtx t = ctx_of(k.tx, k.sg);
        const int lgR = sa.w;
        xwoff[b] = sb.x; xmask[b] = (1 << lgR) - 1;
        const int q0 = t.nabs >> lgR, qb = (t.nabs0 >> lgR) - (TILE_W / 2 - 1);
        // the sample this lane holds (lanes past the widest window any lane needs repeat the last one)
        const int wlast = (63 >> lgR) + TILE_W;
        const unsigned uidx = (unsigned)((qb + (c < wlast ? c : wlast)) & sb.z);
        xbaddr[b] = (q0 - (TILE_W / 2 - 1) - qb) * 4;            // lane that holds tap 0
        const float2* Ub = A.U + sb.w + (int64_t)t.sg * sb.y;
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) {
            const int4 d = rec[r];
            xr[b][r] = d.x; xkc[b][r] = d.z;
            xu[b][r] = Ub[(unsigned)d.y + uidx];
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    Ticket kc_ = grab();                      // the step computed
    if (kc_.p >= ptotal) return;
    load_rec(kc_); load(B0{}, kc_);
    Ticket kn = grab();                       // the step whose samples are loaded next
    load_rec(kn);
    // (phi_t, phi'_t / (R dt)) of the step in hand: fetched for the next step when the last taps of
    // this one are done -- unconditionally (64 bytes per lane from an L1 / L2-resident table): a
    // load under a condition would make the compiler drain all loads at the join
    ssq_f2 wt[TILE_W];
    auto load_wt = [&](auto BB, const Ticket& k) {
        constexpr int b = decltype(BB)::value;
        const TileCtx t = ctx_of(k.tx, k.sg);
        const float4* wp = A.wtab + (int64_t)(xwoff[b] + (t.nabs & xmask[b])) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = wp[q];
            wt[2 * q].x = v.x; wt[2 * q].y = v.y; wt[2 * q + 1].x = v.z; wt[2 * q + 1].y = v.w;
        }
    };
    load_wt(B0{}, kc_);
    auto step = [&](auto BB, auto BN) {
        constexpr int b = decltype(BB)::value;
        const TileCtx tc = ctx_of(kc_.tx, kc_.sg);
        const bool trk = kc_.p / nps == TRACE_TILE && tr;
        const int trj = kc_.ps;
        TILE_STAMP(trk, wv, trj, 0);
        float2* Wx = A.Wx + tc.obase;
        float2* dWx = STORE_D ? A.dWx + tc.obase : nullptr;
        const int baddr = xbaddr[b];
        int kout[TILE_G];
        Ticket knn; knn.p = ptotal; knn.ps = 0; knn.tx = 0; knn.sg = 0;
#pragma unroll
        for (int r = 0; r < ((SSQ_TILE_EXP & 4) ? 0 : TILE_G); ++r) {
            if (r == TILE_G / 2) {
                // the next step: its samples now (its records came in during the first rows), then
                // a ticket and the records of the one after
                load(BN, kn);
                knn = grab();
                load_rec(knn);
                TILE_STAMP(trk, wv, trj, 1);
            }
            // (a, a') = sum_t (phi_t, phi'_t) u[q0 - 3 + t]  (baseband): real and imaginary
            // parts as two packed accumulators (a_re, a'_re), (a_im, a'_im)
            ssq_f2 are2, aim2;
            {
                int fr[TILE_W], fi[TILE_W];
                const int ur = __float_as_int(xu[b][r].x), ui = __float_as_int(xu[b][r].y);
                SSQ_BPERMUTE_OFF(fr[0], baddr, ur, 0);  SSQ_BPERMUTE_OFF(fi[0], baddr, ui, 0);
                SSQ_BPERMUTE_OFF(fr[1], baddr, ur, 4);  SSQ_BPERMUTE_OFF(fi[1], baddr, ui, 4);
                SSQ_BPERMUTE_OFF(fr[2], baddr, ur, 8);  SSQ_BPERMUTE_OFF(fi[2], baddr, ui, 8);
                SSQ_BPERMUTE_OFF(fr[3], baddr, ur, 12); SSQ_BPERMUTE_OFF(fi[3], baddr, ui, 12);
                SSQ_BPERMUTE_OFF(fr[4], baddr, ur, 16); SSQ_BPERMUTE_OFF(fi[4], baddr, ui, 16);
                SSQ_BPERMUTE_OFF(fr[5], baddr, ur, 20); SSQ_BPERMUTE_OFF(fi[5], baddr, ui, 20);
                SSQ_BPERMUTE_OFF(fr[6], baddr, ur, 24); SSQ_BPERMUTE_OFF(fi[6], baddr, ui, 24);
                SSQ_BPERMUTE_OFF(fr[7], baddr, ur, 28); SSQ_BPERMUTE_OFF(fi[7], baddr, ui, 28);
                SSQ_LDS_WAIT();
#pragma unroll
                for (int t = 0; t < TILE_W; ++t) {
                    ssq_f2 sv; sv.x = __int_as_float(fr[t]); sv.y = __int_as_float(fi[t]);
                    if (t == 0) { SSQ_PK_MUL_LO(are2, wt[0], sv); SSQ_PK_MUL_HI(aim2, wt[0], sv); }
                    else { SSQ_PK_FMA_LO(are2, wt[t], sv); SSQ_PK_FMA_HI(aim2, wt[t], sv); }
                }
            }
            if (r == TILE_G - 1) load_wt(BN, kn);
            const float are = are2.x, aim = aim2.x;
            float dre = are2.y, dim = aim2.y;
            // d/dt of e^{i theta n} a(n):  e^{i theta n} (i theta a + a'),  theta = 2 pi kc / (M dt)
            const float theta = (float)xkc[b][r] * A.theta_scale;
            dre = __builtin_fmaf(-theta, aim, dre);
            dim = __builtin_fmaf(theta, are, dim);
            // e^{2 i pi kc n / M}: the phase kc n mod M is exact in integers and in float
            // (M <= 2^24, checked by the host), v_sin_f32 / v_cos_f32 take revolutions (measured on
            // the M = 2^18 circle: max abs error 1.2e-7, as good as a float table)
            const float rev = (float)(__umul24((unsigned)xkc[b][r], (unsigned)tc.nabs) & (unsigned)A.mmask) * A.inv_m;   // (both < 2^24: full-rate multiply)
            const float2 tw = make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev));
            const float2 Wv = cmulf(tw, make_float2(are, aim));
            const float2 Dv = cmulf(tw, make_float2(dre, dim));
            // (rows that only pad a step repeat the previous row -- same address, same value --
            // and lanes past the last column repeat its point; neither contributes below)
            const int row = xr[b][r] & 0xFFFF;
            const bool pad = xr[b][r] < 0;
            const unsigned o = (unsigned)row * nN + (unsigned)tc.colc;
            Wx[o] = Wv;
            if (STORE_D) dWx[o] = Dv;
            // phase transform and bin: as emit_point<LEAN> of the block kernels
            const float cc = Wv.x, dd = Wv.y, aa = Dv.x, bb = Dv.y;
            const float m2 = cc * cc + dd * dd, num = bb * cc - aa * dd;
            const bool above = m2 > m2hi, below = m2 < m2lo;
            const float w32 = fabsf(num * __builtin_amdgcn_rcpf(m2 * 6.2831855f));
            bool ok;
            const int kb = bin_screen_cwt<GRID>(w32, sp, omax, ok);
            const int kf = (kb ^ fx) + fa;
            const bool live = tc.colok && !pad;
            kout[r] = (above && live) ? kf : TILE_NOBIN;
            // undecided by the float32 screens (~0.05 % of the points, one row in 30): the exact
            // double path
            const bool pend = live && !(below | (above & ok));
            if (__builtin_amdgcn_ballot_w64(pend)) {
                if (pend) {
                    const int ke = exact_bin(Wv, Dv, sp, omax, A.gamma);
                    kout[r] = ke >= 0 ? ke : TILE_NOBIN;
                }
            }
        }
        if (SSQ_TILE_EXP & 4) {
            load(BN, kn); knn = grab(); load_rec(knn); load_wt(BN, kn);
#pragma unroll
            for (int r = 0; r < TILE_G; ++r) kout[r] = TILE_NOBIN;
        }
        // hand the step to the updater: bins into the ring slot (free once the updater has
        // consumed step p - TILE_RING), then the flag behind a release (Wx and bins in memory)
        const int pc = kc_.p;
        const int slot = pc & (TILE_RING - 1);
        TILE_STAMP(trk, wv, trj, 2);
        for (;;) {                                            // (the slowest of the updaters counts)
            if (lds_load_relaxed(L.upd_done) > pc - TILE_RING) break;
            // (a long sleep: the ring is two tiles deep, and a dozen wavefronts polling LDS in a
            // tight loop take the LDS and the issue slots from the updaters they are waiting for)
            __builtin_amdgcn_s_sleep(SSQ_TILE_BPSLEEP);
        }
        unsigned short* rb = ring + (size_t)slot * (TILE_G * TILE_COLS) + c;
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) rb[r * TILE_COLS] = (unsigned short)kout[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (c == 0)
            __scoped_atomic_fetch_add(&L.gcnt[(pc >> 2) & (TILE_RING / 4 - 1)], 1, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
        TILE_STAMP(trk, wv, trj, 3);
        kc_ = kn; kn = knn;
    };
    for (;;) {
        if (kc_.p >= ptotal) break;
        step(B0{}, B1{});
        if (kc_.p >= ptotal) break;
        step(B1{}, B0{});
    }
}

// ---------------------------------------------------------------------------- host side
int TilePlan::create(const ssq_cwt_tiles_desc& d, int64_t M_, int64_t N_, int64_t n1_, int64_t na_, int group_,
                     double dt_, int64_t& bytes) {
    M = M_; N = N_; n1 = n1_; na = na_; group = group_; dt = dt_;
    nsegs = d.n_segs; nsteps = d.n_steps; n_irows = d.n_irows; u_total = d.u_total;
    SSQ_REQUIRE(nsegs >= 1 && nsteps >= 1 && n_irows >= 1 && d.n_classes >= 1, "empty tile tables");
    {
        int dev = 0; hipDeviceProp_t pr;
        SSQ_CHECK_HIP(hipGetDevice(&dev));
        SSQ_CHECK_HIP(hipGetDeviceProperties(&pr, dev));
        ncu = pr.multiProcessorCount;
        if (const char* e = getenv("SSQ_TILE_GRID")) if (atoi(e) > 0) ncu = atoi(e);
    }
    SSQ_REQUIRE(tile_lds_bytes(na, nsegs) <= 160 * 1024 && na * N < ((int64_t)1 << 29),
                "na = %lld: the Tx tile exceeds the LDS", (long long)na);
    // the modulation phase kc * n mod M is formed with a 24-bit multiply and carried in a float
    SSQ_REQUIRE(M <= ((int64_t)1 << 24), "the tile path needs a padded length <= 2^24");
    SSQ_REQUIRE((int64_t)group * u_total < ((int64_t)1 << 31), "tile intermediates exceed 2^31 entries");
    auto up = [&](void** dst, const void* src, size_t nbytes) -> int {
        SSQ_CHECK_HIP(hipMalloc(dst, nbytes ? nbytes : 1));
        if (nbytes) SSQ_CHECK_HIP(hipMemcpy(*dst, src, nbytes, hipMemcpyHostToDevice));
        bytes += (int64_t)nbytes;
        return 0;
    };
    int rc;
    static_assert(sizeof(TileSeg) == 32 && sizeof(TileRow) == 16 && sizeof(TileIRow) == 32, "table layout");
    {   // one record per step; `first` of the device copy = the step's index among the producer
        // steps of a tile (-1: rows read back)
        std::vector<TileSeg> hs((size_t)nsteps);
        std::vector<int32_t> hp;
        const TileSeg* sg = reinterpret_cast<const TileSeg*>(d.segs);
        const TileRow* rw = reinterpret_cast<const TileRow*>(d.rows);
        int64_t covered = 0;
        for (int i = 0; i < nsegs; ++i) {
            SSQ_REQUIRE(sg[i].first == covered && sg[i].nsteps >= 1 && sg[i].first + sg[i].nsteps <= nsteps,
                        "tile segment %d does not continue the step list", i);
            SSQ_REQUIRE(sg[i].kind == 0 || sg[i].kind == 1, "tile segment %d: bad kind", i);
            for (int t = 0; t < sg[i].nsteps; ++t) {
                const size_t st = (size_t)sg[i].first + t;
                hs[st] = sg[i];
                hs[st].first = sg[i].kind == 1 ? (int32_t)hp.size() : -1;
                if (sg[i].kind == 1) hp.push_back((int32_t)st);
                // the kernel derives the rows of a step from the first one
                const int32_t r0 = rw[st * TILE_G].row;
                SSQ_REQUIRE(r0 >= 0 && r0 < na, "tile step %zu: bad first row", st);
                for (int r = 1; r < TILE_G; ++r) {
                    const int32_t rr = rw[st * TILE_G + r].row;
                    SSQ_REQUIRE(rr < 0 ? true : (rr == r0 + r && rw[st * TILE_G + r - 1].row >= 0 && rr < na),
                                "tile step %zu: rows are not consecutive", st);
                }
            }
            covered += sg[i].nsteps;
        }
        SSQ_REQUIRE(covered == nsteps, "tile segments cover %lld of %d steps", (long long)covered, nsteps);
        SSQ_REQUIRE(!hp.empty(), "tile tables without interpolated rows");
        npsteps = (int)hp.size();
        if ((rc = up((void**)&steps, hs.data(), sizeof(TileSeg) * nsteps))) return rc;
        // the producers' own dense copies of the records (no indirection in their pipeline)
        std::vector<TileSeg> hps(hp.size());
        std::vector<TileRow> hpr(hp.size() * TILE_G);
        for (size_t i = 0; i < hp.size(); ++i) {
            hps[i] = hs[(size_t)hp[i]];
            for (int r = 0; r < TILE_G; ++r) hpr[i * TILE_G + r] = rw[(size_t)hp[i] * TILE_G + r];
        }
        // the updaters' view of the row list: per segment
        std::vector<int32_t> hu((size_t)nsegs * 4);
        for (int i = 0, pbefore = 0; i < nsegs; ++i) {
            int nrows = 0;
            for (int t = 0; t < sg[i].nsteps; ++t)
                for (int r = 0; r < TILE_G; ++r) nrows += rw[((size_t)sg[i].first + t) * TILE_G + r].row >= 0 ? 1 : 0;
            // (only the last step of a segment may be short: the kernel derives a step's rows from the segment)
            SSQ_REQUIRE(nrows > (sg[i].nsteps - 1) * TILE_G, "tile segment %d: a short step inside the segment", i);
            hu[4 * i] = rw[(size_t)sg[i].first * TILE_G].row; hu[4 * i + 1] = nrows;
            hu[4 * i + 2] = sg[i].kind; hu[4 * i + 3] = pbefore;
            if (sg[i].kind == 1) pbefore += sg[i].nsteps;
        }
        if ((rc = up((void**)&usegs, hu.data(), hu.size() * 4))) return rc;
        if ((rc = up((void**)&psegs, hps.data(), sizeof(TileSeg) * hps.size()))) return rc;
        if ((rc = up((void**)&prows, hpr.data(), sizeof(TileRow) * hpr.size()))) return rc;
    }
    if ((rc = up((void**)&rows, d.rows, sizeof(TileRow) * TILE_G * nsteps))) return rc;
    if ((rc = up(&wtab, d.wtab, (size_t)64 * d.n_phases))) return rc;
    if ((rc = up(&tbank, d.tbank, (size_t)4 * d.n_tbank))) return rc;
    cls.resize(d.n_classes);
    for (int c = 0; c < d.n_classes; ++c) {
        cls[c] = {d.classes[4 * c], d.classes[4 * c + 1], d.classes[4 * c + 2]};
        SSQ_REQUIRE(cls[c].L >= 2 && (cls[c].L & (cls[c].L - 1)) == 0 && cls[c].nrows >= 1, "bad tile class %d", c);
        lmax = std::max(lmax, cls[c].L);
    }
    std::vector<TileIRow> hi((size_t)n_irows);
    for (int r = 0; r < n_irows; ++r) {
        const int64_t* q = d.irows + 8 * r;
        const int c = (int)q[6];
        SSQ_REQUIRE(c >= 0 && c < d.n_classes && q[4] == cls[c].L && q[2] >= 1 && 2 * q[2] <= q[4]
                    && q[1] >= 0 && q[1] + q[2] <= M / 2 + 1 && q[5] >= 0 && q[5] + q[2] <= d.n_tbank,
                    "bad tile row %d", r);
        hi[r] = {(int32_t)q[0], (int32_t)q[1], (int32_t)q[2], (int32_t)q[3], (int32_t)q[4], (int32_t)q[5],
                 (int32_t)((int64_t)group * cls[c].upre + q[7] * cls[c].L), (int32_t)(cls[c].nrows * cls[c].L)};
    }
    if ((rc = up((void**)&irows, hi.data(), sizeof(TileIRow) * n_irows))) return rc;
    SSQ_CHECK_HIP(hipMalloc(&U, (size_t)8 * group * u_total)); bytes += 8 * group * u_total;
    SSQ_CHECK_HIP(hipMemset(U, 0, (size_t)8 * group * u_total));
    {
        const size_t rb = (size_t)ncu * TILE_RING * TILE_G * TILE_COLS * 2;
        SSQ_CHECK_HIP(hipMalloc(&ring, rb)); bytes += (int64_t)rb;
        SSQ_CHECK_HIP(hipMalloc((void**)&counters, 64));
        SSQ_CHECK_HIP(hipMemset(counters, 0, 64));
    }
    for (size_t c = 0; c < cls.size(); ++c) {
        FftPlan fp;
        rc = fp.create(1, SSQ_F32, (size_t)cls[c].L, (size_t)(group * cls[c].nrows), 1.0);
        if (rc) return rc;
        bytes += (int64_t)fp.work_bytes;
        ffts.push_back(fp);
    }
    for (int t = 0; t < 5; ++t) n_items_tile[t] = d.n_items_tile[t];
    if (!getenv("SSQ_TILE_SERIAL")) {
        SSQ_CHECK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        SSQ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        SSQ_CHECK_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    return 0;
}

void TilePlan::destroy() {
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); side = nullptr; }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    ev_fork = ev_join = nullptr;
    for (auto& f : ffts) f.destroy();
    ffts.clear();
    void* ptrs[] = {steps, rows, psegs, prows, usegs, irows, wtab, tbank, U, ring, counters};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    steps = nullptr; rows = nullptr; psegs = nullptr; prows = nullptr; usegs = nullptr; irows = nullptr; wtab = tbank = U = ring = nullptr;
    counters = nullptr;
}

int TilePlan::spectra(int sig, int nsig, const void* xh_all, hipStream_t stream) {
    const dim3 grid((unsigned)std::min<int64_t>((lmax + 255) / 256, 64), (unsigned)n_irows, (unsigned)nsig);
    hipLaunchKernelGGL(tile_spectra_kernel, grid, dim3(256), 0, stream, (const float2*)xh_all, M / 2 + 1, sig,
                       irows, (const float*)tbank, (float2*)U);
    SSQ_LAUNCH_CHECK();
    for (size_t c = 0; c < cls.size(); ++c) {
        // the planned batch covers `group` signals; slots past nsig hold stale finite data
        int rc = ffts[c].execute((float2*)U + (size_t)group * cls[c].upre, nullptr, stream);
        if (rc) return rc;
    }
    return 0;
}

// wavefronts per workgroup (one of them the updater): 16 by default = 4 per SIMD (128 VGPRs);
// SSQ_TILE_NW = 8 | 12 | 16 selects another build of the kernel (tuning aid)
template <int GRID, bool STORE_D, int NW, int CSTK>
static int launch_tile_c(const TilePlan& P, const TileArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    auto kern = tile_kernel<GRID, STORE_D, NW, CSTK>;
    const size_t lds = tile_lds_bytes(P.na, P.nsegs);
    static bool attr_set = false;            // per instantiation
    if (!attr_set) {
        SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds > 64 * 1024 ? 160 * 1024 : 64 * 1024));
        attr_set = true;
    }
    // persistent workgroups, one per CU (the tile fills the LDS)
    const int64_t ntot = ((P.N + TILE_COLS - 1) / TILE_COLS) * nsig;
    const dim3 grid((unsigned)std::min<int64_t>(ntot, P.ncu));
    TileArgs B = A; B.nsig = nsig;
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, B, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}
template <int GRID, bool STORE_D, int NW>
static int launch_tile_nw(const TilePlan& P, const TileArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    if (A.cstk == 0) return launch_tile_c<GRID, STORE_D, NW, 0>(P, A, sp, nsig, stream);
    if (A.cstk == 1) return launch_tile_c<GRID, STORE_D, NW, 1>(P, A, sp, nsig, stream);
    return launch_tile_c<GRID, STORE_D, NW, 2>(P, A, sp, nsig, stream);
}
template <int GRID, bool STORE_D>
static int launch_tile(const TilePlan& P, const TileArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    static const int nw = [] { const char* e = getenv("SSQ_TILE_NW"); int v = e ? atoi(e) : 16; return v == 8 || v == 12 ? v : 16; }();
    if (nw == 8) return launch_tile_nw<GRID, STORE_D, 8>(P, A, sp, nsig, stream);
    if (nw == 12) return launch_tile_nw<GRID, STORE_D, 12>(P, A, sp, nsig, stream);
    return launch_tile_nw<GRID, STORE_D, 16>(P, A, sp, nsig, stream);
}

int TilePlan::run(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                  const void* cst, float cst0, const SsqParams& sp, hipStream_t stream) {
    TileArgs A;
    A.steps = steps; A.rows = rows; A.psegs = psegs; A.prows = prows;
    A.usegs = (const int4*)usegs; A.nusegs = nsegs;
    A.wtab = (const float4*)wtab; A.U = (const float2*)U; A.cst = cst;
    A.Wx = (float2*)Wx; A.dWx = (float2*)dWx; A.Tx = (float2*)Tx; A.kidx = kidx;
    A.ring = (unsigned short*)ring;
    A.N = N; A.na = na; A.nsteps = nsteps; A.npsteps = npsteps; A.n1 = (int)n1; A.mmask = (int)(M - 1);
    A.sig0 = sig; A.nsig = nsig; A.inv_m = 1.0f / (float)M;
    A.cstk = sp.cst_f64 ? 2 : (sp.cst_uniform ? 0 : 1);
    A.theta_scale = (float)(6.283185307179586 / ((double)M * dt)); A.cst0 = cst0;
    A.counters = counters;
    A.gamma = sp.gamma;
    static unsigned long long* trace_buf = nullptr;
    const char* trace_path = getenv("SSQ_TILE_TRACE");
    if (trace_path && !trace_buf) SSQ_CHECK_HIP(hipMalloc((void**)&trace_buf, 8 * TRACE_WORDS));
    if (trace_buf) SSQ_CHECK_HIP(hipMemsetAsync(trace_buf, 0, 8 * TRACE_WORDS, stream));
    A.trace = trace_buf;
    auto dump_trace = [&]() -> int {
        if (!trace_buf) return 0;
        SSQ_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h(TRACE_WORDS);
        SSQ_CHECK_HIP(hipMemcpy(h.data(), trace_buf, 8 * h.size(), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(trace_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        return 0;
    };
#define TILE_LAUNCH(G)                                                                      \
    { int rc_ = dWx ? launch_tile<G, true>(*this, A, sp, nsig, stream)                     \
                    : launch_tile<G, false>(*this, A, sp, nsig, stream);                   \
      return rc_ ? rc_ : dump_trace(); }
    if (sp.grid == SSQ_GRID_LOG) { TILE_LAUNCH(SSQ_GRID_LOG) }
    if (sp.grid == SSQ_GRID_LOG_PIECEWISE) { TILE_LAUNCH(SSQ_GRID_LOG_PIECEWISE) }
    TILE_LAUNCH(SSQ_GRID_LIN)
#undef TILE_LAUNCH
}

int64_t TilePlan::tiles_done(hipStream_t stream) {
    unsigned long long v = 0;
    if (!counters) return 0;
    if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, counters, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

}  // namespace ssq
