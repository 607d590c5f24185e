// ssq_tile_pair.hip -- tile3_kernel: the column-tile kernel of the fused ssq_cwt form with TWO neighbouring columns
// per lane (float32 data, float64 Tx tile in LDS, unordered ds_add_f64; gfx950). Same work and same results as
// tile2_kernel (ssq_tile_f64.hip): interpolates Wx (and dWx) of most rows from decimated baseband samples
// (ssq_tile_fft.hip; math and planning: ssqueezepy_amd/_tiles.py), reads the other rows' Wx + bin back, reassigns all
// of them -- replaces the reference's cwt + phase_cwt + ssqueeze loop nests for these rows
// (ssqueezepy/_cwt.py:167-177, algos.py:859-953).
//
// Why another geometry (round 6). tile2_kernel's counters (profiles/r5z_pmc_summary.txt): as many scalar as vector
// instructions, 43 % of a wavefront's life parked at a wait, every pipe half busy -- four wavefronts per SIMD, each ONE
// chain of dependent instructions per 64 points; round 5's trims of that chain all lost. Here an item is FOUR
// consecutive rows of a class x the tile's 32 columns and a lane owns the column pair (2 cp, 2 cp + 1) of sub-row h
// (lane = 16 h + cp):
//   * the two columns share their eight samples (n and n + 1 lie in the same decimation interval for every R >= 2 when
//     n is even): ONE gather of 16 ds_bpermute serves 128 points instead of 64;
//   * the item's scalar work (record, cursors, address bases, priority toggle: ~73 instructions in tile2_kernel) is
//     paid once per 128 points;
//   * the two columns' chains (taps -> modulation -> phase transform -> bin -> ds_add_f64) are independent: the SIMD
//     has two instructions to choose from per wavefront instead of one;
//   * Wx leaves as one 16-byte store per lane, the tile's write-out as 16 bytes per lane too.
// A lane's weights now depend on two column phases: 32 registers per class. The kernel keeps the class in hand only and
// re-reads 512 bytes per lane at a class change (16 wavefronts at 123 registers); the host deals the items so that few
// wavefronts hold more than one class (ssq_cwt_tiles.hip). Measured beside it and dropped: 12 wavefronts at 168
// registers with the two classes of a list resident -- the same time (profiles/r6_ab_history.txt).
//
// The Tx tile: row k = 512 bytes = [re of the even columns | re of the odd ones | im even | im odd], 16 doubles each:
// the 16 lanes of a sub-row add at consecutive 8-byte addresses whatever rows their points go to (a row's stride is a
// multiple of the bank span), both for a lane's first and for its second column.
//
// A pair never straddles a decimation interval because its first padded index is even: when the left padding n1 is odd
// the tiles start one column early (column -1 dead). Wx / Tx pairs and bin pairs are accessed as 16- / 4-byte words at
// whatever 8- / 2-byte boundary the signal's length and that shift leave them. Compiled with -ffp-contract=off (bin indices); explicit fmaf where a multiply-add may fuse.
#include "ssq_common.h"
#include "ssq_tiles.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace ssq {

#include "ssq_point_math.inl"
#include "ssq_tile_dev.h"

constexpr int T3_COLS = 32, T3_RPI = 4, T3_LGC = 5;
typedef float ssq_f4u __attribute__((ext_vector_type(4), aligned(8)));     // 16 bytes at an 8-byte boundary (samples; Wx, Tx
                                                                            // pairs at an odd column)
typedef unsigned ssq_u32u __attribute__((aligned(2)));                      // two 16-bit bins at a 2-byte boundary
// Wx and Tx are written once and never read here: nontemporal stores (measured, one box: tile stage 179.2-179.5 us with
// plain stores, Wx alone 180.9, Tx alone 175.7, both 172.0-176.1 -- profiles/r6_ab_history.txt r6n; the L2 keeps the
// sample windows and the rows read back instead)
// (not a template: deduction would drop the pointer type's 8-byte alignment and let the compiler assume 16)
__device__ __forceinline__ void t3_store(ssq_f4u* p, const ssq_f4u v) { __builtin_nontemporal_store(v, p); }

struct Tile3Args {
    const int* items;        // [n_items][8]: row0 | npad << 9 | kind << 12 | lgR << 13 | weights' offset << 18, samples'
                             // offset of sub-row 0 (class + row), row0 * N * 8, entries between two signals' rows of
                             // the class (these four: scalar loads), kc of the four sub-rows (read per lane)
    const int4* waves;       // [NW]: first item, end, first item of the wavefront's second class (= end: none), 0
    const float4* wtab; const float2* U;
    const void* cst;
    float2* Wx; float2* dWx; float2* Tx; const unsigned short* kidx;
    unsigned short* kdump;   // STORE_K builds: the bin of every point as it is consumed, (signal, row, column); else null
    int64_t N, na;
    int n_items, n1, mmask, lgM, sig0, nsig, group;
    int carry;               // the walk b, b + G, ... runs through the signals' boundaries
    int xcd;                 // first tiles permuted per XCD
    float inv_m, theta_scale, cst0;
    unsigned long long* counters;
    double gamma;
};

// the bin of one point from (Wx, dWx) = (W, V): the float32 screen of the block kernels' lean epilogue; `pend`: the
// screens could not decide (exact double sequence, rare)
template <int GRID>
__device__ __forceinline__ int pair_bin(const ssq_f2 W, const ssq_f2 V, bool live, float m2hi, float m2lo,
                                        const SsqParams& sp, int omax, int fx, int fa, bool& pend) {
    const float cc = W.x, dd = W.y, aa = V.x, bb = V.y;
    const float m2 = cc * cc + dd * dd, num = bb * cc - aa * dd;
    const bool above = m2 > m2hi, below = m2 < m2lo;
    const float w32 = fabsf(num * __builtin_amdgcn_rcpf(m2 * 6.2831855f));
    bool ok;
    const int kb = bin_screen_cwt<GRID>(w32, sp, omax, ok);
    const int kf = (kb ^ fx) + fa;
    pend = live && !(below | (above & ok));
    return (above && live) ? kf : -1;
}

__device__ __forceinline__ ssq_f4u as_f4u(const float4 v) { ssq_f4u r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }

template <int GRID, bool STORE_D, int CSTK, bool STORE_K = false>
__global__ __launch_bounds__(64 * TILE3_NW) void tile3_kernel(Tile3Args A, SsqParams sp) {
    constexpr int NW = TILE3_NW;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int COLS = T3_COLS, RPI = T3_RPI, LGC = T3_LGC;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = lane & 15, h = lane >> 4, hb4 = (lane & 48) * 4;
    const int64_t N = A.N;
    const unsigned nN = (unsigned)N;
    const int na = (int)A.na, omax = na - 1;
    {   // the tile, (na + 1) rows of 512 bytes (the last: scratch): cleared
        double2* T = reinterpret_cast<double2*>(lds_raw);
        for (int k = threadIdx.x; k < (na + 1) * COLS; k += 64 * NW) T[k] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    // (LDS byte addresses of the lane's column pair in row 0 and in the scratch row; + 128: the odd column, + 256: im)
    const int c8 = cp * 8 + (int)SSQ_LDS_ADDR(lds_raw);
    const int scratch8 = na * 512 + c8;
    constexpr int RR = NW * RPI;                               // rows per write-out round
    const int full_rounds = na / RR;

    // A pair must start at an even padded index (then n and n + 1 share their decimation interval): when the left
    // padding n1 is odd the tiles start one column early -- tile t = columns 32 t - sh .. 32 t - sh + 31, column -1 dead
    const int sh = A.n1 & 1, n1e = A.n1 - sh;
    const int ntx = (int)((N + sh + COLS - 1) / COLS);
    const int G = (int)gridDim.x;
    // (the walk over the tiles, the XCD permutation of the first tiles and the carry through the signals' boundaries:
    // as tile2_kernel, see there)
    const int bid = (A.xcd && (G & 7) == 0) ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int per_sig = bid < ntx ? (ntx - bid + G - 1) / G : 0;
    const int ntl = A.carry ? (int)(((int64_t)A.nsig * ntx - bid + G - 1) / G) : per_sig * A.nsig;
    const auto* waves = SSQ_CONST_PTR(int4, A.waves);
    const int i0 = waves[wv].x, i1 = waves[wv].y, isp = waves[wv].z, ni = i1 - i0;
    // (no issue priorities: tile2_kernel's two levels swapped with every item measured the same here as none -- 173.0
    // against 172.8 us -- and a static priority by age rank, youngest highest, worse: 186; profiles/r6_ab_history.txt r6t)
    const float g2 = (float)(A.gamma * A.gamma);
    const float m2hi = g2 * 1.000004f, m2lo = g2 * 0.999996f;
    const int fx = sp.flipud ? -1 : 0, fa = sp.flipud ? na : 0;
    using TM = TileTerm<CSTK == 2>;
    using w_t = typename TM::wtype;
    const w_t* const cstv = reinterpret_cast<const w_t*>(A.cst);

    // ---- a tile's end: all terms in (barrier), every wavefront writes its share of the rows to Tx and clears them,
    // tile free again (barrier). A lane takes its column pair of a row: four LDS reads, one 16-byte store; a wavefront
    // instruction = 4 rows.
    auto finish_tile = [&](int tx, int sg) {
        SSQ_WG_BARRIER();
        float2* Tx = A.Tx + (int64_t)(A.sig0 + sg) * na * N;
        constexpr int NA_CAP = 320;
        constexpr int ROUNDS = (NA_CAP + RR - 1) / RR;
        const int k0 = wv * RPI + h;                           // the lane's row in round 0
        auto take = [&](int k) {                               // the lane's pair of row k: read, rounded, and cleared
            double* p = reinterpret_cast<double*>(lds_raw + (size_t)k * 512 + (size_t)cp * 8);
            const double r0 = p[0], r1 = p[16], q0 = p[32], q1 = p[48];
            p[0] = 0.0; p[16] = 0.0; p[32] = 0.0; p[48] = 0.0;
            float4 v;
            v.x = (float)r0; v.y = (float)q0; v.z = (float)r1; v.w = (float)q1;
            return v;
        };
        const int tcol0 = tx * COLS - sh;                      // the tile's first column
        if (tcol0 >= 0 && tcol0 + COLS <= (int)nN) {
            // every column of the tile exists: the rounds below the last need no masks
            char* tb = reinterpret_cast<char*>(Tx) + (int64_t)tcol0 * 8;
            const unsigned voff = ((unsigned)k0 * nN + (unsigned)cp * 2u) * 8u;
            int fr = full_rounds;
            size_t step = (size_t)RR * (size_t)N * 8;
#pragma unroll
            for (int m = 0; m < ROUNDS - 1; ++m) {
                SSQ_OPAQUE_S(fr); SSQ_OPAQUE_S(step);          // (re-read as scalars at every use: see tile2_kernel)
                if (m < fr) {                                  // (wave-uniform)
                    const float4 v = take(k0 + m * RR);
                    t3_store(reinterpret_cast<ssq_f4u*>(tb + (size_t)voff), as_f4u(v));
                    tb += step;
                    asm volatile("" ::: "memory");             // (keeps the rounds from being batched into registers)
                }
            }
            {   // the last round: the rows left, and the scratch row cleared by the lanes past them
                const int k = k0 + fr * RR;
                const float4 v = take(k < na ? k : na);
                if (k < na) t3_store(reinterpret_cast<ssq_f4u*>(tb + (size_t)voff), as_f4u(v));
            }
        } else {
            // a signal's first tile when n1 is odd, its last when it is partial: column by column
            const int col = tcol0 + cp * 2;
            const bool ok0 = (unsigned)col < nN, ok1 = (unsigned)(col + 1) < nN;
#pragma unroll 1
            for (int m = 0; m * RR < na + 1; ++m) {
                const int k = k0 + m * RR;
                const float4 v = take(k < na ? k : na);
                if (ok0 && k < na) Tx[(int64_t)k * N + col] = make_float2(v.x, v.y);
                if (ok1 && k < na) Tx[(int64_t)k * N + col + 1] = make_float2(v.z, v.w);
            }
        }
        if (threadIdx.x == 0 && A.counters)
            __scoped_atomic_fetch_add(A.counters, 1ull, __ATOMIC_RELAXED, __MEMORY_SCOPE_DEVICE);
        SSQ_WG_BARRIER();
    };

    if (ni <= 0) {                                             // more wavefronts than items: write-outs only
        int tx = bid, sg = 0;
        for (int j = 0; j < ntl; ++j) {
            finish_tile(tx, sg);
            tx += G;
            if (tx >= ntx) { tx = A.carry ? tx - ntx : bid; ++sg; }
        }
        return;
    }

    // ---- the wavefront's sequence of (tile, item) positions: two cursors (the loads' two positions ahead of the
    // arithmetic's), each an item index and the tile as the kernel uses it -- see tile2_kernel
    struct Pos { int nabs0, sg; int64_t off8; };
    const int nabs_step = G * COLS, nabs_first = n1e + bid * COLS, nabs_last = n1e + (ntx - 1) * COLS;
    const int64_t off8_step = (int64_t)G * COLS * 8;
    const int64_t off8_wrap = A.carry ? ((int64_t)na * N + (int64_t)(G - ntx) * COLS) * 8
                                      : ((int64_t)na * N - (int64_t)(per_sig - 1) * G * COLS) * 8;
    const int nabs_back = ntx * COLS;
    auto next_tile = [&](Pos q) {
        Pos r = q;
        r.nabs0 += nabs_step;
        const bool wrap = r.nabs0 > nabs_last;
        r.off8 += wrap ? off8_wrap : off8_step;
        if (wrap) { r.nabs0 = A.carry ? r.nabs0 - nabs_back : nabs_first; ++r.sg; }
        return (wrap && r.sg >= A.nsig) ? q : r;               // (the tile after the last: the last)
    };
    // a tile with dead columns: a signal's first when the tiles start a column early, its last when it is partial
    auto edge_tile = [&](int nabs0) { return (sh != 0 && nabs0 == n1e) || nabs0 - n1e - sh + COLS > (int)N; };
    const int total = ntl * ni;                                // positions of this wavefront
    // (a record's first four words through the scalar cache; its last four -- the sub-rows' centre bins -- are read per lane
    // with the position's data, see load_data)
    typedef int int8v __attribute__((ext_vector_type(4)));
    struct alignas(32) Rec { int8v lo, hi; };
    const auto* recs = SSQ_CONST_PTR(Rec, A.items);
    auto items_at = [&](int it) { return recs[it].lo; };
    // per-lane constants of the addresses: the lane's place inside an item's rows
    const unsigned lane_row16 = ((unsigned)h * nN + (unsigned)cp * 2u) * 8u;   // bytes: sub-row h, first column of the pair
    const unsigned lane_col16 = (unsigned)cp * 16u;

    // data of a position: (interpolated) the lane's sample of its sub-row's window (.xy; .zw: the sample behind it,
    // unused), or (rows read back) Wx of the lane's two points and their bins (two 16-bit words)
    struct Data { ssq_f4u u; int kq; };
    const char* const U8 = reinterpret_cast<const char*>(A.U);
    const char* const WX8 = reinterpret_cast<const char*>(A.Wx) + (size_t)((int64_t)A.sig0 * na * N) * 8u;
    const char* const KX8 = reinterpret_cast<const char*>(A.kidx);
    const unsigned lane_h4 = (unsigned)h * 4u;
    auto load_data = [&](const int8v R, int it, const Pos& q) {
        Data d;
        const int w0 = R[0];
        const int kind = (w0 >> 12) & 1;
        const char* base; unsigned voff;
        // (interpolated rows: the lane's second load fetches the centre bin of its sub-row -- words 4 .. 7 of the item's record)
        const char* kbase = reinterpret_cast<const char*>(A.items) + (size_t)(unsigned)it * 32u + 16u; unsigned koff = lane_h4;
        if (kind) {                                            // (wave-uniform; the loads themselves stay outside)
            // sample (qb + cp) mod L of row h of the item, h * L entries on: the 16 lanes of a sub-row hold the
            // window every column of the tile takes its eight taps from ((31 >> lgR) + 8 <= 15 for R >= 4)
            const int lgR = (w0 >> 13) & 31;
            const int qb = (q.nabs0 >> lgR) - (TILE_W / 2 - 1);
            const int lmask = A.mmask >> lgR;                  // L - 1, L = M / R
            voff = (((unsigned)((qb + cp) & lmask)) + ((unsigned)h << (A.lgM - lgR))) * 8u;
            base = U8 + ((size_t)(unsigned)R[1] + (size_t)((unsigned)q.sg * (unsigned)R[3])) * 8u;
        } else {
            // points (row0 + h, column pair) -- the last pair's for lanes past it, the last real row's for padded
            // sub-rows -- and their bins
            const int npad = (w0 >> 9) & 7;
            unsigned lr = lane_row16;
            if (npad) lr = (unsigned)min(h, RPI - 1 - npad) * nN * 8u + lane_col16;
            lr += 16u;                                         // (the bases below start 16 bytes early: lr stays >= 0)
            if (edge_tile(q.nabs0)) {
                // pairs with a dead column read the nearest pair inside the row (the body puts the halves in place)
                const int col = q.nabs0 - n1e - sh + cp * 2;
                const int cc = min(max(col, 0), (int)N - 2);
                lr += (unsigned)((cc - col) * 8);
            }
            voff = lr;
            base = WX8 + ((int64_t)q.off8 + (unsigned)R[2] - 16);
            kbase = KX8 + (((int64_t)q.off8 + (unsigned)R[2] - 16) >> 2);
            koff = lr >> 2;
        }
        d.u = *reinterpret_cast<const ssq_f4u*>(base + (size_t)voff);
        // (one load either way: the bins of rows read back, the centre bin of an interpolated sub-row)
        d.kq = (int)*reinterpret_cast<const ssq_u32u*>(kbase + (size_t)koff);
        return d;
    };
    // the weights of a class for the lane's two column phases (every tile of this workgroup: the same n mod R)
    auto load_wt = [&](ssq_f2 (&wa)[TILE_W], ssq_f2 (&wb)[TILE_W], int w0) {
        const int lgR = (w0 >> 13) & 31, woff = (int)((unsigned)w0 >> 18);
        const int nabs = n1e + bid * COLS + cp * 2;
        const int R = 1 << lgR;
        const float4* wp = A.wtab + (int64_t)woff * 4 + (nabs & (R - 1));   // (nabs even: nabs + 1 is the next phase)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = wp[t * R], u = wp[t * R + 1];
            wa[2 * t].x = v.x; wa[2 * t].y = v.y; wa[2 * t + 1].x = v.z; wa[2 * t + 1].y = v.w;
            wb[2 * t].x = u.x; wb[2 * t].y = u.y; wb[2 * t + 1].x = u.z; wb[2 * t + 1].y = u.w;
        }
    };
    ssq_f2 wa0[TILE_W], wb0[TILE_W];
    int wcls;                                                  // the class whose weights are resident: w0 >> 13
    {
        int wsel = items_at(i0)[0];                            // (a list that opens with rows read back: its first interpolated item)
        if (!((wsel >> 12) & 1) && isp < i1) wsel = items_at(isp)[0];
        load_wt(wa0, wb0, wsel);
        wcls = wsel >> 13;
    }

    Data D[3];
    Pos tc, tl;                                                // the tile of the arithmetic's cursor, of the loads'
    tc.nabs0 = nabs_first; tc.sg = 0; tc.off8 = ((int64_t)bid * COLS - sh) * 8;
    tl = tc;
    if (total <= 0) return;
    int it_c = i0, it_l = i0;
    bool tc_edge = edge_tile(tc.nabs0);                        // the arithmetic's tile has dead columns
    int left = total;                                          // positions not yet finished
    auto step_loads = [&]() { if (++it_l >= i1) { it_l = i0; tl = next_tile(tl); } };
    int8v Rc = items_at(i0);
    // (the ring's slots are loaded in the order the loop leaves them at its back edge -- slot 2 oldest, then 0, then 1:
    // the compiler merges the two states at the loop's head per register, and with slot 2 the youngest here the first
    // body waited for all but one of the loads in flight -- s_waitcnt vmcnt(1) -- before it reused slot 2's registers.
    // Slot 2 gets position 0 again, a load like the loop's; its data are never used.)
    D[2] = load_data(Rc, i0, tc);
    D[0] = load_data(Rc, i0, tl);
    step_loads();
    D[1] = load_data(items_at(it_l), it_l, tl);
    step_loads();
    int8v Rn = items_at(it_l);
    // the reassignment weight of the lane's row of the position in hand, when there is one per row: read per lane (four
    // addresses per wavefront), asked for with the position's records. (Four scalar loads and a select by the lane's
    // sub-row were compiled into a detour through scratch memory -- two stores and a load per item: 312 us instead of
    // 250 for the reference's default call.)
    w_t csv = (w_t)A.cst0;
    auto load_cs = [&](int row0) {
        if (CSTK != 0) csv = cstv[min(row0 + h, omax)];
    };
    load_cs(Rc[0] & 0x1FF);
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    auto body = [&](auto KK) {
        constexpr int k0 = decltype(KK)::value, k2 = (k0 + 2) % 3;
        const Pos pc = tc;
        D[k2] = load_data(Rn, it_l, tl);                       // the data of p + 2
        step_loads();
        const Data dc = D[k0];
        const int w0 = Rc[0];
        const int npad = (w0 >> 9) & 7, kind = (w0 >> 12) & 1;
        const int nabs = pc.nabs0 + cp * 2;                    // (lanes past the last column: results unused)
        // (every lane's two points count, except in a class's last item -- padded sub-rows --, in a tile with dead
        // columns and past the wavefront's last position: a wave-uniform test keeps the rest free)
        const bool alive = left > 0;                          // (past the wavefront's last position: see the loop below)
        const bool rare = (w0 & 0xE00) != 0 || tc_edge || !alive;
        const int col = nabs - n1e - sh;                       // the pair's first column
        bool live0 = true, live1 = true;
        if (rare) {
            const bool rowok = alive && h < RPI - npad;
            live0 = rowok && (unsigned)col < nN;
            live1 = rowok && (unsigned)(col + 1) < nN;
        }
        // (byte offset of (signal, the item's first row, the tile's first column) in Wx, dWx; a quarter of it in the bins)
        const int64_t row8 = (int64_t)A.sig0 * na * N * 8 + pc.off8 + (unsigned)Rc[2];
        auto dump_bins = [&](unsigned kk) {                    // (STORE_K builds)
            char* kd = reinterpret_cast<char*>(A.kdump) + (row8 >> 2) + (lane_row16 >> 2);
            if (!rare) *reinterpret_cast<ssq_u32u*>(kd) = kk;
            else {
                if (live0) *reinterpret_cast<unsigned short*>(kd) = (unsigned short)kk;
                if (live1) *reinterpret_cast<unsigned short*>(kd + 2) = (unsigned short)(kk >> 16);
            }
        };
        int cell0, cell1; float t0x, t0y, t1x, t1y;
        if (kind == 0) {
            ssq_f4u u = dc.u;
            unsigned kq = (unsigned)dc.kq;
            if (tc_edge) {                                     // (a pair with a dead column read its neighbour: see load_data)
                if (col < 0) { u.z = u.x; u.w = u.y; kq <<= 16; }
                else if (col == (int)N - 1) { u.x = u.z; u.y = u.w; kq >>= 16; }
            }
            const int ka = (int)(kq & 0xFFFFu), kb = (int)(kq >> 16);
            cell0 = (live0 && ka != TILE_NOBIN) ? ka * 512 + c8 : scratch8;
            cell1 = (live1 && kb != TILE_NOBIN) ? kb * 512 + c8 : scratch8;
            t0x = u.x; t0y = u.y; t1x = u.z; t1y = u.w;
            if constexpr (STORE_K) dump_bins(kq);
        } else {
            const int lgR = (w0 >> 13) & 31;
            const int qb3 = pc.nabs0 >> lgR;                   // window start + 3: tap 0 of sample q0 sits in lane q0 - qb3
            const int baddr = (((nabs >> lgR) - qb3) << 2) + hb4;
            ssq_f2 A0, D0, A1, D1;
            {
                int fr[TILE_W], fi[TILE_W];
                int ur = __float_as_int(dc.u.x), ui = __float_as_int(dc.u.y);
                SSQ_BPERMUTE_OFF(fr[0], baddr, ur, 0);  SSQ_BPERMUTE_OFF(fi[0], baddr, ui, 0);
                SSQ_BPERMUTE_OFF(fr[1], baddr, ur, 4);  SSQ_BPERMUTE_OFF(fi[1], baddr, ui, 4);
                SSQ_BPERMUTE_OFF(fr[2], baddr, ur, 8);  SSQ_BPERMUTE_OFF(fi[2], baddr, ui, 8);
                SSQ_BPERMUTE_OFF(fr[3], baddr, ur, 12); SSQ_BPERMUTE_OFF(fi[3], baddr, ui, 12);
                SSQ_BPERMUTE_OFF(fr[4], baddr, ur, 16); SSQ_BPERMUTE_OFF(fi[4], baddr, ui, 16);
                SSQ_BPERMUTE_OFF(fr[5], baddr, ur, 20); SSQ_BPERMUTE_OFF(fi[5], baddr, ui, 20);
                SSQ_BPERMUTE_OFF(fr[6], baddr, ur, 24); SSQ_BPERMUTE_OFF(fi[6], baddr, ui, 24);
                SSQ_BPERMUTE_OFF(fr[7], baddr, ur, 28); SSQ_BPERMUTE_OFF(fi[7], baddr, ui, 28);
                SSQ_LDS_WAIT();
                // (A = (a_re, a_im), D = (a'_re, a'_im) of the pair's two columns: what the modulation multiplies)
                ssq_f2 sv[TILE_W];
#pragma unroll
                for (int t = 0; t < TILE_W; ++t) { sv[t].x = __int_as_float(fr[t]); sv[t].y = __int_as_float(fi[t]); }
                SSQ_TAPS8X2(A0, D0, A1, D1, wa0, wb0, sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]);
            }
            const int kcs = dc.kq;                             // centre bin of the lane's row
            const float theta = (float)kcs * A.theta_scale;
            // d/dt of e^{i theta n} a(n):  e^{i theta n} (i theta a + a')
            D0.x = __builtin_fmaf(-theta, A0.y, D0.x);
            D0.y = __builtin_fmaf(theta, A0.x, D0.y);
            D1.x = __builtin_fmaf(-theta, A1.y, D1.x);
            D1.y = __builtin_fmaf(theta, A1.x, D1.y);
            // phases kc n mod M of the two columns, exact in integers
            const unsigned ph0 = __umul24((unsigned)kcs, (unsigned)nabs) & (unsigned)A.mmask;
            const unsigned ph1 = (ph0 + (unsigned)kcs) & (unsigned)A.mmask;
            const float rev0 = (float)ph0 * A.inv_m, rev1 = (float)ph1 * A.inv_m;
            ssq_f2 tw0, tw1, W0, V0, W1, V1;
            tw0.x = __builtin_amdgcn_cosf(rev0); tw0.y = __builtin_amdgcn_sinf(rev0);
            tw1.x = __builtin_amdgcn_cosf(rev1); tw1.y = __builtin_amdgcn_sinf(rev1);
            SSQ_CMUL_PK(W0, tw0, A0);
            SSQ_CMUL_PK(W1, tw1, A1);
            SSQ_CMUL_PK(V0, tw0, D0);
            SSQ_CMUL_PK(V1, tw1, D1);
            // (lanes past the last column hold other columns' weights, padded sub-rows another row's samples: their
            // values go nowhere)
            // (one 16-byte store per lane; a pair with a dead point: the live one alone)
            auto store_pair = [&](char* dst, const ssq_f2 a, const ssq_f2 b) {
                char* q = dst + row8 + (size_t)lane_row16;
                if (!rare) { ssq_f4u v; v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y; t3_store(reinterpret_cast<ssq_f4u*>(q), v); }
                else {
                    if (live0) *reinterpret_cast<float2*>(q) = make_float2(a.x, a.y);
                    if (live1) *reinterpret_cast<float2*>(q + 8) = make_float2(b.x, b.y);
                }
            };
            store_pair(reinterpret_cast<char*>(A.Wx), W0, W1);
            if (STORE_D) store_pair(reinterpret_cast<char*>(A.dWx), V0, V1);
            // phase transform and bin: as emit_point<LEAN> of the block kernels, per column
            bool pend0, pend1;
            int ko0 = pair_bin<GRID>(W0, V0, live0, m2hi, m2lo, sp, omax, fx, fa, pend0);
            int ko1 = pair_bin<GRID>(W1, V1, live1, m2hi, m2lo, sp, omax, fx, fa, pend1);
            if (pend0) ko0 = exact_bin(make_float2(W0.x, W0.y), make_float2(V0.x, V0.y), sp, omax, A.gamma);
            if (pend1) ko1 = exact_bin(make_float2(W1.x, W1.y), make_float2(V1.x, V1.y), sp, omax, A.gamma);
            cell0 = ko0 >= 0 ? ko0 * 512 + c8 : scratch8;
            cell1 = ko1 >= 0 ? ko1 * 512 + c8 : scratch8;
            t0x = W0.x; t0y = W0.y; t1x = W1.x; t1y = W1.y;
            if constexpr (STORE_K)
                dump_bins((unsigned)(ko0 >= 0 ? ko0 : TILE_NOBIN) | ((unsigned)(ko1 >= 0 ? ko1 : TILE_NOBIN) << 16));
        }
        {
            w_t cs = (w_t)A.cst0;
            if (CSTK != 0) {
                cs = csv;
            }
            const double a0 = (double)TM::make(t0x, cs), b0 = (double)TM::make(t0y, cs);
            const double a1 = (double)TM::make(t1x, cs), b1 = (double)TM::make(t1y, cs);
            SSQ_LDS_ADD_F64_AT(cell0, 0, a0);
            SSQ_LDS_ADD_F64_AT(cell0, 256, b0);
            SSQ_LDS_ADD_F64_AT(cell1, 128, a1);
            SSQ_LDS_ADD_F64_AT(cell1, 384, b1);
        }
        --left;
        const bool tile_end = alive && ++it_c >= i1;           // (the block's last item: the tile is complete)
        if (tile_end) it_c = i0;
        Rc = items_at(it_c);                                      // the next position's records
        Rn = items_at(it_l);
        load_cs(Rc[0] & 0x1FF);                                // ... and its rows' weights, when there is one per row
        {
            // the next position is of another class: its weights replace the ones in hand (512 bytes per lane out of
            // the L2 -- at most twice per tile and wavefront, the host cuts the row blocks that way)
            const int wn = Rc[0];
            if (((wn >> 12) & 1) && (wn >> 13) != wcls) {
                load_wt(wa0, wb0, wn); wcls = wn >> 13;
                // (the weights are waited for HERE, inside the rare branch: left pending, every item's taps would wait
                // for all loads older than the newest data -- s_waitcnt vmcnt(2) -- whether a class changed or not)
#pragma unroll
                for (int t = 0; t < TILE_W; ++t) { SSQ_OPAQUE_V(wa0[t]); SSQ_OPAQUE_V(wb0[t]); }
            }
        }
        if (tile_end) {
            finish_tile((pc.nabs0 - n1e) >> LGC, pc.sg);
            tc = next_tile(tc);
            tc_edge = edge_tile(tc.nabs0);
        }
    };
    // Whole turns of the ring, ONE back edge: with an exit behind every body the compiler's control flow has edges from
    // the middle of a turn to the loop's head, the count of loads in flight it can rely on there drops to one, and the
    // first body of every turn drained the prefetch (s_waitcnt vmcnt(1)). The up to two positions past the
    // wavefront's last are dead: valid loads, no lane alive, no tile end.
    for (int turn = (total + 2) / 3; turn > 0; --turn) {
        body(K0{});
        body(K1{});
        body(K2{});
    }
}

// ---------------------------------------------------------------------------- host side
template <int GRID, bool STORE_D, int CSTK, bool STORE_K = false>
static int launch_tile3_c(const TilePlan& P, const Tile3Args& A, const SsqParams& sp, hipStream_t stream) {
    auto kern = tile3_kernel<GRID, STORE_D, CSTK, STORE_K>;
    const size_t lds = tile2_lds_bytes(P.na, T3_COLS);
    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int64_t ntx = (P.N + (P.n1 & 1) + T3_COLS - 1) / T3_COLS;     // (the tiles start a column early when n1 is odd)
    // persistent workgroups, one per CU; G * COLS a multiple of the largest R (the lanes keep their weights' phase)
    const int64_t cap = (int64_t)P.ncu;
    const int64_t q = std::max<int64_t>(1, ((int64_t)1 << P.lgr_max2) / T3_COLS);
    const int64_t G = ntx <= cap ? ntx : std::max<int64_t>(q, cap / q * q);
    const char* ce = getenv("SSQ_DEBUG_TILE2_CARRY");              // (read per launch: tests switch it)
    const bool carry_on = !(ce && atoi(ce) == 0);
    Tile3Args B = A;
    B.carry = (carry_on && ntx > G && ntx % q == 0) ? 1 : 0;
    const char* xe = getenv("SSQ_DEBUG_TILE2_XCD");                 // (read per launch)
    B.xcd = !(xe && atoi(xe) == 0) && G >= 16;
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(64 * TILE3_NW), lds, stream, B, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}
template <int GRID, bool STORE_D>
static int launch_tile3_k(const TilePlan& P, const Tile3Args& A, const SsqParams& sp, hipStream_t stream) {
    const int cstk = sp.cst_f64 ? 2 : (sp.cst_uniform ? 0 : 1);
    if (A.kdump) {
        SSQ_REQUIRE(cstk == 0, "bin dump: built for uniform reassignment weights ('log' scales)");
        return launch_tile3_c<GRID, STORE_D, 0, true>(P, A, sp, stream);
    }
    if (cstk == 0) return launch_tile3_c<GRID, STORE_D, 0>(P, A, sp, stream);
    if (cstk == 1) return launch_tile3_c<GRID, STORE_D, 1>(P, A, sp, stream);
    return launch_tile3_c<GRID, STORE_D, 2>(P, A, sp, stream);
}

int TilePlan::run_pair(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                       const void* cst, float cst0, const SsqParams& sp, hipStream_t stream, unsigned short* kdump) {
    SSQ_REQUIRE(pair_ok(), "the pair kernel does not take this plan (more than 318 rows, or fewer than 64 columns)");
    Tile3Args B;
    B.kdump = kdump;
    B.items = reinterpret_cast<const int*>(items3);
    B.waves = reinterpret_cast<const int4*>(wave_first3);
    B.wtab = (const float4*)wtab; B.U = (const float2*)U; B.cst = cst;
    B.Wx = (float2*)Wx; B.dWx = (float2*)dWx; B.Tx = (float2*)Tx; B.kidx = kidx;
    B.N = N; B.na = na; B.n_items = n_items3; B.n1 = (int)n1; B.mmask = (int)(M - 1);
    B.lgM = 0; while (((int64_t)1 << B.lgM) < M) ++B.lgM;
    B.sig0 = sig; B.nsig = nsig; B.group = group; B.inv_m = 1.0f / (float)M;
    B.theta_scale = (float)(6.283185307179586 / ((double)M * dt)); B.cst0 = cst0;
    B.counters = counters; B.gamma = sp.gamma; B.carry = 0; B.xcd = 0;
#define TILE3_LAUNCH(G) return dWx ? launch_tile3_k<G, true>(*this, B, sp, stream) : launch_tile3_k<G, false>(*this, B, sp, stream);
    if (sp.grid == SSQ_GRID_LOG) { TILE3_LAUNCH(SSQ_GRID_LOG) }
    if (sp.grid == SSQ_GRID_LOG_PIECEWISE) { TILE3_LAUNCH(SSQ_GRID_LOG_PIECEWISE) }
    TILE3_LAUNCH(SSQ_GRID_LIN)
#undef TILE3_LAUNCH
}

}  // namespace ssq
