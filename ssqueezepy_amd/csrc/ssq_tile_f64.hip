// ssq_tile_f64.hip -- the default column-tile kernel of the fused ssq_cwt form (float32 data, gfx950): `tile2_kernel`,
// a float64 Tx tile in LDS with unordered ds_add_f64. Interpolates Wx (and dWx) of most rows from decimated baseband
// samples (ssq_tile_fft.hip; math and planning: ssqueezepy_amd/_tiles.py), reads the other rows' Wx + bin back, and
// reassigns all of them -- replaces the reference's cwt + phase_cwt + ssqueeze loop nests for these rows
// (ssqueezepy/_cwt.py:167-177, algos.py:859-953).
//
// Compiled with -ffp-contract=off (bin indices); multiply-adds that may fuse are written as explicit fmaf so every
// instantiation rounds identically. The tuning switches of rounds 3-5 (ablation builds, shader-clock stamps) are not
// in this file: tools/r5/tile_switches.diff re-applies them.
//
// tile2_kernel (round 4): the work of the ticketed kernel (ssq_tile_ordered.hip) without its ticket chain.
//
// What round 4 measured on the MI355X (profiles/r4_ab_history.txt): the ticketed kernel
// spends a quarter of every tile at its boundary and is otherwise paced by the hand-overs (177 us
// of chain alone, 230 us of arithmetic alone, 256 us together); its time does not change when
// the gather, the modulation or the bin arithmetic are taken out, it is the same on 64 and on 256
// CUs (per tile), and LDS *float32* atomics, the obvious way around the tickets, take 193 cycles
// per wavefront instruction -- while ds_add_f64 takes 13.6 and ds_add_u64 10.8
// (tools/probes/lds_atomic_probe.hip).
//
// So the tile is kept in float64 and every wavefront adds its terms as soon as it has them
// (ds_add_f64, no return value): 16 bytes per cell, hence COLS = 32 columns per tile (16 when
// na > 318) and 64 / COLS consecutive rows per wavefront instruction (lane = sub-row h x column).
// Nothing orders the wavefronts inside a tile, so
//   * a wavefront owns a CONTIGUOUS block of the tile's rows (cost-balanced by the host), the
//     same block for every tile: consecutive rows share their decimation class, and the
//     interpolation weights of a class depend on the column only through n mod R -- the same
//     for every tile of a persistent workgroup whose tile stride (gridDim x COLS columns) is a
//     multiple of R: weights are re-read at class changes only, not per step;
//   * an item (= one wavefront instruction's rows) carries 16 bytes of state (one packed
//     record), the pipeline is: record two items ahead, samples one item ahead;
//   * a tile ends with two hardware barriers (all terms in / tile written out and cleared)
//     instead of 76 hand-overs.
// The sum of a cell is the float64 sum of its float32 (or float64) terms, rounded once: it
// differs from the reference's running float32 sum (algos.py:912-924) by that sum's own
// rounding, ~1e-7 of the largest cell (tests bound it at 1e-6); the bins are the same integers.
// float64 addition is not associative either, but with 300 terms of 24-bit mantissas the
// double sum's own rounding error is ~1e-16 relative: the float32 result differs between two
// arrival orders only when the exact sum lies within that of a float32 rounding boundary.
#include "ssq_common.h"
#include "ssq_tiles.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace ssq {

#include "ssq_point_math.inl"
#include "ssq_tile_dev.h"

template <int COLS> struct Tile2Geo {
    static constexpr int RPI = 64 / COLS;              // rows per wavefront instruction
    static constexpr int LGC = COLS == 32 ? 5 : 4;
};
struct Tile2Args {
    const int* items;        // [n_items][8]: row0 | npad << 9 | kind << 12 | lgR << 13, samples' offset of sub-row 0
                             // (class + row), row0 * N * 8, entries between two signals' rows of the class, kc of the sub-rows
    const int4* waves;       // [NW]: first item, end, first item of the wavefront's second class (= end: none), 0
    const float4* wtab; const float2* U;
    const void* cst;
    float2* Wx; float2* dWx; float2* Tx; const unsigned short* kidx;
    unsigned short* kdump;   // STORE_K builds: the bin of every point as it is consumed, (signal, row, column); else null
    int64_t N, na;
    int n_items, n1, mmask, lgM, sig0, nsig, group;
    int carry;                                       // the walk b, b + G, ... runs through the signals' boundaries
    int xcd;                                         // first tiles permuted per XCD (see the kernel)
    float inv_m, theta_scale, cst0;
    unsigned long long* counters;
    double gamma;
};

// Both tile kernels are bound by the instructions they issue, of every kind (round 4,
// profiles/r4_ab_history.txt: one instruction per cycle and CU; 210 per 64 points in the ticketed
// kernel). This one is built to issue few:
//   * an item (64 / COLS consecutive rows x COLS columns) has ONE scalar record (s_load through the
//     constant address space: the index is wavefront-uniform): the sub-rows are consecutive rows of
//     one class, so a lane's addresses are scalar bases + per-lane constants;
//   * one load of samples (or Wx + bin for rows read back) and one store of Wx per item, one
//     16-byte-per-lane store of Tx per 4 (8) rows x COLS columns of a finished tile;
//   * the interpolation weights stay in registers for the whole launch: a wavefront's block of
//     rows spans at most two decimation classes (the host cuts the blocks that way), and a lane's
//     weights depend on its column only through n mod R, the same for every tile of a workgroup
//     whose tile stride is a multiple of R (the launcher picks the grid that way).
// STORE_K (diagnostic builds, ssq_cwt_plan_set_bin_dump): every point's bin index goes to A.kdump as the
// reassignment consumes it -- what pins the kernel's index work as integers against the oracle's map.
template <int GRID, bool STORE_D, int NW, int CSTK, int COLS, bool STORE_K = false>
__global__ __launch_bounds__(64 * NW) void tile2_kernel(Tile2Args A, SsqParams sp) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int RPI = Tile2Geo<COLS>::RPI, LGC = Tile2Geo<COLS>::LGC;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & (COLS - 1), h = lane >> LGC, hb4 = (lane & ~(COLS - 1)) * 4;
    const int64_t N = A.N;
    const unsigned nN = (unsigned)N;
    const int na = (int)A.na, omax = na - 1;
    double2* T = reinterpret_cast<double2*>(lds_raw);          // (na + 1) x COLS cells, the last row: scratch
    for (int k = threadIdx.x; k < (na + 1) * COLS; k += 64 * NW) T[k] = make_double2(0.0, 0.0);
    __syncthreads();
    // (LDS byte addresses of the lane's column in row 0 and in the scratch row)
    const int c16 = c * 16 + (int)SSQ_LDS_ADDR(lds_raw);
    const int scratch16 = na * COLS * 16 + c16;
    const int full_rounds = na / (NW * RPI);                   // write-out rounds (NW * RPI rows each) that are complete

    const int ntx = (int)((N + COLS - 1) / COLS);
    const int G = (int)gridDim.x;
    // Workgroup b walks tiles b, b + G, ... -- of each signal (then a signal's last round is short for the
    // workgroups past ntx mod G, launch after launch: 304 against 320 tiles at config 2), or, A.carry, of the
    // signals laid end to end (the launcher allows it when the lanes' weights survive the boundary).
    // Workgroup b runs on XCD b mod 8 (each XCD has its own L2). The workgroup's FIRST tile is permuted so that the 32
    // workgroups of an XCD walk 32 ADJACENT tiles at a time: neighbouring tiles read overlapping windows of the decimated
    // samples (8 taps of halo; for R >= 64 the very same samples), which then meet in one L2 instead of being fetched
    // from HBM once per tile. The stride between a workgroup's tiles stays G, so the lanes' weight phase is kept.
    // (SSQ_DEBUG_TILE2_XCD=0 in the launcher's environment: the identity.)
    const int bid = (A.xcd && (G & 7) == 0) ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int per_sig = bid < ntx ? (ntx - bid + G - 1) / G : 0;
    const int ntl = A.carry ? (int)(((int64_t)A.nsig * ntx - bid + G - 1) / G)
                            : per_sig * A.nsig;                // tiles of this workgroup
    const auto* waves = SSQ_CONST_PTR(int4, A.waves);
    const int i0 = waves[wv].x, i1 = waves[wv].y, isp = waves[wv].z, ni = i1 - i0;
    // The wavefronts of a SIMD compete for its issue slots and the oldest wins: left alone, the four
    // youngest wavefronts of the workgroup finish their rows of every tile last and the others wait
    // for them at the barrier (measured: 14.7 k of 37 k cycles per tile); fixed priorities against the
    // age only turn the order around. So the priorities rotate: the four wavefronts of a SIMD (w, w + 4,
    // w + 8, w + 12) alternate between two levels, two high and two low at any time, swapped with every
    // item (+2 %; four rotating levels measured the same and cost three more branches per item).
    int prio = (wv >> 2) & 1;
    auto rotate_priority = [&]() {
        SSQ_PRIO_TOGGLE(prio);                                 // (two levels, swapped with every item)
    };
    const float g2 = (float)(A.gamma * A.gamma);
    const float m2hi = g2 * 1.000004f, m2lo = g2 * 0.999996f;
    const int fx = sp.flipud ? -1 : 0, fa = sp.flipud ? na : 0;
    using TM = TileTerm<CSTK == 2>;
    using w_t = typename TM::wtype;
    const auto* cstv = SSQ_CONST_PTR(w_t, A.cst);

    // ---- a tile's end: all terms in (barrier), every wavefront writes its share of the rows to
    // Tx and clears them, tile free again (barrier). A lane takes two neighbouring columns of a row:
    // one 16-byte store, a wavefront instruction = 128 / COLS rows (N even; otherwise column by column).
    auto finish_tile = [&](int tx, int sg) {
        SSQ_WG_BARRIER();
        float2* Tx = A.Tx + (int64_t)(A.sig0 + sg) * na * N;
        constexpr int NA_CAP = COLS == 32 ? 320 : 512;
        {
            constexpr int ROUNDS = (NA_CAP + NW * RPI - 1) / (NW * RPI), RR = NW * RPI;    // RR rows per round
            const int k0 = wv * RPI + h;                       // the lane's row in round 0
            if ((tx + 1) * COLS <= (int)nN) {
                // every column of the tile exists (all but a signal's last tile when COLS does not divide N):
                // the rounds below the last need no masks -- scalar base per round + a per-lane constant
                char* tb = reinterpret_cast<char*>(Tx) + (size_t)tx * (COLS * 8);
                const unsigned voff = ((unsigned)k0 * nN + (unsigned)c) * 8u;
                // (the round count and the rounds' distance are re-read as scalars at every use: hoisted out
                // of the item loop, the compiler keeps ten lane masks and ten 64-bit offsets in spilled registers)
                int fr = full_rounds;
                size_t step = (size_t)RR * (size_t)N * 8;
#pragma unroll
                for (int m = 0; m < ROUNDS - 1; ++m) {
                    SSQ_OPAQUE_S(fr); SSQ_OPAQUE_S(step);
                    if (m < fr) {                              // (wave-uniform)
                        const int k = k0 + m * RR;
                        const double2 v = T[k * COLS + c];
                        T[k * COLS + c] = make_double2(0.0, 0.0);
                        *reinterpret_cast<float2*>(tb + (size_t)voff) = make_float2((float)v.x, (float)v.y);
                        tb += step;
                        asm volatile("" ::: "memory");         // (keeps the rounds from being batched into registers)
                    }
                }
                {   // the last round: the rows left, and the scratch row cleared by the lanes past them
                    const int k = k0 + fr * RR;
                    const int kc_ = k < na ? k : na;
                    const double2 v = T[kc_ * COLS + c];
                    T[kc_ * COLS + c] = make_double2(0.0, 0.0);
                    if (k < na)
                        *reinterpret_cast<float2*>(tb + (size_t)voff) = make_float2((float)v.x, (float)v.y);
                }
            } else {
                const unsigned col = (unsigned)(tx * COLS + c);
                const bool ok = col < nN;
#pragma unroll 1
                for (int m = 0; m * RR < na + 1; ++m) {
                    const int k = k0 + m * RR;
                    const int kc_ = k < na ? k : na;
                    const double2 v = T[kc_ * COLS + c];
                    T[kc_ * COLS + c] = make_double2(0.0, 0.0);
                    if (ok && k < na) Tx[(unsigned)k * nN + col] = make_float2((float)v.x, (float)v.y);
                }
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

    // ---- the wavefront's sequence of (tile, item) positions, software-pipelined over a ring of three
    // data slots (the loop is unrolled three times, the slots are compile-time): while position p is
    // computed, the data of p + 2 go out. Two cursors walk the same sequence, the loads' two positions
    // ahead of the arithmetic's; each is an item index and the tile as the kernel uses it: n of the
    // tile's first column (n1 + first column), the signal, and the byte offset of (signal, row 0, first
    // column) in Wx -- moved by constants when the cursor's item index wraps (no 64-bit products, and no
    // position records copied around per item). Past the last tile the loads' cursor stays on it (all
    // loads unconditional, see the note in tile_kernel: what they fetch there is valid and unused).
    struct Pos { int nabs0, sg; int64_t off8; };
    const int nabs_step = G * COLS, nabs_first = A.n1 + bid * COLS, nabs_last = A.n1 + (ntx - 1) * COLS;
    const int64_t off8_step = (int64_t)G * COLS * 8;
    // (a signal's end: back to the workgroup's first tile, or -- carry -- on by the same stride into the next signal)
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
    const int total = ntl * ni;                                // positions of this wavefront
    typedef int int8v __attribute__((ext_vector_type(8)));
    const auto* items = SSQ_CONST_PTR(int8v, A.items);
    // per-lane constants of the addresses: the lane's place inside an item's rows
    const unsigned lane_row8 = (unsigned)h * nN * 8u + (unsigned)c * 8u;       // bytes: sub-row h, column c
    const unsigned lane_col8 = (unsigned)c * 8u;

    // data of a position: (interpolated) the lane's sample of its sub-row's window, or (rows read
    // back) Wx and the bin of the lane's point
    struct Data { float2 u; int kq; };
    const char* const U8 = reinterpret_cast<const char*>(A.U);
    const char* const WX8 = reinterpret_cast<const char*>(A.Wx) + (size_t)((int64_t)A.sig0 * na * N) * 8u;
    const char* const KX8 = reinterpret_cast<const char*>(A.kidx);
    // (ANY0: the wavefront's block holds rows read back; a wavefront of interpolated rows only -- most
    // are -- runs a loop without the bin load and the kind tests: one vector-memory instruction less per
    // item, and the CU's vector-memory path takes one wavefront instruction per ~20 cycles)
    auto load_data = [&](auto any0, const int8v R, const Pos& q) {
        constexpr bool ANY0 = decltype(any0)::value;
        Data d;
        const int w0 = R[0];
        const int kind = ANY0 ? (w0 >> 12) & 1 : 1;
        const char* base; unsigned voff;
        const char* kbase = reinterpret_cast<const char*>(A.items); unsigned koff = (unsigned)lane * 2u;
        if (kind) {                                            // (wave-uniform; the loads themselves stay outside)
            // sample (qb + min(c, wlast)) mod L of row h of the item, h * L entries on
            const int lgR = (w0 >> 13) & 31;
            const int qb = (q.nabs0 >> lgR) - (TILE_W / 2 - 1);
            const int wlast = ((COLS - 1) >> lgR) + TILE_W;
            const int lmask = A.mmask >> lgR;                  // L - 1, L = M / R
            voff = (((unsigned)((qb + (c < wlast ? c : wlast)) & lmask)) + ((unsigned)h << (A.lgM - lgR))) * 8u;
            base = U8 + ((size_t)(unsigned)R[1] + (size_t)((unsigned)q.sg * (unsigned)R[3])) * 8u;
        } else {
            // point (row0 + h, column) -- the last column's for lanes past it, the last real row's for
            // padded sub-rows -- and its bin
            const int npad = (w0 >> 9) & 7;
            unsigned lr8 = lane_row8;
            if (RPI == 2) { if (npad) lr8 = lane_col8; }
            else if (npad) lr8 = (unsigned)min(h, RPI - 1 - npad) * nN * 8u + lane_col8;
            if (q.nabs0 == nabs_last) {                        // (the last tile may be partial)
                const int col = q.nabs0 - A.n1 + c;
                if (col >= (int)N) lr8 -= (unsigned)(col - ((int)N - 1)) * 8u;
            }
            voff = lr8;
            base = WX8 + ((size_t)q.off8 + (unsigned)R[2]);
            kbase = KX8 + (((size_t)q.off8 + (unsigned)R[2]) >> 2);
            koff = lr8 >> 2;
        }
        d.u = *reinterpret_cast<const float2*>(base + (size_t)voff);
        // (the bin: a load either way, from a harmless address for interpolated rows -- a conditional
        // load costs the compiler its count of loads in flight)
        if constexpr (ANY0) d.kq = (int)*reinterpret_cast<const unsigned short*>(kbase + (size_t)koff);
        else d.kq = 0;
        return d;
    };
    // the weights of the wavefront's (up to) two classes, for the lane's column phase: once
    ssq_f2 wta[TILE_W], wtb[TILE_W];
    auto load_wt = [&](ssq_f2 (&wt)[TILE_W], int it, int woff) {
        const int lgR = (items[it][0] >> 13) & 31;
        const int nabs = A.n1 + bid * COLS + c;                // (every tile of this workgroup: the same n mod R)
        const int R = 1 << lgR;
        const float4* wp = A.wtab + (int64_t)woff * 4 + (nabs & (R - 1));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = wp[t * R];
            wt[2 * t].x = v.x; wt[2 * t].y = v.y; wt[2 * t + 1].x = v.z; wt[2 * t + 1].y = v.w;
        }
    };
    load_wt(wta, i0, waves[wv].w & 0xFFFF);
    load_wt(wtb, isp < i1 ? isp : i0, (int)((unsigned)waves[wv].w >> 16));

    using Yes = std::true_type; using No = std::false_type;
    auto run = [&](auto any0) {
    constexpr bool ANY0 = decltype(any0)::value;
    Data D[3];
    Pos tc, tl;                                                // the tile of the arithmetic's cursor, of the loads'
    tc.nabs0 = nabs_first; tc.sg = 0; tc.off8 = (int64_t)bid * COLS * 8;
    tl = tc;
    if (total <= 0) return;
    int it_c = i0, it_l = i0;
    bool tc_last = tc.nabs0 == nabs_last && (N & (COLS - 1)) != 0;   // the arithmetic's tile is a signal's last, partial one
    int left = total;                                          // positions not yet finished
    auto step_loads = [&]() { if (++it_l >= i1) { it_l = i0; tl = next_tile(tl); } };
    // the records of the position in hand and of the one whose data go out next: asked for (through
    // the scalar cache) at the end of the position before, so that they are there when it starts
    int8v Rc = items[i0];
    D[0] = load_data(any0, Rc, tl);
    step_loads();
    D[1] = load_data(any0, items[it_l], tl);
    step_loads();
    // (the third slot: position 0 again -- a load like the loop's, so that the compiler's count of the loads in
    // flight at the loop's head is the loop's own; a plain copy made the first body wait for one load too many)
    D[2] = load_data(any0, Rc, tc);
    int8v Rn = items[it_l];
    // the per-row reassignment weights of the position in hand (scalar loads, asked for with its records)
    w_t csn[RPI];
    auto load_cs = [&](int row0) {
        if (CSTK != 0) {
#pragma unroll
            for (int k = 0; k < RPI; ++k) csn[k] = cstv[min(row0 + k, omax)];
        }
    };
    load_cs(Rc[0] & 0x1FF);
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    bool more = true;
    auto body = [&](auto KK) {
        constexpr int k0 = decltype(KK)::value, k1 = (k0 + 1) % 3, k2 = (k0 + 2) % 3;
        const Pos pc = tc;
        rotate_priority();
        D[k2] = load_data(any0, Rn, tl);                       // the data of p + 2
        step_loads();
        const Data dc = D[k0];
        const int w0 = Rc[0];
        const int npad = (w0 >> 9) & 7, kind = ANY0 ? (w0 >> 12) & 1 : 1;
        const int nabs = pc.nabs0 + c;                         // (lanes past the last column: results unused)
        // (every lane's point counts, except in a class's last item -- padded sub-rows -- and in the last
        // tile of a signal when N is not a multiple of the tile: a wave-uniform test keeps the rest free)
        bool livept = true;
        if ((w0 & 0xE00) != 0 || tc_last) livept = h < RPI - npad && nabs - A.n1 < (int)N;
        int cell16; float tvx, tvy;
        if (kind == 0) {
            const int kk = dc.kq & 0xFFFF;
            cell16 = (livept && kk != TILE_NOBIN) ? kk * (COLS * 16) + c16 : scratch16;
            tvx = dc.u.x; tvy = dc.u.y;
            if constexpr (STORE_K) {
                char* kd8 = reinterpret_cast<char*>(A.kdump) + (((size_t)((int64_t)A.sig0 * na * N) * 8u + (size_t)pc.off8 + (unsigned)Rc[2]) >> 2);
                if (livept) *reinterpret_cast<unsigned short*>(kd8 + (size_t)(lane_row8 >> 2)) = (unsigned short)kk;
            }
        } else {
            const int lgR = (w0 >> 13) & 31;
            const int qb3 = pc.nabs0 >> lgR;                   // window start + 3: tap 0 of sample q0 sits in lane q0 - qb3
            const int baddr = (((nabs >> lgR) - qb3) << 2) + hb4;
            ssq_f2 A2, D2;
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
                // (A2 = (a_re, a_im), D2 = (a'_re, a'_im): the pairs the modulation multiplies)
                ssq_f2 sv[TILE_W];
#pragma unroll
                for (int t = 0; t < TILE_W; ++t) { sv[t].x = __int_as_float(fr[t]); sv[t].y = __int_as_float(fi[t]); }
                if (it_c < isp) {   // (wave-uniform: the wavefront's first or second class)
                    SSQ_TAPS8(A2, D2, wta, sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]);
                } else {
                    SSQ_TAPS8(A2, D2, wtb, sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]);
                }
            }
            int kcs = Rc[4];                                    // centre bin of the lane's row
#pragma unroll
            for (int k = 1; k < RPI; ++k) if (h == k) kcs = Rc[4 + k];
            const float theta = (float)kcs * A.theta_scale;
            // d/dt of e^{i theta n} a(n):  e^{i theta n} (i theta a + a')
            D2.x = __builtin_fmaf(-theta, A2.y, D2.x);
            D2.y = __builtin_fmaf(theta, A2.x, D2.y);
            const float rev = (float)(__umul24((unsigned)kcs, (unsigned)nabs) & (unsigned)A.mmask) * A.inv_m;
            ssq_f2 tw2, W2, V2;
            tw2.x = __builtin_amdgcn_cosf(rev); tw2.y = __builtin_amdgcn_sinf(rev);
            SSQ_CMUL_PK(W2, tw2, A2);
            SSQ_CMUL_PK(V2, tw2, D2);
            const float2 Wv = make_float2(W2.x, W2.y);
            const float2 Dv = make_float2(V2.x, V2.y);
            // (lanes past the last column hold another column's weights, padded sub-rows another row's
            // samples: their values go nowhere)
            char* wx8 = const_cast<char*>(WX8) + ((size_t)pc.off8 + (unsigned)Rc[2]);
            if (livept) *reinterpret_cast<float2*>(wx8 + (size_t)lane_row8) = Wv;
            if (STORE_D) {
                char* dwx8 = reinterpret_cast<char*>(A.dWx) + ((size_t)((int64_t)A.sig0 * na * N) * 8u + (size_t)pc.off8 + (unsigned)Rc[2]);
                if (livept) *reinterpret_cast<float2*>(dwx8 + (size_t)lane_row8) = Dv;
            }
            // phase transform and bin: as emit_point<LEAN> of the block kernels
            const float cc = Wv.x, dd = Wv.y, aa = Dv.x, bb = Dv.y;
            const float m2 = cc * cc + dd * dd, num = bb * cc - aa * dd;
            const bool above = m2 > m2hi, below = m2 < m2lo;
            const float w32 = fabsf(num * __builtin_amdgcn_rcpf(m2 * 6.2831855f));
            bool ok;
            const int kb = bin_screen_cwt<GRID>(w32, sp, omax, ok);
            const int kf = (kb ^ fx) + fa;
            int kout = (above && livept) ? kf : -1;
            const bool pend = livept && !(below | (above & ok));
            if (pend) kout = exact_bin(Wv, Dv, sp, omax, A.gamma);
            cell16 = kout >= 0 ? kout * (COLS * 16) + c16 : scratch16;
            tvx = Wv.x; tvy = Wv.y;
            if constexpr (STORE_K) {
                char* kd8 = reinterpret_cast<char*>(A.kdump) + (((size_t)((int64_t)A.sig0 * na * N) * 8u + (size_t)pc.off8 + (unsigned)Rc[2]) >> 2);
                if (livept) *reinterpret_cast<unsigned short*>(kd8 + (size_t)(lane_row8 >> 2)) = (unsigned short)(kout >= 0 ? kout : TILE_NOBIN);
            }
        }
        {
            w_t cs = (w_t)A.cst0;
            if (CSTK != 0) {
                cs = csn[0];
#pragma unroll
                for (int k = 1; k < RPI; ++k) if (h == k) cs = csn[k];
            }
            const double ax = (double)TM::make(tvx, cs), ay = (double)TM::make(tvy, cs);
            SSQ_LDS_ADD_F64_AT(cell16, 0, ax);
            SSQ_LDS_ADD_F64_AT(cell16, 8, ay);
        }
        more = --left > 0;
        const bool tile_end = ++it_c >= i1;                    // (the block's last item: the tile is complete)
        if (tile_end) it_c = i0;
        Rc = items[it_c];                                      // the next position's records (see above)
        Rn = items[it_l];
        load_cs(Rc[0] & 0x1FF);                                // ... and its rows' weights, when there is one per row
        if (tile_end) {
            finish_tile((pc.nabs0 - A.n1) >> LGC, pc.sg);
            tc = next_tile(tc);
            tc_last = tc.nabs0 == nabs_last && (N & (COLS - 1)) != 0;
        }
    };
    for (;;) {
        body(K0{}); if (!more) break;
        body(K1{}); if (!more) break;
        body(K2{}); if (!more) break;
    }
    };
    // (a block spans at most two classes: its first item and the first of its second class tell)
    const bool has0 = !((items[i0][0] >> 12) & 1) || (isp < i1 && !((items[isp][0] >> 12) & 1));
    // (measured: a second loop without the bin load and the kind tests for the wavefronts of interpolated
    // rows only -- one vector-memory instruction and ten scalar ones less per item -- is SLOWER, 230 vs 220 us)
    (void)has0;
    run(Yes{});
}

// ---------------------------------------------------------------------------- host side
// ---- tile2_kernel launch
template <int GRID, bool STORE_D, int NW, int CSTK, int COLS, bool STORE_K = false>
static int launch_tile2_c(const TilePlan& P, const Tile2Args& A, const SsqParams& sp, hipStream_t stream) {
    auto kern = tile2_kernel<GRID, STORE_D, NW, CSTK, COLS, STORE_K>;
    const size_t lds = tile2_lds_bytes(P.na, COLS);
    // (set at every launch: the attribute belongs to the function ON THE CURRENT DEVICE, a flag per instantiation
    // would leave a second device of the process without it)
    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int64_t ntx = (P.N + COLS - 1) / COLS;
    // Persistent workgroups: as many as fit a CU's LDS side by side, workgroup b walks tiles b, b + G,
    // ... of every signal. The kernel keeps a lane's interpolation weights for the whole launch, so
    // the columns of a workgroup's tiles must agree mod R for every class: G * COLS a multiple of the
    // largest R (or a single tile per signal and workgroup).
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, (size_t)(32 / NW)));
    const int64_t cap = (int64_t)P.ncu * per_cu;
    const int64_t q = std::max<int64_t>(1, ((int64_t)1 << P.lgr_max2) / COLS);
    const int64_t G = ntx <= cap ? ntx : std::max<int64_t>(q, cap / q * q);
    // ... and through the signals' boundaries when a signal's tile count keeps that phase too
    // (SSQ_DEBUG_TILE2_CARRY=0: every signal's walk starts at the workgroup's own tile)
    const char* ce = getenv("SSQ_DEBUG_TILE2_CARRY");              // (read per launch: tests switch it)
    const bool carry_on = !(ce && atoi(ce) == 0);
    Tile2Args B = A;
    B.carry = (carry_on && ntx > G && ntx % q == 0) ? 1 : 0;
    const char* xe = getenv("SSQ_DEBUG_TILE2_XCD");                 // (read per launch)
    B.xcd = !(xe && atoi(xe) == 0) && G >= 16;
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(64 * NW), lds, stream, B, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}
template <int GRID, bool STORE_D, int NW, int COLS>
static int launch_tile2_k(const TilePlan& P, const Tile2Args& A, const SsqParams& sp, hipStream_t stream) {
    const int cstk = sp.cst_f64 ? 2 : (sp.cst_uniform ? 0 : 1);
    if (A.kdump) {
        // the diagnostic builds exist for one weight per transform (the bins do not depend on the weights:
        // 'log' scales, what the full-size index test runs)
        SSQ_REQUIRE(cstk == 0, "bin dump: built for uniform reassignment weights ('log' scales)");
        return launch_tile2_c<GRID, STORE_D, NW, 0, COLS, true>(P, A, sp, stream);
    }
    if (cstk == 0) return launch_tile2_c<GRID, STORE_D, NW, 0, COLS>(P, A, sp, stream);
    if (cstk == 1) return launch_tile2_c<GRID, STORE_D, NW, 1, COLS>(P, A, sp, stream);
    return launch_tile2_c<GRID, STORE_D, NW, 2, COLS>(P, A, sp, stream);
}
template <int GRID, bool STORE_D>
static int launch_tile2(const TilePlan& P, Tile2Args& A, const SsqParams& sp, hipStream_t stream) {
    // 16 wavefronts = one workgroup of 1024 work-items per CU (measured, round 4: 12 wavefronts 235 us against 220;
    // 16-column tiles with 16 wavefronts 245 us, with 8 wavefronts and two workgroups per CU 260-275 us --
    // profiles/r4_ab_history.txt)
    A.waves = reinterpret_cast<const int4*>(P.wave_first2);
    if (P.cols2 == 32) return launch_tile2_k<GRID, STORE_D, TILE2_NW, 32>(P, A, sp, stream);
    return launch_tile2_k<GRID, STORE_D, TILE2_NW, 16>(P, A, sp, stream);
}

int TilePlan::run_f64(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                      const void* cst, float cst0, const SsqParams& sp, hipStream_t stream, unsigned short* kdump) {
    SSQ_REQUIRE(tile2_ok, "the tile tables do not split into row blocks of at most two classes");
    Tile2Args B;
    B.kdump = kdump;
    B.items = reinterpret_cast<const int*>(items2); B.waves = nullptr;
    B.wtab = (const float4*)wtab; B.U = (const float2*)U; B.cst = cst;
    B.Wx = (float2*)Wx; B.dWx = (float2*)dWx; B.Tx = (float2*)Tx; B.kidx = kidx;
    B.N = N; B.na = na; B.n_items = n_items2; B.n1 = (int)n1; B.mmask = (int)(M - 1);
    B.lgM = 0; while (((int64_t)1 << B.lgM) < M) ++B.lgM;
    B.sig0 = sig; B.nsig = nsig; B.group = group; B.inv_m = 1.0f / (float)M;
    B.theta_scale = (float)(6.283185307179586 / ((double)M * dt)); B.cst0 = cst0;
    B.counters = counters; B.gamma = sp.gamma; B.carry = 0;
#define TILE2_LAUNCH(G)                                                                     \
    return dWx ? launch_tile2<G, true>(*this, B, sp, stream) : launch_tile2<G, false>(*this, B, sp, stream);
    if (sp.grid == SSQ_GRID_LOG) { TILE2_LAUNCH(SSQ_GRID_LOG) }
    if (sp.grid == SSQ_GRID_LOG_PIECEWISE) { TILE2_LAUNCH(SSQ_GRID_LOG_PIECEWISE) }
    TILE2_LAUNCH(SSQ_GRID_LIN)
#undef TILE2_LAUNCH
}

}  // namespace ssq
