// ssq_tile_ordered.hip -- the column-tile kernel that adds in the reference's order (SSQ_TILE_ORDER=ordered;
// float32, gfx950): Tx bit for bit the CPU loop's (ssqueezepy/algos.py:859-953) on the same Wx / dWx.
//
//   tile_kernel           one persistent workgroup per CU (12 wavefronts, 3 per SIMD) walks
//                         64-column tiles of one signal after the other. The 64 columns x na bins
//                         of the tile's Tx live in LDS (which is why there is one workgroup per
//                         CU); lane = column. The steps (4 consecutive rows) of all tiles are
//                         dealt to the wavefronts round-robin. Per step and lane:
//                           interpolated rows: ONE 8-byte load of u_i per row (the lanes hold a
//                             window of consecutive samples, taps come from the neighbours with
//                             ds_bpermute), 8 taps x (phi, phi') as packed FMAs, modulation by
//                             hardware sin / cos of the exact phase kc n mod M, Wx stored
//                             (512-byte runs), phase transform + bin exactly as the other fused
//                             kernels do (ssq_point_math.inl);
//                           rows read back: Wx and the 2-byte bin the block / exact kernels left.
//                         The arithmetic of different steps runs concurrently; only the
//                         reassignment T[bin] += Wx * const is ordered, by a ticket in LDS (step
//                         S may update the tile once `turn` says so): every cell receives its
//                         contributions in ascending row order, the reference's
//                         (algos.py:859-953), so the float sums are bit-identical to the CPU loop
//                         on the same Wx / dWx. Inside a step the four rows' cells are read
//                         together and chained in registers when they coincide (same lane = same
//                         column: no cross-lane traffic), then written in row order. Between two
//                         tiles all wavefronts write their share of the finished tile to Tx.
//                         No workgroup barrier after the prologue, no atomics on data.
//
//                         Round 3 tried the other split -- producer wavefronts that only compute,
//                         one (or four) updater wavefronts that only reassign, hand-over through an
//                         L2-resident ring -- and measured it slower (400-550 us per transform
//                         against 320): a single wavefront issues a dependent instruction every
//                         8-10 cycles, so a serial stream of ~100 instructions per step cannot keep
//                         up with fifteen producers. What that round kept from it: loads issued
//                         unconditionally so that the compiler's wait counts stay static, the
//                         float64-weight fold, one step per ticket. Everything else that was
//                         measured and dropped: DESIGN.md section 4.8, profiles/r3_ab_history.txt.
//
// Compiled with -ffp-contract=off (bin indices); multiply-adds that may fuse are written as
// explicit fmaf so every instantiation rounds identically.
#include "ssq_common.h"
#include "ssq_tiles.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace ssq {

#include "ssq_point_math.inl"
#include "ssq_tile_dev.h"

struct TileArgs {
    const int4* pstep;                               // one packed record per step (see TilePlan::create)
    const int2* prow;                                // 4 packed row records per step (see TilePlan::create)
    const float4* wtab; const float2* U;
    const void* cst;
    float2* Wx; float2* dWx; float2* Tx; const unsigned short* kidx;
    int64_t N, na;
    int nsteps, n1, mmask, sig0, nsig;
    float inv_m;         // 1 / M
    float theta_scale;   // 2 pi / (M dt): theta of a row = kc * theta_scale
    float cst0;          // the reassignment weight when it is the same for every row
    unsigned long long* counters;   // [0] += tiles finished (what actually ran)
    double gamma;
};

// Workgroup-scope synchronisation through LDS words. The tile and the ticket both live in LDS, so the fences name
// the LDS address space only ("local"): they cost a wait for the wavefront's own LDS operations (s_waitcnt
// lgkmcnt(0)) and leave the vector-memory loads in flight -- the samples of the next steps -- alone; all wavefronts
// of a workgroup share the CU's LDS, so no cache maintenance is involved.
//
// The ticket: `turn` says which step may update the tile. The hand-over is a release (every tile access of the step
// ordered before the new ticket) / acquire (the next step's tile reads ordered after it has seen its ticket) pair --
// round 5; rounds 2-4 relied on the hardware alone (LDS serves one wavefront's operations in program order and the
// CU's wavefronts from one queue, so relaxed accesses in program order were enough on gfx950, and the cells were
// read in the same round trip as the ticket); the language-level ordering costs one more LDS round trip per step.
__device__ __forceinline__ void lds_release_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void lds_acquire_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
__device__ __forceinline__ void lds_store_release(int* p, int v) {
    lds_release_fence();
    __scoped_atomic_store_n(p, v, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
}
// (a look at the ticket that orders nothing: priorities, distance to the turn)
__device__ __forceinline__ int ticket_peek(const int* p) {
    const int v = __scoped_atomic_load_n(p, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
    asm volatile("" ::: "memory");
    return v;
}
// Earliest deadline first: the wavefronts of a SIMD compete for its issue slots (the oldest wins by
// default, so the youngest would always be late for its turn and everybody would wait for it); a
// wavefront raises its priority as its turn comes closer.
__device__ __forceinline__ void ticket_priority(const int* turn, int ticket) {
    const int d = __builtin_amdgcn_readfirstlane(ticket - ticket_peek(turn));
    if (d <= 3) __builtin_amdgcn_s_setprio(3);
    else if (d <= 6) __builtin_amdgcn_s_setprio(2);
    else if (d <= 9) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
// wait until the ticket shows `ticket`; what follows is ordered after the tile accesses of the step that passed it
__device__ __forceinline__ void ticket_wait(const int* turn, int ticket) {
    for (;;) {
        const int d = ticket - ticket_peek(turn);
        if (d == 0) break;
        // (s_sleep 0 is the shortest pause there is; under the CPU emulation it is where the other wavefronts run)
        if (d > 1) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(0);
    }
    lds_acquire_fence();
}
// pass the ticket on: every tile access of this wavefront so far is ordered before it
__device__ __forceinline__ void ticket_pass(int* turn, int next, int lane) {
    lds_release_fence();
    if (lane == 0) __scoped_atomic_store_n(turn, next, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
    asm volatile("" ::: "memory");
}

// ---- the reassignment of one step (4 rows) into the tile, in row order; lane = column. Rows of
// a step that hit the same cell are chained in registers: the cells are read together, a row
// that hits the cell of an earlier row of the step starts from that row's result, the cells are
// written back in row order. `cell` of a point without contribution is the lane's scratch cell.
// Everything that does not depend on the tile is prepared BEFORE the step's turn (the ticket
// section is the serial part of a tile): the cells' addresses, the terms, and the "same cell"
// tests as bit masks, so that inside the turn a select is one v_bfi_b32 per word.
struct Update4Prep {
    int off[TILE_G];                   // byte offset of the cell in the tile
    int same[TILE_G][TILE_G];          // [r][q], q < r: all ones if row q hits the cell of row r
};
__device__ __forceinline__ void update4_prepare(const int (&cell)[TILE_G], Update4Prep& u) {
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) {
        u.off[r] = cell[r] * 8;
        SSQ_OPAQUE_V(u.off[r]);        // (materialised here, not behind the ticket)
#pragma unroll
        for (int q = 0; q < r; ++q) { u.same[r][q] = cell[q] == cell[r] ? -1 : 0; SSQ_OPAQUE_V(u.same[r][q]); }
    }
}
__device__ __forceinline__ float bit_select(int m, float a, float b) {      // m ? a : b, per bit
    int d;
    SSQ_BFI(d, m, __float_as_int(a), __float_as_int(b));
    return __int_as_float(d);
}
// The step's turn, then its four cells (read together: one LDS round trip behind the acquire).
__device__ __forceinline__ void ticket_wait_read4(const int* turn, int ticket, unsigned char* tile,
                                                  const Update4Prep& u, float2 (&t)[TILE_G]) {
    ticket_wait(turn, ticket);
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) t[r] = *reinterpret_cast<const float2*>(tile + u.off[r]);
}
template <typename TM>
__device__ __forceinline__ void update4_finish(unsigned char* tile, const Update4Prep& u, float2 (&t)[TILE_G],
                                               const typename TM::type (&vx)[TILE_G], const typename TM::type (&vy)[TILE_G]) {
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) {
#pragma unroll
        for (int q = 0; q < r; ++q) {
            t[r].x = bit_select(u.same[r][q], t[q].x, t[r].x);
            t[r].y = bit_select(u.same[r][q], t[q].y, t[r].y);
        }
        t[r].x = TM::fold(t[r].x, vx[r]); t[r].y = TM::fold(t[r].y, vy[r]);
    }
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) *reinterpret_cast<float2*>(tile + u.off[r]) = t[r];
}
// (pins a term in a register before the ticket)
__device__ __forceinline__ void keep_term(float& x) { SSQ_OPAQUE_V(x); }
__device__ __forceinline__ void keep_term(double& x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    SSQ_OPAQUE_V(lo); SSQ_OPAQUE_V(hi);
    x = __hiloint2double(hi, lo);
}

// CSTK: reassignment weights -- 0 one float (cst0), 1 a float per row, 2 a double per row
template <int GRID, bool STORE_D, int NW, int CSTK>
__global__ __launch_bounds__(64 * NW) void tile_kernel(TileArgs A, SsqParams sp) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int c = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t N = A.N;
    const unsigned nN = (unsigned)N;
    const int na = (int)A.na, omax = na - 1;
    float2* T = reinterpret_cast<float2*>(lds_raw);           // (na + 1) x 64 cells, the last row: scratch
    int* turn = reinterpret_cast<int*>(lds_raw + (size_t)(na + 1) * TILE_COLS * 8);
    int* wdone = turn + 1;
    for (int k = wv; k <= na; k += NW) T[k * TILE_COLS + c] = make_float2(0.f, 0.f);
    if (threadIdx.x == 0) { *turn = 0; *wdone = 0; }
    __syncthreads();
    const int scratch = na * TILE_COLS + c;

    // The workgroup is persistent: it walks the tiles blockIdx.x, + gridDim.x, ... of the launch
    // group (tile = 64 columns of one signal). All steps of all its tiles form one sequence
    // S = 0, 1, ...: wavefront w takes S = w, w + NW, ... (positions advance monotonically, so
    // divisions are replaced by repeated subtraction). Ticket of step S of tile itl: S + itl --
    // one extra ticket per tile, during which the finished tile is written out.
    const int ntx = (int)((N + TILE_COLS - 1) / TILE_COLS);
    const int ntot = ntx * A.nsig;
    // (first tiles permuted per XCD -- workgroup b runs on XCD b mod 8 -- so that the workgroups of one XCD walk adjacent
    // tiles and their sample windows meet in one L2, as in tile2_kernel: ssq_tile_f64.hip)
    const int G_ = (int)gridDim.x;
    const int bid = ((G_ & 7) == 0 && G_ >= 16) ? ((int)blockIdx.x & 7) * (G_ >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int ntl = ntot > bid ? (ntot - bid + G_ - 1) / G_ : 0;
    const int nst = A.nsteps;
    const int total = nst * ntl;
    struct Pos { int S, itl, st, tx, sg; };                   // a step: sequence number, tile, step in tile, tile position
    auto advance = [&](Pos& q, int by) {
        q.S += by; q.st += by;
        while (q.st >= nst && q.itl < ntl) {
            q.st -= nst; ++q.itl; q.tx += (int)gridDim.x;
            while (q.tx >= ntx) { q.tx -= ntx; ++q.sg; }
        }
    };

    // rows k = wv, wv + NW, ... of the finished tile go to Tx and are cleared; the last
    // wavefront to finish opens the next tile's tickets. (Round 3 also measured the write-out by
    // ONE wavefront, inside the turn of the tile's last step, so that the others never meet: a
    // single wavefront stores 300 x 512 bytes in ~17 k cycles -- 345 us per transform against 275.)
    auto write_out = [&](int itl, int tx, int sg) {
        const int boundary = (itl + 1) * nst + itl;           // the ticket after the tile's last step
        ticket_wait(turn, boundary);                           // (acquire: the tile's last update is in)
        const unsigned col = (unsigned)(tx * TILE_COLS + c);
        const bool ok = col < nN;
        float2* Tx = A.Tx + (int64_t)(A.sig0 + sg) * na * N;
        for (int k0 = wv; k0 < na; k0 += 4 * NW) {
            float2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int k = k0 + q * NW; v[q] = T[(k < na ? k : na) * TILE_COLS + c]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + q * NW;
                if (k < na) {
                    T[k * TILE_COLS + c] = make_float2(0.f, 0.f);
                    if (ok) Tx[(unsigned)k * nN + col] = v[q];
                }
            }
        }
        lds_release_fence();                                   // (the rows cleared above, before the count)
        __builtin_amdgcn_wave_barrier();
        if (c == 0) {
            const int before = __scoped_atomic_fetch_add(wdone, 1, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP);
            lds_acquire_fence();                               // (... and every other wavefront's, before the tickets reopen)
            if (before + 1 == NW * (itl + 1)) {
                lds_store_release(turn, boundary + 1);
                if (A.counters) __scoped_atomic_fetch_add(A.counters, 1ull, __ATOMIC_RELAXED, __MEMORY_SCOPE_DEVICE);
            }
        }
    };

    const float g2 = (float)(A.gamma * A.gamma);
    const float m2hi = g2 * 1.000004f, m2lo = g2 * 0.999996f;
    const int fx = sp.flipud ? -1 : 0, fa = sp.flipud ? na : 0;
    using TM = TileTerm<CSTK == 2>;
    using term_t = typename TM::type;
    using w_t = typename TM::wtype;
    const w_t* cstv = (const w_t*)A.cst;

    // Step and row records are the same for every lane. They are fetched with vector loads from a
    // lane-independent address (one request per wavefront) rather than scalar loads: scalar and
    // LDS operations share one counter (lgkmcnt) and scalar loads return out of order, so a
    // scalar load in flight turns every wait for a ds_bpermute result into a full drain.
    int vz = 0;
    SSQ_OPAQUE_V(vz);
    const int2* rows2 = reinterpret_cast<const int2*>(A.prow) + vz;   // per row: row | pad << 9 | kc << 10, offset of its samples
    const int4* steps4 = reinterpret_cast<const int4*>(A.pstep) + vz;   // per step: kind | lgR << 1 | weight offset << 8, L - 1, stride, base
    const w_t* cstu = cstv + vz;

    // Software pipeline over this wavefront's steps: the records of a step are fetched while the
    // step before it is computed, its samples (or Wx and bins) half a step ahead, the
    // interpolation weights when the last taps of the step before are done. ALL loads are issued
    // unconditionally (past the last step: the last step again, results unused): the compiler
    // counts the loads in flight per path, and a path that skips some turns every wait into a
    // full drain.
    int4 sa; int2 rec[TILE_G];                // next step: its packed record, its rows
    auto load_rec = [&](const Pos& q) {
        const int g = q.st;
        sa = steps4[g];
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) rec[r] = rows2[g * TILE_G + r];
    };
    float2 xu[2][TILE_G];                     // samples of the interpolated rows / Wx of the rows read back
    // per row: the packed record (interpolated rows) or the bin (rows read back) -- one register
    // either way (16 wavefronts need the step pipeline under 128 registers)
    int xq[2][TILE_G];
    int xnv[2];                               // rows of the step that are not padding
    w_t xc[2][CSTK == 0 ? 1 : TILE_G];        // per-row weights
    int xkind[2], xbaddr[2], xwoff[2], xmask[2];
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    auto load = [&](auto BB, const Pos& q) {   // data of the step whose records are in (sa, sb, rec)
        constexpr int b = decltype(BB)::value;
        const int sax = __builtin_amdgcn_readfirstlane(sa.x);
        const int kind = sax & 1;
        xkind[b] = kind;
        const int col0 = q.tx * TILE_COLS, col = col0 + c;
        const int colc = col < (int)N ? col : (int)N - 1;         // loads stay in range
        const int nabs = A.n1 + colc, nabs0 = A.n1 + col0;
        const int lgR = (sax >> 1) & 31;
        xwoff[b] = sax >> 8; xmask[b] = (1 << lgR) - 1;
        const int q0 = nabs >> lgR, qb = (nabs0 >> lgR) - (TILE_W / 2 - 1);
        // interpolated rows: the sample this lane holds (lanes past the widest window any lane
        // needs repeat the last one); rows read back: the lane's own point
        const int wlast = (63 >> lgR) + TILE_W;
        const unsigned uidx = (unsigned)((qb + (c < wlast ? c : wlast)) & __builtin_amdgcn_readfirstlane(sa.y));
        xbaddr[b] = (q0 - (TILE_W / 2 - 1) - qb) * 4;            // lane that holds tap 0
        // Addresses: a wave-uniform 64-bit base (scalar arithmetic on the records, which arrived half
        // a step ago) + one 32-bit byte offset per lane that is the same for the four rows.
        const char* Ub8 = reinterpret_cast<const char*>(A.U + __builtin_amdgcn_readfirstlane(sa.w)
                                                        + (int64_t)q.sg * __builtin_amdgcn_readfirstlane(sa.z));
        const char* Wx8 = reinterpret_cast<const char*>(A.Wx + (int64_t)(A.sig0 + q.sg) * na * N);
        const char* kx8 = reinterpret_cast<const char*>(A.kidx + (int64_t)q.sg * na * N);
        const unsigned vo = kind ? uidx * 8u : (unsigned)colc * 8u;
        int nv = 0;
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) {
            const int2 d = rec[r];
            const int dx = __builtin_amdgcn_readfirstlane(d.x);
            const unsigned row = (unsigned)dx & 0x1FFu;
            nv += ((dx >> 9) & 1) ^ 1;
            // one 8-byte load either way: a sample of u (interpolated) or Wx (read back)
            const char* base = kind ? Ub8 + (size_t)(unsigned)__builtin_amdgcn_readfirstlane(d.y) * 8u
                                    : Wx8 + (size_t)row * (nN * 8u);
            xu[b][r] = *reinterpret_cast<const float2*>(base + vo);
            if (kind) xq[b][r] = d.x;
            else xq[b][r] = *reinterpret_cast<const unsigned short*>(kx8 + (size_t)row * (nN * 2u) + (unsigned)colc * 2u);
            if (CSTK != 0) xc[b][r] = cstu[row];
        }
        xnv[b] = nv;
    };
    ssq_f2 wt[TILE_W];                        // (phi_t, phi'_t / (R dt)) of the step in hand
    auto load_wt = [&](auto BB, const Pos& q) {
        constexpr int b = decltype(BB)::value;
        const int col = q.tx * TILE_COLS + c;
        const int nabs = A.n1 + (col < (int)N ? col : (int)N - 1);
        // (the table is stored tap pair by tap pair, [4][R] float4 per class: the 64 lanes of a load
        // read one run of consecutive phases, not 64 separate 64-byte rows)
        const float4* wp = A.wtab + (int64_t)xwoff[b] * 4 + (nabs & xmask[b]);
        const int wstride = xmask[b] + 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = wp[t * wstride];
            wt[2 * t].x = v.x; wt[2 * t].y = v.y; wt[2 * t + 1].x = v.z; wt[2 * t + 1].y = v.w;
        }
    };

    Pos pc; pc.S = 0; pc.itl = 0; pc.st = 0;
    pc.sg = bid / ntx; pc.tx = bid - pc.sg * ntx;
    int w_itl = 0, w_tx = pc.tx, w_sg = pc.sg;                // next tile to write out
    advance(pc, wv);                          // the step computed
    auto write_outs_before = [&](int itl) {   // every finished tile before tile `itl`, in order
        while (w_itl < itl) {
            write_out(w_itl, w_tx, w_sg);
            ++w_itl; w_tx += (int)gridDim.x;
            while (w_tx >= ntx) { w_tx -= ntx; ++w_sg; }
        }
    };
    Pos pl = pc;                              // a valid step for the loads past the end
    auto clampp = [&](const Pos& q) { return q.S < total ? q : pl; };
    if (pc.S < total) {
        load_rec(pc); load(B0{}, pc); load_wt(B0{}, pc);
        Pos pn = pc; advance(pn, NW);         // the step whose data are loaded next
        load_rec(clampp(pn));
        auto step = [&](auto BB, auto BN) {
            constexpr int b = decltype(BB)::value;
            ticket_priority(turn, pc.S + pc.itl);
            const int col0 = pc.tx * TILE_COLS, col = col0 + c;
            const bool colok = col < (int)N;
            const int colc = colok ? col : (int)N - 1;
            const int nabs = A.n1 + colc;
            int cell[TILE_G]; term_t vx[TILE_G], vy[TILE_G];
            Pos pnn = pn;
            if (__builtin_amdgcn_readfirstlane(xkind[b]) == 0) {
                // rows read back: Wx and the bin are there
                load(BN, clampp(pn)); load_wt(BN, clampp(pn)); advance(pnn, NW); load_rec(clampp(pnn));
#pragma unroll
                for (int r = 0; r < TILE_G; ++r) {
                    const int kk = xq[b][r] & 0xFFFF;
                    const bool act = r < xnv[b] && colok && kk != TILE_NOBIN;
                    cell[r] = act ? kk * TILE_COLS + c : scratch;
                    const w_t cs = CSTK == 0 ? (w_t)A.cst0 : xc[b][CSTK == 0 ? 0 : r];
                    vx[r] = TM::make(xu[b][r].x, cs); vy[r] = TM::make(xu[b][r].y, cs);
                }
            } else {
                char* Wx8 = reinterpret_cast<char*>(A.Wx + (int64_t)(A.sig0 + pc.sg) * na * N);
                char* dWx8 = STORE_D ? reinterpret_cast<char*>(A.dWx + (int64_t)(A.sig0 + pc.sg) * na * N) : nullptr;
                const unsigned colc8 = (unsigned)colc * 8u;
                const int baddr = xbaddr[b];
#pragma unroll
                for (int r = 0; r < TILE_G; ++r) {
                    if (r == TILE_G / 2) {
                        // the next step: its data now (its records came in at the end of the step before)
                        load(BN, clampp(pn));
                        ticket_priority(turn, pc.S + pc.itl);
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
                    if (r == TILE_G - 1) {
                        // the taps of this step are done: the next step's weights, then the records of
                        // the step after it. Loads return in issue order: what the next step needs first
                        // (samples, weights) must not queue behind loads it needs later.
                        load_wt(BN, clampp(pn)); advance(pnn, NW); load_rec(clampp(pnn));
                    }
                    const float are = are2.x, aim = aim2.x;
                    float dre = are2.y, dim = aim2.y;
                    // d/dt of e^{i theta n} a(n):  e^{i theta n} (i theta a + a'),  theta = 2 pi kc / (M dt)
                    const int xrow = __builtin_amdgcn_readfirstlane(xq[b][r]);
                    const int kcs = (int)((unsigned)xrow >> 10);               // centre bin (wave-uniform)
                    const float theta = (float)kcs * A.theta_scale;
                    dre = __builtin_fmaf(-theta, aim, dre);
                    dim = __builtin_fmaf(theta, are, dim);
                    // e^{2 i pi kc n / M}: the phase kc n mod M is exact in integers and in float
                    // (M <= 2^24, checked by the host), v_sin_f32 / v_cos_f32 take revolutions (measured on
                    // the M = 2^18 circle: max abs error 1.2e-7, as good as a float table)
                    const float rev = (float)(__umul24((unsigned)kcs, (unsigned)nabs) & (unsigned)A.mmask) * A.inv_m;   // (both < 2^24: full-rate multiply)
                    const float2 tw = make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev));
                    const float2 Wv = cmulf(tw, make_float2(are, aim));
                    const float2 Dv = cmulf(tw, make_float2(dre, dim));
                    // (rows that only pad a step repeat the previous row -- same address, same value --
                    // and lanes past the last column repeat its point; neither contributes below)
                    const bool pad = (xrow >> 9) & 1;
                    const size_t rowoff = (size_t)((unsigned)xrow & 0x1FFu) * (nN * 8u);   // wave-uniform
                    *reinterpret_cast<float2*>(Wx8 + rowoff + colc8) = Wv;
                    if (STORE_D) *reinterpret_cast<float2*>(dWx8 + rowoff + colc8) = Dv;
                    // phase transform and bin: as emit_point<LEAN> of the block kernels
                    const float cc = Wv.x, dd = Wv.y, aa = Dv.x, bb = Dv.y;
                    const float m2 = cc * cc + dd * dd, num = bb * cc - aa * dd;
                    const bool above = m2 > m2hi, below = m2 < m2lo;
                    const float w32 = fabsf(num * __builtin_amdgcn_rcpf(m2 * 6.2831855f));
                    bool ok;
                    const int kb = bin_screen_cwt<GRID>(w32, sp, omax, ok);
                    const int kf = (kb ^ fx) + fa;
                    const bool live = colok && !pad;
                    int kout = (above && live) ? kf : -1;
                    // undecided by the float32 screens (~0.05 % of the points, one row in 30): the exact
                    // double path
                    const bool pend = live && !(below | (above & ok));
                    if (__builtin_amdgcn_ballot_w64(pend)) {
                        if (pend) kout = exact_bin(Wv, Dv, sp, omax, A.gamma);
                    }
                    // (a point without contribution adds to the lane's scratch cell)
                    cell[r] = kout >= 0 ? kout * TILE_COLS + c : scratch;
                    const w_t cs = CSTK == 0 ? (w_t)A.cst0 : xc[b][CSTK == 0 ? 0 : r];
                    vx[r] = TM::make(Wv.x, cs); vy[r] = TM::make(Wv.y, cs);
                }
            }
            // the step's update, in ticket order (tiles finished before it are written out first)
            Update4Prep up4;
            update4_prepare(cell, up4);
#pragma unroll
            for (int r = 0; r < TILE_G; ++r) { keep_term(vx[r]); keep_term(vy[r]); }
            write_outs_before(pc.itl);
            const int ticket = pc.S + pc.itl;
            float2 tcell[TILE_G];
            ticket_wait_read4(turn, ticket, lds_raw, up4, tcell);
            __builtin_amdgcn_wave_barrier();
            update4_finish<TM>(lds_raw, up4, tcell, vx, vy);
            __builtin_amdgcn_wave_barrier();
            ticket_pass(turn, ticket + 1, c);
            pl = pc; pc = pn; pn = pnn;
        };
        for (;;) {
            if (pc.S >= total) break;
            step(B0{}, B1{});
            if (pc.S >= total) break;
            step(B1{}, B0{});
        }
    }
    write_outs_before(ntl);
}


// ---------------------------------------------------------------------------- host side
// wavefronts per workgroup: 12 = 3 per SIMD (168 VGPRs: the step pipeline needs ~160; at 16 wavefronts / 128
// registers it spills and measured slower)
template <int GRID, bool STORE_D, int NW, int CSTK>
static int launch_tile_c(const TilePlan& P, const TileArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    auto kern = tile_kernel<GRID, STORE_D, NW, CSTK>;
    const size_t lds = tile_lds_bytes(P.na);
    // (set at every launch: the attribute belongs to the function ON THE CURRENT DEVICE, a flag per instantiation
    // would leave a second device of the process without it; the call costs well under a microsecond)
    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
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
    const int cstk = sp.cst_f64 ? 2 : (sp.cst_uniform ? 0 : 1);
    if (cstk == 0) return launch_tile_c<GRID, STORE_D, NW, 0>(P, A, sp, nsig, stream);
    if (cstk == 1) return launch_tile_c<GRID, STORE_D, NW, 1>(P, A, sp, nsig, stream);
    return launch_tile_c<GRID, STORE_D, NW, 2>(P, A, sp, nsig, stream);
}
template <int GRID, bool STORE_D>
static int launch_tile(const TilePlan& P, const TileArgs& A, const SsqParams& sp, int nsig, hipStream_t stream) {
    return launch_tile_nw<GRID, STORE_D, 12>(P, A, sp, nsig, stream);
}


int TilePlan::run_ordered(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                          const void* cst, float cst0, const SsqParams& sp, hipStream_t stream) {
    SSQ_REQUIRE(tile_lds_bytes(na) <= 160 * 1024, "na = %lld: the ordered tile kernel's Tx tile exceeds the LDS", (long long)na);
    TileArgs A;
    A.pstep = reinterpret_cast<const int4*>(steps); A.prow = reinterpret_cast<const int2*>(rows);
    A.wtab = (const float4*)wtab; A.U = (const float2*)U; A.cst = cst;
    A.Wx = (float2*)Wx; A.dWx = (float2*)dWx; A.Tx = (float2*)Tx; A.kidx = kidx;
    A.N = N; A.na = na; A.nsteps = nsteps; A.n1 = (int)n1; A.mmask = (int)(M - 1);
    A.sig0 = sig; A.nsig = nsig; A.inv_m = 1.0f / (float)M;
    A.theta_scale = (float)(6.283185307179586 / ((double)M * dt)); A.cst0 = cst0;
    A.counters = counters;
    A.gamma = sp.gamma;
#define TILE_LAUNCH(G)                                                                      \
    return dWx ? launch_tile<G, true>(*this, A, sp, nsig, stream) : launch_tile<G, false>(*this, A, sp, nsig, stream);
    if (sp.grid == SSQ_GRID_LOG) { TILE_LAUNCH(SSQ_GRID_LOG) }
    if (sp.grid == SSQ_GRID_LOG_PIECEWISE) { TILE_LAUNCH(SSQ_GRID_LOG_PIECEWISE) }
    TILE_LAUNCH(SSQ_GRID_LIN)
#undef TILE_LAUNCH
}

}  // namespace ssq
