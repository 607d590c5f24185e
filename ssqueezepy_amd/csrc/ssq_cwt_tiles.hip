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
//   tile_kernel           one workgroup (4 wavefronts) = 64 columns x every row of one signal.
//                         A wavefront owns 16 columns: its lanes are (column c, row slot r), 4 rows
//                         per step, and its 16 columns x na bins of Tx live in LDS (na * 128 B
//                         per wavefront, 150 KiB per workgroup at na = 300: one workgroup per
//                         CU -- the LDS-resident tile is what removes the HBM round trip of Wx,
//                         and it is also what bounds the kernel to one wavefront per SIMD, so all
//                         loads are software-pipelined). Per step and lane:
//                           interpolated rows: ONE 8-byte load of u_i (the 16 lanes of a row slot
//                             hold a window of 16 consecutive samples; taps are fetched from the
//                             neighbours with ds_bpermute), 8 taps x (phi, phi') real weights kept
//                             in registers for the whole decimation class, modulation by
//                             e^{2i pi kc n / M} = (tile twiddle) x (lane twiddle), Wx stored
//                             (128-byte runs), phase transform + bin exactly as the other fused
//                             kernels do (ssq_point_math.inl);
//                           read-back rows: Wx and the 2-byte bin written by the block / exact
//                             kernels.
//                         Reassignment: ds_add_f32 into the wavefront's private tile, row slot after
//                         row slot (exec-masked, in program order), i.e. in ascending row order
//                         per cell -- the reference's summation order, so float sums are
//                         bit-identical to the CPU loop (algos.py:859-953) on the same Wx / dWx.
//                         No workgroup barrier, no global atomics; Tx is written once at the end.
//
// Compiled with -ffp-contract=off (bin indices); multiply-adds that may fuse are written as
// explicit fmaf so every instantiation rounds identically.
#include "ssq_common.h"
#include "ssq_tiles.h"
#include <algorithm>

namespace ssq {

#include "ssq_point_math.inl"

constexpr int TILE_COLS = 64;     // columns per workgroup (one per lane)
// rows per step x steps per ticket (a wavefront takes TILE_B consecutive steps). Measured on one
// box (config 2, tile stage): 4 x 2 -> 330 us; 2 x 4 -> 361 us, with 8 or with 12 wavefronts alike;
// 4 x 4 -> 413 us (20 tickets for 8 wavefronts: uneven shares, longer waits)
constexpr int TILE_G = 4;
constexpr int TILE_B = 2;
constexpr int TILE_W = 8;         // taps

struct TileArgs {
    const TileSeg* steps; const TileRow* rows;       // one TileSeg record per step
    const float4* wtab; const float2* U;
    const float* cst;
    float2* Wx; float2* dWx; float2* Tx; const unsigned short* kidx;
    int64_t N, na;
    int nsteps, n1, mmask, sig0, nsig;
    float inv_m;         // 1 / M
    float theta_scale;   // 2 pi / (M dt): theta of a row = kc * theta_scale
    float cst0;          // the reassignment weight when it is the same for every row
    unsigned long long* trace;   // tuning aid (SSQ_TILE_TRACE): shader-clock stamps of one workgroup
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
__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

// The tile of Tx (64 columns x na bins, + one scratch row for points that contribute
// nothing) is shared by all NW wavefronts of the workgroup. A lane is a column, a wavefront
// takes the steps (4 consecutive rows) of the row list round-robin: everything that depends
// on the row alone -- descriptor, theta, tile twiddle, weight -- is wavefront-uniform (SGPRs,
// scalar loads), loads and stores are 512-byte runs, and the arithmetic of different steps
// runs concurrently on the SIMDs. Only the reassignment itself is ordered, by a ticket in
// LDS: step g may update the tile once `turn == g` (acquire / release at workgroup scope
// order the tile accesses around it), so every cell receives its contributions in
// ascending row order -- the reference's -- and the float sums are bit-identical to the CPU
// loop. Inside a step the four rows' cells are read together and chained in registers when
// they coincide (same lane = same column: no cross-lane traffic), then written in row order.
// (LDS float atomics were measured first: ds_add_f32 retires about one lane per 3-4 cycles
// on gfx950; and a layout with a private 16-column tile per wavefront, lanes = 16 columns x 4
// rows: per-lane descriptors and exec-masked row slots doubled the instruction count.)
__device__ __forceinline__ void take_turn(const int* turn, int g) {
    while (__atomic_load_n(turn, __ATOMIC_ACQUIRE) != g) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void pass_turn(int* turn, int g, int lane) {
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __atomic_store_n(turn, g + 1, __ATOMIC_RELEASE);
}
// T[cell[r]] += v[r], r = 0..3 in order; `cell` of a point without contribution is the
// lane's scratch cell and its v is 0
// `src[r]`: the latest earlier row of the step on the same cell, or -1 (computed before the
// turn is taken: the comparisons do not depend on the tile)
__device__ __forceinline__ void forward4(const int (&cell)[TILE_G], int (&src)[TILE_G]) {
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) {
        src[r] = -1;
#pragma unroll
        for (int q = 0; q < r; ++q) if (cell[q] == cell[r]) src[r] = q;
    }
}
__device__ __forceinline__ void update4(float2* T, const int (&cell)[TILE_G], const float2 (&v)[TILE_G],
                                        const int (&src)[TILE_G]) {
    float2 t[TILE_G];
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) t[r] = T[cell[r]];
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) {
#pragma unroll
        for (int q = 0; q < r; ++q) if (src[r] == q) t[r] = t[q];
        t[r].x += v[r].x; t[r].y += v[r].y;
    }
#pragma unroll
    for (int r = 0; r < TILE_G; ++r) T[cell[r]] = t[r];
}

// bin of a point the float32 screens could not decide (flipped as Tx wants it), or -1 when it
// does not contribute: the exact double sequence of the CPU path. Kept out of line -- one
// copy per kernel, ~0.05 % of the points.
__device__ __attribute__((noinline)) int exact_bin(float2 W, float2 D, const SsqParams& sp, int omax, double gamma) {
    if (!(mag_of(W.x, W.y) > gamma)) return -1;
    const int ke = (int)bin_of_point_exact(D.x, D.y, W.x, W.y, sp, (int64_t)omax);
    return sp.flipud ? omax - ke : ke;
}

// trace slot: [wave][16 steps][8 stamps]
#define TILE_STAMP(j, k)                                                                        \
    do { if (tr && (j) < 16 && c == 0) tr[((size_t)wv * 16 + (j)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

// what a step needs to know about its tile (64 columns of one signal of the launch group)
struct TileCtx {
    int tx, sg;              // tile along time, signal of the group
    int colc, nabs, nabs0;   // column of the lane (clamped), its padded index, padded index of column 0
    bool colok;
    int64_t obase;           // element offset of the signal in Wx / dWx / Tx
    int64_t kbase;           // ... in the bin map of the group
};

template <int GRID, bool STORE_D, int NW, bool CSTU>
__global__ __launch_bounds__(64 * NW) void tile_kernel(TileArgs A, SsqParams sp) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int c = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t N = A.N;
    const unsigned nN = (unsigned)N;
    const int na = (int)A.na, omax = na - 1;
    float2* T = reinterpret_cast<float2*>(lds_raw);
    int* turn = reinterpret_cast<int*>(lds_raw + (size_t)(na + 1) * TILE_COLS * 8);
    int* done = turn + 1;
    for (int k = wv; k <= na; k += NW) T[k * TILE_COLS + c] = make_float2(0.f, 0.f);
    if (threadIdx.x == 0) { *turn = 0; *done = 0; }
    __syncthreads();
    const int scratch = na * TILE_COLS + c;

    // The workgroup is persistent: it walks the tiles blockIdx.x, + gridDim.x, ... One
    // workgroup fills a CU (the Tx tile), so nothing else hides the head of a tile (first
    // loads, first pair of steps: ~17 k cycles before the first update) and its tail (the last
    // updates, 150 KiB of Tx written out): a wavefront that has finished its steps of tile i
    // therefore goes on with the loads and the arithmetic of tile i + 1 and meets the others
    // again only at the ticket.
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
    auto advance = [&](TileCtx& t) {                         // the workgroup's next tile
        int tx = t.tx + (int)gridDim.x, sg = t.sg;
        while (tx >= ntx) { tx -= ntx; ++sg; }
        t = ctx_of(tx, sg);
    };
    unsigned long long* tr = (A.trace && blockIdx.x == 100) ? A.trace : nullptr;
    if (tr && threadIdx.x == 0) tr[16 * 16 * 8] = __builtin_amdgcn_s_memtime();

    const float g2 = (float)(A.gamma * A.gamma);
    const float m2hi = g2 * 1.000004f, m2lo = g2 * 0.999996f;
    const int fx = sp.flipud ? -1 : 0, fa = sp.flipud ? na : 0;

    // Step and row records and the reassignment weight are the same for every lane. They are
    // fetched with vector loads from a lane-independent address (one request per wavefront)
    // rather than scalar loads: scalar and LDS operations share one counter (lgkmcnt) and
    // scalar loads return out of order, so a scalar load in flight turns every wait for a
    // ds_bpermute result into a full drain.
    int vz = 0;
    SSQ_OPAQUE_V(vz);
    const int4* rows4 = reinterpret_cast<const int4*>(A.rows) + vz;
    const int4* steps4 = reinterpret_cast<const int4*>(A.steps) + vz;
    const float* cstv = A.cst + vz;

    // This wavefront's steps of a tile: the wv-th pair of consecutive steps of every NW pairs
    // (the step list has an even number of steps) -- the ticket is taken once per pair.
    // Consecutive pairs of one wavefront are NW pairs apart in the row list and usually of
    // different decimation classes, so ONE software pipeline runs over all of them, across
    // tiles: records two steps ahead (they hold the addresses), samples and interpolation
    // weights one step ahead.
    const int npairs = A.nsteps / TILE_B;                    // groups of TILE_B steps ("pairs" when it was 2)
    const int mypairs = npairs > wv ? (npairs - wv + NW - 1) / NW : 0;
    const int nmine = TILE_B * mypairs;                      // steps per tile
    const int jtot = nmine * ntl;
    auto gstep = [&](int j) { return TILE_B * (wv + (j / TILE_B) * NW) + (j % TILE_B); };
    auto tbase = [&](int itl) { return itl * (npairs + 2); };   // ticket of the tile's first pair

    // rows k = wv, wv + NW, ... of the finished tile go to Tx and are cleared; the last
    // wavefront to finish opens the next tile's tickets
    auto write_out = [&](const TileCtx& t, int itl) {
        take_turn(turn, tbase(itl) + npairs);
        float2* Tx = A.Tx + t.obase;
        const int col = t.tx * TILE_COLS + c;
        for (int k0 = wv; k0 < na; k0 += 8 * NW) {          // 8 rows in flight per wavefront
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + u * NW; v[u] = T[(k < na ? k : na) * TILE_COLS + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * NW;
                if (k < na) {
                    T[k * TILE_COLS + c] = make_float2(0.f, 0.f);
                    if (t.colok) Tx[(unsigned)k * nN + (unsigned)col] = v[u];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (c == 0) {
            const int before = __atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL);
            if (before + 1 == NW * (itl + 1)) __atomic_store_n(turn, tbase(itl + 1), __ATOMIC_RELEASE);
        }
    };
    if (nmine == 0) {                                        // more wavefronts than pairs of steps
        TileCtx t = ctx_of((int)blockIdx.x % ntx, (int)blockIdx.x / ntx);
        for (int itl = 0; itl < ntl; ++itl) { write_out(t, itl); advance(t); }
        return;
    }

    // Software pipeline over this wavefront's steps, across tiles. Registers decide how many
    // wavefronts a SIMD holds (168 for three), so the pipeline keeps only what it must: the
    // records of the NEXT step (loaded at the top of a step), the samples of the next step
    // (loaded in the middle of a step, when the records have arrived); interpolation weights
    // are fetched when the class changes (once per pair of steps, L2-resident), theta and the
    // uniform reassignment weight are derived / passed as scalars.
    int4 sa, sb, rec[TILE_G];                 // next step: (kind, first, nsteps, lgR | wtab, stride, L-1, base), rows
    auto load_rec = [&](int j) {
        const int g = gstep(j);
        sa = steps4[2 * g]; sb = steps4[2 * g + 1];
#pragma unroll
        for (int r = 0; r < TILE_G; ++r) rec[r] = rows4[g * TILE_G + r];
    };
    float2 xu[2][TILE_G];
    int xr[2][TILE_G], xkc[2][TILE_G]; unsigned short xk[2][TILE_G];
    float xc[2][CSTU ? 1 : TILE_G];
    int xkind[2], xbaddr[2], xwoff[2], xmask[2];
    auto load = [&](int b, const TileCtx& t) {
        const int kind = sa.x;
        xkind[b] = kind;
        if (kind == 0) {                                     // rows read back: Wx, bin
            const float2* Wx = A.Wx + t.obase;
            const unsigned short* kidx = A.kidx + t.kbase;
#pragma unroll
            for (int r = 0; r < TILE_G; ++r) {
                const int row = rec[r].x & 0xFFFF;
                const unsigned o = (unsigned)row * nN + (unsigned)t.colc;
                xr[b][r] = rec[r].x;
                xu[b][r] = Wx[o];
                xk[b][r] = kidx[o];
                if (!CSTU) xc[b][r] = cstv[row];
            }
        } else {                                             // rows interpolated
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
                if (!CSTU) xc[b][r] = cstv[d.x & 0xFFFF];
            }
        }
    };

    TileCtx tc = ctx_of((int)blockIdx.x % ntx, (int)blockIdx.x / ntx);   // tile of the step computed
    TileCtx tl = tc;                                                       // tile of the step loaded
    TileCtx tp = tc;                                                       // previous tile (to write out)
    int jl = 0, jc = 0, itl = 0;              // step inside the tile (loads / arithmetic), tile count
    load_rec(0); load(0, tl);
    if (++jl == nmine) { jl = 0; advance(tl); }
    load_rec(jl);
    ssq_f2 wt[TILE_W];                        // (phi_t, phi'_t / (R dt)) of the class in hand
    int wt_off = -1, wt_phase = -1;
    for (int jj = 0; jj < jtot; jj += TILE_B) {
        int cells[TILE_B][TILE_G]; float2 vs[TILE_B][TILE_G];
#pragma unroll
        for (int st = 0; st < TILE_B; ++st) {
            const int b = st & 1;                               // sample buffer of this step
            if (tr && itl == 2) TILE_STAMP(jc + st, 0);
            int (&cell)[TILE_G] = cells[st]; float2 (&v)[TILE_G] = vs[st];
            // the next step: its samples now (its records came in during the previous step),
            // then the records of the one after
            auto prefetch = [&]() {
                if (jj + st + 1 < jtot) load(b ^ 1, tl);
                if (++jl == nmine) { jl = 0; advance(tl); }
                load_rec(jl);
            };
            if (__builtin_amdgcn_readfirstlane(xkind[b]) == 0) {
                prefetch();
#pragma unroll
                for (int r = 0; r < TILE_G; ++r) {
                    const int kk = xk[b][r];
                    const bool act = xr[b][r] >= 0 && tc.colok && kk != 0xFFFF;
                    const float cs = act ? (CSTU ? A.cst0 : xc[b][CSTU ? 0 : r]) : 0.f;
                    cell[r] = act ? kk * TILE_COLS + c : scratch;
                    v[r] = make_float2(xu[b][r].x * cs, xu[b][r].y * cs);
                }
            } else {
                float2* Wx = A.Wx + tc.obase;
                float2* dWx = STORE_D ? A.dWx + tc.obase : nullptr;
                const int phase = tc.nabs & xmask[b];
                if (__builtin_amdgcn_readfirstlane(xwoff[b]) != wt_off || phase != wt_phase) {
                    const float4* wp = A.wtab + (int64_t)(xwoff[b] + phase) * 4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 q = wp[t];
                        wt[2 * t].x = q.x; wt[2 * t].y = q.y; wt[2 * t + 1].x = q.z; wt[2 * t + 1].y = q.w;
                    }
                    wt_off = __builtin_amdgcn_readfirstlane(xwoff[b]); wt_phase = phase;
                }
                const int baddr = xbaddr[b];
                unsigned pend = 0;
                float2 Wk[TILE_G], Dk[TILE_G];
#pragma unroll
                for (int r = 0; r < TILE_G; ++r) {
                    if (r == TILE_G / 2) prefetch();
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
                        if (tr && itl == 2 && r == 0) TILE_STAMP(jc + st, 1);
                        if (tr && itl == 2 && r == 1) TILE_STAMP(jc + st, 7);
#pragma unroll
                        for (int t = 0; t < TILE_W; ++t) {
                            ssq_f2 sv; sv.x = __int_as_float(fr[t]); sv.y = __int_as_float(fi[t]);
                            if (t == 0) { SSQ_PK_MUL_LO(are2, wt[0], sv); SSQ_PK_MUL_HI(aim2, wt[0], sv); }
                            else { SSQ_PK_FMA_LO(are2, wt[t], sv); SSQ_PK_FMA_HI(aim2, wt[t], sv); }
                        }
                    }
                    const float are = are2.x, aim = aim2.x;
                    float dre = are2.y, dim = aim2.y;
                    // d/dt of e^{i theta n} a(n):  e^{i theta n} (i theta a + a'),  theta = 2 pi kc / (M dt)
                    const float theta = (float)xkc[b][r] * A.theta_scale;
                    dre = __builtin_fmaf(-theta, aim, dre);
                    dim = __builtin_fmaf(theta, are, dim);
                    // e^{2 i pi kc n / M}: the phase kc n mod M is exact in integers and in float
                    // (M <= 2^24), v_sin_f32 / v_cos_f32 take revolutions (measured on the
                    // M = 2^18 circle: max abs error 1.2e-7, as good as a float table)
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
                    const bool act = above && live;
                    // undecided by the float32 screens (~0.05 % of the points): the exact double
                    // path, looked at once per step
                    if (live && !(below | (above & ok))) pend |= 1u << r;
                    // (a point without contribution adds 0 * Wx to the lane's scratch cell)
                    const float cs = act ? (CSTU ? A.cst0 : xc[b][CSTU ? 0 : r]) : 0.f;
                    cell[r] = act ? kf * TILE_COLS + c : scratch;
                    v[r] = make_float2(Wv.x * cs, Wv.y * cs);
                    Wk[r] = Wv; Dk[r] = Dv;
                }
                if (__builtin_amdgcn_ballot_w64(pend != 0)) {
#pragma unroll
                    for (int r = 0; r < TILE_G; ++r)
                        if (pend & (1u << r)) {
                            const int ke = exact_bin(Wk[r], Dk[r], sp, omax, A.gamma);
                            const float cs = ke >= 0 ? (CSTU ? A.cst0 : xc[b][CSTU ? 0 : r]) : 0.f;
                            cell[r] = ke >= 0 ? ke * TILE_COLS + c : scratch;
                            v[r] = make_float2(Wk[r].x * cs, Wk[r].y * cs);
                        }
                }
            }
            if (tr && itl == 2) TILE_STAMP(jc + st, 2);
        }
        // the pair's update, in ticket order (the previous tile is written out first)
        int src[TILE_B][TILE_G];
#pragma unroll
        for (int st = 0; st < TILE_B; ++st) forward4(cells[st], src[st]);
        if (jc == 0 && itl > 0) {
            if (tr && itl == 2) TILE_STAMP(0, 5);
            write_out(tp, itl - 1);
            if (tr && itl == 2) TILE_STAMP(0, 6);
        }
        const int g = tbase(itl) + gstep(jc) / TILE_B;
        take_turn(turn, g);
        if (tr && itl == 2) TILE_STAMP(jc, 3);
#pragma unroll
        for (int st = 0; st < TILE_B; ++st) update4(T, cells[st], vs[st], src[st]);
        pass_turn(turn, g, c);
        if (tr && itl == 2) TILE_STAMP(jc, 4);
        jc += TILE_B;
        if (jc == nmine) { jc = 0; ++itl; tp = tc; advance(tc); }
    }
    write_out(tp, ntl - 1);
    if (tr && c == 0) tr[16 * 16 * 8 + 20 + wv] = __builtin_amdgcn_s_memtime();
}

// ---------------------------------------------------------------------------- host side
int TilePlan::create(const ssq_cwt_tiles_desc& d, int64_t M_, int64_t N_, int64_t n1_, int64_t na_, int group_,
                     double dt_, int64_t& bytes) {
    M = M_; N = N_; n1 = n1_; na = na_; group = group_; dt = dt_;
    nsegs = d.n_segs; nsteps = d.n_steps; n_irows = d.n_irows; u_total = d.u_total;
    SSQ_REQUIRE(nsegs >= 1 && nsteps >= TILE_B && nsteps % TILE_B == 0 && n_irows >= 1 && d.n_classes >= 1,
                "empty tile tables, or the number of steps is not a multiple of %d", TILE_B);
    SSQ_REQUIRE((size_t)(na + 1) * TILE_COLS * 8 + 16 <= 160 * 1024 && na * N < ((int64_t)1 << 29), "na = %lld: the Tx tile exceeds the LDS",
                (long long)na);
    SSQ_REQUIRE((int64_t)group * u_total < ((int64_t)1 << 31), "tile intermediates exceed 2^31 entries");
    auto up = [&](void** dst, const void* src, size_t nbytes) -> int {
        SSQ_CHECK_HIP(hipMalloc(dst, nbytes ? nbytes : 1));
        if (nbytes) SSQ_CHECK_HIP(hipMemcpy(*dst, src, nbytes, hipMemcpyHostToDevice));
        bytes += (int64_t)nbytes;
        return 0;
    };
    int rc;
    static_assert(sizeof(TileSeg) == 32 && sizeof(TileRow) == 16 && sizeof(TileIRow) == 32, "table layout");
    {   // one record per step (a wavefront's consecutive steps are usually of different segments)
        std::vector<TileSeg> hs((size_t)nsteps);
        const TileSeg* sg = reinterpret_cast<const TileSeg*>(d.segs);
        int64_t covered = 0;
        for (int i = 0; i < nsegs; ++i) {
            SSQ_REQUIRE(sg[i].first == covered && sg[i].nsteps >= 1 && sg[i].first + sg[i].nsteps <= nsteps,
                        "tile segment %d does not continue the step list", i);
            for (int t = 0; t < sg[i].nsteps; ++t) hs[(size_t)sg[i].first + t] = sg[i];
            covered += sg[i].nsteps;
        }
        SSQ_REQUIRE(covered == nsteps, "tile segments cover %lld of %d steps", (long long)covered, nsteps);
        if ((rc = up((void**)&steps, hs.data(), sizeof(TileSeg) * nsteps))) return rc;
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
    for (size_t c = 0; c < cls.size(); ++c) {
        FftPlan fp;
        rc = fp.create(1, SSQ_F32, (size_t)cls[c].L, (size_t)(group * cls[c].nrows), 1.0);
        if (rc) return rc;
        bytes += (int64_t)fp.work_bytes;
        ffts.push_back(fp);
    }
    for (int t = 0; t < 5; ++t) n_items_tile[t] = d.n_items_tile[t];
    n_exact_tile = d.n_exact_tile;
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
    void* ptrs[] = {steps, rows, irows, wtab, tbank, U};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    steps = nullptr; rows = nullptr; irows = nullptr; wtab = tbank = U = nullptr;
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

template <int GRID, bool STORE_D, int NW, bool CSTU>
static int launch_tile_c(const TileArgs& A, const SsqParams& sp, int64_t N, int64_t na, int nsig, hipStream_t stream);
template <int GRID, bool STORE_D, int NW>
static int launch_tile_k(const TileArgs& A, const SsqParams& sp, int64_t N, int64_t na, int nsig, hipStream_t stream) {
    if (sp.cst_uniform) return launch_tile_c<GRID, STORE_D, NW, true>(A, sp, N, na, nsig, stream);
    return launch_tile_c<GRID, STORE_D, NW, false>(A, sp, N, na, nsig, stream);
}
template <int GRID, bool STORE_D, int NW, bool CSTU>
static int launch_tile_c(const TileArgs& A, const SsqParams& sp, int64_t N, int64_t na, int nsig, hipStream_t stream) {
    auto kern = tile_kernel<GRID, STORE_D, NW, CSTU>;
    const size_t lds = (size_t)(na + 1) * TILE_COLS * 8 + 16;
    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent workgroups, one per CU (the tile fills the LDS)
    static const int ncu = [] {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
        if (const char* e = getenv("SSQ_TILE_GRID")) return atoi(e) > 0 ? atoi(e) : pr.multiProcessorCount;
        return pr.multiProcessorCount;
    }();
    const int64_t ntot = ((N + TILE_COLS - 1) / TILE_COLS) * nsig;
    const dim3 grid((unsigned)std::min<int64_t>(ntot, ncu));
    TileArgs B = A; B.nsig = nsig;
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, B, sp);
    SSQ_LAUNCH_CHECK();
    return 0;
}
// wavefronts per workgroup: 8 by default (SSQ_TILE_K = 1 -> 4: tuning aid; 12 need more than the
// 168 VGPRs three wavefronts per SIMD leave and measured no faster)
template <int GRID, bool STORE_D>
static int launch_tile(const TileArgs& A, const SsqParams& sp, int64_t N, int64_t na, int nsig, hipStream_t stream) {
    static const int k = [] { const char* e = getenv("SSQ_TILE_K"); int v = e ? atoi(e) : 2; return v < 1 || v > 3 ? 2 : v; }();
    if (k == 1) return launch_tile_k<GRID, STORE_D, 4>(A, sp, N, na, nsig, stream);
    return launch_tile_k<GRID, STORE_D, 8>(A, sp, N, na, nsig, stream);
}

int TilePlan::run(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                  const void* cst, float cst0, const SsqParams& sp, hipStream_t stream) {
    TileArgs A;
    A.steps = steps; A.rows = rows;
    A.wtab = (const float4*)wtab; A.U = (const float2*)U; A.cst = (const float*)cst;
    A.Wx = (float2*)Wx; A.dWx = (float2*)dWx; A.Tx = (float2*)Tx; A.kidx = kidx;
    A.N = N; A.na = na; A.nsteps = nsteps; A.n1 = (int)n1; A.mmask = (int)(M - 1); A.sig0 = sig; A.inv_m = 1.0f / (float)M;
    A.theta_scale = (float)(6.283185307179586 / ((double)M * dt)); A.cst0 = cst0;
    A.gamma = sp.gamma;
    static unsigned long long* trace_buf = nullptr;
    const char* trace_path = getenv("SSQ_TILE_TRACE");
    if (trace_path && !trace_buf) { SSQ_CHECK_HIP(hipMalloc((void**)&trace_buf, 8 * (16 * 16 * 8 + 64))); }
    if (trace_buf) SSQ_CHECK_HIP(hipMemsetAsync(trace_buf, 0, 8 * (16 * 16 * 8 + 64), stream));
    A.trace = trace_buf;
    auto dump_trace = [&]() -> int {
        if (!trace_buf) return 0;
        SSQ_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h(16 * 16 * 8 + 64);
        SSQ_CHECK_HIP(hipMemcpy(h.data(), trace_buf, 8 * h.size(), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(trace_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        return 0;
    };
#define TILE_LAUNCH(G)                                                                      \
    { int rc_ = dWx ? launch_tile<G, true>(A, sp, N, na, nsig, stream)                     \
                    : launch_tile<G, false>(A, sp, N, na, nsig, stream);                   \
      return rc_ ? rc_ : dump_trace(); }
    if (sp.grid == SSQ_GRID_LOG) { TILE_LAUNCH(SSQ_GRID_LOG) }
    if (sp.grid == SSQ_GRID_LOG_PIECEWISE) { TILE_LAUNCH(SSQ_GRID_LOG_PIECEWISE) }
    TILE_LAUNCH(SSQ_GRID_LIN)
#undef TILE_LAUNCH
}

}  // namespace ssq
