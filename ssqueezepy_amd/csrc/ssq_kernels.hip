// ssq_kernels.hip -- reassignment / phase / framing / padding kernels for gfx950 and
// the runtime half of the C ABI (include/ssq_hip.h).
//
// Compiled with -ffp-contract=off: the per-point arithmetic here is specified
// operation by operation (see oracle/ssq_oracle.c, "numba typing") so that bin
// indices are reproducible bit-for-bit against the CPU path; no fused multiply-adds
// may be introduced behind our back.
//
// Kernel inventory (all HBM-bound streaming kernels; no MFMA -- there is no dense
// contraction on this path):
//   accumulate_tile_kernel   fused phase transform + bin map + accumulate. One
//       wavefront (64 lanes) owns a tile of TC time columns x all `na` frequency
//       bins of Tx, held in LDS (<=160 KiB/CU on gfx950); it streams rows of
//       Wx/dWx through registers with several loads in flight per lane, applies
//       updates in ascending row order (the reference's summation order) and writes
//       the tile out once. HBM traffic = read Wx + dWx (or w / bin map) once, write
//       Tx once -- no read-modify-write, no atomics, deterministic.
//   accumulate_global_kernel fallback for `na` too large for an LDS tile.
//   phase_kernel             w = |Im(dWx/Wx)|/2pi (CWT) or |Sfs - ...| (STFT).
//   replace_under_abs_kernel, buffer_kernel, pad_kernel.
#include "ssq_common.h"
#include <cstdlib>
#include <mutex>
#include <string>

namespace ssq {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

#include "ssq_point_math.inl"

// out += z * cst   in the CPU path's arithmetic
template <typename T, bool CST64>
__device__ __forceinline__ void weighted_add(T& o_re, T& o_im, T c, T d, const void* cst, int64_t i) {
    if constexpr (CST64 && sizeof(T) == 4) {
        double w = ((const double*)cst)[i];
        o_re = (T)((double)o_re + (double)c * w);
        o_im = (T)((double)o_im + (double)d * w);
    } else {
        T w = ((const T*)cst)[i];
        o_re = o_re + c * w;
        o_im = o_im + d * w;
    }
}

// Per-point side input of the accumulate kernels, loaded up front so that the loads
// of several rows are in flight together.
template <typename T, int BINSRC> struct SideVal;
template <typename T> struct SideVal<T, BIN_FROM_DWX> {
    static constexpr size_t stride = 2 * sizeof(T);
    T a, b;
    __device__ __forceinline__ void load(const void* src, int64_t q, int64_t, int64_t, int64_t) {
        const T* dz = (const T*)src + 2 * q; a = dz[0]; b = dz[1];
    }
};
template <typename T> struct SideVal<T, BIN_FROM_W> {
    static constexpr size_t stride = sizeof(T);
    T w;
    __device__ __forceinline__ void load(const void* src, int64_t q, int64_t, int64_t, int64_t) { w = ((const T*)src)[q]; }
};
template <typename T> struct SideVal<T, BIN_FROM_KIDX> {
    static constexpr size_t stride = 2;
    unsigned short k;
    __device__ __forceinline__ void load(const void* src, int64_t q, int64_t, int64_t, int64_t) {
        k = ((const unsigned short*)src)[q];
    }
};

// One point of the fused kernel: returns bin (or -1) for row i
template <typename T, int BINSRC, bool STFT>
__device__ __forceinline__ int64_t point_bin(T c, T d, const SideVal<T, BINSRC>& sv, int64_t i,
                                             const T* Sfs, const SsqParams& sp, int64_t omax) {
    int64_t k;
    if constexpr (BINSRC == BIN_FROM_DWX) {
        if (!mag_gt(c, d, sp.gamma)) return -1;
        T sf = T(0);
        if constexpr (STFT) sf = Sfs[i];
        k = bin_of_point(sv.a, sv.b, c, d, STFT, sf, sp, omax);
    } else if constexpr (BINSRC == BIN_FROM_W) {
        if (isinf(sv.w)) return -1;
        k = bin_from_stored_w(sv.w, sp, omax);
    } else {
        if (sv.k == 0xFFFFu) return -1;
        return (int64_t)sv.k;             // already flipped by the producer
    }
    return sp.flipud ? omax - k : k;
}

// ------------------------------------------------- accumulate, LDS-tile form
// block = 64 threads = RL row-lanes x TC columns; tile[k][c] complex<T> in LDS.
template <typename T, int BINSRC, bool STFT, bool CST64, int TC>
__global__ __launch_bounds__(64) void accumulate_tile_kernel(
    const T* __restrict__ Wx, const void* __restrict__ src, const T* __restrict__ Sfs,
    T* __restrict__ Tx, const void* __restrict__ cst, SsqParams sp, int64_t na, int64_t n,
    int32_t* __restrict__ kmap) {
    constexpr int RL = 64 / TC;
    constexpr int U = 4;                       // row batches in flight per lane
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* tile = reinterpret_cast<T*>(lds_raw);   // [na][TC][2]

    const int lane = threadIdx.x;
    const int c = lane % TC, rl = lane / TC;
    const int64_t j = (int64_t)blockIdx.x * TC + c;
    const int64_t b = blockIdx.y;
    const bool col_ok = j < n;
    const int64_t omax = na - 1;
    const int64_t base = b * na * n;

    for (int64_t t = lane; t < na * TC; t += 64) {
        tile[2 * t] = T(0);
        tile[2 * t + 1] = T(0);
    }
    __builtin_amdgcn_wave_barrier();

    for (int64_t i0 = 0; i0 < na; i0 += RL * U) {
        T zc[U], zd[U];
        SideVal<T, BINSRC> sv[U];
        int64_t kk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = i0 + u * RL + rl;
            bool ok = col_ok && i < na;
            zc[u] = T(0); zd[u] = T(0); kk[u] = -1;
            if (ok) {
                int64_t q = base + i * n + j;
                zc[u] = Wx[2 * q];
                zd[u] = Wx[2 * q + 1];
                sv[u].load(src, q, i, j, na);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = i0 + u * RL + rl;
            bool ok = col_ok && i < na;
            if (ok) {
                kk[u] = point_bin<T, BINSRC, STFT>(zc[u], zd[u], sv[u], i, Sfs, sp, omax);
                if (kmap) kmap[base + i * n + j] = (int32_t)kk[u];
            }
        }
        // apply in ascending row order: batch u, then row-lane r
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                if (rl == r && kk[u] >= 0) {
                    T* o = tile + 2 * (kk[u] * TC + c);
                    T ore = o[0], oim = o[1];
                    weighted_add<T, CST64>(ore, oim, zc[u], zd[u], cst, i0 + u * RL + r);
                    o[0] = ore; o[1] = oim;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (col_ok) {
        for (int64_t k = rl; k < na; k += RL) {
            int64_t q = base + k * n + j;
            Tx[2 * q] = tile[2 * (k * TC + c)];
            Tx[2 * q + 1] = tile[2 * (k * TC + c) + 1];
        }
    }
}

// ------------------------------------- accumulate, LDS-tile form, fast variant
// Same contract as accumulate_tile_kernel, organised for latency hiding. The chain
// "load -> bin -> LDS read -> fold -> LDS write" of one column is serial by
// definition (ascending-row summation order), so the kernel (a) makes the chain short
// and (b) keeps many independent chains per SIMD:
//   * a wavefront owns WC adjacent columns x RL row-lanes (RL = 8, WC = 8 for float32;
//     RL = 16, WC = 4 for float64; lane = RL*col + rl); the RL rows of a step are combined
//     in registers: every lane reads its target cell once, folds in -- in ascending row
//     order, the reference's summation order, bit for bit -- the terms of the lower
//     row-lanes of its column that hit the same cell (DPP row_shr moves), and only the
//     highest lane of a cell writes it back: one LDS read + one LDS write per RL rows;
//   * a workgroup is 16/WC such wavefronts (16 columns, 128-byte row segments between
//     them in float32); its Tx tile (na x 16 cells) lives in LDS, 4 workgroups per CU, and
//     the wavefronts never synchronise with each other;
//   * a few row batches are kept in flight in registers per lane (rolling prefetch of
//     depth U; loads unconditional so that the compiler can count them);
//   * cells are stored skewed (cell (k, c) at WC*k + ((c + k) & (WC-1)) of the wave's
//     slab) to spread the columns of a wave over LDS banks.
template <int CTRL> __device__ __forceinline__ int dpp_mov(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(dpp_mov<CTRL>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double old, double v) {
    int lo = dpp_mov<CTRL>(__double2loint(old), __double2loint(v));
    int hi = dpp_mov<CTRL>(__double2hiint(old), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// the additive term of one point and how it is folded into a cell, in the CPU path's
// arithmetic: float32 data with a float64 weight vector accumulates through double
template <typename T, bool CST64> struct Term {
    using type = T;
    using wtype = T;
    static __device__ __forceinline__ T make(T z, T w) { return z * w; }
    static __device__ __forceinline__ T fold(T o, T t) { return o + t; }
};
template <> struct Term<float, true> {
    using type = double;
    using wtype = double;
    static __device__ __forceinline__ double make(float z, double w) { return (double)z * w; }
    static __device__ __forceinline__ float fold(float o, double t) { return (float)((double)o + t); }
};

// fold in the terms of row-lanes rl-15 .. rl-1 that target the same cell, ascending.
// The match test of distance N yields a wave mask (one compare); from it come, for
// free on the scalar unit, (a) the decision whether anything has to move at this
// distance and (b) the "a higher row-lane hits my cell" mask: lane L+N matching at
// distance N is lane L having a higher partner at distance N (row_shr never crosses a
// 16-lane row, so mask >> N stays inside the column).
// SC (scalar combine): the per-distance decision is one DPP move + one compare, the rest
// scalar -- measured faster with 8 row-lanes (245 vs 263 us), slower with 16 (263 vs 257),
// so each layout keeps the form that suits it.
template <int N, bool SC, typename TM, typename T, typename term_t>
struct FoldLower {
    static __device__ __forceinline__ void run(int k, unsigned long long valid, term_t tr, term_t ti,
                                               T& ore, T& oim, unsigned long long& higher) {
        // row_shr:N with bound_ctrl: lanes without a source in their row read 0. Keys are
        // bin + 1 (0 = "takes no part", `valid` = the wave mask of k != 0), so a lane that
        // takes part never matches a missing source.
        const int ks = __builtin_amdgcn_update_dpp(0, k, 0x110 + N, 0xF, 0xF, true);
        bool m;
        unsigned long long mask;
        if constexpr (SC) {
            const bool e = ks == k;
            mask = __builtin_amdgcn_ballot_w64(e) & valid;
            m = e & (k != 0);
        } else {
            m = (ks == k) & (k != 0);
            mask = __ballot(m);
        }
        if (mask) {                                         // wave-uniform
            higher |= mask >> N;
            term_t rs = dpp_mov<0x110 + N>(term_t(0), tr), is = dpp_mov<0x110 + N>(term_t(0), ti);
            if (m) { ore = TM::fold(ore, rs); oim = TM::fold(oim, is); }
        }
        FoldLower<N - 1, SC, TM, T, term_t>::run(k, valid, tr, ti, ore, oim, higher);
    }
};
template <bool SC, typename TM, typename T, typename term_t>
struct FoldLower<0, SC, TM, T, term_t> {
    static __device__ __forceinline__ void run(int, unsigned long long, term_t, term_t, T&, T&,
                                               unsigned long long&) {}
};

// RL = row-lanes per column (16 or 8): a wavefront covers WC = 64/RL columns and the
// 16-column tile takes 16/WC wavefronts. With RL = 8 two columns share a 16-lane DPP
// row; their keys are made distinct so that a shift across the boundary never matches.
// Fold work per column falls with RL (RL-1 distances per RL rows) and so does the
// number of resident wavefronts (LDS per wavefront grows with WC).
template <typename T, int BINSRC, bool STFT, bool CST64, int U, int WPS, int RL, int TC = 16>
__global__ __launch_bounds__(64 * (TC / (64 / RL)), WPS) void accumulate_tile16_kernel(
    const T* __restrict__ Wx, const void* __restrict__ src, const T* __restrict__ Sfs,
    T* __restrict__ Tx, const void* __restrict__ cst, SsqParams sp, int64_t na64, int64_t n64,
    int32_t* __restrict__ kmap) {
    constexpr int WC = 64 / RL;                    // columns per wavefront
    constexpr int NTH = 64 * (TC / WC);            // threads per workgroup
    using TM = Term<T, CST64>;
    using term_t = typename TM::type;
    using w_t = typename TM::wtype;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int na = (int)na64, n = (int)n64;        // host guarantees na * n < 2^31
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T* tile = reinterpret_cast<T*>(lds_raw);       // [4 waves][na][4][2], cells skewed
    T* slab = tile + (size_t)wave * na * WC * 2;
    const int cl = lane / RL, rl = lane % RL;
    const int colkey = (RL < 16) ? ((cl & (16 / RL - 1)) << 20) : 0;   // column tag inside a DPP row

    // XCD-aware tile order (workgroup b runs on XCD b % 8: used for speed only):
    // consecutive 16-column tiles are issued to the same XCD back to back
    const int per = gridDim.x >> 3;                // grid.x is a multiple of 8
    const int tile_id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile_id * TC >= n) return;
    const int j = tile_id * TC + wave * WC + cl;
    const bool col_ok = j < n;
    const int64_t omax = na - 1;
    const size_t boff = (size_t)blockIdx.y * (size_t)na * (size_t)n;
    const T* Wb = Wx + 2 * boff;
    T* Tb = Tx + 2 * boff;
    int32_t* kb = kmap ? kmap + boff : nullptr;
    const char* sb = (const char*)src + boff * SideVal<T, BINSRC>::stride;

    for (int t = lane; t < na * WC * 2; t += 64) slab[t] = T(0);
    __builtin_amdgcn_wave_barrier();

    // rolling prefetch: slot u holds rows i0 + 16*u + rl; after a slot is consumed the
    // row 16*U further down is requested into it, so 16*U rows per column are in flight
    T zc[U], zd[U];
    w_t wt[U];
    SideVal<T, BINSRC> sv[U];
    const bool uni = sp.cst_uniform != 0;          // scalar weight: one load per kernel
    const w_t w0 = ((const w_t*)cst)[0];
    // Loads are unconditional (row and column clamped into the array, the point is
    // discarded at its use when it lies outside): with a branch around them the compiler
    // loses count of the loads in flight and drains them all (s_waitcnt vmcnt(0)) before
    // every use, which turns the rolling prefetch into one batch at a time.
    const int jc = col_ok ? j : n - 1;
    auto request = [&](int u, int i) {
        const int ic = i < na ? i : na - 1;
        const unsigned q = (unsigned)ic * (unsigned)n + (unsigned)jc;
        zc[u] = Wb[2 * (size_t)q];
        zd[u] = Wb[2 * (size_t)q + 1];
        sv[u].load(sb, q, ic, jc, na);
        if (!uni) wt[u] = ((const w_t*)cst)[ic];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) request(u, u * RL + rl);

    for (int i0 = 0; i0 < na; i0 += RL * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * RL + rl;
            int k = -1;
            if (col_ok && i < na) {
                k = (int)point_bin<T, BINSRC, STFT>(zc[u], zd[u], sv[u], i, Sfs, sp, omax);
                if (kb) kb[(unsigned)i * (unsigned)n + (unsigned)j] = k;
            }
            term_t tr = term_t(0), ti = term_t(0);
            T ore = T(0), oim = T(0);
            T* cell = slab;
            if (k >= 0) {
                const w_t wsel = uni ? w0 : wt[u];
                tr = TM::make(zc[u], wsel);
                ti = TM::make(zd[u], wsel);
                cell = slab + 2 * (k * WC + ((cl + k) & (WC - 1)));
                ore = cell[0]; oim = cell[1];
            }
            request(u, i + RL * U);                 // refill the slot
            unsigned long long higher = 0;
            const int key = k >= 0 ? (k | colkey) + 1 : 0;
            FoldLower<RL - 1, (RL < 16), TM, T, term_t>::run(key, __builtin_amdgcn_ballot_w64(key != 0), tr, ti,
                                                             ore, oim, higher);
            ore = TM::fold(ore, tr); oim = TM::fold(oim, ti);
            const bool last = !((higher >> lane) & 1ull);
            if (k >= 0 && last) { cell[0] = ore; cell[1] = oim; }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // write-out by the whole workgroup: every row of the tile leaves as one 16-column
    // segment (a full 128-byte line in float32), 16 rows per pass
    __syncthreads();
    {
        const int cc = threadIdx.x % TC, rr = threadIdx.x / TC;
        const int jj = tile_id * TC + cc;
        const T* ws = tile + (size_t)(cc / WC) * na * WC * 2;        // owning wave's slab
        const int c4 = cc % WC;
        if (jj < n) {
#pragma unroll 4
            for (int k = rr; k < na; k += NTH / TC) {
                const T* cell = ws + 2 * (k * WC + ((c4 + k) & (WC - 1)));
                size_t q = (size_t)((unsigned)k * (unsigned)n + (unsigned)jj);
                Tb[2 * q] = cell[0];
                Tb[2 * q + 1] = cell[1];
            }
        }
    }
}

// ------------------------------------------- accumulate, LDS-tile form, quad variant
// Same tile and same exactness contract as accumulate_tile16_kernel, organised the
// other way round: one wavefront owns the whole 16-column tile, lane = 4*col + rl, so
// the four rows of a step that belong to one column are a DPP quad. A step therefore
// needs only 3 pair tests (quad_perm moves) instead of 15, loads and stores are full
// 128-byte row segments (4 rows per instruction), and the per-step instruction count
// drops ~2.5x. The price is one wavefront per SIMD (4 tiles per CU), so memory latency
// is covered by a deep rolling prefetch (4*U rows in flight per column) rather than by
// other waves.
template <typename T, int BINSRC, bool STFT, bool CST64, int U>
__global__ __launch_bounds__(64, 1) void accumulate_quad_kernel(
    const T* __restrict__ Wx, const void* __restrict__ src, const T* __restrict__ Sfs,
    T* __restrict__ Tx, const void* __restrict__ cst, SsqParams sp, int64_t na64, int64_t n64,
    int32_t* __restrict__ kmap) {
    constexpr int RL = 4, TC = 16;
    using TM = Term<T, CST64>;
    using term_t = typename TM::type;
    using w_t = typename TM::wtype;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* tile = reinterpret_cast<T*>(lds_raw);       // [na][16][2], cells skewed
    const int na = (int)na64, n = (int)n64;        // host guarantees na * n < 2^31
    const int lane = threadIdx.x;
    const int c = lane >> 2, rl = lane & 3;
    const int per = gridDim.x >> 3;                // grid.x is a multiple of 8
    const int tile_id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile_id * TC >= n) return;
    const int j = tile_id * TC + c;
    const bool col_ok = j < n;
    const int64_t omax = na - 1;
    const size_t boff = (size_t)blockIdx.y * (size_t)na * (size_t)n;
    const T* Wb = Wx + 2 * boff;
    T* Tb = Tx + 2 * boff;
    int32_t* kb = kmap ? kmap + boff : nullptr;
    const char* sb = (const char*)src + boff * SideVal<T, BINSRC>::stride;

    for (int t = lane; t < na * TC; t += 64) { tile[2 * t] = T(0); tile[2 * t + 1] = T(0); }
    __builtin_amdgcn_wave_barrier();

    T zc[U], zd[U];
    w_t wt[U];
    SideVal<T, BINSRC> sv[U];
    const bool uni = sp.cst_uniform != 0;          // scalar weight: one load per kernel
    const w_t w0 = ((const w_t*)cst)[0];
    // Loads are unconditional (row and column clamped into the array, the point is
    // discarded at its use when it lies outside): with a branch around them the compiler
    // loses count of the loads in flight and drains them all (s_waitcnt vmcnt(0)) before
    // every use, which turns the rolling prefetch into one batch at a time.
    const int jc = col_ok ? j : n - 1;
    auto request = [&](int u, int i) {
        const int ic = i < na ? i : na - 1;
        const unsigned q = (unsigned)ic * (unsigned)n + (unsigned)jc;
        zc[u] = Wb[2 * (size_t)q];
        zd[u] = Wb[2 * (size_t)q + 1];
        sv[u].load(sb, q, ic, jc, na);
        if (!uni) wt[u] = ((const w_t*)cst)[ic];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) request(u, u * RL + rl);

    for (int i0 = 0; i0 < na; i0 += RL * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * RL + rl;
            int k = -1;
            if (col_ok && i < na) {
                k = (int)point_bin<T, BINSRC, STFT>(zc[u], zd[u], sv[u], i, Sfs, sp, omax);
                if (kb) kb[(unsigned)i * (unsigned)n + (unsigned)j] = k;
            }
            term_t tr = term_t(0), ti = term_t(0);
            T ore = T(0), oim = T(0);
            T* cell = tile;
            if (k >= 0) {
                const w_t wsel = uni ? w0 : wt[u];
                tr = TM::make(zc[u], wsel);
                ti = TM::make(zd[u], wsel);
                cell = tile + 2 * (k * TC + ((c + k) & 15));
                ore = cell[0]; oim = cell[1];
            }
            request(u, i + RL * U);                 // refill the slot
            // lower rows of this column, ascending: quad lanes rl-3, rl-2, rl-1
            // (quad_perm [0,0,0,0], [0,0,0,1], [0,0,1,2]); -1 never matches a valid bin
            const int k3 = dpp_mov<0x00>(-1, k), k2 = dpp_mov<0x40>(-1, k), k1 = dpp_mov<0x90>(-1, k);
            const term_t r3 = dpp_mov<0x00>(term_t(0), tr), i3 = dpp_mov<0x00>(term_t(0), ti);
            const term_t r2 = dpp_mov<0x40>(term_t(0), tr), i2 = dpp_mov<0x40>(term_t(0), ti);
            const term_t r1 = dpp_mov<0x90>(term_t(0), tr), i1 = dpp_mov<0x90>(term_t(0), ti);
            if (rl >= 3 && k3 == k) { ore = TM::fold(ore, r3); oim = TM::fold(oim, i3); }
            if (rl >= 2 && k2 == k) { ore = TM::fold(ore, r2); oim = TM::fold(oim, i2); }
            if (rl >= 1 && k1 == k) { ore = TM::fold(ore, r1); oim = TM::fold(oim, i1); }
            ore = TM::fold(ore, tr); oim = TM::fold(oim, ti);
            // a higher row of the quad hitting the same cell writes it instead
            // (quad_perm [1,2,3,3], [2,3,3,3], [3,3,3,3])
            const int h1 = dpp_mov<0xF9>(-1, k), h2 = dpp_mov<0xFE>(-1, k), h3 = dpp_mov<0xFF>(-1, k);
            const bool last = !((rl <= 2 && h1 == k) || (rl <= 1 && h2 == k) || (rl == 0 && h3 == k));
            if (k >= 0 && last) { cell[0] = ore; cell[1] = oim; }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int cc = lane & 15, rr = lane >> 4;
        const int jj = tile_id * TC + cc;
        if (jj < n) {
#pragma unroll 4
            for (int k = rr; k < na; k += 4) {
                const T* cell = tile + 2 * (k * TC + ((cc + k) & 15));
                size_t q = (size_t)((unsigned)k * (unsigned)n + (unsigned)jj);
                Tb[2 * q] = cell[0];
                Tb[2 * q + 1] = cell[1];
            }
        }
    }
}

// --------------------------------- accumulate from a bin map, float64 tile, unordered
// The bins are known (2-byte map written by the producer), so a point costs two LDS float64
// adds (ds_add_f64, no return value) into a tile of `na` x COLS float64 cells held as a real and
// an imaginary plane: no ordering between rows, hence no combining across lanes and no ticket.
// Each cell is the float64 sum of its terms in whatever order the hardware served them, rounded
// to the data type once at the write-out -- for float32 data that is within 1e-6 of the cell's
// ordered float32 sum (float64 rounding, 1e-16 per add, is far below float32's), for float64
// data within n_terms * 2^-53. SSQ_TILE_ORDER=ordered keeps the bit-exact kernels above.
// A wavefront instruction covers 64 / COLS consecutive rows x COLS columns; loads are
// unconditional (clamped) and refilled as consumed, NW wavefronts share a tile.
bool reassign_ordered() {
    const char* e = getenv("SSQ_TILE_ORDER");       // (read at every launch: tests switch it)
    return e && !strcmp(e, "ordered");
}

template <typename T, bool CST64, int COLS, int NW, int U>
__global__ __launch_bounds__(64 * NW) void accumulate_f64_kernel(
    const T* __restrict__ Wx, const unsigned short* __restrict__ kidx, T* __restrict__ Tx,
    const void* __restrict__ cst, int cst_uniform, int na, int n) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* plane = reinterpret_cast<double*>(lds_raw);
    using TM = Term<T, CST64>;
    using w_t = typename TM::wtype;
    constexpr int RPI = 64 / COLS;                 // rows per wavefront instruction
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = lane % COLS, h = lane / COLS;
    const int per = gridDim.x >> 3;                // grid.x is a multiple of 8: tiles of one XCD adjoin
    const int tile_id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile_id * COLS >= n) return;
    const int j = tile_id * COLS + c;
    const bool col_ok = j < n;
    const int jc = col_ok ? j : n - 1;
    const size_t boff = (size_t)blockIdx.y * (size_t)na * (size_t)n;
    const T* Wb = Wx + 2 * boff;
    const unsigned short* kb = kidx + boff;
    T* Tb = Tx + 2 * boff;
    const int cells = na * COLS;
    const unsigned im_off = (unsigned)cells * 8u;  // byte offset of the imaginary plane

    for (int t = tid; t < 2 * cells; t += 64 * NW) plane[t] = 0.0;
    __syncthreads();

    T zc[U], zd[U];
    w_t wt[U];
    unsigned short kk[U];
    const bool uni = cst_uniform != 0;
    auto request = [&](int u, int i) {
        const int ic = i < na ? i : na - 1;
        const unsigned q = (unsigned)ic * (unsigned)n + (unsigned)jc;
        if constexpr (sizeof(T) == 4) {
            const float2 z = reinterpret_cast<const float2*>(Wb)[q];
            zc[u] = z.x; zd[u] = z.y;
        } else {
            // (float64: read once, written once -- nontemporal, config 5 -1.5 ... -2 %, "r6y22")
            typedef double ssq_d2v __attribute__((ext_vector_type(2)));
            const ssq_d2v z = __builtin_nontemporal_load(reinterpret_cast<const ssq_d2v*>(Wb) + q);
            zc[u] = z.x; zd[u] = z.y;
        }
        kk[u] = kb[q];
        wt[u] = ((const w_t*)cst)[uni ? 0 : ic];   // (unconditional: a branch around a load costs a full drain)
    };
    const int step = NW * RPI;                     // rows between two groups of one wavefront
#pragma unroll
    for (int u = 0; u < U; ++u) request(u, (wv + u * NW) * RPI + h);
    for (int i0 = wv * RPI; i0 < na; i0 += U * step) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * step + h;
            const unsigned k = kk[u];
            const w_t wsel = wt[u];
            // the product in the CPU path's arithmetic (float32 data x float32 weight is a float32
            // product; a float64 weight vector makes it a float64 one), the sum in float64
            const double tr = (double)TM::make(zc[u], wsel), ti = (double)TM::make(zd[u], wsel);
            const bool live = col_ok && i < na && k != 0xFFFFu;
            if (live) {
                const unsigned off = (k * (unsigned)COLS + (unsigned)c) * 8u;
                SSQ_LDS_ADD_F64(lds_raw, off, tr);
                SSQ_LDS_ADD_F64(lds_raw, off + im_off, ti);
            }
            request(u, i + U * step);              // refill the slot
        }
    }
    __syncthreads();
    if (col_ok) {
        for (int k = wv * RPI + h; k < na; k += step) {
            const double re = plane[k * COLS + c], im = plane[cells + k * COLS + c];
            const size_t q = (size_t)((unsigned)k * (unsigned)n + (unsigned)j);
            if constexpr (sizeof(T) == 4) reinterpret_cast<float2*>(Tb)[q] = make_float2((float)re, (float)im);
            else {
                typedef double ssq_d2v __attribute__((ext_vector_type(2)));
                const ssq_d2v v = {re, im};
                __builtin_nontemporal_store(v, reinterpret_cast<ssq_d2v*>(Tb) + q);
            }
        }
    }
}

// columns per tile: rows of at least 128 bytes of data per tile, two or more workgroups per CU
// when the tile allows; 0 = the tile does not fit the LDS
template <typename T> static int f64_tile_cols(int64_t na, size_t lds_cap) {
    for (int cols = 32; cols >= 16; cols >>= 1)
        if ((size_t)na * cols * 16 <= lds_cap / 2) return cols;
    // (float64 data, 512 rows: 16 columns in one workgroup per CU 4.3 ms, 8 columns in two 4.9 ms)
    if ((size_t)na * 16 * 16 <= lds_cap) return 16;
    if (sizeof(T) == 8 && (size_t)na * 8 * 16 <= lds_cap) return 8;
    return 0;
}

template <typename T, bool CST64, int NW>
static int launch_accumulate_f64_w(const void* Wx, const void* kidx, void* Tx, const void* cst,
                                   const SsqParams& sp, int64_t batch, int64_t na, int64_t n,
                                   int cols, hipStream_t stream) {
    constexpr int U = 4;
    const size_t lds = (size_t)na * cols * 16;
    dim3 grid((unsigned)(((n + cols - 1) / cols + 7) / 8 * 8), (unsigned)batch);
#define SSQ_ACC64(C)                                                                               \
    {                                                                                              \
        auto kern = accumulate_f64_kernel<T, CST64, C, NW, U>;                                     \
        SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                     \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, (const T*)Wx,                   \
                           (const unsigned short*)kidx, (T*)Tx, cst, (int)sp.cst_uniform, (int)na, \
                           (int)n);                                                                \
    }
    if (cols == 32) SSQ_ACC64(32) else if (cols == 16) SSQ_ACC64(16) else SSQ_ACC64(8)
#undef SSQ_ACC64
    SSQ_LAUNCH_CHECK();
    return 0;
}

template <typename T, bool CST64>
static int launch_accumulate_f64(const void* Wx, const void* kidx, void* Tx, const void* cst,
                                 const SsqParams& sp, int64_t batch, int64_t na, int64_t n,
                                 int cols, hipStream_t stream) {
    // wavefronts per tile (SSQ_DEBUG_ACC64_NW = 4 / 8 / 16, tuning aid; default 8)
    static const int nw = getenv("SSQ_DEBUG_ACC64_NW") ? atoi(getenv("SSQ_DEBUG_ACC64_NW")) : 8;
    if (nw == 16) return launch_accumulate_f64_w<T, CST64, 16>(Wx, kidx, Tx, cst, sp, batch, na, n, cols, stream);
    if (nw == 4) return launch_accumulate_f64_w<T, CST64, 4>(Wx, kidx, Tx, cst, sp, batch, na, n, cols, stream);
    return launch_accumulate_f64_w<T, CST64, 8>(Wx, kidx, Tx, cst, sp, batch, na, n, cols, stream);
}

// ---------------------------------------------- accumulate, global fallback
// one thread per time column, serial over rows, Tx (pre-zeroed) updated in place.
template <typename T, int BINSRC, bool STFT, bool CST64>
__global__ __launch_bounds__(256) void accumulate_global_kernel(
    const T* __restrict__ Wx, const void* __restrict__ src, const T* __restrict__ Sfs,
    T* __restrict__ Tx, const void* __restrict__ cst, SsqParams sp, int64_t na, int64_t n,
    int32_t* __restrict__ kmap) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t base = (int64_t)blockIdx.y * na * n;
    const int64_t omax = na - 1;
    for (int64_t i = 0; i < na; ++i) {
        int64_t q = base + i * n + j;
        T c = Wx[2 * q], d = Wx[2 * q + 1];
        SideVal<T, BINSRC> sv;
        sv.load(src, q, i, j, na);
        int64_t k = point_bin<T, BINSRC, STFT>(c, d, sv, i, Sfs, sp, omax);
        if (kmap) kmap[q] = (int32_t)k;
        if (k < 0) continue;
        T* o = Tx + 2 * (base + k * n + j);
        T ore = o[0], oim = o[1];
        weighted_add<T, CST64>(ore, oim, c, d, cst, i);
        o[0] = ore; o[1] = oim;
    }
}

template <typename T, int BINSRC, bool STFT, bool CST64>
static int launch_accumulate_t(const void* Wx, const void* src, const void* Sfs, void* Tx,
                               const void* cst, const SsqParams& sp, int64_t batch,
                               int64_t na, int64_t n, int32_t* kmap, hipStream_t stream) {
    const size_t cell = 2 * sizeof(T);
    const size_t lds_cap = 160 * 1024;
    if constexpr (BINSRC == BIN_FROM_KIDX) {
        // known bins: the unordered float64 tile (see accumulate_f64_kernel) unless the caller
        // asked for the ordered sums or wants the bin map back
        int cols = f64_tile_cols<T>(na, lds_cap);
        static const int cols_env = getenv("SSQ_DEBUG_ACC64_COLS") ? atoi(getenv("SSQ_DEBUG_ACC64_COLS")) : 0;   // (tuning aid)
        if ((cols_env == 8 || cols_env == 16 || cols_env == 32) && (size_t)na * cols_env * 16 <= lds_cap) cols = cols_env;
        if (!reassign_ordered() && !kmap && cols && (size_t)na * (size_t)n < ((size_t)1 << 31))
            return launch_accumulate_f64<T, CST64>(Wx, src, Tx, cst, sp, batch, na, n, cols, stream);
    }
    auto launch_tile = [&](auto tc_tag) -> int {
        constexpr int TC = decltype(tc_tag)::value;
        size_t lds = (size_t)na * TC * cell;
        auto kern = accumulate_tile_kernel<T, BINSRC, STFT, CST64, TC>;
        SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid((unsigned)((n + TC - 1) / TC), (unsigned)batch);
        hipLaunchKernelGGL(kern, grid, dim3(64), lds, stream, (const T*)Wx, src, (const T*)Sfs,
                           (T*)Tx, cst, sp, na, n, kmap);
        SSQ_LAUNCH_CHECK();
        return 0;
    };
    // 16-column tiles use the DPP-combined, double-buffered kernel
    if ((size_t)na * 16 * cell <= lds_cap) {
        size_t lds = (size_t)na * 16 * cell;
        if ((size_t)na * (size_t)n < ((size_t)1 << 31)) {      // 32-bit offsets inside
            dim3 grid((unsigned)(((n + 15) / 16 + 7) / 8 * 8), (unsigned)batch);
            static const int variant = getenv("SSQ_DEBUG_ACC_VARIANT") ? atoi(getenv("SSQ_DEBUG_ACC_VARIANT")) : 0;
            if (variant != 2) {        // row-lane layouts (variant 2: one wavefront per tile, quads)
                // Row batches in flight per lane. Measured (config 2, float32), 16 row-lanes:
                // U = 2: 257 us, 3: 263, 4: 259, 8: 278; 8 row-lanes: U = 1: 314, 2: 246, 3: 240,
                // 4: 243, 8: 267 -- the resident wavefronts cover most of the latency, and a
                // deep prefetch spreads each wavefront's requests over more DRAM pages.
                constexpr int U = sizeof(T) == 4 ? 2 : 4;       // 16 row-lanes
                constexpr int U8 = 3;                           // 8 row-lanes
                // float32 with a tall tile (na * 32 cells over half the LDS): 8 row-lanes x 8 columns
                // per wavefront, 2 wavefronts per 16-column tile (240 us at config 2 vs 250 with 16
                // row-lanes); float64 -> 16 row-lanes x 4 columns, 4 wavefronts (config 5: 22.7 vs
                // 23.6 ms).
                // SSQ_DEBUG_ACC_VARIANT = 1 / 3 force the 16- / 8-lane layout.
                if ((variant == 0 || variant == 9) && sizeof(T) == 4 &&
                    (size_t)na * 32 * cell <= lds_cap / 2) {
                    // float32 default: 32-column tiles of four 8-lane wavefronts (256-byte row
                    // segments per workgroup, 2 workgroups per CU): 232 us at config 2 vs 240 with
                    // 16-column tiles (64-column tiles, one workgroup per CU: 231)
                    auto kern = accumulate_tile16_kernel<T, BINSRC, STFT, CST64, U8, 2, 8, 32>;
                    const size_t lds32 = (size_t)na * 32 * cell;
                    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
                    dim3 grid32((unsigned)(((n + 31) / 32 + 7) / 8 * 8), (unsigned)batch);
                    hipLaunchKernelGGL(kern, grid32, dim3(256), lds32, stream, (const T*)Wx, src, (const T*)Sfs,
                                       (T*)Tx, cst, sp, na, n, kmap);
                } else
                if (variant == 3 || (variant == 0 && sizeof(T) == 4)) {
                    auto kern = accumulate_tile16_kernel<T, BINSRC, STFT, CST64, U8, 2, 8>;
                    SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(kern, grid, dim3(128), lds, stream, (const T*)Wx, src, (const T*)Sfs,
                                       (T*)Tx, cst, sp, na, n, kmap);
                } else {
                auto kern = accumulate_tile16_kernel<T, BINSRC, STFT, CST64, U, 4, 16>;
                SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const T*)Wx, src, (const T*)Sfs,
                                   (T*)Tx, cst, sp, na, n, kmap);
                }
            } else {                   // SSQ_DEBUG_ACC_VARIANT=2: one wave per tile, quads (tuning aid)
                constexpr int U = sizeof(T) == 4 ? 16 : 8;
                auto kern = accumulate_quad_kernel<T, BINSRC, STFT, CST64, U>;
                SSQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, grid, dim3(64), lds, stream, (const T*)Wx, src, (const T*)Sfs,
                                   (T*)Tx, cst, sp, na, n, kmap);
            }
            SSQ_LAUNCH_CHECK();
            return 0;
        }
    }
    // larger `na`: shrink the tile before giving up on LDS
    if ((size_t)na * 16 * cell <= lds_cap / 2) return launch_tile(std::integral_constant<int, 16>{});
    if ((size_t)na * 8 * cell <= lds_cap / 2) return launch_tile(std::integral_constant<int, 8>{});
    if ((size_t)na * 16 * cell <= lds_cap) return launch_tile(std::integral_constant<int, 16>{});
    if ((size_t)na * 8 * cell <= lds_cap) return launch_tile(std::integral_constant<int, 8>{});
    if ((size_t)na * 4 * cell <= lds_cap) return launch_tile(std::integral_constant<int, 4>{});
    SSQ_CHECK_HIP(hipMemsetAsync(Tx, 0, (size_t)batch * na * n * cell, stream));
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL((accumulate_global_kernel<T, BINSRC, STFT, CST64>), grid, dim3(256), 0,
                       stream, (const T*)Wx, src, (const T*)Sfs, (T*)Tx, cst, sp, na, n, kmap);
    SSQ_LAUNCH_CHECK();
    return 0;
}

template <typename T, int BINSRC>
static int launch_accumulate_b(const void* Wx, const void* src, const void* Sfs, void* Tx,
                               const void* cst, const SsqParams& sp, int64_t batch,
                               int64_t na, int64_t n, int32_t* kmap, hipStream_t stream) {
    const bool c64 = sp.cst_f64 && sizeof(T) == 4;
    if (Sfs) {
        if constexpr (BINSRC == BIN_FROM_DWX) {
            return c64 ? launch_accumulate_t<T, BINSRC, true, true>(Wx, src, Sfs, Tx, cst, sp, batch, na, n, kmap, stream)
                       : launch_accumulate_t<T, BINSRC, true, false>(Wx, src, Sfs, Tx, cst, sp, batch, na, n, kmap, stream);
        }
    }
    return c64 ? launch_accumulate_t<T, BINSRC, false, true>(Wx, src, nullptr, Tx, cst, sp, batch, na, n, kmap, stream)
               : launch_accumulate_t<T, BINSRC, false, false>(Wx, src, nullptr, Tx, cst, sp, batch, na, n, kmap, stream);
}

int launch_accumulate(int dtype, int binsrc, const void* Wx, const void* src, const void* Sfs,
                      void* Tx, const void* cst, const SsqParams& sp, int64_t batch, int64_t na,
                      int64_t n, int32_t* kmap, hipStream_t stream) {
    SSQ_REQUIRE(na >= 1 && n >= 1 && batch >= 1, "accumulate: empty shape (%lld, %lld, %lld)",
                (long long)batch, (long long)na, (long long)n);
    SSQ_REQUIRE(batch <= 65535, "accumulate: batch %lld > 65535", (long long)batch);
    SSQ_REQUIRE(binsrc != BIN_FROM_KIDX || na < 65535, "bin map needs na < 65535");
#define SSQ_ACC(T)                                                                                 \
    switch (binsrc) {                                                                              \
        case BIN_FROM_DWX: return launch_accumulate_b<T, BIN_FROM_DWX>(Wx, src, Sfs, Tx, cst, sp, batch, na, n, kmap, stream); \
        case BIN_FROM_W: return launch_accumulate_b<T, BIN_FROM_W>(Wx, src, Sfs, Tx, cst, sp, batch, na, n, kmap, stream);     \
        default: return launch_accumulate_b<T, BIN_FROM_KIDX>(Wx, src, Sfs, Tx, cst, sp, batch, na, n, kmap, stream);          \
    }
    if (dtype == SSQ_F32) { SSQ_ACC(float) }
    SSQ_ACC(double)
#undef SSQ_ACC
}

// ------------------------------------------------------------------ phase
template <typename T, bool STFT>
__global__ __launch_bounds__(256) void phase_kernel(const T* __restrict__ Wx,
                                                    const T* __restrict__ dWx,
                                                    const T* __restrict__ Sfs, T* __restrict__ w,
                                                    int64_t na, int64_t n, int64_t total,
                                                    double gamma) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
         q += (int64_t)gridDim.x * blockDim.x) {
        T c = Wx[2 * q], d = Wx[2 * q + 1];
        // the two-step path thresholds with `abs(Wx) < gamma`, gamma in the data dtype
        if (mag_lt(c, d, (T)gamma)) { w[q] = (T)INFINITY; continue; }
        double r = phase_ratio(dWx[2 * q], dWx[2 * q + 1], c, d);
        if constexpr (STFT) {
            int64_t i = (q / n) % na;
            w[q] = (T)fabs((double)Sfs[i] - r);
        } else {
            w[q] = (T)fabs(r);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void replace_under_abs_kernel(T* __restrict__ w,
                                                                const T* __restrict__ ref,
                                                                int64_t total, double value,
                                                                T replacement) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
         q += (int64_t)gridDim.x * blockDim.x)
        if (mag_of(ref[2 * q], ref[2 * q + 1]) < value) w[q] = replacement;
}

// ---------------------------------------------------------------- framing
template <typename T>
__global__ __launch_bounds__(256) void buffer_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                     int64_t n_x, int64_t seg_len, int64_t n_segs,
                                                     int64_t hop, int64_t s20, int64_t s21,
                                                     int modulated, int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = t % n_segs, r = (t / n_segs) % seg_len, b = t / (n_segs * seg_len);
        int64_t start = hop * c, s;
        if (!modulated) s = start + r;
        else if (r < s20) s = start + s21 + r;
        else s = start + (r - s20);
        out[t] = x[b * n_x + s];
    }
}

// ---------------------------------------------------------------- padding
__device__ __forceinline__ int64_t pad_source(int64_t t, int64_t n, int padtype) {
    // t in [-n1, n + n2); returns source index in [0, n) or -1 for zero fill
    if (t >= 0 && t < n) return t;
    switch (padtype) {
        case SSQ_PAD_REFLECT: {
            if (n == 1) return 0;
            int64_t period = 2 * (n - 1);
            int64_t m = t % period; if (m < 0) m += period;
            return m < n ? m : period - m;
        }
        case SSQ_PAD_SYMMETRIC: {
            int64_t period = 2 * n;
            int64_t m = t % period; if (m < 0) m += period;
            return m < n ? m : period - 1 - m;
        }
        case SSQ_PAD_REPLICATE: return t < 0 ? 0 : n - 1;
        case SSQ_PAD_WRAP: { int64_t m = t % n; if (m < 0) m += n; return m; }
        default: return -1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pad_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                  int64_t n, int64_t n1, int64_t m, int padtype,
                                                  int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t b = t / m, p = t % m;
        int64_t s = pad_source(p - n1, n, padtype);
        out[t] = s < 0 ? T(0) : x[b * n + s];
    }
}

static inline unsigned stream_grid(int64_t total, int block = 256) {
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 256 * 8 * 4;   // 256 CUs x 8 blocks, grid-stride the rest
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace ssq

using namespace ssq;

// ======================================================================= C ABI
// the build's stamp: ssqueezepy_amd/build.py links a strong definition (the last commit that touched
// csrc/ or include/) over this one
extern "C" __attribute__((weak)) const char ssq_build_sha_value[] = "unknown";

extern "C" {

const char* ssq_build_sha(void) { return ssq_build_sha_value; }
int ssq_version(void) { return 105; }   // 105: ssq_cwt_plan_tile_kernel; 104: ssq_build_sha, ssq_cwt_plan_set_bin_dump; 103: ssq_ridge_*_batch; 102: ssq_cwt_plan_tile_cols; 101: ssq_cwt_blocks_desc.classes has 5 columns (analytic classes)
const char* ssq_last_error(void) { return g_last_error.c_str(); }

int ssq_device_count(int* count) {
    SSQ_REQUIRE(count, "ssq_device_count: null pointer");
    SSQ_CHECK_HIP(hipGetDeviceCount(count));
    return 0;
}
int ssq_set_device(int device) { SSQ_CHECK_HIP(hipSetDevice(device)); return 0; }

int ssq_device_info(int device, char* name, int len, int* cus, int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    SSQ_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (name && len > 0) { strncpy(name, prop.gcnArchName, len - 1); name[len - 1] = 0; }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
}

int ssq_malloc(void** ptr, int64_t bytes) {
    SSQ_REQUIRE(ptr && bytes >= 0, "ssq_malloc: bad arguments");
    SSQ_CHECK_HIP(hipMalloc(ptr, (size_t)(bytes > 0 ? bytes : 1)));
    return 0;
}
int ssq_free(void* ptr) { SSQ_CHECK_HIP(hipFree(ptr)); return 0; }
int ssq_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream) {
    SSQ_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return 0;
}
int ssq_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream) {
    SSQ_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return 0;
}
int ssq_memset(void* dst, int value, int64_t bytes, void* stream) {
    SSQ_CHECK_HIP(hipMemsetAsync(dst, value, (size_t)bytes, as_stream(stream)));
    return 0;
}
int ssq_stream_synchronize(void* stream) {
    SSQ_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

static int check_dtype(int dtype) {
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "dtype must be SSQ_F32 or SSQ_F64 (got %d)", dtype);
    return 0;
}

int ssq_phase_cwt(int dtype, const void* Wx, const void* dWx, void* w, int64_t batch, int64_t na,
                  int64_t n, double gamma, void* stream) {
    if (check_dtype(dtype)) return -1;
    SSQ_REQUIRE(Wx && dWx && w, "ssq_phase_cwt: null pointer");
    int64_t total = batch * na * n;
    if (total == 0) return 0;
    if (dtype == SSQ_F32)
        hipLaunchKernelGGL((phase_kernel<float, false>), dim3(stream_grid(total)), dim3(256), 0, as_stream(stream),
                           (const float*)Wx, (const float*)dWx, (const float*)nullptr, (float*)w, na, n, total, gamma);
    else
        hipLaunchKernelGGL((phase_kernel<double, false>), dim3(stream_grid(total)), dim3(256), 0, as_stream(stream),
                           (const double*)Wx, (const double*)dWx, (const double*)nullptr, (double*)w, na, n, total, gamma);
    SSQ_LAUNCH_CHECK();
    return 0;
}

int ssq_phase_stft(int dtype, const void* Sx, const void* dSx, const void* Sfs, void* w,
                   int64_t batch, int64_t na, int64_t n, double gamma, void* stream) {
    if (check_dtype(dtype)) return -1;
    SSQ_REQUIRE(Sx && dSx && Sfs && w, "ssq_phase_stft: null pointer");
    int64_t total = batch * na * n;
    if (total == 0) return 0;
    if (dtype == SSQ_F32)
        hipLaunchKernelGGL((phase_kernel<float, true>), dim3(stream_grid(total)), dim3(256), 0, as_stream(stream),
                           (const float*)Sx, (const float*)dSx, (const float*)Sfs, (float*)w, na, n, total, gamma);
    else
        hipLaunchKernelGGL((phase_kernel<double, true>), dim3(stream_grid(total)), dim3(256), 0, as_stream(stream),
                           (const double*)Sx, (const double*)dSx, (const double*)Sfs, (double*)w, na, n, total, gamma);
    SSQ_LAUNCH_CHECK();
    return 0;
}

static int fill_params(SsqParams& sp, int grid, const double* params, int flipud, double gamma, int cst_f64) {
    SSQ_REQUIRE(grid >= SSQ_GRID_LOG && grid <= SSQ_GRID_LIN, "unknown grid kind %d", grid);
    SSQ_REQUIRE(params, "grid params must not be null");
    for (int t = 0; t < 5; ++t) sp.p[t] = params[t];
    sp.grid = grid; sp.flipud = flipud ? 1 : 0; sp.gamma = gamma; sp.cst_f64 = cst_f64 ? 1 : 0;
    sp.cst_uniform = 0;                   // weights are a device array here: not inspected
    finalize_params(sp);
    return 0;
}

int ssq_ssqueeze(int dtype, const void* Wx, const void* dWx, const void* Sfs, void* Tx,
                 const void* cst, int cst_f64, int64_t batch, int64_t na, int64_t n, double gamma,
                 int grid, const double* params, int flipud, int32_t* kmap, void* stream) {
    if (check_dtype(dtype)) return -1;
    SSQ_REQUIRE(Wx && dWx && Tx && cst, "ssq_ssqueeze: null pointer");
    SsqParams sp;
    if (fill_params(sp, grid, params, flipud, gamma, cst_f64)) return -1;
    return launch_accumulate(dtype, BIN_FROM_DWX, Wx, dWx, Sfs, Tx, cst, sp, batch, na, n, kmap, as_stream(stream));
}

int ssq_indexed_sum(int dtype, const void* Wx, const void* w, void* Tx, const void* cst, int cst_f64,
                    int64_t batch, int64_t na, int64_t n, int grid, const double* params, int flipud,
                    void* stream) {
    if (check_dtype(dtype)) return -1;
    SSQ_REQUIRE(Wx && w && Tx && cst, "ssq_indexed_sum: null pointer");
    SsqParams sp;
    if (fill_params(sp, grid, params, flipud, 0.0, cst_f64)) return -1;
    return launch_accumulate(dtype, BIN_FROM_W, Wx, w, nullptr, Tx, cst, sp, batch, na, n, nullptr, as_stream(stream));
}

int ssq_replace_under_abs(int dtype, void* w, const void* ref, int64_t count, double value,
                          double replacement, void* stream) {
    if (check_dtype(dtype)) return -1;
    SSQ_REQUIRE(w && ref, "ssq_replace_under_abs: null pointer");
    if (count == 0) return 0;
    if (dtype == SSQ_F32)
        hipLaunchKernelGGL((replace_under_abs_kernel<float>), dim3(stream_grid(count)), dim3(256), 0, as_stream(stream),
                           (float*)w, (const float*)ref, count, value, (float)replacement);
    else
        hipLaunchKernelGGL((replace_under_abs_kernel<double>), dim3(stream_grid(count)), dim3(256), 0, as_stream(stream),
                           (double*)w, (const double*)ref, count, value, replacement);
    SSQ_LAUNCH_CHECK();
    return 0;
}

int ssq_buffer(int dtype, const void* x, void* out, int64_t batch, int64_t n_x, int64_t seg_len,
               int64_t n_overlap, int modulated, void* stream) {
    if (check_dtype(dtype)) return -1;
    SSQ_REQUIRE(x && out, "ssq_buffer: null pointer");
    int64_t hop = seg_len - n_overlap;
    SSQ_REQUIRE(seg_len >= 1 && hop >= 1 && n_x >= seg_len, "ssq_buffer: bad framing (n_x=%lld seg_len=%lld n_overlap=%lld)",
                (long long)n_x, (long long)seg_len, (long long)n_overlap);
    int64_t n_segs = (n_x - seg_len) / hop + 1;
    int64_t s20 = (seg_len + 1) / 2, s21 = (seg_len % 2 == 1) ? s20 - 1 : s20;
    int64_t total = batch * seg_len * n_segs;
    if (dtype == SSQ_F32)
        hipLaunchKernelGGL((buffer_kernel<float>), dim3(stream_grid(total)), dim3(256), 0, as_stream(stream),
                           (const float*)x, (float*)out, n_x, seg_len, n_segs, hop, s20, s21, modulated, total);
    else
        hipLaunchKernelGGL((buffer_kernel<double>), dim3(stream_grid(total)), dim3(256), 0, as_stream(stream),
                           (const double*)x, (double*)out, n_x, seg_len, n_segs, hop, s20, s21, modulated, total);
    SSQ_LAUNCH_CHECK();
    return 0;
}

int ssq_pad_signal(int dtype, const void* x, void* out, int64_t batch, int64_t n, int64_t n1,
                   int64_t n2, int padtype, void* stream) {
    if (check_dtype(dtype)) return -1;
    SSQ_REQUIRE(x && out, "ssq_pad_signal: null pointer");
    SSQ_REQUIRE(padtype >= SSQ_PAD_ZERO && padtype <= SSQ_PAD_WRAP, "unknown padtype %d", padtype);
    int64_t m = n1 + n + n2, total = batch * m;
    if (total == 0) return 0;
    if (dtype == SSQ_F32)
        hipLaunchKernelGGL((pad_kernel<float>), dim3(stream_grid(total)), dim3(256), 0, as_stream(stream),
                           (const float*)x, (float*)out, n, n1, m, padtype, total);
    else
        hipLaunchKernelGGL((pad_kernel<double>), dim3(stream_grid(total)), dim3(256), 0, as_stream(stream),
                           (const double*)x, (double*)out, n, n1, m, padtype, total);
    SSQ_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
