// ssq_cwt_tiles.hip -- the column-tile path of the fused ssq_cwt form (float32, gfx950): the plan object.
//
// Math and planning: ssqueezepy_amd/_tiles.py. The device code lives in three translation units:
//   ssq_tile_fft.hip      the intermediates (decimated baseband samples of the interpolated rows; the analytic signal)
//   ssq_tile_f64.hip      tile2_kernel -- float64 Tx tile in LDS, unordered ds_add_f64: the default
//   ssq_tile_ordered.hip  tile_kernel  -- float32 tile, terms added in the reference's row order by a ticket
//                         (SSQ_TILE_ORDER=ordered: Tx bit for bit the CPU loop's)
// Here: the tables both kernels walk (TilePlan::create), the choice between them (usable / tile_cols / run), what
// executed (tiles_done).
#include "ssq_common.h"
#include "ssq_tiles.h"
#include <algorithm>
#include <cmath>

namespace ssq {

int tile_rows_per_step() { return TILE_G; }
int TilePlan::create(const ssq_cwt_tiles_desc& d, int64_t M_, int64_t N_, int64_t n1_, int64_t na_, int group_,
                     double dt_, int64_t& bytes) {
    M = M_; N = N_; n1 = n1_; na = na_; group = group_; dt = dt_;
    nsegs = d.n_segs; nsteps = d.n_steps; n_irows = d.n_irows; u_total = d.u_total;
    SSQ_REQUIRE(nsegs >= 1 && nsteps >= 1 && n_irows >= 1 && d.n_classes >= 1, "empty tile tables");
    SSQ_REQUIRE((d.reserved ? d.reserved : 4) == TILE_G, "tile tables hold %d rows per step, this build walks %d",
                d.reserved ? d.reserved : 4, TILE_G);
    {
        int dev = 0; hipDeviceProp_t pr;
        SSQ_CHECK_HIP(hipGetDevice(&dev));
        SSQ_CHECK_HIP(hipGetDeviceProperties(&pr, dev));
        ncu = pr.multiProcessorCount;
        if (const char* e = getenv("SSQ_DEBUG_TILE_GRID")) if (atoi(e) > 0) ncu = atoi(e);
        // both tile kernels keep a tile of up to 160 KB in a workgroup's LDS (gfx950); a device with
        // less refuses the tile path here instead of failing at the first launch
        SSQ_REQUIRE((size_t)pr.maxSharedMemoryPerMultiProcessor >= 160 * 1024,
                    "the tile path needs 160 KB of LDS per workgroup, the device has %zu",
                    (size_t)pr.maxSharedMemoryPerMultiProcessor);
    }
    SSQ_REQUIRE(na * N < ((int64_t)1 << 29) && na < 512, "na = %lld, N = %lld: outside the tile path's 32-bit offsets",
                (long long)na, (long long)N);
    // the modulation phase kc * n mod M is formed with a 24-bit multiply and carried in a float
    // (and the centre bin, < M / 2, shares a word with the row: 22 bits)
    SSQ_REQUIRE(M <= ((int64_t)1 << 23), "the tile path needs a padded length <= 2^23");
    SSQ_REQUIRE((int64_t)group * u_total < ((int64_t)1 << 31), "tile intermediates exceed 2^31 entries");
    auto up = [&](void** dst, const void* src, size_t nbytes) -> int {
        SSQ_CHECK_HIP(hipMalloc(dst, nbytes ? nbytes : 1));
        if (nbytes) SSQ_CHECK_HIP(hipMemcpy(*dst, src, nbytes, hipMemcpyHostToDevice));
        bytes += (int64_t)nbytes;
        return 0;
    };
    int rc;
    static_assert(sizeof(TileSeg) == 32 && sizeof(TileRow) == 16 && sizeof(TileIRow) == 32, "table layout");
    {   // one packed record per step (a wavefront's consecutive steps are usually of different
        // segments): kind | log2 R << 1 | weight-table offset << 8, L - 1, entries between two
        // signals' rows of the class, entries before the class
        std::vector<int32_t> hs((size_t)nsteps * 4);
        const TileSeg* sg = reinterpret_cast<const TileSeg*>(d.segs);
        int64_t covered = 0;
        for (int i = 0; i < nsegs; ++i) {
            SSQ_REQUIRE(sg[i].first == covered && sg[i].nsteps >= 1 && sg[i].first + sg[i].nsteps <= nsteps,
                        "tile segment %d does not continue the step list", i);
            SSQ_REQUIRE(sg[i].kind == 0 || sg[i].kind == 1, "tile segment %d: bad kind", i);
            SSQ_REQUIRE(sg[i].lgR >= 0 && sg[i].lgR < 32 && sg[i].wtab_off >= 0 && sg[i].wtab_off < (1 << 23),
                        "tile segment %d does not fit its packed record", i);
            for (int t = 0; t < sg[i].nsteps; ++t) {
                int32_t* q = &hs[((size_t)sg[i].first + t) * 4];
                q[0] = sg[i].kind | (sg[i].lgR << 1) | (sg[i].wtab_off << 8);
                q[1] = sg[i].lmask; q[2] = sg[i].sig_stride; q[3] = sg[i].cls_base;
            }
            covered += sg[i].nsteps;
        }
        SSQ_REQUIRE(covered == nsteps, "tile segments cover %lld of %d steps", (long long)covered, nsteps);
        if ((rc = up((void**)&steps, hs.data(), hs.size() * 4))) return rc;
    }
    {   // row records as the kernel reads them: row | padding << 9 | centre bin << 10, offset of the
        // row's samples inside its class
        const TileRow* rw = reinterpret_cast<const TileRow*>(d.rows);
        std::vector<int32_t> hp((size_t)nsteps * TILE_G * 2);
        for (size_t i = 0; i < (size_t)nsteps * TILE_G; ++i) {
            const int32_t row = rw[i].row & 0xFFFF;
            SSQ_REQUIRE(row < 512 && rw[i].kc >= 0 && rw[i].kc < (1 << 22), "tile row %zu does not fit its packed record", i);
            hp[2 * i] = row | (rw[i].row < 0 ? 0x200 : 0) | (int32_t)((uint32_t)rw[i].kc << 10);
            hp[2 * i + 1] = rw[i].ubase;
        }
        if ((rc = up((void**)&rows, hp.data(), hp.size() * 4))) return rc;
    }
    {   // the default kernels' items: `rpi` consecutive rows of a step, one record of 8 words per item:
        // row0 | padded sub-rows << 9 | kind << 12 | lgR << 13 (| the weights' table offset << 18: tile3_kernel),
        // samples of sub-row 0 (class offset + row in class * L), row0 * N * 8, entries between two signals' rows of
        // the class, centre bins -- and the wavefronts' blocks of items, [nw][4] = first item, end, first item of the
        // second class, the weights' table offsets of the two classes (16 bits each).
        // tile2_kernel (ssq_tile_f64.hip): rpi = 64 / columns per tile; tile3_kernel (ssq_tile_pair.hip): 4 rows x 32
        // columns, two columns per lane.
        const TileRow* rw = reinterpret_cast<const TileRow*>(d.rows);
        const TileSeg* sg = reinterpret_cast<const TileSeg*>(d.segs);
        cols2 = tile2_lds_bytes(na, 32) <= 160 * 1024 ? 32 : 16;
        if (const char* e = getenv("SSQ_DEBUG_TILE2_COLS")) if (atoi(e) == 16) cols2 = 16;     // (tuning aid)
        SSQ_REQUIRE(tile2_lds_bytes(na, cols2) <= 160 * 1024, "na = %lld: the Tx tile exceeds the LDS", (long long)na);
        lgr_max2 = 0;
        for (int i = 0; i < nsegs; ++i) if (sg[i].kind) lgr_max2 = std::max(lgr_max2, (int)sg[i].lgR);
        // (what a row read back costs next to an interpolated one when the rows are dealt to the wavefronts; measured
        // 0.5 .. 1.2: 221 / 223 / 222 / 228 / 225 us, round 4)
        // chg_cost < 0: tile2_kernel's contiguous blocks; >= 0: tile3_kernel's dealt lists (see below), a class change
        // inside a wavefront's list costing that many items (max_cls = 2: both classes' weights resident, no cost)
        auto build = [&](int rpi, int nw, bool woff_in_record, float rb_cost, float chg_cost, int max_cls,
                         void** items_dev, int32_t** waves_dev, int& n_items, bool& ok) -> int {
            SSQ_REQUIRE(TILE_G % rpi == 0, "tile tables: %d rows per step, %d per item", TILE_G, rpi);
            n_items = nsteps * TILE_G / rpi;
            std::vector<int32_t> hi8((size_t)n_items * 8, 0);
            std::vector<float> cost((size_t)n_items, 0.f);
            std::vector<int32_t> icls((size_t)n_items, 0), woff((size_t)n_items, 0);
            ok = true;
            for (int i = 0; i < nsegs; ++i) {
                const int64_t L = (int64_t)M >> sg[i].lgR;
                if (sg[i].kind && (sg[i].wtab_off >= (woff_in_record ? 16384 : 65536) || sg[i].sig_stride % L)) ok = false;
                for (int t = 0; t < sg[i].nsteps * TILE_G / rpi; ++t) {
                    const size_t it = (size_t)sg[i].first * TILE_G / rpi + t;
                    const TileRow* r = rw + it * rpi;
                    int npad = 0;
                    for (int k = 0; k < rpi; ++k) {
                        if (r[k].row < 0) ++npad;
                        else if (npad) ok = false;                        // padding trails
                        // the sub-rows are consecutive rows of the class (the padding repeats the last one)
                        if (r[k].row >= 0 && ((r[k].row & 0xFFFF) != (r[0].row & 0xFFFF) + k
                                              || (sg[i].kind && r[k].ubase != r[0].ubase + k * L))) ok = false;
                        hi8[8 * it + 4 + k] = r[k].kc;
                    }
                    const int32_t row0 = r[0].row & 0xFFFF;
                    hi8[8 * it] = row0 | (npad << 9) | (sg[i].kind << 12) | (sg[i].lgR << 13)
                                  | (woff_in_record && sg[i].kind ? (int32_t)((uint32_t)sg[i].wtab_off << 18) : 0);
                    hi8[8 * it + 1] = sg[i].kind ? sg[i].cls_base + r[0].ubase : 0;
                    hi8[8 * it + 2] = (int32_t)(uint32_t)((int64_t)row0 * N * 8);
                    hi8[8 * it + 3] = sg[i].kind ? sg[i].sig_stride : 0;
                    cost[it] = sg[i].kind ? 1.0f : rb_cost;           // what rows read back cost next to interpolated ones
                    icls[it] = sg[i].kind ? 1 + sg[i].lgR : 0;
                    woff[it] = sg[i].kind ? sg[i].wtab_off : 0;
                }
            }
            int rcb;
            if (chg_cost >= 0.f) {
                // tile3_kernel: per-wavefront LISTS instead of contiguous row blocks. Round 6's stamps showed the
                // wavefronts waiting 22 % of their time at the tile's end for the slowest of them (4-row items: a block
                // is 4 or 5 items, a class change costs 0.65 of one, rows read back 0.45). So:
                //   1. the interpolated items, class by class, are cut into `nw` chunks minimising the largest
                //      (a chunk of several classes pays `chg_cost` per class: the weights are re-read per tile);
                //   2. the items of rows read back need no weights: each goes to the wavefront with the least work;
                //   3. the wavefronts w, w + 4, w + 8, ... share a SIMD: the lists are dealt so that the SIMDs' sums agree;
                //   4. the item table is permuted so that a wavefront's list is contiguous (chunk, then rows read back).
                std::vector<int> iin, irb;
                for (int it = 0; it < n_items; ++it) (icls[it] ? iin : irb).push_back(it);
                const int ni = (int)iin.size();
                auto chunk_cost = [&](int a, int b) -> double {        // interpolated items [a, b) of `iin`
                    if (b <= a) return 0.0;
                    int ncl = 1;
                    for (int j = a + 1; j < b; ++j) if (icls[iin[j]] != icls[iin[j - 1]]) ++ncl;
                    if (ncl > max_cls) return 1e30;
                    return (double)(b - a) + (ncl > 1 && max_cls > 2 ? chg_cost * ncl : 0.0);
                };
                // (the wavefronts of a SIMD do not run at one speed: the arbiter serves the oldest first, and the stamps show the
                // youngest taking a third longer per item. speed[w]: what wavefront w gets done relative to the mean, by its
                // age rank w / 4 -- chunk k goes to wavefront k, and "largest" above means largest time = cost / speed.)
                // (measured at config 2, one box: skew 0 / 0.1 / 0.2 / 0.3 -> 193-196 / 187 / 184-186 / 191 us with 16 wavefronts;
                // no gain with 12)
                float skew = nw == 16 ? 0.2f : 0.f;
                if (const char* e = getenv("SSQ_DEBUG_TILE3_SKEW")) skew = (float)atof(e);
                std::vector<double> speed(nw, 1.0);
                {
                    const int nr = nw / 4;
                    for (int w = 0; w < nw; ++w) speed[w] = 1.0 + skew * (nr > 1 ? 1.0 - 2.0 * (w / 4) / (double)(nr - 1) : 0.0);
                    if (const char* e = getenv("SSQ_DEBUG_TILE3_SPEEDS")) {     // (A/B: a speed per age rank, "1.2,1.07,0.93,0.8")
                        double v[8]; int n = 0;
                        for (const char* q = e; *q && n < 8; ) { v[n++] = atof(q); while (*q && *q != ',' && *q != '/') ++q; if (*q) ++q; }
                        if (n == nr) for (int w = 0; w < nw; ++w) speed[w] = v[w / 4];
                    }
                }
                std::vector<std::vector<double>> f(nw + 1, std::vector<double>(ni + 1, 1e30));
                std::vector<std::vector<int>> arg(nw + 1, std::vector<int>(ni + 1, 0));
                f[0][0] = 0.0;
                for (int k = 1; k <= nw; ++k)
                    for (int j = 0; j <= ni; ++j)
                        for (int a = 0; a <= j; ++a) {
                            if (f[k - 1][a] >= 1e30) continue;
                            const double c = std::max(f[k - 1][a], chunk_cost(a, j) / speed[k - 1]);
                            // (ties: the later cut -- chunks of equal size rather than one long and one empty)
                            if (c < f[k][j] - 1e-12 || (c <= f[k][j] + 1e-12 && a >= arg[k][j])) { f[k][j] = c; arg[k][j] = a; }
                        }
                if (f[nw][ni] >= 1e30) ok = false;
                std::vector<std::vector<int>> lists(nw);
                std::vector<double> load(nw, 0.0);
                if (ok) {
                    int j = ni;
                    for (int k = nw; k >= 1; --k) {
                        const int a = arg[k][j];
                        for (int q = a; q < j; ++q) lists[k - 1].push_back(iin[q]);
                        load[k - 1] = chunk_cost(a, j);
                        j = a;
                    }
                    for (int it : irb) {
                        int best = 0;
                        for (int w = 1; w < nw; ++w)
                            if ((load[w] + rb_cost) / speed[w] < (load[best] + rb_cost) / speed[best] - 1e-12) best = w;
                        lists[best].push_back(it);
                        load[best] += rb_cost;
                    }
                }
                // the SIMDs: heaviest list first, each to the SIMD with the least work that still has a slot
                const int nsimd = 4, per = nw / nsimd;
                std::vector<int> order(nw), slot_of(nw, -1);
                for (int w = 0; w < nw; ++w) order[w] = w;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return load[a] > load[b]; });
                std::vector<double> ssum(nsimd, 0.0);
                std::vector<int> scnt(nsimd, 0);
                for (int w : order) {
                    int g = -1;
                    for (int q = 0; q < nsimd; ++q) if (scnt[q] < per && (g < 0 || ssum[q] < ssum[g] - 1e-12)) g = q;
                    slot_of[w] = scnt[g] * nsimd + g;
                    ssum[g] += load[w]; ++scnt[g];
                }
                if (skew != 0.f) for (int w = 0; w < nw; ++w) slot_of[w] = w;      // (the speeds were the slots')
                std::vector<int32_t> hp((size_t)n_items * 8, 0), wrec((size_t)nw * 4, 0);
                int pos = 0;
                std::vector<int> list_of_slot(nw, 0);
                for (int w = 0; w < nw; ++w) list_of_slot[slot_of[w]] = w;
                for (int sl = 0; sl < nw; ++sl) {
                    const std::vector<int>& L = lists[list_of_slot[sl]];
                    const int first = pos;
                    int isp = -1;
                    for (size_t q = 0; q < L.size(); ++q) {
                        if (q > 0 && icls[L[q]] && icls[L[q - 1]] && icls[L[q]] != icls[L[q - 1]] && isp < 0) isp = pos;
                        memcpy(&hp[(size_t)pos * 8], &hi8[(size_t)L[q] * 8], 32);
                        ++pos;
                    }
                    // (isp: the first item of the list's second class -- or of its rows read back, or its end)
                    int e_int = first;
                    for (size_t q = 0; q < L.size(); ++q) if (icls[L[q]]) e_int = first + (int)q + 1;
                    if (isp < 0) isp = e_int;
                    wrec[4 * sl] = first; wrec[4 * sl + 1] = pos; wrec[4 * sl + 2] = isp; wrec[4 * sl + 3] = 0;
                }
                if (pos != n_items) ok = false;
                if (getenv("SSQ_DEBUG_TILE_PLAN_PRINT")) {
                    for (int sl = 0; sl < nw; ++sl)
                        fprintf(stderr, "tile3 wave %2d (simd %d): items %3d..%3d second class at %3d, load %.2f\n", sl, sl % nsimd,
                                wrec[4 * sl], wrec[4 * sl + 1], wrec[4 * sl + 2], load[list_of_slot[sl]]);
                }
                if ((rcb = up(items_dev, hp.data(), hp.size() * 4))) return rcb;
                return up((void**)waves_dev, wrec.data(), wrec.size() * 4);
            }
            if ((rcb = up(items_dev, hi8.data(), hi8.size() * 4))) return rcb;
            // Contiguous, cost-balanced blocks of items per wavefront, each spanning at most TWO classes
            // (kind / decimation): the kernels keep the weights of (up to) two classes in registers.
            std::vector<int> run_start;                        // maximal runs of one class
            for (int it = 0; it < n_items; ++it)
                if (it == 0 || icls[it] != icls[it - 1]) run_start.push_back(it);
            const int nruns = (int)run_start.size();
            run_start.push_back(n_items);
            std::vector<double> pre((size_t)n_items + 1, 0.0);
            for (int it = 0; it < n_items; ++it) pre[it + 1] = pre[it] + cost[it];
            std::vector<int32_t> wt_;
            // (round 4 also sized the blocks by the wavefronts' measured speeds -- the older wavefronts of a SIMD finish the
            // same work 10-19 % sooner -- to no effect: 221 +- 3 us for every weighting; a SIMD's total is what counts)
            {
                int cur = 0;
                const double stot = nw;
                double sacc = 0;
                for (int w = 0; w < nw; ++w) {
                    sacc += 1.0;
                    if (cur >= n_items) { wt_.insert(wt_.end(), {n_items, n_items, n_items, 0}); continue; }
                    int r0 = 0;
                    while (run_start[r0 + 1] <= cur) ++r0;
                    const int maxe = run_start[std::min(r0 + 2, nruns)];          // at most the rest of this run and the next
                    const int rem = nw - w - 1;
                    const int rmin = std::max(r0, nruns - 2 * rem);               // the rest must fit the remaining wavefronts
                    int mine = rem == 0 ? n_items : run_start[std::min(rmin, nruns)];
                    mine = std::max(mine, cur + 1);
                    int e = cur;
                    const double want = pre[n_items] * sacc / stot;
                    while (e < n_items && pre[e + 1] <= want + 1e-9) ++e;
                    e = std::min(std::max(e, mine), maxe);
                    if (rem == 0) { e = n_items; if (e > maxe) ok = false; }
                    const int isp = run_start[r0 + 1] < e ? run_start[r0 + 1] : e;
                    wt_.insert(wt_.end(), {cur, e, isp, woff[cur] | (woff[std::min(isp, n_items - 1)] << 16)});
                    cur = e;
                }
                if (cur < n_items) ok = false;
            }
            return up((void**)waves_dev, wt_.data(), wt_.size() * 4);
        };
        if ((rc = build(64 / cols2, TILE2_NW, false, 0.7f, -1.f, 2, &items2, &wave_first2, n_items2, tile2_ok))) return rc;
        tile3_ok = false;
        if (cols2 == 32 && TILE_G == 4) {
            // (measured with shader-clock stamps, round 6: an item of rows read back costs 0.45 of an interpolated one,
            // re-reading a class's weights 0.65)
            float rb3 = 0.45f, chg3 = 0.65f;
            if (const char* e = getenv("SSQ_DEBUG_TILE3_RB")) if (atof(e) > 0) rb3 = (float)atof(e);
            if (const char* e = getenv("SSQ_DEBUG_TILE3_CHG")) if (atof(e) >= 0) chg3 = (float)atof(e);
            if ((rc = build(4, TILE3_NW, true, rb3, chg3, 1 << 20, &items3, &wave_first3, n_items3, tile3_ok))) return rc;
            // (the 16 lanes of a sub-row hold the sample window of the tile's 32 columns: (31 >> lgR) + 8 + 1 <= 16 needs
            // a decimation of 4 or more -- R_MIN of _tiles.py; SSQ_DEBUG_TILE_RMIN=2 plans go to tile2_kernel)
            for (int i = 0; i < nsegs; ++i) if (sg[i].kind && sg[i].lgR < 2) tile3_ok = false;
        }
    }
    {   // weights, per class (R phases from wtab_off on): [phase][4 tap pairs] -> [tap pair][phase]
        const TileSeg* sg = reinterpret_cast<const TileSeg*>(d.segs);
        const float* src = reinterpret_cast<const float*>(d.wtab);
        std::vector<float> w((size_t)16 * d.n_phases, 0.f);
        std::vector<char> done((size_t)d.n_phases, 0);
        for (int i = 0; i < nsegs; ++i) {
            if (sg[i].kind != 1) continue;
            const int64_t R = (int64_t)1 << sg[i].lgR, o = sg[i].wtab_off;
            SSQ_REQUIRE(o >= 0 && o + R <= d.n_phases, "tile segment %d: weights outside the table", i);
            if (done[(size_t)o]) continue;
            done[(size_t)o] = 1;
            for (int64_t ph = 0; ph < R; ++ph)
                for (int t = 0; t < 4; ++t)
                    for (int k = 0; k < 4; ++k)
                        w[(size_t)(16 * o + (t * R + ph) * 4 + k)] = src[(size_t)(16 * (o + ph) + 4 * t + k)];
        }
        if ((rc = up(&wtab, w.data(), (size_t)64 * d.n_phases))) return rc;
    }
    if ((rc = up(&tbank, d.tbank, (size_t)4 * d.n_tbank))) return rc;
    cls.resize(d.n_classes);
    // classes of 2^14 entries and more: four-step kernels, L = A B with A <= B, both 128 .. 2048
    // (SSQ_DEBUG_TILE_FFT=rocfft keeps every class on rocFFT)
    const bool own_fft = !(getenv("SSQ_DEBUG_TILE_FFT") && !strcmp(getenv("SSQ_DEBUG_TILE_FFT"), "rocfft"));
    int64_t y_entries = 0;
    for (int c = 0; c < d.n_classes; ++c) {
        cls[c] = {d.classes[4 * c], d.classes[4 * c + 1], d.classes[4 * c + 2], 0, 0, 0};
        SSQ_REQUIRE(cls[c].L >= 2 && (cls[c].L & (cls[c].L - 1)) == 0 && cls[c].nrows >= 1, "bad tile class %d", c);
        lmax = std::max(lmax, cls[c].L);
        int lg = 0;
        while (((int64_t)1 << lg) < cls[c].L) ++lg;
        if (own_fft && lg >= 13 && lg <= 22) {
            cls[c].B = 1 << ((lg + 1) / 2); cls[c].A = 1 << (lg / 2);
            y_entries += (int64_t)group * cls[c].nrows * cls[c].L;
        } else if (own_fft && lg >= 6 && lg <= 12) {
            cls[c].B = 1;                              // one-pass kernel
        }
    }
    std::vector<TileIRow> hi;
    hi.reserve((size_t)n_irows);
    for (int pass = 0; pass < 2; ++pass)            // rows sorted by class, the four-step classes first
        for (int c = 0; c < d.n_classes; ++c) {
            if ((cls[c].A > 0 || cls[c].B > 0) != (pass == 0)) continue;
            if (pass == 1 && n_irows_fft == 0) first_irow_fft = (int)hi.size();
            cls[c].first = (int)hi.size();
            for (int r = 0; r < n_irows; ++r) {
                const int64_t* q = d.irows + 8 * r;
                if ((int)q[6] != c) continue;
                SSQ_REQUIRE(q[4] == cls[c].L && q[2] >= 1 && 2 * q[2] <= q[4]
                            && q[1] >= 0 && q[1] + q[2] <= M / 2 + 1 && q[5] >= 0 && q[5] + q[2] <= d.n_tbank
                            && q[7] >= 0 && q[7] < cls[c].nrows, "bad tile row %d", r);
                hi.push_back({(int32_t)q[0], (int32_t)q[1], (int32_t)q[2], (int32_t)q[3], (int32_t)q[4], (int32_t)q[5],
                              (int32_t)((int64_t)group * cls[c].upre + q[7] * cls[c].L), (int32_t)(cls[c].nrows * cls[c].L)});
            }
            SSQ_REQUIRE((int64_t)hi.size() - cls[c].first == cls[c].nrows, "tile class %d: %lld rows listed, %lld declared",
                        c, (long long)((int64_t)hi.size() - cls[c].first), (long long)cls[c].nrows);
            if (pass == 1) n_irows_fft += (int)cls[c].nrows;
        }
    SSQ_REQUIRE((int)hi.size() == n_irows, "tile rows of unknown classes");
    if ((rc = up((void**)&irows, hi.data(), sizeof(TileIRow) * n_irows))) return rc;
    if (own_fft) {
        SSQ_CHECK_HIP(hipMalloc(&Y, (size_t)8 * std::max<int64_t>(y_entries, 1))); bytes += 8 * y_entries;
        std::vector<float> tw;
        for (int s = 0; s < 7; ++s) {
            const int Lp = 64 << s;
            ftw_off[s] = (int64_t)tw.size() / 2;
            for (int q = 0; q < Lp; ++q) {
                const double a = 6.283185307179586 * (double)q / (double)Lp;
                tw.push_back((float)std::cos(a)); tw.push_back((float)std::sin(a));
            }
        }
        if ((rc = up(&ftw, tw.data(), tw.size() * 4))) return rc;
    }
    // (+ 4 rows of slack: tile2_kernel's padded sub-rows read, and discard, the rows behind a class's last)
    SSQ_CHECK_HIP(hipMalloc(&U, (size_t)8 * (group * u_total + 4 * lmax))); bytes += 8 * (group * u_total + 4 * lmax);
    SSQ_CHECK_HIP(hipMemset(U, 0, (size_t)8 * (group * u_total + 4 * lmax)));
    SSQ_CHECK_HIP(hipMalloc((void**)&counters, 4096));       // [0]: tiles done
    SSQ_CHECK_HIP(hipMemset(counters, 0, 4096));
    for (size_t c = 0; c < cls.size(); ++c) {
        FftPlan fp;
        if (!cls[c].A && !cls[c].B) {
            rc = fp.create(1, SSQ_F32, (size_t)cls[c].L, (size_t)(group * cls[c].nrows), 1.0);
            if (rc) return rc;
            bytes += (int64_t)fp.work_bytes;
        }
        ffts.push_back(fp);
    }
    for (int t = 0; t < 5; ++t) n_items_tile[t] = d.n_items_tile[t];
    SSQ_CHECK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    SSQ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    SSQ_CHECK_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    return 0;
}

void TilePlan::destroy() {
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); side = nullptr; }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    ev_fork = ev_join = nullptr;
    for (auto& f : ffts) f.destroy();
    ffts.clear();
    void* ptrs[] = {steps, rows, irows, wtab, tbank, U, counters, Y, ftw, items2, wave_first2, items3, wave_first3};
    items2 = nullptr; wave_first2 = nullptr; items3 = nullptr; wave_first3 = nullptr;
    for (void* p : ptrs) if (p) (void)hipFree(p);
    steps = nullptr; rows = nullptr; irows = nullptr; wtab = tbank = U = Y = ftw = nullptr;
    counters = nullptr;
}

// SSQ_TILE_ORDER = ordered: the ticketed kernel (float32 sums in the reference's order, bit for bit; na <=
// 318); default: tile2_kernel (float64 tile, unordered adds: the same bins, sums rounded once)
bool tile_ordered() { return reassign_ordered(); }
// tile3_kernel (two columns per lane): 32-column tiles (up to 318 rows); SSQ_DEBUG_TILE_PAIR=0 keeps tile2_kernel (read at
// every call)
bool TilePlan::pair_ok() const {
    if (!tile3_ok || cols2 != 32 || N < 64) return false;
    const char* e = getenv("SSQ_DEBUG_TILE_PAIR");
    return !(e && atoi(e) == 0);
}
bool TilePlan::usable() const {
    if (!tile_ordered() && (tile2_ok || pair_ok())) return true;
    return tile_lds_bytes(na) <= 160 * 1024;
}
int TilePlan::tile_kernel() const {
    if (!usable()) return 0;
    if (tile_ordered()) return 1;
    return pair_ok() ? 3 : (tile2_ok ? 2 : 1);
}
int TilePlan::tile_cols() const { return !usable() ? 0 : (tile_ordered() || !(tile2_ok || pair_ok())) ? TILE_COLS : cols2; }

int TilePlan::run(int sig, int nsig, float* Wx, float* dWx, float* Tx, const unsigned short* kidx,
                  const void* cst, float cst0, const SsqParams& sp, hipStream_t stream, unsigned short* kdump) {
    SSQ_REQUIRE(usable(), "na = %lld: no tile kernel can run in this mode (the executor asks usable() first)", (long long)na);
    if (!tile_ordered() && pair_ok()) return run_pair(sig, nsig, Wx, dWx, Tx, kidx, cst, cst0, sp, stream, kdump);
    if (!tile_ordered() && tile2_ok) return run_f64(sig, nsig, Wx, dWx, Tx, kidx, cst, cst0, sp, stream, kdump);
    SSQ_REQUIRE(!kdump, "bin dump: the default tile kernel only (unset SSQ_TILE_ORDER)");
    return run_ordered(sig, nsig, Wx, dWx, Tx, kidx, cst, cst0, sp, stream);
}

int64_t TilePlan::tiles_done(hipStream_t stream) {
    unsigned long long v = 0;
    if (!counters) return 0;
    if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, counters, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

}  // namespace ssq
