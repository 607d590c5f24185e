// ssq_inverse.hip -- device side of the inverse transforms (C ABI: ssq_colsum,
// ssq_band_colsum, ssq_istft). The inverses of the reference are reductions over the
// scale / frequency axis of arrays that already live on the device:
//   icwt (one integral)  x[j] = sum_i Re(Wx[i,j]) / norm(scale_i)      _cwt.py:472-476
//   issq_cwt, issq_stft  x[j] = sum_i Re(Tx[i,j])  (optionally inside curve bands)
//                                                  _ssq_cwt.py:368-408, _ssq_stft.py:190-197
//   istft                irfft of every column, overlap-add with window^a, divided by
//                        the overlap-added window^(a+1)        _stft.py:238-256
// Sums run in the reference's order (ascending row, one accumulator per column, in the
// array's own precision), so the reductions are bit-identical to the NumPy results.
// Compiled with -ffp-contract=off.
#include "ssq_common.h"
#include "ssq_fft.h"
#include <rocfft/rocfft.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

namespace ssq {

// out[b][j] = sum_i Re(Z[b][i][j]) (/ div[i]); one thread per column, rows in order
template <typename T, bool DIV>
__global__ __launch_bounds__(64) void colsum_kernel(const T* __restrict__ Z, const T* __restrict__ div,
                                                     T* __restrict__ out, int64_t na, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const T* z = Z + 2 * ((int64_t)blockIdx.y * na * n + j);
    T acc = T(0);
    // the additions are a dependent chain (fixed order); the loads are not: 16 rows in flight
    constexpr int UN = 16;
    int64_t i = 0;
    for (; i + UN <= na; i += UN) {
        T v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = z[2 * (i + u) * n];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            T t = v[u];
            if (DIV) t = t / div[i + u];
            acc = acc + t;
        }
    }
    for (; i < na; ++i) {
        T v = z[2 * i * n];
        if (DIV) v = v / div[i];
        acc = acc + v;
    }
    out[(int64_t)blockIdx.y * n + j] = acc;
}

// blockIdx.y = k < K: rows lo[k][j] .. hi[k][j] of column j, accumulated in double (the
// reference builds the mask in complex128); blockIdx.y == K: the rows no band covers,
// accumulated in the data's precision (the reference zeroes them in a copy of Tx)
template <typename T>
__global__ __launch_bounds__(256) void band_colsum_kernel(const T* __restrict__ Z,
                                                          const int32_t* __restrict__ lo,
                                                          const int32_t* __restrict__ hi, int K,
                                                          double* __restrict__ out, int64_t na,
                                                          int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int k = (int)blockIdx.y;
    const T* z = Z + 2 * j;
    if (k < K) {
        double acc = 0.0;
        const int64_t a = lo[(int64_t)k * n + j], b = hi[(int64_t)k * n + j];
        for (int64_t i = a; i <= b && i < na; ++i) acc = acc + (double)z[2 * i * n];
        out[(int64_t)k * n + j] = acc;
    } else {
        T acc = T(0);
        for (int64_t i = 0; i < na; ++i) {
            bool covered = false;
            for (int c = 0; c < K; ++c)
                covered |= (i >= lo[(int64_t)c * n + j]) & (i <= hi[(int64_t)c * n + j]);
            if (!covered) acc = acc + z[2 * i * n];
        }
        out[(int64_t)K * n + j] = (double)acc;
    }
}

// St[c][f] = Sx[f][c] (one contiguous half-spectrum per frame for the C2R transform);
// the imaginary parts of DC and, for even n_fft, Nyquist do not enter an inverse real FFT
template <typename T>
__global__ __launch_bounds__(256) void spec_transpose_kernel(const T* __restrict__ Sx, T* __restrict__ St,
                                                             int64_t rows, int64_t n_hops, int even) {
    const int64_t total = rows * n_hops;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = t % rows, c = t / rows;
        T re = Sx[2 * (f * n_hops + c)], im = Sx[2 * (f * n_hops + c) + 1];
        if (f == 0 || (even && f == rows - 1)) im = T(0);
        St[2 * t] = re; St[2 * t + 1] = im;
    }
}

// overlap-add of the (unnormalised) inverse real transforms, window modulation, trim
template <typename T>
__global__ __launch_bounds__(256) void istft_ola_kernel(const T* __restrict__ frames,
                                                        const T* __restrict__ win_a,
                                                        const T* __restrict__ win_a1, T* __restrict__ x,
                                                        int64_t n_fft, int64_t n_hops, int64_t hop,
                                                        int64_t N, int modulated, T inv_n, T tiny) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int64_t p = s + n_fft / 2;                    // index in the untrimmed signal
    const int64_t half = n_fft / 2;
    const int64_t max_hops = (N - 1) / hop + 1;         // frames the norm counts (len(wn) = N + n_fft - 1)
    int64_t i0 = p - n_fft + 1;
    i0 = i0 <= 0 ? 0 : (i0 + hop - 1) / hop;
    const int64_t i1 = p / hop;
    T acc = T(0);
    double wn = 0.0;
    for (int64_t i = i0; i <= i1; ++i) {
        const int64_t r = p - i * hop;
        if (i < max_hops) wn = wn + (double)win_a1[r];
        if (i < n_hops) {
            int64_t src = r;
            if (modulated) { src = r - half; if (src < 0) src += n_fft; }   // fftshift along the frame
            const T v = frames[i * n_fft + src] * inv_n;
            acc = acc + v * win_a[r];
        }
    }
    if (wn > (double)tiny) acc = (T)((double)acc / wn);
    x[s] = acc;
}

// S[k] = sum_a F[a][k] * psih[a][k] (rows in order): the double-integral iCWT summed in
// the frequency domain, so that one inverse transform serves all scales
template <typename T>
__global__ __launch_bounds__(256) void mulsum_rows_kernel(const T* __restrict__ F, const T* __restrict__ psih,
                                                          T* __restrict__ S, int64_t na, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    T re = T(0), im = T(0);
    for (int64_t a = 0; a < na; ++a) {
        const T p = psih[a * n + k];
        re = re + F[2 * (a * n + k)] * p;
        im = im + F[2 * (a * n + k) + 1] * p;
    }
    S[2 * k] = re; S[2 * k + 1] = im;
}

template <typename T>
__global__ __launch_bounds__(256) void real_part_kernel(const T* __restrict__ S, T* __restrict__ out, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = S[2 * k];
}

static std::mutex g_icwt2_mu;
static std::map<std::tuple<int, int64_t, int64_t, hipStream_t>, std::pair<FftPlan, FftPlan>> g_icwt2_plans;

template <typename T>
static int icwt2_t(int dtype, void* Wp, const void* psih, void* out, int64_t na, int64_t n,
                   hipStream_t stream) {
    // plans (and their rocFFT work buffers) are per stream; the lock is held until everything
    // is enqueued, so two host threads cannot interleave set_stream / execute on one plan
    std::pair<FftPlan, FftPlan>* pp = nullptr;
    std::lock_guard<std::mutex> lock(g_icwt2_mu);
    {
        auto key = std::make_tuple(dtype, na, n, stream);
        if (g_icwt2_plans.size() >= 16 && !g_icwt2_plans.count(key)) {
            (void)hipDeviceSynchronize();
            for (auto& kv : g_icwt2_plans) { kv.second.first.destroy(); kv.second.second.destroy(); }
            g_icwt2_plans.clear();
        }
        auto it = g_icwt2_plans.find(key);
        if (it == g_icwt2_plans.end()) {
            std::pair<FftPlan, FftPlan> pr;
            int rc = pr.first.create(2, dtype, (size_t)n, (size_t)na, 1.0);
            if (rc) return rc;
            rc = pr.second.create(1, dtype, (size_t)n, 1, 1.0 / (double)n);
            if (rc) return rc;
            it = g_icwt2_plans.emplace(key, pr).first;
        }
        pp = &it->second;
    }
    int rc = pp->first.execute(Wp, nullptr, stream);              // forward FFT of every row, in place
    if (rc) return rc;
    T* S = nullptr;
    SSQ_CHECK_HIP(hipMallocAsync((void**)&S, (size_t)n * 2 * sizeof(T), stream));
    dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL((mulsum_rows_kernel<T>), grid, dim3(256), 0, stream, (const T*)Wp, (const T*)psih, S, na, n);
    rc = (hipGetLastError() == hipSuccess) ? pp->second.execute(S, nullptr, stream) : -3;
    if (!rc) {
        hipLaunchKernelGGL((real_part_kernel<T>), grid, dim3(256), 0, stream, (const T*)S, (T*)out, n);
        if (hipGetLastError() != hipSuccess) rc = -3;
    }
    if (rc == -3) set_error("icwt2 kernel launch failed");
    (void)hipFreeAsync(S, stream);
    return rc;
}

// ------------------------------------------------------------------- trigdiff
// F[r][k] *= 1j * xi[k] * fs, formed as the reference's complex expression
// `A_freqdom * 1j * xi * fs` rounds it (utils/common.py:220): the rotation is exact, then one
// rounding per real factor
template <typename T>
__global__ __launch_bounds__(256) void mul_ixi_kernel(T* __restrict__ F, const T* __restrict__ xi, T fs,
                                                      int64_t rows, int64_t n) {
    const int64_t total = rows * n;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const T x = xi[q % n];
        const T re = F[2 * q], im = F[2 * q + 1];
        F[2 * q] = (-im * x) * fs;
        F[2 * q + 1] = (re * x) * fs;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void unpad_rows_kernel(const T* __restrict__ F, T* __restrict__ out,
                                                         int64_t rows, int64_t n_up, int64_t n1, int64_t N) {
    const int64_t total = rows * N;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t r = q / N, j = q - r * N;
        out[2 * q] = F[2 * (r * n_up + n1 + j)];
        out[2 * q + 1] = F[2 * (r * n_up + n1 + j) + 1];
    }
}

static std::mutex g_trig_mu;
static std::map<std::tuple<int, int64_t, int64_t, hipStream_t>, std::pair<FftPlan, FftPlan>> g_trig_plans;

template <typename T>
static int trigdiff_t(int dtype, void* Ap, const void* xi, double fs, void* out, int64_t rows,
                      int64_t n_up, int64_t n1, int64_t N, hipStream_t stream) {
    std::pair<FftPlan, FftPlan>* pp = nullptr;
    std::lock_guard<std::mutex> lock(g_trig_mu);          // per-stream plans; held until enqueued
    {
        auto key = std::make_tuple(dtype, rows, n_up, stream);
        if (g_trig_plans.size() >= 16 && !g_trig_plans.count(key)) {
            (void)hipDeviceSynchronize();
            for (auto& kv : g_trig_plans) { kv.second.first.destroy(); kv.second.second.destroy(); }
            g_trig_plans.clear();
        }
        auto it = g_trig_plans.find(key);
        if (it == g_trig_plans.end()) {
            std::pair<FftPlan, FftPlan> pr;
            int rc = pr.first.create(2, dtype, (size_t)n_up, (size_t)rows, 1.0);
            if (rc) return rc;
            rc = pr.second.create(1, dtype, (size_t)n_up, (size_t)rows, 1.0 / (double)n_up);
            if (rc) return rc;
            it = g_trig_plans.emplace(key, pr).first;
        }
        pp = &it->second;
    }
    int rc = pp->first.execute(Ap, nullptr, stream);
    if (rc) return rc;
    const int64_t total = rows * n_up;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL((mul_ixi_kernel<T>), dim3(blocks), dim3(256), 0, stream, (T*)Ap, (const T*)xi, (T)fs,
                       rows, n_up);
    SSQ_LAUNCH_CHECK();
    rc = pp->second.execute(Ap, nullptr, stream);
    if (rc) return rc;
    const unsigned blocks2 = (unsigned)std::min<int64_t>((rows * N + 255) / 256, 65536);
    hipLaunchKernelGGL((unpad_rows_kernel<T>), dim3(blocks2), dim3(256), 0, stream, (const T*)Ap, (T*)out, rows,
                       n_up, n1, N);
    SSQ_LAUNCH_CHECK();
    return 0;
}

struct IstftFft {
    rocfft_plan plan = nullptr; rocfft_execution_info info = nullptr; void* work = nullptr;
};
static std::mutex g_istft_mu;
static std::map<std::tuple<int, int64_t, int64_t, hipStream_t>, IstftFft> g_istft_plans;   // one work buffer per stream

// (the caller holds g_istft_mu until the transform is enqueued)
static int istft_plan(int dtype, int64_t n_fft, int64_t n_hops, hipStream_t stream, IstftFft** out) {
    auto key = std::make_tuple(dtype, n_fft, n_hops, stream);
    if (g_istft_plans.size() >= 16 && !g_istft_plans.count(key)) {      // bounded cache
        (void)hipDeviceSynchronize();
        for (auto& kv : g_istft_plans) {
            rocfft_execution_info_destroy(kv.second.info); rocfft_plan_destroy(kv.second.plan);
            if (kv.second.work) (void)hipFree(kv.second.work);
        }
        g_istft_plans.clear();
    }
    auto it = g_istft_plans.find(key);
    if (it == g_istft_plans.end()) {
        if (fft_global_setup()) return -4;
        IstftFft f;
        size_t len = (size_t)n_fft;
        rocfft_status st = rocfft_plan_create(&f.plan, rocfft_placement_notinplace,
                rocfft_transform_type_real_inverse,
                dtype == SSQ_F32 ? rocfft_precision_single : rocfft_precision_double, 1, &len,
                (size_t)n_hops, nullptr);
        if (st != rocfft_status_success) { set_error("rocfft_plan_create (istft) failed: %d", (int)st); return -4; }
        size_t wb = 0;
        rocfft_plan_get_work_buffer_size(f.plan, &wb);
        rocfft_execution_info_create(&f.info);
        if (wb) {
            SSQ_CHECK_HIP(hipMalloc(&f.work, wb));
            rocfft_execution_info_set_work_buffer(f.info, f.work, wb);
        }
        it = g_istft_plans.emplace(key, f).first;
    }
    *out = &it->second;
    return 0;
}

template <typename T>
static int istft_t(int dtype, const void* Sx, const void* win_a, const void* win_a1, void* x,
                   int64_t n_fft, int64_t n_hops, int64_t hop, int64_t N, int modulated,
                   hipStream_t stream) {
    const int64_t rows = n_fft / 2 + 1;
    T* St = nullptr; T* frames = nullptr;
    SSQ_CHECK_HIP(hipMallocAsync((void**)&St, (size_t)rows * n_hops * 2 * sizeof(T), stream));
    SSQ_CHECK_HIP(hipMallocAsync((void**)&frames, (size_t)(n_fft + 2) * n_hops * sizeof(T), stream));
    const int64_t total = rows * n_hops;
    unsigned g = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL((spec_transpose_kernel<T>), dim3(g), dim3(256), 0, stream, (const T*)Sx, St, rows,
                       n_hops, (int)(n_fft % 2 == 0));
    SSQ_LAUNCH_CHECK();
    IstftFft* f = nullptr;
    std::unique_lock<std::mutex> lock(g_istft_mu);
    int rc = istft_plan(dtype, n_fft, n_hops, stream, &f);
    if (!rc) {
        rocfft_execution_info_set_stream(f->info, stream);
        void* ins[1] = {St}; void* outs[1] = {frames};
        if (rocfft_execute(f->plan, ins, outs, f->info) != rocfft_status_success) {
            set_error("rocfft_execute (istft) failed"); rc = -4;
        }
    }
    lock.unlock();
    if (!rc) {
        const T tiny = sizeof(T) == 4 ? (T)1.17549435e-38f : (T)2.2250738585072014e-308;
        hipLaunchKernelGGL((istft_ola_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream,
                           (const T*)frames, (const T*)win_a, (const T*)win_a1, (T*)x, n_fft, n_hops,
                           hop, N, modulated, (T)(T(1) / (T)n_fft), tiny);
        if (hipGetLastError() != hipSuccess) { set_error("istft_ola launch failed"); rc = -3; }
    }
    (void)hipFreeAsync(St, stream);
    (void)hipFreeAsync(frames, stream);
    return rc;
}

}  // namespace ssq

using namespace ssq;

extern "C" {

int ssq_colsum(int dtype, const void* Z, const void* divisor, void* out, int64_t batch, int64_t na,
               int64_t n, void* stream) {
    SSQ_REQUIRE(Z && out, "ssq_colsum: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(batch >= 1 && batch <= 65535 && na >= 1 && n >= 1, "colsum: bad shape (%lld, %lld, %lld)",
                (long long)batch, (long long)na, (long long)n);
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)batch);
    hipStream_t s = as_stream(stream);
#define LAUNCH(T, DIV) hipLaunchKernelGGL((colsum_kernel<T, DIV>), grid, dim3(64), 0, s, (const T*)Z, \
                                          (const T*)divisor, (T*)out, na, n)
    if (dtype == SSQ_F32) { if (divisor) LAUNCH(float, true); else LAUNCH(float, false); }
    else { if (divisor) LAUNCH(double, true); else LAUNCH(double, false); }
#undef LAUNCH
    SSQ_LAUNCH_CHECK();
    return 0;
}

int ssq_band_colsum(int dtype, const void* Z, const int32_t* lo, const int32_t* hi, int64_t ncomp,
                    double* out, int64_t na, int64_t n, void* stream) {
    SSQ_REQUIRE(Z && lo && hi && out, "ssq_band_colsum: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(ncomp >= 1 && ncomp <= 65534 && na >= 1 && n >= 1, "band_colsum: bad shape");
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)(ncomp + 1));
    hipStream_t s = as_stream(stream);
    if (dtype == SSQ_F32)
        hipLaunchKernelGGL((band_colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)Z, lo, hi,
                           (int)ncomp, out, na, n);
    else
        hipLaunchKernelGGL((band_colsum_kernel<double>), grid, dim3(256), 0, s, (const double*)Z, lo, hi,
                           (int)ncomp, out, na, n);
    SSQ_LAUNCH_CHECK();
    return 0;
}

int ssq_icwt2(int dtype, void* Wp, const void* psih, void* out, int64_t na, int64_t n_up, void* stream) {
    SSQ_REQUIRE(Wp && psih && out, "ssq_icwt2: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(na >= 1 && n_up >= 2, "icwt2: bad shape (%lld, %lld)", (long long)na, (long long)n_up);
    if (dtype == SSQ_F32) return icwt2_t<float>(dtype, Wp, psih, out, na, n_up, as_stream(stream));
    return icwt2_t<double>(dtype, Wp, psih, out, na, n_up, as_stream(stream));
}

int ssq_trigdiff(int dtype, void* Ap, const void* xi, double fs, void* out, int64_t rows, int64_t n_up,
                 int64_t n1, int64_t N, void* stream) {
    SSQ_REQUIRE(Ap && xi && out, "ssq_trigdiff: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(rows >= 1 && n_up >= 2 && n1 >= 0 && N >= 1 && n1 + N <= n_up,
                "trigdiff: bad shape (%lld, %lld) / slice (%lld, %lld)", (long long)rows, (long long)n_up,
                (long long)n1, (long long)N);
    if (dtype == SSQ_F32) return trigdiff_t<float>(dtype, Ap, xi, fs, out, rows, n_up, n1, N, as_stream(stream));
    return trigdiff_t<double>(dtype, Ap, xi, fs, out, rows, n_up, n1, N, as_stream(stream));
}

int ssq_istft(int dtype, const void* Sx, const void* win_a, const void* win_a1, void* x, int64_t n_fft,
              int64_t n_hops, int64_t hop_len, int64_t N, int modulated, void* stream) {
    SSQ_REQUIRE(Sx && win_a && win_a1 && x, "ssq_istft: null pointer");
    SSQ_REQUIRE(dtype == SSQ_F32 || dtype == SSQ_F64, "bad dtype %d", dtype);
    SSQ_REQUIRE(n_fft >= 2 && n_hops >= 1 && hop_len >= 1 && N >= 1, "istft: bad sizes");
    if (dtype == SSQ_F32)
        return istft_t<float>(dtype, Sx, win_a, win_a1, x, n_fft, n_hops, hop_len, N, modulated, as_stream(stream));
    return istft_t<double>(dtype, Sx, win_a, win_a1, x, n_fft, n_hops, hop_len, N, modulated, as_stream(stream));
}

}  // extern "C"
