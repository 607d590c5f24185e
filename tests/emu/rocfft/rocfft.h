// Host stand-in for <rocfft/rocfft.h>: the subset of the rocFFT API that csrc/ uses (1-D
// batched real-forward / real-inverse / complex transforms with strides, distances and a
// scale factor), computed in double precision on the CPU. TEST INFRASTRUCTURE ONLY
// (tests/emu/): lets the plans run where there is no GPU.
#pragma once
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdlib>
#include <vector>

typedef enum { rocfft_status_success = 0, rocfft_status_failure = 1 } rocfft_status;
typedef enum { rocfft_transform_type_complex_forward, rocfft_transform_type_complex_inverse,
               rocfft_transform_type_real_forward, rocfft_transform_type_real_inverse } rocfft_transform_type;
typedef enum { rocfft_precision_single, rocfft_precision_double } rocfft_precision;
typedef enum { rocfft_placement_inplace, rocfft_placement_notinplace } rocfft_result_placement;
typedef enum { rocfft_array_type_complex_interleaved, rocfft_array_type_real,
               rocfft_array_type_hermitian_interleaved } rocfft_array_type;

struct rocfft_plan_description_t {
    bool has_layout = false;
    size_t in_stride = 1, in_dist = 0, out_stride = 1, out_dist = 0;
    double scale = 1.0;
};
struct rocfft_plan_t {
    rocfft_result_placement place; rocfft_transform_type type; rocfft_precision prec;
    size_t n, batch; rocfft_plan_description_t d;
};
struct rocfft_execution_info_t { int unused; };
typedef rocfft_plan_description_t* rocfft_plan_description;
typedef rocfft_plan_t* rocfft_plan;
typedef rocfft_execution_info_t* rocfft_execution_info;

inline rocfft_status rocfft_setup() { return rocfft_status_success; }
inline rocfft_status rocfft_plan_description_create(rocfft_plan_description* d) {
    *d = new rocfft_plan_description_t; return rocfft_status_success;
}
inline rocfft_status rocfft_plan_description_destroy(rocfft_plan_description d) { delete d; return rocfft_status_success; }
inline rocfft_status rocfft_plan_description_set_data_layout(
    rocfft_plan_description d, rocfft_array_type, rocfft_array_type, const size_t*, const size_t*,
    size_t, const size_t* in_strides, size_t in_dist, size_t, const size_t* out_strides, size_t out_dist) {
    d->has_layout = true;
    d->in_stride = in_strides[0]; d->in_dist = in_dist;
    d->out_stride = out_strides[0]; d->out_dist = out_dist;
    return rocfft_status_success;
}
inline rocfft_status rocfft_plan_description_set_scale_factor(rocfft_plan_description d, double s) {
    d->scale = s; return rocfft_status_success;
}
inline rocfft_status rocfft_plan_create(rocfft_plan* p, rocfft_result_placement place,
                                        rocfft_transform_type type, rocfft_precision prec, size_t dims,
                                        const size_t* lengths, size_t batch, rocfft_plan_description d) {
    if (dims != 1) return rocfft_status_failure;
    rocfft_plan_t* q = new rocfft_plan_t;
    q->place = place; q->type = type; q->prec = prec; q->n = lengths[0]; q->batch = batch;
    if (d) q->d = *d;
    const size_t n = q->n, h = n / 2 + 1;
    if (!q->d.has_layout || !q->d.in_dist)
        q->d.in_dist = type == rocfft_transform_type_real_inverse ? h : n;
    if (!q->d.has_layout || !q->d.out_dist)
        q->d.out_dist = type == rocfft_transform_type_real_forward ? h : n;
    *p = q;
    return rocfft_status_success;
}
inline rocfft_status rocfft_plan_destroy(rocfft_plan p) { delete p; return rocfft_status_success; }
inline rocfft_status rocfft_plan_get_work_buffer_size(rocfft_plan, size_t* b) { *b = 0; return rocfft_status_success; }
inline rocfft_status rocfft_execution_info_create(rocfft_execution_info* i) {
    *i = new rocfft_execution_info_t; return rocfft_status_success;
}
inline rocfft_status rocfft_execution_info_destroy(rocfft_execution_info i) { delete i; return rocfft_status_success; }
inline rocfft_status rocfft_execution_info_set_work_buffer(rocfft_execution_info, void*, size_t) { return rocfft_status_success; }
inline rocfft_status rocfft_execution_info_set_stream(rocfft_execution_info, void*) { return rocfft_status_success; }

namespace emu_fft {
// unnormalised DFT in working precision W, sign = -1 forward / +1 inverse; radix-2 when n is a
// power of two, else direct
template <typename W>
inline void dft(std::vector<std::complex<W>>& a, int sign) {
    typedef std::complex<W> cd;
    const size_t n = a.size();
    if (n <= 1) return;
    const W PI = (W)3.141592653589793238462643383279502884L;
    if ((n & (n - 1)) == 0) {
        for (size_t i = 1, j = 0; i < n; ++i) {
            size_t bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            std::vector<cd> w(len / 2);
            for (size_t k = 0; k < len / 2; ++k) {
                const W ang = sign * (W)2 * PI * (W)k / (W)len;
                w[k] = cd(std::cos(ang), std::sin(ang));
            }
            for (size_t i = 0; i < n; i += len)
                for (size_t k = 0; k < len / 2; ++k) {
                    const cd u = a[i + k], v = a[i + k + len / 2] * w[k];
                    a[i + k] = u + v; a[i + k + len / 2] = u - v;
                }
        }
    } else {
        std::vector<cd> out(n);
        for (size_t k = 0; k < n; ++k) {
            cd s = 0;
            for (size_t t = 0; t < n; ++t) {
                const W ang = sign * (W)2 * PI * (W)((k * t) % n) / (W)n;
                s += a[t] * cd(std::cos(ang), std::sin(ang));
            }
            out[k] = s;
        }
        a.swap(out);
    }
}
// R: data precision; W: working precision (double by default; SSQ_EMU_FFT32=1 computes float32
// plans in float32, so that tolerances are exercised at rocFFT-like accuracy)
template <typename R, typename W>
void run(rocfft_plan p, void* in, void* out) {
    typedef std::complex<W> cd;
    const size_t n = p->n, h = n / 2 + 1;
    const auto& d = p->d;
    std::vector<cd> a(n);
    for (size_t b = 0; b < p->batch; ++b) {
        if (p->type == rocfft_transform_type_real_forward) {
            const R* x = (const R*)in + b * d.in_dist;
            for (size_t t = 0; t < n; ++t) a[t] = cd((W)x[t * d.in_stride], (W)0);
            dft(a, -1);
            R* y = (R*)out + 2 * b * d.out_dist;
            for (size_t k = 0; k < h; ++k) {
                y[2 * k * d.out_stride] = (R)(a[k].real() * (W)d.scale);
                y[2 * k * d.out_stride + 1] = (R)(a[k].imag() * (W)d.scale);
            }
        } else if (p->type == rocfft_transform_type_real_inverse) {
            const R* x = (const R*)in + 2 * b * d.in_dist;
            for (size_t k = 0; k < h; ++k) a[k] = cd((W)x[2 * k * d.in_stride], (W)x[2 * k * d.in_stride + 1]);
            for (size_t k = h; k < n; ++k) a[k] = std::conj(a[n - k]);
            a[0] = cd(a[0].real(), (W)0);
            if (n % 2 == 0) a[n / 2] = cd(a[n / 2].real(), (W)0);
            dft(a, +1);
            R* y = (R*)out + b * d.out_dist;
            for (size_t t = 0; t < n; ++t) y[t * d.out_stride] = (R)(a[t].real() * (W)d.scale);
        } else {
            const int sign = p->type == rocfft_transform_type_complex_forward ? -1 : +1;
            const R* x = (const R*)in + 2 * b * d.in_dist;
            for (size_t t = 0; t < n; ++t) a[t] = cd((W)x[2 * t * d.in_stride], (W)x[2 * t * d.in_stride + 1]);
            dft(a, sign);
            R* y = (R*)(p->place == rocfft_placement_inplace ? in : out) + 2 * b * d.out_dist;
            for (size_t t = 0; t < n; ++t) {
                y[2 * t * d.out_stride] = (R)(a[t].real() * (W)d.scale);
                y[2 * t * d.out_stride + 1] = (R)(a[t].imag() * (W)d.scale);
            }
        }
    }
}
}  // namespace emu_fft

inline rocfft_status rocfft_execute(rocfft_plan p, void** in, void** out, rocfft_execution_info) {
    void* o = out ? out[0] : nullptr;
    static const bool fft32 = getenv("SSQ_EMU_FFT32") != nullptr;
    if (p->prec == rocfft_precision_single) {
        if (fft32) emu_fft::run<float, float>(p, in[0], o);
        else emu_fft::run<float, double>(p, in[0], o);
    } else {
        emu_fft::run<double, double>(p, in[0], o);
    }
    return rocfft_status_success;
}
