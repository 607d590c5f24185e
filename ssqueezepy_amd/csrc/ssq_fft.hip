// ssq_fft.hip -- rocFFT plan wrapper (host code only).
#include "ssq_fft.h"
#include <mutex>

namespace ssq {

#define SSQ_CHECK_FFT(expr)                                                        \
    do {                                                                           \
        rocfft_status _s = (expr);                                                 \
        if (_s != rocfft_status_success) {                                         \
            set_error("%s failed: rocfft_status %d (%s:%d)", #expr, (int)_s,       \
                      __FILE__, __LINE__);                                         \
            return -4;                                                             \
        }                                                                          \
    } while (0)

int fft_global_setup() {
    static std::once_flag once;
    static int rc = 0;
    std::call_once(once, [] {
        if (rocfft_setup() != rocfft_status_success) rc = -4;
    });
    if (rc) set_error("rocfft_setup failed");
    return rc;
}

int FftPlan::create(int kind, int dtype, size_t length, size_t batch, double scale,
                    size_t in_dist, size_t out_dist) {
    if (fft_global_setup()) return -4;
    rocfft_plan_description desc = nullptr;
    SSQ_CHECK_FFT(rocfft_plan_description_create(&desc));
    rocfft_transform_type tt;
    rocfft_result_placement place;
    rocfft_array_type in_t, out_t;
    if (kind == 0) {
        tt = rocfft_transform_type_real_forward;
        place = rocfft_placement_notinplace;
        in_t = rocfft_array_type_real;
        out_t = rocfft_array_type_hermitian_interleaved;
        if (!in_dist) in_dist = length;
        if (!out_dist) out_dist = length / 2 + 1;
    } else {
        tt = kind == 1 ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward;
        place = rocfft_placement_inplace;
        in_t = out_t = rocfft_array_type_complex_interleaved;
        if (!in_dist) in_dist = length;
        if (!out_dist) out_dist = length;
    }
    size_t stride = 1, offs = 0;
    SSQ_CHECK_FFT(rocfft_plan_description_set_data_layout(desc, in_t, out_t, &offs, &offs, 1, &stride,
                                                          in_dist, 1, &stride, out_dist));
    if (scale != 1.0) SSQ_CHECK_FFT(rocfft_plan_description_set_scale_factor(desc, scale));
    SSQ_CHECK_FFT(rocfft_plan_create(&plan, place, tt,
                                     dtype == SSQ_F32 ? rocfft_precision_single : rocfft_precision_double,
                                     1, &length, batch, desc));
    rocfft_plan_description_destroy(desc);
    SSQ_CHECK_FFT(rocfft_plan_get_work_buffer_size(plan, &work_bytes));
    SSQ_CHECK_FFT(rocfft_execution_info_create(&info));
    if (work_bytes) {
        SSQ_CHECK_HIP(hipMalloc(&work, work_bytes));
        SSQ_CHECK_FFT(rocfft_execution_info_set_work_buffer(info, work, work_bytes));
    }
    return 0;
}

int FftPlan::execute(void* in, void* out, hipStream_t stream) {
    SSQ_CHECK_FFT(rocfft_execution_info_set_stream(info, stream));
    void* ins[1] = {in};
    void* outs[1] = {out};
    SSQ_CHECK_FFT(rocfft_execute(plan, ins, out ? outs : nullptr, info));
    return 0;
}

void FftPlan::destroy() {
    if (info) rocfft_execution_info_destroy(info);
    if (plan) rocfft_plan_destroy(plan);
    if (work) (void)hipFree(work);
    info = nullptr; plan = nullptr; work = nullptr; work_bytes = 0;
}

}  // namespace ssq
