// ssq_fft.h -- thin RAII-free wrapper over rocFFT plans used by the CWT/STFT plans.
#pragma once
#include "ssq_common.h"
#include <rocfft/rocfft.h>

namespace ssq {

int fft_global_setup();

struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    size_t work_bytes = 0;

    // 1-D batched transform. kind: 0 = real->hermitian forward (out of place),
    // 1 = complex inverse in place, 2 = complex forward in place.
    // `scale` multiplies the result (1/M for a normalised inverse).
    int create(int kind, int dtype, size_t length, size_t batch, double scale,
               size_t in_dist = 0, size_t out_dist = 0);
    int execute(void* in, void* out, hipStream_t stream);
    void destroy();
};

}  // namespace ssq
