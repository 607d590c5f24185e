// ssq_common.h -- internal helpers shared by the translation units of libssq_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/ssq_hip.h"

namespace ssq {

void set_error(const char* fmt, ...);

#define SSQ_CHECK_HIP(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            ssq::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                           __FILE__, __LINE__);                                    \
            return -2;                                                             \
        }                                                                          \
    } while (0)

#define SSQ_REQUIRE(cond, ...)                                                     \
    do {                                                                           \
        if (!(cond)) {                                                             \
            ssq::set_error(__VA_ARGS__);                                           \
            return -1;                                                             \
        }                                                                          \
    } while (0)

#define SSQ_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t _e = hipGetLastError();                                         \
        if (_e != hipSuccess) {                                                    \
            ssq::set_error("kernel launch failed: %s (%s:%d)",                     \
                           hipGetErrorString(_e), __FILE__, __LINE__);             \
            return -3;                                                             \
        }                                                                          \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T> struct cplx { T re, im; };

// reassignment parameters as the kernels consume them
struct SsqParams {
    double p[5];      // see SSQ_GRID_* in ssq_hip.h
    double gamma;
    int    grid;
    int    flipud;
    int    cst_f64;
};

// ---- launchers implemented in ssq_kernels.hip, used by the plans --------------
// bin source for the accumulate kernel
enum BinSrc { BIN_FROM_DWX = 0, BIN_FROM_W = 1, BIN_FROM_KIDX = 2 };

// Tx <- accumulate(Wx, src) ; src is dWx (complex), w (real) or kidx (uint16)
int launch_accumulate(int dtype, int binsrc, const void* Wx, const void* src,
                      const void* Sfs, void* Tx, const void* cst,
                      const SsqParams& sp, int64_t batch, int64_t na, int64_t n,
                      int32_t* kmap, hipStream_t stream);

}  // namespace ssq
