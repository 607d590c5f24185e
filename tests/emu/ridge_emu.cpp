// ridge_emu.cpp -- csrc/ssq_ridge.hip compiled for the host (see hip/hip_runtime.h here):
// the same kernels and the same C entry points, operating on host memory.
// TEST INFRASTRUCTURE ONLY (tests/test_ridge_kernels_emulated.py).
#include "hip/hip_runtime.h"
#include <cstdarg>
#include <cstdio>

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local Block* t_block = nullptr;
}
// the workgroup's dynamic LDS (blocks run one after another)
namespace ssq { alignas(64) unsigned char smem[160 * 1024]; }

#include "../../ssqueezepy_amd/csrc/ssq_common.h"
namespace ssq {
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}
}
#include "../../ssqueezepy_amd/csrc/ssq_ridge.hip"
