// kernels_emu.cpp -- csrc/ssq_kernels.hip (phase transforms, reassignment, framing, padding)
// compiled for the host (see hip/hip_runtime.h here): the same kernels and the same C entry
// points, operating on host memory. TEST INFRASTRUCTURE ONLY
// (tests/test_reassign_kernels_emulated.py).
#include "hip/hip_runtime.h"
#include <cstdarg>
#include <cstdio>

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local Block* t_block = nullptr;
}
// the workgroup's dynamic LDS (blocks run one after another)
namespace ssq { alignas(64) unsigned char lds_raw[160 * 1024]; }

#include "../../ssqueezepy_amd/csrc/ssq_kernels.hip"
