// emu_globals.cpp -- state shared by the translation units of the emulated library (see
// hip/hip_runtime.h): the per-OS-thread fiber scheduler and the dynamic-LDS buffers that the
// `extern __shared__` arrays of csrc/ssq_kernels.hip and csrc/ssq_ridge.hip resolve to (one
// workgroup at a time per OS thread). TEST INFRASTRUCTURE ONLY.
#include "hip/hip_runtime.h"

namespace emu {
thread_local Worker t_worker;
}
namespace ssq {
alignas(64) thread_local unsigned char lds_raw[160 * 1024];
alignas(64) thread_local unsigned char smem[160 * 1024];
}
