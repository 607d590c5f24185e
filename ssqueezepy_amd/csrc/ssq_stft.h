// ssq_stft.h -- what the STFT translation units share: the fused kernels' argument block and the entry points of
// the mixed-radix kernel (ssq_stft_generic.hip) the plan (ssq_stft.hip) calls.
#pragma once
#include "ssq_common.h"
#include "ssq_ldsfft.h"

namespace ssq {

struct StftFusedArgs {
    const float* xp; const float* window; const float* diff_window; const c32* ftw;
    float2* Sx; float2* dSx;            // dSx null: derivative not stored
    // fused ssq_stft: the bin of every point (2 bytes) instead of dSx (8 bytes)
    unsigned short* kidx; const float* Sfs; double gamma;
    int64_t padlen, n_hops, rows;
    int hop, s20, s21, modulated;
    int xcd;                            // grid.x padded to a multiple of 8, remapped per XCD
    // REASSIGN instantiation: Tx of the workgroup's frames is summed in LDS (float64, unordered adds -- see
    // accumulate_f64_kernel) and written here; neither the bin map nor a second pass over Sx is needed
    float2* Tx; const void* cst; int cst_uniform;
    // the frames read the signal itself through the padding rule (round 6: no padded copy -- config 3's pad_kernel was
    // 7 % of the call): x (batch, n), the left padding, the rule (SSQ_PAD_*); null: xp holds the padded batch
    const float* x; int n, n1, padtype;
    const float2* wd;                   // (window, diff_window) interleaved, or null (stft_fused_kernel's staged loads)
    int batch;                          // signals of the call (stft_fused_kernel's items: batch x groups of frames)
};

// padded sample t - n1 of a signal of n samples: the source index in [0, n), or -1 for a zero (the rule of pad_kernel,
// ssq_kernels.hip, in 32-bit arithmetic)
__device__ __forceinline__ int stft_pad_source(int t, int n, int padtype) {
    if ((unsigned)t < (unsigned)n) return t;
    switch (padtype) {
        case SSQ_PAD_REFLECT: {
            if (n == 1) return 0;
            const int period = 2 * (n - 1);
            int m = t % period; if (m < 0) m += period;
            return m < n ? m : period - m;
        }
        case SSQ_PAD_SYMMETRIC: {
            const int period = 2 * n;
            int m = t % period; if (m < 0) m += period;
            return m < n ? m : period - 1 - m;
        }
        case SSQ_PAD_REPLICATE: return t < 0 ? 0 : n - 1;
        case SSQ_PAD_WRAP: { int m = t % n; if (m < 0) m += n; return m; }
        default: return -1;
    }
}

// ---- mixed-radix fused kernel (any n_fft with prime factors <= 31, float32)
constexpr int GEN_NT = 256;
constexpr int GEN_MAX_PASSES = 10;
constexpr int GEN_MAX_PPT = 40;                     // complex points a work-item holds in a pass (n_fft * G / 256 <= 39)
constexpr int GEN_LDS_BYTES = 78 * 1024;            // n_fft * G * 8 bytes at most: two workgroups per CU
constexpr int GEN_MAX_EPI = 10;                     // epilogue points a work-item holds: (n_fft / 2 + 1) * G / 256, rounded up
// factors of n (16s, 8s, 4s, a 2, then the odd primes up to 31) and the frames per workgroup; false when n has a
// larger prime factor, too many factors, or does not fit the LDS
bool stft_generic_plan(int64_t n, int* radix, int* npass, int* G);
int launch_stft_generic(const StftFusedArgs& A, const SsqParams& sp, const c32* tw, int n, const int* radix,
                        int npass, int G, int64_t batch, hipStream_t stream);

}  // namespace ssq
