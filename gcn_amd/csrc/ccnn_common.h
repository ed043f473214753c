// Shared host/device helpers for libccnn_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ccnn.h"

namespace ccnn {

constexpr int F = CCNN_FILTERS;       // 20 filters
constexpr int K = CCNN_FILTER_SIZE;   // 171 taps = 19 bands x 9 neighbours
constexpr int H = CCNN_HIDDEN;        // 100
constexpr int OMAX = CCNN_MAX_CLASSES;   // 16: the output layer fits one 16-wide tile / 16 lanes whatever the class count
constexpr int NB = CCNN_PER_LAYER;    // 9 values per band
constexpr int WSTEP = NB * CCNN_STRIDE;  // 18: cube offset between conv windows
constexpr int PCHUNK = 64;            // pooled features per FC1 K-chunk

struct Dims : ccnn_dims_t {};

// `geometry` is a band count, or CCNN_GEOMETRY(bands, classes) for a class count other than 13.
inline int make_dims(int geometry, ccnn_dims_t *d)
{
    if (!d || geometry < 0) return CCNN_E_INVALID;
    const int z = geometry & 0xffff;
    const int classes = (geometry >> 16) ? (geometry >> 16) : CCNN_CLASSES;
    if (z < CCNN_MIN_BANDS || z > CCNN_MAX_BANDS) return CCNN_E_INVALID;
    if (classes < CCNN_MIN_CLASSES || classes > CCNN_MAX_CLASSES) return CCNN_E_INVALID;
    d->z = z;
    d->n_classes = classes;
    d->cube = NB * z;
    d->re = (z - CCNN_COV_LEN - 1) / CCNN_STRIDE + 1;   // gcn_mbgd_fp64.cu:199
    d->mre = (d->re - 1) / CCNN_POOL_LEN + 1;
    d->pin = d->re * F;
    d->pout = d->mre * F;
    d->off_w0 = 0;
    d->off_b0 = d->off_w0 + F * K;
    d->off_w1 = d->off_b0 + F;
    d->off_b1 = d->off_w1 + d->pout * H;
    d->off_w2 = d->off_b1 + H;
    d->off_b2 = d->off_w2 + H * classes;
    d->nparams = d->off_b2 + classes;
    return 0;
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline int nchunks(int pout) { return (pout + PCHUNK - 1) / PCHUNK; }

// Carves typed sub-buffers out of a caller-provided workspace.
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <typename T>
    T *take(size_t n)
    {
        T *r = reinterpret_cast<T *>(base + off);
        off = align_up(off + n * sizeof(T));
        return r;
    }
};

// Kernels whose dynamic LDS can exceed the 64 KB default need the limit raised once
// per (kernel, device).
int raise_smem_limit(const void *kernel, size_t bytes);
template <typename Kern>
inline int allow_smem(Kern k, size_t bytes)
{
    return bytes <= 64 * 1024 ? 0 : raise_smem_limit(reinterpret_cast<const void *>(k), bytes);
}

// Workgroups of `kernel` (threads, dynamic LDS bytes) the current device holds at once.
int resident_blocks(const void *kernel, int threads, size_t smem);

#define CCNN_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return (int)e_;       \
    } while (0)

#ifdef __HIPCC__
// tanh in the reference's spelling (gcn_mbgd_fp64.cu:233): 2/(1+1/exp(2v)) - 1.
// fp32: same expression on the hardware exp2 / reciprocal units (v_exp_f32, v_rcp_f32, ~1 ulp
// each) instead of the ~50-instruction correctly-rounded expf and two IEEE divisions; the
// activation epilogue otherwise costs more than half of the conv MFMA time.  Saturates like the
// reference: exp -> inf gives 1, exp -> 0 gives -1.
__device__ __forceinline__ float act(float v)
{
    const float e = __builtin_amdgcn_exp2f(v * 2.885390081777927f);   // exp(2v) = 2^(2v log2 e)
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_rcpf(e)) - 1.0f;
}
__device__ __forceinline__ double act(double v) { return 2.0 / (1.0 + (1.0 / exp(2.0 * v))) - 1.0; }
__device__ __forceinline__ float rexp(float v) { return expf(v); }
__device__ __forceinline__ double rexp(double v) { return exp(v); }
#endif

}  // namespace ccnn
