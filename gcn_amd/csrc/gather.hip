// Cube gather: out[i][b][j] = scene[idx9[i][j] + b*x*y]
// (preprocessing, gcn_mbgd_fp64.cu:599-615; processing, preprocess048.cu:50-65).
//
// The reference gives one thread a whole sample, so neither its reads nor its
// writes coalesce.  Here a workgroup moves 16 samples at a time through LDS, 32
// bands per pass:
//   read   lanes walk the 16 samples first, so when the samples are neighbouring
//          pixels (labelled pixels come in patches) each neighbour slot j of a band
//          is one contiguous 16-element run of the scene; all of a thread's loads of
//          a pass (18) are unconditional and in flight together;
//   write  each sample's 32 x 9 slab is one contiguous run of the output.
// The training list is shuffled, so with a workspace the samples are first ordered
// by centre pixel (radix sort of n keys) purely to decide which 16 samples share a
// workgroup; each cube still lands at its own row i of `out`.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "ccnn_common.h"
#include "profile.h"

namespace ccnn {

constexpr int GS = 16;   // samples per workgroup
constexpr int GBANDS = 32;   // bands per LDS pass
constexpr int GU = GS * GBANDS * NB / 256;   // loads in flight per thread (18)

template <typename T>
__global__ __launch_bounds__(256) void gather_tile(const T *__restrict__ scene,
                                                   const int *__restrict__ idx9,
                                                   const int *__restrict__ order, long n, long plane,
                                                   int z, T *__restrict__ out)
{
    // grid: x = group of 16 samples (grid-stride), y = pass of 32 bands
    __shared__ int sidx[NB * GS];            // [j][s] pixel index
    __shared__ long sid[GS];                 // output row of sample s
    __shared__ T tile[GS][GBANDS * NB + 1];  // odd row stride: lanes walking s hit distinct banks
    const int tid = threadIdx.x;
    const long cube_len = (long)NB * z;
    const long ngroups = (n + GS - 1) / GS;
    const int b0 = blockIdx.y * GBANDS;
    const int nb = (z - b0) < GBANDS ? (z - b0) : GBANDS;
    const int items = nb * NB * GS;
    const int row_len = nb * NB;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long s0 = g * GS;
        const int ns = (int)((n - s0) < GS ? (n - s0) : GS);
        __syncthreads();   // the previous group's readers of sid/sidx/tile are done
        if (tid < NB * GS) {
            const int s = tid / NB, j = tid - s * NB;
            const long k = s0 + (s < ns ? s : ns - 1);
            const long id = order ? order[k] : k;
            sidx[j * GS + s] = idx9[id * NB + j];
            if (j == 0) sid[s] = id;
        }
        __syncthreads();
        T tmp[GU];
#pragma unroll
        for (int u = 0; u < GU; u++) {
            // clamped and unconditional: a guarded load costs a full memory round trip each
            const int w = min(u * 256 + tid, items - 1);
            const int s = w & (GS - 1), bj = w >> 4;
            const int bb = bj / NB, jj = bj - bb * NB;
            tmp[u] = scene[sidx[jj * GS + s] + (b0 + bb) * plane];
        }
#pragma unroll
        for (int u = 0; u < GU; u++) {
            const int w = u * 256 + tid;
            if (w < items) tile[w & (GS - 1)][w >> 4] = tmp[u];
        }
        __syncthreads();
        for (int e = tid; e < GS * row_len; e += 256) {
            const int s = e / row_len, k = e - s * row_len;
            if (s < ns) out[sid[s] * cube_len + (long)b0 * NB + k] = tile[s][k];
        }
    }
}

__global__ void gather_keys(const int *__restrict__ idx9, long n, int *__restrict__ keys,
                            int *__restrict__ vals)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) {
        keys[i] = idx9[i * NB + NB / 2];   // centre pixel
        vals[i] = (int)i;
    }
}

struct GatherWs {
    int *keys, *vals, *keys_sorted, *order;
    void *sort_tmp;
    size_t sort_bytes;
    static size_t sort_tmp_bytes(long n)
    {
        size_t bytes = 0;
        (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int *)nullptr, (int *)nullptr, (const int *)nullptr,
                                        (int *)nullptr, (size_t)n, 0u, 32u, (hipStream_t) nullptr);
        return bytes;
    }
    static size_t bytes(long n)
    {
        return 4 * align_up((size_t)n * sizeof(int)) + align_up(sort_tmp_bytes(n));
    }
    GatherWs(void *ws, long n)
    {
        Carver c(ws);
        keys = c.take<int>(n);
        vals = c.take<int>(n);
        keys_sorted = c.take<int>(n);
        order = c.take<int>(n);
        sort_bytes = sort_tmp_bytes(n);
        sort_tmp = c.take<char>(sort_bytes);
    }
};

template <typename T>
int gather_impl(const T *scene, const int *idx9, long n, int x, int y, int z, T *out, void *ws,
                size_t ws_bytes, hipStream_t st)
{
    if (n < 0 || x <= 0 || y <= 0 || z <= 0) return CCNN_E_INVALID;
    if (n == 0) return 0;                                  // empty sample list: nothing to read
    if (!scene || !idx9 || !out) return CCNN_E_INVALID;
    if ((long)x * y > 0x7fffffffL || n > 0x7fffffffL) return CCNN_E_RANGE;   // idx9 and order are int32
    const int *order = nullptr;
    if (ws && n > GS) {
        if (ws_bytes < GatherWs::bytes(n)) return CCNN_E_WORKSPACE;
        GatherWs w(ws, n);
        {
            launch("gather_keys", gather_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx9, n, w.keys,
                   w.vals);
        }
        CCNN_LAUNCH_CHECK();
        unsigned bits = 1;
        while (bits < 32 && ((long)1 << bits) < (long)x * y) bits++;
        hipError_t e = rocprim::radix_sort_pairs(w.sort_tmp, w.sort_bytes, (const int *)w.keys, w.keys_sorted,
                                                 (const int *)w.vals, w.order, (size_t)n, 0u, bits, st);
        if (e != hipSuccess) return (int)e;
        order = w.order;
    }
    long blocks = (n + GS - 1) / GS;
    if (blocks > 8192) blocks = 8192;
    {
        launch("gather_tile", gather_tile<T>, dim3((unsigned)blocks, (unsigned)((z + GBANDS - 1) / GBANDS)), dim3(256), 0,
               st, scene, idx9, order, n, (long)x * y, z, out);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace ccnn

extern "C" {

size_t ccnn_gather_workspace_bytes(long n) { return n > ccnn::GS ? ccnn::GatherWs::bytes(n) : 0; }

int ccnn_gather_f32(const float *scene, const int *idx9, long n, int x, int y, int z, float *out,
                    void *workspace, size_t workspace_bytes, void *stream)
{
    return ccnn::gather_impl<float>(scene, idx9, n, x, y, z, out, workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_gather_f64(const double *scene, const int *idx9, long n, int x, int y, int z, double *out,
                    void *workspace, size_t workspace_bytes, void *stream)
{
    return ccnn::gather_impl<double>(scene, idx9, n, x, y, z, out, workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_get_dims(int z, ccnn_dims_t *dims) { return ccnn::make_dims(z, dims); }

const char *ccnn_error_string(int code)
{
    switch (code) {
        case 0: return "success";
        case CCNN_E_INVALID: return "ccnn: invalid argument (null pointer, bad size, or band count out of range)";
        case CCNN_E_WORKSPACE: return "ccnn: workspace too small";
        case CCNN_E_RANGE: return "ccnn: size exceeds 32-bit index range";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ccnn: unknown error";
    }
}

}  // extern "C"
