// Cube gather: out[i][b][j] = scene[idx9[i][j] + b*x*y]
// (preprocessing, gcn_mbgd_fp64.cu:599-615; processing, preprocess048.cu:50-65).
//
// The reference gives one thread a whole sample, so neither its reads nor its
// writes coalesce.  Here a workgroup moves 8 samples at a time through LDS, 64
// bands per pass:
//   read   lanes walk the 8 samples first, so when the samples are neighbouring
//          pixels (labelled pixels come in patches) each neighbour slot j of a band
//          is one contiguous run of the scene; all of a thread's loads of
//          a pass (18) are unconditional and in flight together;
//   write  each sample's 64 x 9 slab is one contiguous run of the output.
// The training list is shuffled, so with a workspace the samples are first ordered
// by centre pixel (radix sort of n keys) purely to decide which 8 samples share a
// workgroup; each cube still lands at its own row i of `out`.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "ccnn_common.h"
#include "profile.h"

#ifndef CCNN_GATHER_PROBE
#define CCNN_GATHER_PROBE 0   // diagnostics: 1 = skip the stores, 2 = skip the scene loads
#endif

namespace ccnn {

#ifndef CCNN_GATHER_GSH
#define CCNN_GATHER_GSH 3   // 8 samples x 64 bands per item: measured 6-12 % faster than 16 x 32 on all three scene shapes (longer contiguous output runs)
#endif
constexpr int GSH = CCNN_GATHER_GSH;
constexpr int GS = 1 << GSH;   // samples per workgroup
constexpr int GBANDS = 512 / GS;   // bands per LDS pass
constexpr int GU = GS * GBANDS * NB / 256;   // loads in flight per thread (18)
#ifndef CCNN_GATHER_GRID
#define CCNN_GATHER_GRID 1024
#endif
constexpr int GATHER_GRID = CCNN_GATHER_GRID;   // persistent workgroups (4 per CU)

template <typename T>
__global__ __launch_bounds__(256) void gather_tile(const T *__restrict__ scene,
                                                   const int *__restrict__ idx9,
                                                   const int *__restrict__ order, long n, long plane,
                                                   int z, T *__restrict__ out)
{
    // A workgroup walks groups of GS samples and, per group, passes of GBANDS bands.
    // The walk is software-pipelined: the scene loads of the next (group, pass) item are issued
    // into registers, and the pixel indices of the next group are fetched, before the current
    // item's tile is stored, so a workgroup always has reads and writes in flight together.
    __shared__ int sidx[2][NB * GS];         // [j][s] pixel index, double-buffered by group
    __shared__ long sid[2][GS];              // output row of sample s
    __shared__ T tile[GS][GBANDS * NB + 1];  // odd row stride: lanes walking s hit distinct banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long cube_len = (long)NB * z;
    const long ngroups = (n + GS - 1) / GS;
    const int npass = (z + GBANDS - 1) / GBANDS;
    // Consecutive workgroup ids land on different XCDs, each with its own L2.  Every XCD gets one
    // contiguous eighth of the (centre-ordered) group list and its workgroups sweep it side by
    // side (workgroup w takes groups w, w + gstep, ..), so the scene rows a group shares with the
    // groups above and below it are fetched once, into one L2, while they are still resident.
    const long chunk = (ngroups + 7) / 8;
    const int xcd = blockIdx.x & 7;
    const long gstep = gridDim.x >> 3;   // the grid is a multiple of 8
    long g = xcd * chunk + (blockIdx.x >> 3);
    const long gend = (xcd + 1) * chunk < ngroups ? (xcd + 1) * chunk : ngroups;
    if (g >= gend) return;
    // short lists: gridDim.y workgroups share a group, each taking a contiguous range of passes
    const int p_lo = (int)((long)blockIdx.y * npass / gridDim.y), p_hi = (int)((long)(blockIdx.y + 1) * npass / gridDim.y);
    if (p_lo >= p_hi) return;
    int pass = p_lo;

    constexpr int IL = (NB * GS + 255) / 256;   // index values a thread fetches per group
    auto load_idx = [&](long grp, int *iv, long *id) {   // clamped, unconditional
        const long gc = grp < ngroups ? grp : ngroups - 1;
#pragma unroll
        for (int u = 0; u < IL; u++) {
            const int e = min(u * 256 + tid, NB * GS - 1);
            const int is = e / NB, ij = e - is * NB;
            const long k = gc * GS + is;
            const long kk = k < n ? k : n - 1;
            id[u] = order ? order[kk] : kk;
            iv[u] = idx9[id[u] * NB + ij];
        }
    };
    auto publish_idx = [&](int b, const int *iv, const long *id) {
#pragma unroll
        for (int u = 0; u < IL; u++) {
            const int e = u * 256 + tid;
            if (e < NB * GS) {
                const int is = e / NB, ij = e - is * NB;
                sidx[b][ij * GS + is] = iv[u];
                if (ij == 0) sid[b][is] = id[u];
            }
        }
    };
    T tmp[GU];
    auto issue = [&](int buf, int pass) {   // scene loads of one item -> registers (clamped, unconditional)
        const int b0 = pass * GBANDS;
        const int nb = (z - b0) < GBANDS ? (z - b0) : GBANDS;
        const int items = nb * NB * GS;
#pragma unroll
        for (int u = 0; u < GU; u++) {
            const int w = min(u * 256 + tid, items - 1);
            const int s = w & (GS - 1), bj = w >> GSH;
            const int bb = bj / NB, jj = bj - bb * NB;
#if CCNN_GATHER_PROBE == 2
            tmp[u] = (T)sidx[buf][jj * GS + s];
#else
            tmp[u] = scene[sidx[buf][jj * GS + s] + (b0 + bb) * plane];
#endif
        }
    };

    int iv[IL];
    long id[IL];
    load_idx(g, iv, id);
    publish_idx(0, iv, id);
    __syncthreads();
    issue(0, pass);
    load_idx(g + gstep, iv, id);   // next group's indices, published when this group's last pass is stored
    int buf = 0;
    while (true) {
        const int b0 = pass * GBANDS;
        const int nb = (z - b0) < GBANDS ? (z - b0) : GBANDS;
        const int items = nb * NB * GS, row_len = nb * NB;
        const int ns = (int)((n - g * GS) < GS ? (n - g * GS) : GS);
#pragma unroll
        for (int u = 0; u < GU; u++) {
            const int w = u * 256 + tid;
            if (w < items) tile[w & (GS - 1)][w >> GSH] = tmp[u];
        }
        const bool new_group = pass + 1 == p_hi;   // the next item starts this workgroup's next group
        const bool more = !new_group || g + gstep < gend;
        if (new_group) publish_idx(buf ^ 1, iv, id);
        __syncthreads();
        // next item's loads (when there is none, the current item's addresses again: harmless)
        const int nbuf = more && new_group ? buf ^ 1 : buf;
        const int npas = more ? (new_group ? p_lo : pass + 1) : pass;
        issue(nbuf, npas);
        if (new_group) load_idx(g + 2 * gstep, iv, id);
        // store the tile: wave w owns samples w, w+4, ..; lanes walk the sample's contiguous run
        for (int s = wave; s < ns; s += 4) {
            T *dst = out + sid[buf][s] * cube_len + (long)b0 * NB;
            for (int k = lane; k < row_len; k += 64) {
#if CCNN_GATHER_PROBE == 1
                if (tile[s][k] == (T)-12345.678) dst[k] = tile[s][k];
#else
                dst[k] = tile[s][k];
#endif
            }
        }
        if (!more) break;
        __syncthreads();   // the tile's readers are done before the next item overwrites it
        if (new_group) g += gstep;
        buf = nbuf;
        pass = npas;
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

// The ordering pass: samples by centre pixel, into the workspace's `order`.
int gather_order(const int *idx9, long n, int x, int y, void *ws, size_t ws_bytes, hipStream_t st)
{
    if (ws_bytes < GatherWs::bytes(n)) return CCNN_E_WORKSPACE;
    GatherWs w(ws, n);
    {
        launch("gather_keys", gather_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx9, n, w.keys, w.vals);
    }
    CCNN_LAUNCH_CHECK();
    unsigned bits = 1;
    while (bits < 32 && ((long)1 << bits) < (long)x * y) bits++;
    hipError_t e = rocprim::radix_sort_pairs(w.sort_tmp, w.sort_bytes, (const int *)w.keys, w.keys_sorted,
                                             (const int *)w.vals, w.order, (size_t)n, 0u, bits, st);
    return (int)e;
}

// planned: the workspace already holds this list's order (ccnn_gather_plan)
template <typename T>
int gather_impl(const T *scene, const int *idx9, long n, int x, int y, int z, T *out, void *ws,
                size_t ws_bytes, hipStream_t st, bool planned = false)
{
    if (n < 0 || x <= 0 || y <= 0 || z <= 0) return CCNN_E_INVALID;
    if (n == 0) return 0;                                  // empty sample list: nothing to read
    if (!scene || !idx9 || !out) return CCNN_E_INVALID;
    if ((long)x * y > 0x7fffffffL || n > 0x7fffffffL) return CCNN_E_RANGE;   // idx9 and order are int32
    if (planned && n > GS && !ws) return CCNN_E_INVALID;
    const int *order = nullptr;
    if (ws && n > GS) {
        if (ws_bytes < GatherWs::bytes(n)) return CCNN_E_WORKSPACE;
        if (!planned)
            if (int e = gather_order(idx9, n, x, y, ws, ws_bytes, st)) return e;
        order = GatherWs(ws, n).order;
    }
    long blocks = ((n + GS - 1) / GS + 7) / 8 * 8;   // a multiple of 8: one slice of the list per XCD
    if (blocks > GATHER_GRID) blocks = GATHER_GRID;
    const int npass = (z + GBANDS - 1) / GBANDS;
    long split = 3 * GATHER_GRID / (2 * blocks);   // enough workgroups to fill the chip on short lists
    split = split < 1 ? 1 : split > npass ? npass : split;
    {
        launch("gather_tile", gather_tile<T>, dim3((unsigned)blocks, (unsigned)split), dim3(256), 0, st, scene, idx9,
               order, n, (long)x * y, z, out);
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

int ccnn_gather_plan(const int *idx9, long n, int x, int y, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || x <= 0 || y <= 0) return CCNN_E_INVALID;
    if ((long)x * y > 0x7fffffffL || n > 0x7fffffffL) return CCNN_E_RANGE;
    if (n <= ccnn::GS) return 0;   // short lists are gathered without an order
    if (!idx9 || !workspace) return CCNN_E_INVALID;
    return ccnn::gather_order(idx9, n, x, y, workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_gather_planned_f32(const float *scene, const int *idx9, long n, int x, int y, int z, float *out,
                            void *workspace, size_t workspace_bytes, void *stream)
{
    return ccnn::gather_impl<float>(scene, idx9, n, x, y, z, out, workspace, workspace_bytes, (hipStream_t)stream, true);
}

int ccnn_gather_planned_f64(const double *scene, const int *idx9, long n, int x, int y, int z, double *out,
                            void *workspace, size_t workspace_bytes, void *stream)
{
    return ccnn::gather_impl<double>(scene, idx9, n, x, y, z, out, workspace, workspace_bytes, (hipStream_t)stream, true);
}

int ccnn_get_dims(int z, ccnn_dims_t *dims) { return ccnn::make_dims(z, dims); }

const char *ccnn_error_string(int code)
{
    switch (code) {
        case 0: return "success";
        case CCNN_E_INVALID: return "ccnn: invalid argument (null pointer, bad size, or band count out of range)";
        case CCNN_E_WORKSPACE: return "ccnn: workspace too small";
        case CCNN_E_RANGE: return "ccnn: size exceeds 32-bit index range";
        case CCNN_E_SCRATCH: return "ccnn: batch does not fit the library's fixed scratch array (CCNN_NOALLOC_SCRATCH_BYTES)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ccnn: unknown error";
    }
}

}  // extern "C"
