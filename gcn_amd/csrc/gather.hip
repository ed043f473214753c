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
#include <cstdlib>
#include <cstring>
#include <new>

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
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    static const long grid_env = getenv("CCNN_GATHER_GRID") ? atol(getenv("CCNN_GATHER_GRID")) : 0;     // diagnostics
    static const long split_env = getenv("CCNN_GATHER_SPLIT") ? atol(getenv("CCNN_GATHER_SPLIT")) : 0;
    const long grid_cap = grid_env > 0 ? grid_env : GATHER_GRID;
    if (blocks > grid_cap) blocks = grid_cap;
    const int npass = (z + GBANDS - 1) / GBANDS;
    long split = 3 * grid_cap / (2 * blocks);   // enough workgroups to fill the chip on short lists
    if (split_env > 0) split = split_env;
    split = split < 1 ? 1 : split > npass ? npass : split;
    {
        launch("gather_tile", gather_tile<T>, dim3((unsigned)blocks, (unsigned)split), dim3(256), 0, st, scene, idx9,
               order, n, (long)x * y, z, out);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Dense path: lists whose samples are plain 3x3 neighbourhoods (mirrored at the scene's edge, as
// ccnn_host_build_index makes them) and cover a good share of the tiles they touch - fully or
// densely labelled scenes.  No sort: one scatter builds a pixel -> sample-row table (head[] /
// next[] chains, so a pixel that occurs twice in the list is served twice) and the list of
// non-empty 32 x 8 pixel tiles; the gather kernel then stages each tile plus a one-pixel halo,
// DB bands at a time, ONCE in LDS (coalesced row reads; every scene value is fetched once per
// tile instead of nine times) and writes every sample's [DB][9] run from there.
constexpr int DTX = 32, DTY = 8;                 // tile: 32 pixels along x (contiguous in memory) x 8 along y
constexpr int DRX = DTX + 2, DRY = DTY + 2;      // with halo
constexpr int DREG = DRX * DRY;                  // region cells per band
constexpr long DENSE_MIN_SAMPLES = 65536;
template <typename T>
struct DenseCfg {
    static constexpr int VW = 16 / (int)sizeof(T);    // values per 16-byte store
    static constexpr int DB = 7 * VW;                 // bands per pass: [DB][9] is 63 lanes x 16 bytes; 38 KB of LDS
};
// 16 bytes of a sample's run: rows start on 4- or 8-byte boundaries only, the store is one
// unaligned dwordx4 all the same
template <typename T>
union Run16 {
    T v[16 / sizeof(T)];
    __attribute__((ext_vector_type(4))) int bits;
};
// (the compiler splits a store it believes unaligned into dwordx3 + dword; the hardware takes
// the dwordx4 at any 4-byte boundary)
__device__ __forceinline__ void store16_unaligned(void *dst, __attribute__((ext_vector_type(4))) int bits)
{
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(bits) : "memory");
}

struct DenseWs {
    int *head, *next, *tile_count, *tiles, *counters;   // counters: [0] non-empty tiles, [1] samples that are not plain neighbourhoods
    static size_t bytes(long n, int x, int y)
    {
        const long nt = (long)((x + DTX - 1) / DTX) * ((y + DTY - 1) / DTY);
        return align_up((size_t)x * y * sizeof(int)) + align_up((size_t)n * sizeof(int)) + 2 * align_up((size_t)nt * sizeof(int)) +
               align_up(16 * sizeof(int));
    }
    DenseWs(void *ws, long n, int x, int y)
    {
        const long nt = (long)((x + DTX - 1) / DTX) * ((y + DTY - 1) / DTY);
        Carver c(ws);
        head = c.take<int>((size_t)x * y);
        next = c.take<int>(n);
        tile_count = c.take<int>(nt);
        tiles = c.take<int>(nt);
        counters = c.take<int>(16);
    }
};

__device__ __forceinline__ int mirror(int v, int d, int lim) { return (v + d < 0 || v + d >= lim) ? v - d : v + d; }

__global__ void dense_build(const int *__restrict__ idx9, long n, int x, int y, DenseWs w)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int row[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) row[j] = idx9[i * NB + j];
    const int p = row[NB / 2];
    bool plain = p >= 0 && p < x * y;
    const int r = plain ? p % x : 0, c = plain ? p / x : 0;
#pragma unroll
    for (int j = 0; j < NB; j++) plain = plain && row[j] == mirror(r, j % 3 - 1, x) + mirror(c, j / 3 - 1, y) * x;
    if (!plain) {
        atomicAdd(&w.counters[1], 1);
        return;
    }
    w.next[i] = atomicExch(&w.head[p], (int)i);
    const int t = (c / DTY) * ((x + DTX - 1) / DTX) + r / DTX;
    if (atomicAdd(&w.tile_count[t], 1) == 0) w.tiles[atomicAdd(&w.counters[0], 1)] = t;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_dense(const T *__restrict__ scene, int x, int y, int z, long plane,
                                                    DenseWs w, T *__restrict__ out)
{
    constexpr int DB = DenseCfg<T>::DB;
    __shared__ T reg[DB * DREG];
    __shared__ int shead[DTX * DTY], snext[DTX * DTY];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntx = (x + DTX - 1) / DTX;
    const int t = w.tiles[blockIdx.x];
    const int tx0 = (t % ntx) * DTX, ty0 = (t / ntx) * DTY;
    const long cube_len = (long)NB * z;
    // this thread's pixel of the tile: head of its chain and the second link (both -1 for most pixels)
    {
        const int px = tid % DTX, py = tid / DTX;
        const int r = tx0 + px, c = ty0 + py;
        int h = -1, nx = -1;
        if (r < x && c < y) h = w.head[c * x + r];
        if (h >= 0) nx = w.next[h];
        shead[tid] = h;
        snext[tid] = nx;
    }
    const int b0 = blockIdx.y * DB;
    const int nb = min(DB, z - b0);
    // the tile with its halo, nb bands: rows of 34 contiguous values (clamped at the scene's edge:
    // cells outside the scene are never read, neighbours are mirrored about the centre pixel)
    const int cells = nb * DREG;
    for (int e0 = 0; e0 < cells; e0 += 16 * 256) {
        T v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int e = min(e0 + u * 256 + tid, cells - 1);
            const int b = e / DREG, rem = e - b * DREG;
            const int ry = rem / DRX, rx = rem - ry * DRX;
            const int gx = min(max(tx0 - 1 + rx, 0), x - 1), gy = min(max(ty0 - 1 + ry, 0), y - 1);
            v[u] = scene[(long)(b0 + b) * plane + (long)gy * x + gx];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int e = e0 + u * 256 + tid;
            if (e < cells) reg[e] = v[u];
        }
    }
    __syncthreads();
    // Each wave serves whole pixels; lane l < 63 holds the run's values VW l .. VW l + VW - 1, each
    // with a fixed (band, neighbour) whatever the pixel is, and writes them with one 16-byte store.
    constexpr int VW = DenseCfg<T>::VW;
    int kb[VW], kdr[VW], kdc[VW];
#pragma unroll
    for (int i = 0; i < VW; i++) {
        const int k = min(VW * lane + i, DB * NB - 1), j = k % NB;
        kb[i] = (k / NB) * DREG;
        kdr[i] = j % 3 - 1;
        kdc[i] = j / 3 - 1;
    }
    const int run = nb * NB;
    const int have = min(max(run - VW * lane, 0), VW);      // this lane's values inside the run
    for (int pix = wave; pix < DTX * DTY; pix += 4) {
        int rowi = shead[pix];
        if (rowi < 0) continue;                    // wave-uniform: no sample at this pixel
        const int r = tx0 + pix % DTX, c = ty0 + pix / DTX;
        Run16<T> val;
#pragma unroll
        for (int i = 0; i < VW; i++) {
            const int rr = mirror(r, kdr[i], x), cc = mirror(c, kdc[i], y);
            const int at = kb[i] + (cc - ty0 + 1) * DRX + (rr - tx0 + 1);
            val.v[i] = reg[i < have ? at : 0];
        }
        int nxt = snext[pix];
        while (rowi >= 0) {                        // the pixel's chain: once, unless the list repeats the pixel
            T *dst = out + (long)rowi * cube_len + (long)b0 * NB + VW * lane;
            if (have == VW) {
                store16_unaligned(dst, val.bits);
            } else {
#pragma unroll
                for (int i = 0; i < VW; i++)
                    if (i < have) dst[i] = val.v[i];
            }
            rowi = nxt;
            if (rowi >= 0) nxt = w.next[rowi];
        }
    }
}

}  // namespace ccnn

// One sample list prepared for repeated gathers (ccnn_gather_plan_create).
struct ccnn_gather_plan {
    const int *idx9;
    long n;
    int x, y;
    int kind;        // 0: samples visited in centre-pixel order (sorted list); 1: dense tiles
    int ntiles;      // dense: non-empty tiles
    void *ws;
    size_t ws_bytes;
};

namespace {
template <typename T>
int gather_planned(const ccnn_gather_plan *pl, const T *scene, int z, T *out, hipStream_t st)
{
    using namespace ccnn;
    if (!pl || z <= 0) return CCNN_E_INVALID;
    if (pl->n == 0) return 0;
    if (!scene || !out) return CCNN_E_INVALID;
    if (pl->kind == 0)
        return gather_impl<T>(scene, pl->idx9, pl->n, pl->x, pl->y, z, out, pl->ws, GatherWs::bytes(pl->n), st, true);
    DenseWs w(static_cast<char *>(pl->ws) + GatherWs::bytes(pl->n), pl->n, pl->x, pl->y);
    const int nchunks = (z + DenseCfg<T>::DB - 1) / DenseCfg<T>::DB;
    launch("gather_dense", gather_dense<T>, dim3((unsigned)pl->ntiles, (unsigned)nchunks), dim3(256), 0, st, scene, pl->x,
           pl->y, z, (long)pl->x * pl->y, w, out);
    CCNN_LAUNCH_CHECK();
    return 0;
}
}  // namespace

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

size_t ccnn_gather_plan_workspace_bytes(long n, int x, int y)
{
    if (n <= 0 || x <= 0 || y <= 0) return 0;
    return ccnn::GatherWs::bytes(n) + ccnn::DenseWs::bytes(n, x, y);
}

int ccnn_gather_plan_create(const int *idx9, long n, int x, int y, void *workspace, size_t workspace_bytes, void *stream,
                            ccnn_gather_plan_t **plan)
{
    using namespace ccnn;
    if (!plan || n < 0 || x <= 0 || y <= 0) return CCNN_E_INVALID;
    if ((long)x * y > 0x7fffffffL || n > 0x7fffffffL) return CCNN_E_RANGE;
    if (n > 0 && (!idx9 || !workspace)) return CCNN_E_INVALID;
    if (n > 0 && workspace_bytes < ccnn_gather_plan_workspace_bytes(n, x, y)) return CCNN_E_WORKSPACE;
    ccnn_gather_plan *pl = new (std::nothrow) ccnn_gather_plan{idx9, n, x, y, 0, 0, workspace, workspace_bytes};
    if (!pl) return (int)hipErrorOutOfMemory;
    hipStream_t st = (hipStream_t)stream;
    if (n > GS) {
        // the dense candidate first: table, tile list and the count of samples that are not plain
        // neighbourhoods (one synchronisation: the host decides which kernel later calls launch)
        char *dense_ws = static_cast<char *>(workspace) + GatherWs::bytes(n);
        DenseWs w(dense_ws, n, x, y);
        const long nt = (long)((x + DTX - 1) / DTX) * ((y + DTY - 1) / DTY);
        hipError_t e = hipMemsetAsync(w.head, 0xff, (size_t)x * y * sizeof(int), st);
        if (e == hipSuccess) e = hipMemsetAsync(w.tile_count, 0, (size_t)nt * sizeof(int), st);
        if (e == hipSuccess) e = hipMemsetAsync(w.counters, 0, 16 * sizeof(int), st);
        if (e != hipSuccess) {
            delete pl;
            return (int)e;
        }
        launch("dense_build", dense_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx9, n, x, y, w);
        int counters[2] = {0, 0};
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(counters, w.counters, sizeof(counters), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            delete pl;
            return (int)e;
        }
        // Tiles when every sample is a plain neighbourhood, the tiles they touch are at least three
        // quarters full and the list is long enough to fill the chip several times over: measured,
        // a fully labelled 610 x 340 x 103 scene gathers in 317 us by tiles against 384 us by the
        // ordered list, while on shorter or sparser lists (a 145 x 145 scene fully labelled, the
        // training lists of the three published scenes) the ordered list is as fast or faster -
        // both kernels are then bound by the scattered 1 KB writes, not by the reads the tiles save.
        const char *force = getenv("CCNN_GATHER_PLAN");   // tests: "dense" / "sorted" override the occupancy rule
        const bool plain = counters[1] == 0 && counters[0] > 0;
        bool dense = plain && n >= DENSE_MIN_SAMPLES && n * 4 >= 3L * counters[0] * DTX * DTY;
        if (force && !strcmp(force, "dense")) dense = plain;
        if (force && !strcmp(force, "sorted")) dense = false;
        if (dense) {
            pl->kind = 1;
            pl->ntiles = counters[0];
        } else if (int rc = gather_order(idx9, n, x, y, workspace, GatherWs::bytes(n), st)) {
            delete pl;
            return rc;
        }
    }
    *plan = pl;
    return 0;
}

int ccnn_gather_plan_kind(const ccnn_gather_plan_t *plan)
{
    return plan ? plan->kind : CCNN_E_INVALID;
}

int ccnn_gather_plan_destroy(ccnn_gather_plan_t *plan)
{
    delete plan;
    return 0;
}

int ccnn_gather_planned_f32(const ccnn_gather_plan_t *plan, const float *scene, int z, float *out, void *stream)
{
    return gather_planned<float>(plan, scene, z, out, (hipStream_t)stream);
}

int ccnn_gather_planned_f64(const ccnn_gather_plan_t *plan, const double *scene, int z, double *out, void *stream)
{
    return gather_planned<double>(plan, scene, z, out, (hipStream_t)stream);
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
