// Forward (inference) path of the Cube CNN, batched over pixels, plus the
// stand-alone parameter updates and the accuracy counter.
//
// Kernel map (reference kernel -> here)
//   convolution + maxpooling   -> conv_pool_fwd  (implicit GEMM on MFMA: positions x taps x filters)
//   fully_connect,
//   output_and_dvalue(false),
//   count_err                  -> fc_head_fwd (FC1 on MFMA with W1 streamed through LDS, then the
//                                 output layer, softmax and device argmax) + count_correct_kernel
//   update_params_row/col      -> update_*_kernel (after an all-reduce; single-GPU training
//                                 fuses the update into grad_reduce, see mbgd.hip)
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "ccnn_device.h"
#include "profile.h"
#include "train_fwd.h"

namespace ccnn {

int raise_smem_limit(const void *fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> raised;   // (kernel, device) -> limit already set
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    std::lock_guard<std::mutex> lk(mu);
    size_t &have = raised[{fn, dev}];
    if (have >= bytes) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return (int)e;
}

// ---------------------------------------------------------------------------
// convolution + tanh + max-pool.  Every WAVE owns whole samples: it keeps its current sample's
// cube and pooled features in a private LDS region, runs that sample's MFMA chains
// (conv_pool_wave), and holds the next sample's cube in registers meanwhile.  The waves of a
// workgroup share only the staged filters, so after the prologue there is no workgroup barrier:
// the waves drift apart and one wave's loads, activations and stores run under the others'
// matrix instructions.  What made the kernel faster than its predecessor (4 samples per
// 256-thread workgroup, two barriers per group, 297 us per 65536 Pavia samples) was, in order of
// size: fewer MFMA instructions (packed tile, ccnn_device.h), a grid of exactly the resident
// workgroups, rows handed out on demand, and 16-byte pooled stores; the barrier-free structure
// by itself measured the same as the barriers at equal instruction count.
// pool[row][f*mre + m] = pooled value.
//
// NREG >= ceil(cube_len/64) is a template parameter so that the prefetch is straight-line code:
// loads under a run-time guard are waited for at the guard's join, which would put the whole
// memory latency back in front of the MFMAs.
//
// SCENE: the samples are not pre-gathered cubes but pixels of the scene: `cubes` is the scene
// [z][plane], idx9 the neighbour lists, and element b*9+j of a sample's cube is read straight
// from scene[idx9[sample][j] + b*plane] (preprocessing, gcn_mbgd_fp64.cu:599-615, folded into
// the load), so full-scene inference never materialises 9z values per pixel in HBM.  A wave's
// neighbour lists sit in LDS one sample ahead of the loads that use them.
// Diagnostic builds only (-DCCNN_CONV_CLOCK): workgroup 7, wave 0 stamps the shader clock and the
// 100 MHz wall clock around its sample loop; ccnn_debug_conv_clock() returns the four stamps.
#ifdef CCNN_CONV_CLOCK
__device__ unsigned long long g_conv_clock[4];
__device__ unsigned long long g_conv_wg[3 * 2048];   // per workgroup: entry, loop start, loop end (100 MHz)
#define CONV_STAMP(k)                                                                 \
    do {                                                                              \
        if (blockIdx.x == 7 && threadIdx.x == 0) {                                    \
            g_conv_clock[2 * (k)] = __builtin_amdgcn_s_memtime();                     \
            g_conv_clock[2 * (k) + 1] = __builtin_amdgcn_s_memrealtime();             \
        }                                                                             \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048)                             \
            atomicMax(&g_conv_wg[3 * blockIdx.x + 1 + (k)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); \
    } while (0)
#else
#define CONV_STAMP(k) do { } while (0)
#endif

template <typename T>
__host__ __device__ inline int conv_px_words(int pout) { return (pout + (int)sizeof(T) - 1) / (int)sizeof(T); }

constexpr int CONV_MAXREG = 64;    // cube elements a lane may hold in flight for the next sample
constexpr int CONV_MAXWAVES = 8;   // waves per workgroup (fewer when the per-wave LDS region is large)
constexpr int CONV_CTR_STRIDE = 32;   // ints between two work counters (one 128-byte line each)

// TRAIN: also stores, per pooled feature, which member of its pair won (0: position 2m, 1: 2m+1)
// as one byte in px[row][f*mre + m] - what bp_maxpooling needs to route the delta back
// (gcn_mbgd_fp64.cu:439-454).
template <typename T, bool FLIP, int NREG, bool SCENE, bool TRAIN>
__global__ __launch_bounds__(64 * CONV_MAXWAVES) void conv_pool_fwd(const T *__restrict__ params,
                                                                     const T *__restrict__ cubes, long first,
                                                                     long nrows, int cube_len, int re, int mre,
                                                                     int off_b0, T *__restrict__ pool,
                                                                     const int *__restrict__ idx9, long plane,
                                                                     int *__restrict__ ctr,
                                                                     unsigned char *__restrict__ px)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int sidx_all[CONV_MAXWAVES][2][16];   // SCENE: lists of the sample being loaded / the one after
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CCNN_CONV_CLOCK
    if (tid == 0 && blockIdx.x < 2048) g_conv_wg[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    const int nwv = blockDim.x >> 6;
    const int col = lane & 15;
    const int pout = F * mre;
    const int cube_pad = conv_cube_pad(re);
    T *wl = reinterpret_cast<T *>(smem);
    const int pxw = TRAIN ? conv_px_words<T>(pout) : 0;   // elements of T that hold the wave's pout winner bytes
    T *cube = wl + WP_LEN + wave * (cube_pad + pout + pxw);   // this wave's sample
    T *pl = cube + cube_pad;                            // and its pooled features
    unsigned char *pxl = reinterpret_cast<unsigned char *>(pl + pout);
    int(*sidx)[16] = sidx_all[wave];

    // Work distribution.  Rows [0, 3W) (W = waves of the grid) are fixed: wave g owns g, W + g and
    // 2W + g, which fills its pipeline (current cube in LDS, next in registers, the one after
    // known) without waiting for anybody.  The rest is handed out one row at a time, because waves
    // do not run equally fast (measured: loop times of 185..234 us for equal shares): eight
    // counters, one per eighth of the remaining rows, workgroup b using counter b % 8 (the
    // workgroups of one XCD share one counter), two grabs ahead of the row being computed.
    const long W = (long)gridDim.x * nwv;
    const long gw = (long)blockIdx.x * nwv + wave;
    const bool active = gw < nrows;   // (every wave helps to stage the filters first)
    const int xcd = blockIdx.x & 7;
    const long rem = nrows > 3 * W ? nrows - 3 * W : 0;
    const long per = (rem + 7) / 8;
    const long xlo = 3 * W + xcd * per;
    const long xhi = xlo + per < nrows ? xlo + per : nrows;
    int *my_ctr = ctr + CONV_CTR_STRIDE * xcd;
    long row = gw;
    long r_nxt = W + gw < nrows ? W + gw : -1;
    long r_nn = 2 * W + gw < nrows ? 2 * W + gw : -1;

    // The next cube waits in registers while this one's chains run, when that many registers
    // are to spare (beyond it the compiler spills them to scratch: then it is loaded after the
    // chains and only the other waves hide its latency).
    constexpr bool PREFETCH = NREG * sizeof(T) <= 48 * 4;
    T creg[PREFETCH ? NREG : 1];
    auto issue = [&](long r, int ib) {   // sample r's cube -> registers (clamped, unconditional)
        if (SCENE) {
#pragma unroll
            for (int u = 0; u < NREG; u++) {
                const int e = min(u * 64 + lane, cube_len - 1);
                const int b = e / NB, j = e - b * NB;
                creg[u] = cubes[sidx[ib][j] + b * plane];
            }
        } else {
            const T *src = cubes + (first + r) * (long)cube_len;
#pragma unroll
            for (int u = 0; u < NREG; u++) creg[u] = src[min(u * 64 + lane, cube_len - 1)];
        }
    };
    auto load_idx = [&](long r) {   // one neighbour index of sample r per lane (clamped, unconditional)
        return idx9[(first + (r >= 0 ? r : 0)) * NB + min(lane, NB - 1)];
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NREG; u++) {
            const int e = u * 64 + lane;
            if (e < cube_len) cube[e] = creg[u];
        }
    };

    auto fetch = [&](long r, int ib) {   // without prefetch: sample r's cube -> LDS, 16 elements per lane at a time
#pragma unroll 1
        for (int u0 = 0; u0 < NREG; u0 += 16) {
            T t[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = min((u0 + u) * 64 + lane, cube_len - 1);
                if (SCENE) {
                    const int b = e / NB, j = e - b * NB;
                    t[u] = cubes[sidx[ib][j] + b * plane];
                } else {
                    t[u] = cubes[(first + r) * (long)cube_len + e];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = (u0 + u) * 64 + lane;
                if (e < cube_len) cube[e] = t[u];
            }
        }
    };

    // Prologue: the first cube's loads are issued before the filters are staged, so the two
    // memory round trips overlap.
    int ib = 0;   // sidx[ib] = list of the sample whose loads are issued next
    if (SCENE && active) {
        const int i0 = load_idx(row), i1 = load_idx(r_nxt);
        if (lane < NB) {
            sidx[0][lane] = i0;
            sidx[1][lane] = i1;
        }
        wave_sync();
    }
    if (PREFETCH && active) issue(row, 0);
    stage_filters_packed<T, FLIP>(params, wl, tid, blockDim.x);
    for (int e = lane; e < cube_pad; e += 64) cube[e] = 0;   // pads stay zero: loads only touch [0, cube_len)
    const T bias0 = params[off_b0 + col];
    const T biasx = params[off_b0 + 16 + (col & 3)];
    __syncthreads();   // the only workgroup barrier: filters staged
    if (!active) return;
    if (PREFETCH) commit();
    else fetch(row, 0);
    ib = 1;
    wave_sync();
    CONV_STAMP(0);
    while (row >= 0) {
        const bool want = rem > 0 && r_nn >= 0;            // wave-uniform: rows are left to hand out
        int got = 0;
        if (want && lane == 0) got = atomicAdd(my_ctr, 1);
        if (PREFETCH && r_nxt >= 0) issue(r_nxt, ib);      // next sample's loads fly during the MFMAs
        int inext = 0;
        if (SCENE) inext = load_idx(r_nn);                 // and the list of the sample after it
        conv_pool_wave<T, TRAIN>(cube, wl, re, bias0, biasx, lane, [&](int f, int m, T v, int bt) {
            pl[f * mre + m] = v;
            if (TRAIN) pxl[f * mre + m] = (unsigned char)(bt - 2 * m);
        });
        wave_sync();                                       // pooled values complete, cube no longer read
        if (SCENE) {
            ib ^= 1;                                       // the list issue(r_nxt) used is dead: reuse its slot
            if (lane < NB) sidx[ib][lane] = inext;
        }
        if (r_nxt >= 0) {
            if (PREFETCH) commit();
            else fetch(r_nxt, ib ^ 1);
        }
        T *dst = pool + row * pout;
        {
            typedef typename Vec16<T>::type V;
            constexpr int VW = Vec16<T>::W;
            const V *src = reinterpret_cast<const V *>(pl);
            V *dv = reinterpret_cast<V *>(dst);
            for (int o = lane; o < pout / VW; o += 64) dv[o] = src[o];   // pout = 20 mre: whole 16-byte vectors
            if (TRAIN) {   // pout is a multiple of 4: whole 32-bit words
                const unsigned *ps = reinterpret_cast<const unsigned *>(pxl);
                unsigned *pd = reinterpret_cast<unsigned *>(px + row * pout);
                for (int o = lane; o < pout / 4; o += 64) pd[o] = ps[o];
            }
        }
        wave_sync();                                       // next cube and list in place, pl free again
        long r_new = -1;
        if (want) {
            const long r = xlo + __builtin_amdgcn_readfirstlane(got);
            r_new = r < xhi ? r : -1;
        }
        row = r_nxt;
        r_nxt = r_nn;
        r_nn = r_new;
    }
    CONV_STAMP(1);
}

template <typename T>
size_t conv_smem(const ccnn_dims_t &d, int nwaves, bool train = false)
{
    return (size_t)(WP_LEN + nwaves * (conv_cube_pad(d.re) + d.pout + (train ? conv_px_words<T>(d.pout) : 0))) * sizeof(T);
}

// ---------------------------------------------------------------------------
// FC1 + tanh + output layer + softmax + count_err argmax, a tile of 32 (short lists: 16) samples
// at a time per persistent workgroup (fully_connect, output_and_dvalue(false), count_err:
// gcn_mbgd_fp64.cu:271-354, 628-652).
//
// FC1 is a [32 x Pout] x [Pout x 100] MFMA product: the K loop walks W1 in chunks of
// 64 pooled features (25.6 KB, contiguous in W1) staged through LDS; the next chunk's
// global loads are issued before the current chunk's MFMAs and stored after them.
// The epilogue runs from LDS: 8 lanes per sample, lane j producing outputs j and j+8.
constexpr int FH_MS = 32;                      // samples per tile (16 when the list is too short to give every CU two tiles)
constexpr int LDZ = H + 1;                     // z1/a1 rows in LDS (odd stride)

template <typename T, int MS>
struct FcHeadLds {
    static constexpr int wl = 0;                                   // [64][LDW1]   W1 chunk; later z1s[32][LDZ]
    static constexpr int pt = wl + PCHUNK * LDW1;                  // [32][LDP]    pool chunk
    static constexpr int w2 = pt + ((MS * LDP + 3) & ~3);       // [100 x classes] W2
    static constexpr int b1 = w2 + H * OMAX;                        // [100]
    static constexpr int b2 = b1 + H;                              // [16]
    static constexpr int es = b2 + 16;                             // [32][16]
    static constexpr int ps = es + MS * 16;                     // [32][16]
    static constexpr int total = ps + MS * 16;
    static constexpr size_t bytes = (size_t)total * sizeof(T);
};

#ifdef CCNN_CONV_CLOCK
__device__ unsigned long long g_fc_wg[4 * 4096];   // per workgroup: entry, K loop start, K loop end, exit (100 MHz)
#define FC_STAMP(k)                                                                                  \
    do {                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_fc_wg[4 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
extern "C" int ccnn_debug_fc_wg(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fc_wg), sizeof(unsigned long long) * 4 * 4096);
}
#else
#define FC_STAMP(k) do { } while (0)
#endif

// TRAIN (the minibatch step's forward half): the epilogue goes on to what output_and_dvalue and the
// first bp_fully_connect produce per sample (gcn_mbgd_fp64.cu:307-354, 401-436) - a1, delta2 =
// (p - y)/classes, delta1 = (class-wide tree of W2 delta2)(1 + a1)(1 - a1) - and to loss_function's
// value (:559-596); `tr` carries the labels and the outputs, all indexed like pool's rows.
template <typename T>
struct FcTrainOut {
    const int *onehot;    // [nrows][classes]
    T *a1, *d1, *d2;      // [nrows][100], [nrows][100], [nrows][classes]
    double *loss_values;  // [nrows]
};

// NO: the class count when it is the reference's 13 (compile-time then), 0 for any other (no_rt).
template <typename T, int MS, bool TRAIN, int NO>
__global__ __launch_bounds__(256) void fc_head_fwd(const T *__restrict__ params, int off_w1, int off_b1,
                                                   int off_w2, int off_b2, int no_rt, const T *__restrict__ pool,
                                                   int nrows, int pout, T *__restrict__ probs,
                                                   int *__restrict__ argmax, int *__restrict__ ctr,
                                                   FcTrainOut<T> tr)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_next;   // the workgroup's next tile, grabbed one tile ahead
    const int no = NO ? NO : no_rt;
    typedef FcHeadLds<T, MS> L;
    typedef typename Vec16<T>::type V;
    constexpr int VW = Vec16<T>::W;
    T *base = reinterpret_cast<T *>(smem);
    T *wl = base + L::wl, *pt = base + L::pt, *w2s = base + L::w2, *b1s = base + L::b1, *b2s = base + L::b2;
    T *es = base + L::es, *ps = base + L::ps;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kq = lane >> 4;
    // Tiles of 32 samples: workgroup b starts with tile b; the tiles beyond the grid are handed
    // out through eight counters (workgroup b uses counter b % 8, one eighth of the remaining
    // tiles each), so the grid can be exactly what the device holds and nobody idles through a
    // partly filled last round.
    const int ntiles = (nrows + MS - 1) / MS;
    const int xcd = blockIdx.x & 7;
    const int rem = ntiles > (int)gridDim.x ? ntiles - (int)gridDim.x : 0;
    const int per = (rem + 7) / 8;
    const int xlo = (int)gridDim.x + xcd * per;
    const int xhi = min(xlo + per, ntiles);
    int tile = blockIdx.x;
    int r0 = tile * MS;
    int nb = min(MS, nrows - r0);
    const T *W1 = params + off_w1;
    const int nch = (pout + PCHUNK - 1) / PCHUNK;
    FC_STAMP(0);

    // 32 samples: wave w owns sample tile w & 1 and neuron tiles {0..3} (w < 2) or {4..6};
    // 16 samples: one sample tile, wave w owns neuron tiles {2w, 2w+1} (tile 7 does not exist)
    constexpr int NJ = MS == 32 ? 4 : 2;
    const int mtile = MS == 32 ? (wave & 1) : 0;
    const int nt0 = MS == 32 ? ((wave >> 1) ? 4 : 0) : 2 * wave;
    const T *ap = pt + (mtile * 16 + col) * LDP + kq;
    const T *bp = wl + kq * LDW1 + col;
    int boff[NJ];          // column offset of this wave's j-th neuron tile (a valid tile when it has none)
    bool jlive[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        jlive[j] = nt0 + j < 7;
        boff[j] = 16 * (jlive[j] ? nt0 + j : 0);
    }
    typename Mma<T>::acc_t acc[NJ];

    constexpr int NWV = (PCHUNK * H / VW + 255) / 256;      // W1 vectors per thread per chunk
    constexpr int NPT = MS * PCHUNK / 256;               // pool elements per thread per chunk
    // Two register sets: while chunk c is multiplied, chunk c + 1 waits in one set and chunk c + 2's
    // loads fly into the other, so a load has two multiply phases to arrive (with one set every
    // chunk cost a full L2 round trip: the K loop was latency-bound).
    struct ChunkRegs {
        V w[NWV];
        T p[NPT];
    } ra, rb;
    auto issue = [&](int c, ChunkRegs &rg) {   // global loads of chunk c into registers (clamped, unconditional)
        const int p0 = c * PCHUNK;
        const int pmax = min(PCHUNK, pout - p0);
        const int nvec = pmax * H / VW;
        const V *src = reinterpret_cast<const V *>(W1 + (long)p0 * H);
#pragma unroll
        for (int u = 0; u < NWV; u++) rg.w[u] = src[min(u * 256 + tid, nvec - 1)];
#pragma unroll
        for (int u = 0; u < NPT; u++) {
            const int e = u * 256 + tid, b = e >> 6, p = e & 63;
            rg.p[u] = pool[(long)(r0 + min(b, nb - 1)) * pout + p0 + min(p, pmax - 1)];
        }
    };
    auto commit = [&](int c, const ChunkRegs &rg) {   // registers -> LDS images of chunk c
        const int p0 = c * PCHUNK;
        const int pmax = min(PCHUNK, pout - p0);
        const int nvec = pmax * H / VW;
#pragma unroll
        for (int u = 0; u < NWV; u++) {
            const int e = u * 256 + tid;
            if (e < nvec) {
                const int p = (e * VW) / H, n = e * VW - p * H;
                *reinterpret_cast<V *>(wl + p * LDW1 + n) = rg.w[u];
            }
        }
        if (pmax < PCHUNK)   // last chunk: rows past the end must read as zero
            for (int e = pmax * LDW1 + tid; e < PCHUNK * LDW1; e += 256) wl[e] = 0;
#pragma unroll
        for (int u = 0; u < NPT; u++) {
            const int e = u * 256 + tid, b = e >> 6, p = e & 63;
            pt[b * LDP + p] = (b < nb && p < pmax) ? rg.p[u] : T(0);
        }
    };

    issue(0, ra);   // first: the chunk's loads fly while the small operands below are staged
    if (nch > 1) issue(1, rb);
    stage<6, 256>(params + off_w2, H * no, tid, [&](int e, T v) { w2s[e] = v; });
    if (tid < H) b1s[tid] = params[off_b1 + tid];
    if (tid < no) b2s[tid] = params[off_b2 + tid];
  for (;;) {   // one tile per iteration; the first chunk of this tile is in wreg / preg
    if (tid == 0) {
        int nx = -1;
        if (rem > 0) {
            nx = xlo + atomicAdd(ctr + CONV_CTR_STRIDE * xcd + CONV_CTR_STRIDE / 2, 1);
            if (nx >= xhi) nx = -1;
        }
        s_next = nx;   // read after the K loop's barriers
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = typename Mma<T>::acc_t{0, 0, 0, 0};
    // neuron columns 100..111 of the W1 image are never loaded: keep them zero (the previous
    // tile's epilogue used the region)
    for (int e = tid; e < PCHUNK * (LDW1 - H); e += 256) wl[(e / (LDW1 - H)) * LDW1 + H + e % (LDW1 - H)] = 0;
    __syncthreads();
    commit(0, ra);
    FC_STAMP(1);
    // chunk c is in LDS (it came from `cur`, which is free again), chunk c + 1 waits in `nxt`
    auto chunk_step = [&](int c, ChunkRegs &cur, ChunkRegs &nxt) {
        __syncthreads();                         // chunk c is in LDS
        if (c + 2 < nch) issue(c + 2, cur);      // the loads of the chunk after next fly during two multiply phases
        for (int s = 0; s < PCHUNK / 4; s += 4) {
            T a[4], b[4][NJ];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = ap[4 * (s + u)];
#pragma unroll
                for (int j = 0; j < NJ; j++) b[u][j] = bp[4 * (s + u) * LDW1 + boff[j]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int j = 0; j < NJ; j++) acc[j] = Mma<T>::mma(a[u], b[u][j], acc[j]);
        }
        __syncthreads();                         // everyone is done reading chunk c
        if (c + 1 < nch) commit(c + 1, nxt);
    };
    for (int c = 0; c < nch; c += 2) {
        chunk_step(c, ra, rb);
        if (c + 1 < nch) chunk_step(c + 1, rb, ra);
    }

    FC_STAMP(2);
    // ---- epilogue: z1 -> LDS, then 8 lanes per sample --------------------------------
    const int tile_next = s_next;
    const int r0_done = r0, nb_done = nb;
    if (tile_next >= 0) {   // the next tile's first chunk flies during this tile's epilogue
        r0 = tile_next * MS;
        nb = min(MS, nrows - r0);
        issue(0, ra);
        if (nch > 1) issue(1, rb);
    }
    // epilogue geometry: LS lanes per sample (8 with 32 samples, 16 with 16: every thread has work),
    // lane jl producing outputs jl and jl + LS where they exist
    constexpr int LS = 256 / MS;
    const int jl = tid & (LS - 1), s = tid / LS;
    const int o0 = min(jl, no - 1), o1 = min(jl + LS, no - 1);
    const bool has0 = jl < no, has1 = jl + LS < no;
    int lab0 = 0, lab1 = 0;
    if (TRAIN) {   // this thread's two labels (clamped, unconditional): in flight during the activations
        const int sl = min(s, nb_done - 1);
        lab0 = tr.onehot[(long)(r0_done + sl) * no + o0];
        lab1 = tr.onehot[(long)(r0_done + sl) * no + o1];
    }
    T *z1s = wl;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int n = (nt0 + j) * 16 + col;
        if (n < H && jlive[j]) {
#pragma unroll
            for (int r = 0; r < 4; r++) z1s[(mtile * 16 + Mma<T>::row(lane, r)) * LDZ + n] = acc[j][r];
        }
    }
    __syncthreads();
    T *zr = z1s + s * LDZ;
    for (int n = jl; n < H; n += LS) zr[n] = act(zr[n] + b1s[n]);   // gcn_mbgd_fp64.cu:302
    __syncthreads();
    T m0 = 0, m1 = 0;
    if (LS == 8) {
        for (int i = 0; i < H; i++) {
            const T a = zr[i];
            m0 = m0 + w2s[o0 + i * no] * a;
            m1 = m1 + w2s[o1 + i * no] * a;
        }
    } else {
        for (int i = 0; i < H; i++) m0 = m0 + w2s[o0 + i * no] * zr[i];
    }
    const T e0 = rexp(m0 + b2s[o0]), e1 = rexp(m1 + b2s[o1]);
    if (has0) es[s * 16 + o0] = e0;
    if (has1) es[s * 16 + o1] = e1;
    __syncthreads();
    const T sum = tree_n(es + s * 16, no);
    const T p0 = e0 / sum, p1 = e1 / sum;
    if (has0) ps[s * 16 + o0] = p0;
    if (has1) ps[s * 16 + o1] = p1;
    if (s < nb_done && probs) {
        if (has0) probs[(long)(r0_done + s) * no + o0] = p0;
        if (has1) probs[(long)(r0_done + s) * no + o1] = p1;
    }
    __syncthreads();
    if (argmax && jl == 0 && s < nb_done) {
        // count_err: running max starts at 0, strict '>' (gcn_mbgd_fp64.cu:636-647)
        T mx = 0;
        int idx = 0;
        for (int i = 0; i < no; i++)
            if (ps[s * 16 + i] > mx) {
                mx = ps[s * 16 + i];
                idx = i;
            }
        argmax[r0_done + s] = idx;
    }
    if (TRAIN) {
        T *d2s = es;   // the exponentials are dead (every lane holds its probabilities in registers)
        const bool live = s < nb_done;
        const T dv0 = (p0 - lab0) / no, dv1 = (p1 - lab1) / no;   // gcn_mbgd_fp64.cu:352
        if (has0) d2s[s * 16 + o0] = dv0;
        if (has1) d2s[s * 16 + o1] = dv1;
        if (live) {
            if (has0) tr.d2[(long)(r0_done + s) * no + o0] = dv0;
            if (has1) tr.d2[(long)(r0_done + s) * no + o1] = dv1;
        }
        // the loss terms (an exp and a log each) spread over the sample's lanes, from the labels already
        // in registers; they go to the part of the W1 image that z1s does not use
        T *lta = wl + MS * LDZ, *ltb = lta + MS * 16;
        int *labs = reinterpret_cast<int *>(ltb + MS * 16);
        if (has0) {
            loss_terms(p0, lab0, &lta[s * 16 + o0], &ltb[s * 16 + o0]);
            labs[s * 16 + o0] = lab0;
        }
        if (has1) {
            loss_terms(p1, lab1, &lta[s * 16 + o1], &ltb[s * 16 + o1]);
            labs[s * 16 + o1] = lab1;
        }
        __syncthreads();
        if (live) {
            T dd[OMAX];   // entries past the class count are never read by the tree
#pragma unroll
            for (int o = 0; o < OMAX; o++) dd[o] = d2s[s * 16 + o];
            for (int n = jl; n < H; n += LS) {
                // bp_fully_connect(out -> hidden), gcn_mbgd_fp64.cu:416-434: class-wide tree
                T prod[OMAX];
#pragma unroll
                for (int o = 0; o < OMAX; o++) prod[o] = w2s[min(o, no - 1) + n * no] * dd[o];
                const T a = zr[n];
                tr.a1[(long)(r0_done + s) * H + n] = a;
                tr.d1[(long)(r0_done + s) * H + n] = tree_n(prod, no) * (1 + a) * (1 - a);
            }
            if (jl == LS - 1) tr.loss_values[r0_done + s] = loss_sum(lta + s * 16, ltb + s * 16, labs + s * 16, no);   // the ordered sum
        }
    }
    FC_STAMP(3);
    if (tile_next < 0) break;
    tile = tile_next;
    __syncthreads();   // everyone is done with z1s / es / ps (and has read s_next)
  }
}

// ---------------------------------------------------------------------------
// Stand-alone parameter updates (after an all-reduce, or with AdaGrad).
__global__ void update_f64_kernel(double *__restrict__ params, const double *__restrict__ grad,
                                  double lr, int divisor, long n)
{
    const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (k < n) params[k] = params[k] - lr * grad[k] / divisor;   // gcn_mbgd_fp64.cu:519
}

__global__ void update_f32_kernel(float *__restrict__ params, const float *__restrict__ grad,
                                  float lr, long n, int frozen_lo, int frozen_hi)
{
    const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (k < n && !(k >= frozen_lo && k < frozen_hi))
        params[k] = params[k] - lr * grad[k];                     // gcn_mbgd_fp32.cu:529, 561
}

__global__ void update_adagrad_f64_kernel(double *__restrict__ params, double *__restrict__ rms,
                                          const double *__restrict__ grad, double lr, long iter,
                                          long n)
{
    const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (k < n) {
        const double g = grad[k];
        const double r = rms[k] + g * g;     // hyper_version_streamwithalda.cu:373-374
        rms[k] = r;
        params[k] = params[k] - lr * g / sqrt(r / iter + 1);
    }
}

__global__ void count_correct_kernel(const int *__restrict__ argmax, const int *__restrict__ labels,
                                     long n, int *__restrict__ count)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    int hit = 0;
    for (; i < n; i += (long)gridDim.x * blockDim.x) hit += (argmax[i] + 1 == labels[i]);
    for (int off = 32; off > 0; off >>= 1) hit += __shfl_down(hit, off);
    if ((threadIdx.x & 63) == 0 && hit) atomicAdd(count, hit);
}

// ---------------------------------------------------------------------------
// host-side orchestration

// Samples per forward pass through the workspace: as many as keep the pooled features within
// 128 MB, so that the persistent conv workgroups amortise their filter staging over many groups
// (CCNN_FWD_CHUNK overrides it: tests use a small value to cross chunk boundaries cheaply).
inline long fwd_chunk_rows(const ccnn_dims_t &d, size_t elem)
{
    if (const char *s = getenv("CCNN_FWD_CHUNK")) {
        const long v = atol(s);
        if (v > 0) return v;
    }
    long rows = (long)((128ull << 20) / ((size_t)d.pout * elem)) & ~1023L;
    return rows < 16384 ? 16384 : rows > 131072 ? 131072 : rows;
}

// Workgroups of `fn` the device holds at once (the grid of a persistent kernel: more would run
// as a second, mostly empty round).
int resident_blocks(const void *fn, int threads, size_t smem)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, std::pair<int, size_t>>, int> known;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    int &have = known[{fn, {dev, smem}}];
    if (have > 0) return have;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, smem) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    have = per_cu * cus;
    return have;
}

template <typename T>
struct FwdWs {
    T *pool;
    int *ctr;   // work counters of conv_pool_fwd
    static size_t bytes(long rows, const ccnn_dims_t &d)
    {
        return align_up((size_t)rows * d.pout * sizeof(T)) + align_up(8 * CONV_CTR_STRIDE * sizeof(int));
    }
    FwdWs(void *ws, long rows, const ccnn_dims_t &d)
    {
        Carver c(ws);
        pool = c.take<T>((size_t)rows * d.pout);
        ctr = c.take<int>(8 * CONV_CTR_STRIDE);
    }
};

// idx9 == null: `cubes` holds n pre-gathered cubes.  idx9 != null: `cubes` is the scene
// [z][plane] and sample i is the pixel neighbourhood idx9[i][0..9).
// Training (tr != null): samples [first, first + n) of `cubes` in ONE pass (the caller's pool
// buffer holds all n rows), winners to px, the per-sample training outputs through tr.
template <typename T, bool FLIP>
int forward_impl(const T *params, const T *cubes, long n, int z, T *probs, int *argmax, void *ws,
                 size_t ws_bytes, hipStream_t st, const int *idx9 = nullptr, long plane = 0,
                 const FcTrainOut<T> *tr = nullptr, long first = 0, T *pool_out = nullptr, unsigned char *px = nullptr,
                 int *ctr_out = nullptr)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    if (!params || !cubes || n < 0 || (n > 0 && !ws && !tr)) return CCNN_E_INVALID;
    if (n == 0) return 0;
    const long chunk = tr ? n : fwd_chunk_rows(d, sizeof(T));
    const long rows = n < chunk ? n : chunk;
    if (!tr && ws_bytes < FwdWs<T>::bytes(rows, d)) return CCNN_E_WORKSPACE;
    FwdWs<T> w(tr ? nullptr : ws, rows, d);
    if (tr) {
        w.pool = pool_out;
        w.ctr = ctr_out;
    }
    const int nreg16 = ((d.cube + 63) / 64 + 15) / 16;   // prefetch registers per lane, in sixteens
    if (nreg16 > CONV_MAXREG / 16) return CCNN_E_RANGE;
    typedef void (*ConvKern)(const T *, const T *, long, long, int, int, int, int, T *, const int *, long, int *,
                             unsigned char *);
    static const ConvKern kern[3][4] = {
        {conv_pool_fwd<T, FLIP, 16, false, false>, conv_pool_fwd<T, FLIP, 32, false, false>,
         conv_pool_fwd<T, FLIP, 48, false, false>, conv_pool_fwd<T, FLIP, 64, false, false>},
        {conv_pool_fwd<T, FLIP, 16, true, false>, conv_pool_fwd<T, FLIP, 32, true, false>,
         conv_pool_fwd<T, FLIP, 48, true, false>, conv_pool_fwd<T, FLIP, 64, true, false>},
        {conv_pool_fwd<T, FLIP, 16, false, true>, conv_pool_fwd<T, FLIP, 32, false, true>,
         conv_pool_fwd<T, FLIP, 48, false, true>, conv_pool_fwd<T, FLIP, 64, false, true>}};
    const bool train = tr != nullptr;
    const ConvKern conv = kern[train ? 2 : idx9 != nullptr][nreg16 - 1];
    // Waves per workgroup: the count (8, 4, 2 or 1) that puts the most waves on the device; LDS
    // (the filter image plus a cube and a pooled row per wave) and registers both limit it.
    int nwv = 0, resident = 0;
    size_t csm = 0;
    for (int w = CONV_MAXWAVES; w >= 1; w >>= 1) {
        const size_t b = conv_smem<T>(d, w, train);
        if (b > 160 * 1024) continue;
        if (int e = allow_smem(conv, b)) return e;
        const int r = resident_blocks(reinterpret_cast<const void *>(conv), 64 * w, b);
        if (r * w > resident * nwv) {
            nwv = w;
            resident = r;
            csm = b;
        }
    }
    if (nwv == 0) return CCNN_E_RANGE;
    typedef void (*FcKern)(const T *, int, int, int, int, int, const T *, int, int, T *, int *, int *, FcTrainOut<T>);
    constexpr int C13 = CCNN_CLASSES;
    const bool c13 = d.n_classes == C13;
    const FcKern fc32 = train ? (c13 ? fc_head_fwd<T, 32, true, C13> : fc_head_fwd<T, 32, true, 0>)
                              : (c13 ? fc_head_fwd<T, 32, false, C13> : fc_head_fwd<T, 32, false, 0>);
    const FcKern fc16 = train ? (c13 ? fc_head_fwd<T, 16, true, C13> : fc_head_fwd<T, 16, true, 0>)
                              : (c13 ? fc_head_fwd<T, 16, false, C13> : fc_head_fwd<T, 16, false, 0>);
    if (int e = allow_smem(fc32, FcHeadLds<T, 32>::bytes)) return e;
    if (int e = allow_smem(fc16, FcHeadLds<T, 16>::bytes)) return e;
    const int fc_res32 = resident_blocks(reinterpret_cast<const void *>(fc32), 256, FcHeadLds<T, 32>::bytes);
    const int fc_res16 = resident_blocks(reinterpret_cast<const void *>(fc16), 256, FcHeadLds<T, 16>::bytes);
    if (fc_res32 <= 0 || fc_res16 <= 0) return CCNN_E_RANGE;
    for (long s0 = 0; s0 < n; s0 += rows) {
        const int nr = (int)((n - s0) < rows ? (n - s0) : rows);
        const int groups = (nr + nwv - 1) / nwv;
        const int pg = groups < resident ? groups : resident;
        // 16-sample tiles while 32-sample ones would leave CUs with fewer than two workgroups
        const bool small = (nr + FH_MS - 1) / FH_MS < 512;
        const int ms = small ? 16 : FH_MS;
        const int ntiles = (nr + ms - 1) / ms;
        const int fc_resident = small ? fc_res16 : fc_res32;
        const int fg = ntiles < fc_resident ? ntiles : fc_resident;
        if ((long)nr > 3L * pg * nwv || ntiles > fg) {   // work beyond the fixed shares is handed out through the counters
            hipError_t e = hipMemsetAsync(w.ctr, 0, 8 * CONV_CTR_STRIDE * sizeof(int), st);
            if (e != hipSuccess) return (int)e;
        }
        {
            launch(train ? "conv_pool_fwd_train" : "conv_pool_fwd", conv, dim3(pg), dim3(64 * nwv), csm, st, params, cubes,
                   first + s0, (long)nr, d.cube, d.re, d.mre, d.off_b0, w.pool, idx9, plane, w.ctr, px);
        }
        CCNN_LAUNCH_CHECK();
        {
            launch(train ? "fc_head_fwd_train" : "fc_head_fwd", small ? fc16 : fc32, dim3(fg), dim3(256),
                   small ? FcHeadLds<T, 16>::bytes : FcHeadLds<T, 32>::bytes, st,
                   params, d.off_w1, d.off_b1, d.off_w2, d.off_b2, d.n_classes, w.pool, nr, d.pout,
                   probs ? probs + s0 * d.n_classes : nullptr, argmax ? argmax + s0 : nullptr, w.ctr,
                   tr ? *tr : FcTrainOut<T>{});
        }
        CCNN_LAUNCH_CHECK();
    }
    return 0;
}

// The forward half of the batched minibatch step (train_fwd.h): conv + pool (+ winners) and
// FC1 + head + deltas over samples [first, first + B) of `cubes`.
template <typename T, bool FLIP>
int train_forward(const T *params, const T *cubes, const int *onehot, long first, int B, int z, T *pool,
                  unsigned char *px, T *a1, T *d1, T *d2, T *probs, double *loss_values, int *ctr, hipStream_t st)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    FcTrainOut<T> tr{onehot + first * d.n_classes, a1, d1, d2, loss_values};
    return forward_impl<T, FLIP>(params, cubes, B, z, probs, nullptr, nullptr, 0, st, nullptr, 0, &tr, first, pool, px, ctr);
}
template int train_forward<float, true>(const float *, const float *, const int *, long, int, int, float *,
                                        unsigned char *, float *, float *, float *, float *, double *, int *, hipStream_t);
template int train_forward<double, false>(const double *, const double *, const int *, long, int, int, double *,
                                          unsigned char *, double *, double *, double *, double *, double *, int *,
                                          hipStream_t);

}  // namespace ccnn

using namespace ccnn;

#ifdef CCNN_CONV_CLOCK
extern "C" int ccnn_debug_conv_clock(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_clock), sizeof(unsigned long long) * 4);
}
extern "C" int ccnn_debug_conv_wg(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_wg), sizeof(unsigned long long) * 3 * 2048);
}
#endif

extern "C" {

size_t ccnn_forward_workspace_bytes(long n, int z, int elem_size)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || n <= 0) return 0;
    const long chunk = fwd_chunk_rows(d, elem_size == 8 ? 8 : 4);
    const long rows = n < chunk ? n : chunk;
    return elem_size == 8 ? FwdWs<double>::bytes(rows, d) : FwdWs<float>::bytes(rows, d);
}

int ccnn_forward_f32(const float *params, const float *cubes, long n, int z, float *probs, int *argmax,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    return forward_impl<float, true>(params, cubes, n, z, probs, argmax, workspace, workspace_bytes,
                                     (hipStream_t)stream);
}

int ccnn_forward_f64(const double *params, const double *cubes, long n, int z, double *probs,
                     int *argmax, void *workspace, size_t workspace_bytes, void *stream)
{
    return forward_impl<double, false>(params, cubes, n, z, probs, argmax, workspace, workspace_bytes,
                                       (hipStream_t)stream);
}

int ccnn_forward_scene_f32(const float *params, const float *scene, const int *idx9, long n, int x, int y, int z,
                           float *probs, int *argmax, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!idx9 && n > 0) return CCNN_E_INVALID;
    if (x <= 0 || y <= 0 || (long)x * y > 0x7fffffffL) return CCNN_E_RANGE;
    return forward_impl<float, true>(params, scene, n, z, probs, argmax, workspace, workspace_bytes,
                                     (hipStream_t)stream, idx9, (long)x * y);
}

int ccnn_forward_scene_f64(const double *params, const double *scene, const int *idx9, long n, int x, int y,
                           int z, double *probs, int *argmax, void *workspace, size_t workspace_bytes,
                           void *stream)
{
    if (!idx9 && n > 0) return CCNN_E_INVALID;
    if (x <= 0 || y <= 0 || (long)x * y > 0x7fffffffL) return CCNN_E_RANGE;
    return forward_impl<double, false>(params, scene, n, z, probs, argmax, workspace, workspace_bytes,
                                       (hipStream_t)stream, idx9, (long)x * y);
}

int ccnn_count_correct(const int *argmax, const int *labels, long n, int *count, void *stream)
{
    if (!count || n < 0 || (n > 0 && (!argmax || !labels))) return CCNN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return 0;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    {
        launch("count_correct_kernel", count_correct_kernel, dim3((unsigned)blocks), dim3(256), 0, st, argmax, labels, n,
               count);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

int ccnn_update_f64(double *params, const double *grad, double lr, int divisor, long n, void *stream)
{
    if (!params || !grad || divisor <= 0 || n < 0) return CCNN_E_INVALID;
    if (n == 0) return 0;
    {
        launch("update_f64_kernel", update_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
               (hipStream_t)stream, params, grad, lr, divisor, n);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

int ccnn_update_f32(float *params, const float *grad, float lr, int z, int freeze_conv_bias, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params || !grad) return CCNN_E_INVALID;
    const int lo = freeze_conv_bias ? d.off_b0 : 0, hi = freeze_conv_bias ? d.off_w1 : 0;
    {
        launch("update_f32_kernel", update_f32_kernel, dim3((d.nparams + 255) / 256), dim3(256), 0, (hipStream_t)stream,
               params, grad, lr, d.nparams, lo, hi);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

int ccnn_update_adagrad_f64(double *params, double *rms, const double *grad, double lr, long iter, long n,
                            void *stream)
{
    if (!params || !rms || !grad || iter <= 0 || n < 0) return CCNN_E_INVALID;
    if (n == 0) return 0;
    {
        launch("update_adagrad_f64_kernel", update_adagrad_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
               (hipStream_t)stream, params, rms, grad, lr, iter, n);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

#ifdef CCNN_DEBUG_LDS_POISON
// Diagnostic builds only: fills the LDS of every CU with 0xFF bytes (NaN as floats), so that a
// kernel which reads LDS it never wrote shows up as a changed result.
__global__ void poison_lds_kernel(int words)
{
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds_words[i] = 0xffffffffu;
    __syncthreads();
    if (lds_words[(threadIdx.x * 7919) % words] != 0xffffffffu) __builtin_trap();
}
extern "C" int ccnn_debug_poison_lds(void *stream)
{
    const size_t bytes = 160 * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(poison_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), bytes, (hipStream_t)stream, (int)(bytes / 4));
    return (int)hipGetLastError();
}
#endif
