// The per-sample part of a minibatch step for batches larger than the chip has compute units:
// ONE launch in which every workgroup takes a tile of up to eight samples through the whole chain
//
//   conv + tanh + pool (one wave per sample, MFMA)            convolution, maxpooling
//   FC1 for the tile as one [ms x Pout] x [Pout x 100] MFMA product, W1 read once per TILE
//                                                              fully_connect
//   output layer, softmax, loss, delta2, delta1 (one wave per sample)
//                                                              output_and_dvalue, loss_function, bp_fully_connect
//   delta at the pooled features as one [Pout x 100] x [100 x ms] MFMA product, W1 read once per tile
//                                                              bp_fully_connect (hidden -> pool)
//   bp_maxpooling scatter + conv gradient (MFMA), summed over the tile in the accumulators
//                                                              bp_maxpooling, bp_convolution
//
// (gcn_mbgd_fp64.cu:193-496, gcn_mbgd_fp32.cu:207-500).  What reaches memory per sample is what
// grad_reduce needs for the batch sums (pool, a1, delta1, delta2 rows); the conv gradient leaves as
// one [F K + F] partial per WORKGROUP.  The per-sample kernel of mbgd.hip walks all of W1 twice
// per sample, the layer-by-layer kernels hand every intermediate through memory and pay five
// launches; here the tile's state stays in LDS from the cube to the conv gradient.
#include <cstdlib>
#include <type_traits>

#include "ccnn_device.h"
#include "profile.h"
#include "train_fwd.h"

// Diagnostic builds only (-DCCNN_TILE_TIMING): thread 0 of workgroup 5 stamps the 100 MHz wall clock
// at TT(k) (and the shader clock for k < 12, at [20 + k]: their ratio is the clock the phase ran
// at); ccnn_debug_tile_ticks() returns the stamps of the last launch.
#ifdef CCNN_TILE_TIMING
__device__ unsigned long long g_tile_ticks[32];
#define TT(k)                                                                        \
    do {                                                                             \
        if (blockIdx.x == 5 && threadIdx.x == 0) {                                   \
            g_tile_ticks[k] = wall_clock64();                                        \
            if ((k) < 12) g_tile_ticks[20 + (k)] = __builtin_amdgcn_s_memtime();     \
        }                                                                            \
    } while (0)
extern "C" int ccnn_debug_tile_ticks(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_ticks), sizeof(unsigned long long) * 32);
}
// ... and every workgroup its first instruction and the end of each of its waves ([block][0], [block][1 + wave])
__device__ unsigned long long g_tile_wg[1024][17];
#define TWG(slot)                                                                                       \
    do {                                                                                                \
        if (blockIdx.x < 1024 && (threadIdx.x & 63) == 0) g_tile_wg[blockIdx.x][slot] = wall_clock64(); \
    } while (0)
extern "C" int ccnn_debug_tile_wg(unsigned long long *out)   // [1024][17]
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_wg), sizeof(unsigned long long) * 1024 * 17);
}
#else
#define TT(k) do { } while (0)
#define TWG(slot) do { } while (0)
#endif

namespace ccnn {

constexpr int TILE_LDS_LIMIT = 160 * 1024;
constexpr int TILE_RED = 2048;    // values per wave in one round of the conv-gradient reduction (4 tap tiles x 2 filter tiles)
constexpr int TILE_ACC_MS = 4;    // up to this many samples per tile the conv gradient is summed in the accumulators
constexpr int TILE_DEFAULT_WAVES = 16;
constexpr int TILE_ZN = 128;      // columns of an FC1 partial-sum row (neurons in the permuted order below: 64 T + 4 col + j)

// LDS carve (elements of T; every region a multiple of 4 elements so 16-byte accesses stay aligned).
struct TileLayout {
    int pld, hld, kc, ldd, cube_pad, b1_len;
    int pl, a1, d1, w2, b1v, b2v, es, d2, bs, rb, cube, total;
    size_t px_off, bytes;
};

template <typename T>
__host__ __device__ inline TileLayout tile_layout(int re, int mre, int nwv, int ms, int nch)
{
    TileLayout L;
    const int pout = F * mre, ks = (re + 3) >> 2;
    L.pld = pout + ((36 - pout % 32) % 32);   // = 4 (mod 32): the 16 sample rows x 4 k of an A fragment spread over all banks
    L.hld = 108;
    L.kc = (ks + nch - 1) / nch;              // k-steps (of four positions) per delta-image chunk
    L.ldd = 4 * L.kc * LDWK;                  // one position-major delta tile [4 kc][16 filters]
    L.cube_pad = (conv_cube_pad(re) + 3) & ~3;
    // sample slots (cube, delta image, reduction row): one per sample for tiles whose conv gradient is
    // summed in the accumulators; larger tiles add all eight waves' accumulator rows, present or not
    const int nsw = nwv < 8 ? nwv : 8;
    const int ns = ms <= TILE_ACC_MS && ms < nsw ? ms : nsw;
    int b1 = WP_LEN;                                              // conv filters (forward)
    if (b1 < (nwv / 2) * 8 * TILE_ZN) b1 = (nwv / 2) * 8 * TILE_ZN;   // FC1 partial sums [K split][8][128]
    if (b1 < ms * pout) b1 = ms * pout;                           // delta x tanh' (the conv-bias path)
    if (b1 < ns * 2 * L.ldd) b1 = ns * 2 * L.ldd;                 // per-sample delta images
    if (ms > TILE_ACC_MS && b1 + ns * L.cube_pad < ns * TILE_RED)
        b1 = ns * (TILE_RED - L.cube_pad);                        // conv-gradient reduction rounds (they run over into the cubes)
    L.b1_len = (b1 + 3) & ~3;
    int o = 0;
    L.pl = o;   o += ms * L.pld;
    L.a1 = o;   o += ms * L.hld;
    L.d1 = o;   o += ms * L.hld;
    L.w2 = o;   o += H * OMAX;
    L.b1v = o;  o += 104;
    L.b2v = o;  o += 16;
    L.es = o;   o += 8 * 16;
    L.d2 = o;   o += 8 * 16;
    L.bs = o;   o += 8 * 32;    // [8][32] conv-bias sums
    L.rb = o;   o += L.b1_len;
    L.cube = o; o += ns * L.cube_pad;
    L.total = o;
    L.px_off = (size_t)o * sizeof(T);
    L.bytes = L.px_off + (size_t)((ms * pout + 15) & ~15);
    return L;
}

// Waves per workgroup to try first.  Sixteen (four per SIMD, 128 registers each) for full fp32 tiles:
// two waves share a sample's conv chains and its conv-gradient products - 45.9 -> 43.1 us at 2048
// samples of 103 bands, 82 -> 76 us at 200 bands.  forced_waves = 8 | 16 overrides (the test switch
// CCNN_TILE_WAVES, read by the caller once per call: tests run both), anything else: the default.
template <typename T>
inline int tile_max_waves(int ms, int forced_waves)
{
    if (sizeof(T) != 4 || ms <= TILE_ACC_MS) return 8;
    return forced_waves == 8 || forced_waves == 16 ? forced_waves : TILE_DEFAULT_WAVES;
}

int tile_waves_env()
{
    const char *e = getenv("CCNN_TILE_WAVES");
    return e ? atoi(e) : 0;
}
// CCNN_TILE_MS=1..8 (diagnostics): samples per tile instead of the count tile_plan would choose
static int tile_ms_env()
{
    const char *e = getenv("CCNN_TILE_MS");
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 8 ? v : 0;
}

template <typename T>
bool tile_plan_waves(int B, const ccnn_dims_t &d, int forced_waves, TilePlan *out)
{
    // samples per tile: enough to cover the batch with one workgroup per compute unit (256 on the
    // part this library is built for), at most one per wave
    int want = (B + 255) / 256;
    if (want < 1) want = 1;
    if (want > 8) want = 8;
    if (const int forced = tile_ms_env()) want = forced;
    for (int ms = want; ms >= 1; ms--) {
        for (int nwv = tile_max_waves<T>(ms, forced_waves); nwv >= 4; nwv >>= 1) {
            if (ms > nwv) continue;
            for (int nch = 1; nch <= 8; nch <<= 1) {
                const TileLayout L = tile_layout<T>(d.re, d.mre, nwv, ms, nch);
                if (L.bytes <= (size_t)TILE_LDS_LIMIT) {
                    out->nwv = nwv;
                    out->ms = ms;
                    out->nch = nch;
                    out->smem = L.bytes;
                    out->tiles = (B + ms - 1) / ms;
                    return true;
                }
            }
        }
    }
    return false;
}
template bool tile_plan_waves<float>(int, const ccnn_dims_t &, int, TilePlan *);
template bool tile_plan_waves<double>(int, const ccnn_dims_t &, int, TilePlan *);

template <typename T>
bool tile_plan(int B, const ccnn_dims_t &d, TilePlan *out)
{
    return tile_plan_waves<T>(B, d, tile_waves_env(), out);
}
template bool tile_plan<float>(int, const ccnn_dims_t &, TilePlan *);
template bool tile_plan<double>(int, const ccnn_dims_t &, TilePlan *);

// Four consecutive values of T from a 16-byte aligned address.
template <typename T>
struct Quad;
template <>
struct Quad<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 v;
    __device__ __forceinline__ void load(const float *p) { v = *reinterpret_cast<const v4 *>(p); }
    __device__ __forceinline__ float operator[](int j) const { return v[j]; }
};
template <>
struct Quad<double> {
    typedef double v2 __attribute__((ext_vector_type(2)));
    v2 lo, hi;
    __device__ __forceinline__ void load(const double *p)
    {
        lo = *reinterpret_cast<const v2 *>(p);
        hi = *reinterpret_cast<const v2 *>(p + 2);
    }
    __device__ __forceinline__ double operator[](int j) const { return j < 2 ? lo[j] : hi[j - 2]; }
};

// NO: the class count when it is the reference's 13 (compile-time then), 0 for any other (d.n_classes).
// (Tried and not kept: an eight-wave instantiation held to 128 registers so that TWO workgroups of
// four samples share a compute unit, one's latency phases under the other's matrix phases - 52.1
// against 38.5 us at 2048 samples of 103 bands: half tiles read W1 twice as often and their conv
// gradient runs 22 full 16-wide products per k-step where a full tile runs 11 and 12 short ones.)
template <typename T, bool FLIP, int NO, int NWV>
__global__ __launch_bounds__(64 * NWV) void tile_fwd_bwd(const T *__restrict__ params, const T *__restrict__ cubes,
                                                    const int *__restrict__ onehot, long first, int B, int ms, int nch,
                                                    ccnn_dims_t d, T *__restrict__ pool_g, T *__restrict__ a1_g,
                                                    T *__restrict__ d1_g, T *__restrict__ d2_g, T *__restrict__ probs_g,
                                                    double *__restrict__ loss_values, T *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec16<T>::type V;
    constexpr int VW = Vec16<T>::W;
    constexpr int nt = 64 * NWV, nwv = NWV;
    constexpr int NS = NWV < 8 ? NWV : 8;   // sample slots
    // (the wave number through v_readfirstlane: the compiler then keeps everything indexed by it -
    // job numbers, sample slots, LDS bases - in scalar registers and scalar instructions)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kq = lane >> 4;
    const int re = d.re, mre = d.mre, pout = d.pout, no = NO ? NO : d.n_classes;
    const TileLayout L = tile_layout<T>(re, mre, nwv, ms, nch);
    T *base = reinterpret_cast<T *>(smem);
    T *pl = base + L.pl, *a1s = base + L.a1, *d1s = base + L.d1, *w2s = base + L.w2, *b1s = base + L.b1v;
    T *b2s = base + L.b2v, *es = base + L.es, *d2s = base + L.d2;
    T *bs = base + L.bs, *rb = base + L.rb;
    const bool slot = wave < NS && (ms > TILE_ACC_MS || wave < ms);   // this wave stages a sample slot's cube
    T *cube = base + L.cube + (slot ? wave : 0) * L.cube_pad;
    unsigned char *px = reinterpret_cast<unsigned char *>(smem + L.px_off);

    const int r0 = blockIdx.x * ms;
    const int nb = min(ms, B - r0);
    const bool mine = wave < nb;                 // this wave owns sample r0 + wave
    const long row = r0 + (mine ? wave : 0);     // (a valid row for the loads of a wave that owns none)
    T *plr = pl + (mine ? wave : 0) * L.pld;
    unsigned char *pxr = px + (mine ? wave : 0) * pout;
    const T *W1 = params + d.off_w1;

    TT(0);
    if (threadIdx.x == 0) TWG(0);
    const T bias0 = params[d.off_b0 + col];
    const T biasx = params[d.off_b0 + 16 + (col & 3)];
    const int lab_g = onehot[(first + row) * no + min(lane, no - 1)];   // needed deep in the chain: fetched now
    // ---- stage: this wave's cube, the conv filters, the output layer ------------------------
    {
        const T *src = cubes + (first + row) * (long)d.cube;
        T t0[16];   // the first 1024 cube elements and the filters: one memory round trip, before any LDS work
#pragma unroll
        for (int u = 0; u < 16; u++) t0[u] = src[min(u * 64 + lane, d.cube - 1)];
        // The conv filters as conv_pool_chain wants them (ccnn_device.h: filters 0..15 tap-major, then
        // the packed tile; the fp32 reference walks the filter backwards in the forward pass only,
        // gcn_mbgd_fp32.cu:241-243).  fp32, eight waves or more: thread t owns elements t, t + nt, .. of
        // the IMAGE and fetches the filter value (or a zero) that belongs there, so consecutive lanes
        // store consecutive words - no zeroing pass, no barrier before the stores, no bank conflicts.
        // (Owning elements of the FILTERS instead, as the other forms below do, puts the 64 stores of a
        // wave 16 words apart: two banks, 32-way conflicts, about 2 us of LDS time per tile with 16 waves.)
        constexpr bool IMG = sizeof(T) == 4 && NWV >= 8;
        static_assert(LDWK == 16, "the image element -> (tap, filter) split below");
        constexpr int NWR = IMG ? (WP_LEN + nt - 1) / nt : (F * K + nt - 1) / nt;
        T wreg[NWR];
        unsigned wok = 0;   // IMG: bit u = element u holds a filter value (else a zero)
#pragma unroll
        for (int u = 0; u < NWR; u++) {
            if constexpr (IMG) {
                const int e = u * nt + tid;
                int f, kk;
                if (e < WL_TILE) {
                    kk = e >> 4;
                    f = e & 15;
                } else {
                    const int x = e - WL_TILE, c = x & 15;
                    f = 16 + (c & 3);
                    kk = (x >> 4) - WSTEP * (c >> 2);   // column c = shift c / 4 of filter 16 + c % 4
                }
                const bool ok = e < WP_LEN && kk >= 0 && kk < K;
                wreg[u] = params[d.off_w0 + (ok ? f * K + (FLIP ? K - 1 - kk : kk) : 0)];
                wok |= (unsigned)ok << u;
            } else {
                wreg[u] = params[d.off_w0 + min(u * nt + tid, F * K - 1)];   // clamped, unconditional
            }
        }
        constexpr int NW2 = (H * OMAX + nt - 1) / nt;
        T w2r[NW2];
#pragma unroll
        for (int u = 0; u < NW2; u++) w2r[u] = params[d.off_w2 + min(u * nt + tid, H * no - 1)];
        const T b1r = params[d.off_b1 + min(tid, H - 1)], b2r = params[d.off_b2 + min(tid, no - 1)];
        if constexpr (!IMG)
            for (int e = tid; e < WP_LEN; e += nt) rb[e] = 0;
#pragma unroll
        for (int u = 0; u < NW2; u++)
            if (u * nt + tid < H * no) w2s[u * nt + tid] = w2r[u];
        if (tid < H) b1s[tid] = b1r;
        if (tid < no) b2s[tid] = b2r;
        TT(11);
        if constexpr (IMG) {
            TT(12);
#pragma unroll
            for (int u = 0; u < NWR; u++) {
                const int e = u * nt + tid;
                if (e < WP_LEN) rb[e] = (wok >> u & 1) ? wreg[u] : T(0);
            }
        } else {
            lds_barrier();
            TT(12);
#pragma unroll
            for (int u = 0; u < NWR; u++) {
                const int e = u * nt + tid;
                if (e < F * K) {
                    const int f = e / K, k = e - f * K;
                    const int kk = FLIP ? K - 1 - k : k;
                    if (f < 16) {
                        rb[kk * LDWK + f] = wreg[u];
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; g++) rb[WL_TILE + (kk + WSTEP * g) * LDWK + 4 * g + (f - 16)] = wreg[u];
                    }
                }
            }
        }
        TT(13);
        for (int e = d.cube + lane; slot && e < L.cube_pad; e += 64) cube[e] = 0;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int e = u * 64 + lane;
            if (slot && e < d.cube) cube[e] = t0[u];
        }
        for (int e0 = 1024; e0 < d.cube; e0 += 1024) {
            T t[16];
#pragma unroll
            for (int u = 0; u < 16; u++) t[u] = src[min(e0 + u * 64 + lane, d.cube - 1)];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = e0 + u * 64 + lane;
                if (slot && e < d.cube) cube[e] = t[u];
            }
        }
    }
    TT(14);
    lds_barrier();
    TT(15);

    // FC1 geometry (the product itself comes after the conv phase; its first W1 loads are issued
    // before it).  Every workgroup reads all of W1: its waves' K ranges are dealt in an order rotated
    // by the workgroup number, so that at any moment the workgroups of an XCD pull from different
    // L2 channels instead of queueing on the same lines.
    constexpr int KQ = nwv >> 1;
    // fp32: FC1 runs on the 16-block 4 x 4 x 1 instruction (below), one pooled feature per instruction,
    // and a register of a set holds one W1 value; fp64: 16 x 16 x 4 tiles, a set register = one k-step
    constexpr bool FC4 = sizeof(T) == 4;
    constexpr int U = FC4 ? (NWV == 16 ? 16 : 32) : 4;   // features (fp32) / k-steps (fp64) per register set, two sets
    const int T2 = wave & 1, q = ((wave >> 1) + blockIdx.x) & (KQ - 1);
    const int ks1 = pout >> 2, kper = (ks1 + KQ - 1) / KQ;
    const int s_lo = q * kper, s_hi = min(s_lo + kper, ks1);
    const int len = s_hi - s_lo;
    const int flen = FC4 ? 4 * len : len;           // length of the range in set registers
    const int ncol = FC4 ? 64 * T2 + lane : 64 * T2 + 4 * col;
    const bool nvalid = FC4 ? ncol < H : ncol + 3 < H;
    // (an empty range still loads from valid rows)
    const T *wp = W1 + (nvalid ? ncol : 0) + (4L * min(s_lo, ks1 - 1) + (FC4 ? 0 : kq)) * H;
    struct W1Set {
        std::conditional_t<FC4, T, Quad<T>> w[U];
    } fa, fb;
    auto fc1_issue = [&](int i, W1Set &st) {   // registers i .. i + U - 1 of this wave's range (clamped, unconditional)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = max(min(i + u, flen - 1), 0);
            if constexpr (FC4) st.w[u] = wp[(long)H * k];
            else st.w[u].load(wp + 4 * H * k);
        }
    };
    fc1_issue(0, fa);
    fc1_issue(U, fb);

    TT(1);
    // ---- conv + tanh + pool: the tile's (sample, chain) jobs dealt to the waves ----------------
    // With a full tile a wave gets the chains of its own sample; with fewer samples than waves a
    // sample's chains spread over several waves.
    if (nb == nwv) {
        // a full tile: every wave runs all chains of its own sample, the position tiles in groups
        // that share their filter fragments
        conv_pool_wave<T>(cube, rb, re, bias0, biasx, lane, [&](int f, int m, T v, int bt) {
            plr[f * mre + m] = v;
            pxr[f * mre + m] = (unsigned char)(bt - 2 * m);   // which member of the pair won
        });
    } else {
        const int nc = conv_chains(re);
        for (int job = wave; job < nb * nc; job += nwv) {
            const int c = job / nb, s = job - c * nb;
            T *ps = pl + s * L.pld;
            unsigned char *pxs = px + s * pout;
            conv_pool_chain<T>(base + L.cube + s * L.cube_pad, rb, re, c, bias0, biasx, lane, [&](int f, int m, T v, int bt) {
                ps[f * mre + m] = v;
                pxs[f * mre + m] = (unsigned char)(bt - 2 * m);
            });
        }
    }
    lds_barrier();
    {
        const int nvr = pout / VW;   // pout = 20 mre: whole 16-byte vectors
        for (int e = tid; e < nb * nvr; e += nt) {
            const int s = e / nvr, o = e - s * nvr;
            reinterpret_cast<V *>(pool_g + (long)(r0 + s) * pout)[o] = reinterpret_cast<const V *>(pl + s * L.pld)[o];
        }
    }

    TT(2);
    // ---- FC1: z1[s][n] = sum_p pool[s][p] W1[n + 100 p] ----------------------------------------
    // Rows of the MFMA tile = samples, K = pooled features, split over the waves' K ranges; the
    // columns are neurons in a permuted order so that every lane fetches W1 in 16-byte pieces: wave
    // half T covers neurons 64 T .. 64 T + 63, column c of its j-th tile is neuron 64 T + 4 c + j,
    // i.e. one vector load feeds four tiles.  A row of W1 (400 bytes) is fetched exactly once per
    // workgroup: 256 bytes by a T = 0 wave, 144 by its T = 1 partner.
    // fp32: the 16-wide tile would be half empty (eight samples), and the product runs at the matrix
    // pipe's rate, so it goes to v_mfma_f32_4x4x1 (16 independent 4 x 4 x 1 blocks, 8 cycles): a block =
    // four samples x four neurons x ONE pooled feature.  Lane l supplies W1[p][64 T + l] (a 256-byte
    // run of the row per load) and pool[l mod 4 + 4 g][p] (sample group g, four features per 16-byte
    // LDS read) and receives z[4 g + i][64 T + l] in register i: two instructions of 8 cycles per
    // feature and 64 neurons, where four half-empty 16 x 16 x 4 tiles took 32 cycles per feature.
    {
        T *zp = rb;   // [KQ][8][TILE_ZN] (the filters are dead)
        if constexpr (FC4) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const T *ap0 = pl + min(lane & 3, ms - 1) * L.pld + 4 * s_lo;
            const T *ap1 = pl + min((lane & 3) + 4, ms - 1) * L.pld + 4 * s_lo;
            f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            auto fc1_mul = [&](int i, const W1Set &st) {
#pragma unroll
                for (int u = 0; u < U; u += 4) {
                    if (i + u < flen) {   // (the range is whole k-steps: four features)
                        const V a0 = *reinterpret_cast<const V *>(ap0 + i + u);
                        const V a1 = *reinterpret_cast<const V *>(ap1 + i + u);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[k], st.w[u + k], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[k], st.w[u + k], acc1, 0, 0, 0);
                        }
                    }
                }
            };
            for (int i = 0; i < flen; i += 2 * U) {
                fc1_mul(i, fa);
                if (i + 2 * U < flen) fc1_issue(i + 2 * U, fa);
                fc1_mul(i + U, fb);
                if (i + 3 * U < flen) fc1_issue(i + 3 * U, fb);
            }
            TT(3);
            if (nvalid) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r < ms) zp[(q * 8 + r) * TILE_ZN + ncol] = acc0[r];
                    if (4 + r < ms) zp[(q * 8 + 4 + r) * TILE_ZN + ncol] = acc1[r];
                }
            }
        } else {
        const T *ap = pl + min(col, ms - 1) * L.pld + 4 * s_lo + kq;
        typename Mma<T>::acc_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = typename Mma<T>::acc_t{0, 0, 0, 0};
        auto fc1_mul = [&](int i, const W1Set &st) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (i + u < len) {
                    const T a = ap[4 * (i + u)];
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[j] = Mma<T>::mma(a, st.w[u][j], acc[j]);
                }
            }
        };
        // sets A and B were issued before the conv phase; each is refilled as soon as it is consumed
        for (int i = 0; i < len; i += 2 * U) {
            fc1_mul(i, fa);
            if (i + 2 * U < len) fc1_issue(i + 2 * U, fa);
            fc1_mul(i + U, fb);
            if (i + 3 * U < len) fc1_issue(i + 3 * U, fb);
        }
        TT(3);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int sr = Mma<T>::row(lane, r);
                if (sr < ms && nvalid) zp[(q * 8 + sr) * TILE_ZN + ncol + j] = acc[j][r];
            }
        }
        }
        lds_barrier();
        for (int e = tid; e < nb * H; e += nt) {
            const int s = e / H, n = e - s * H;
            T z = 0;
            for (int qq = 0; qq < KQ; qq++) z = z + zp[(qq * 8 + s) * TILE_ZN + n];
            const T a = act(z + b1s[n]);   // gcn_mbgd_fp64.cu:302
            a1s[s * L.hld + n] = a;
            a1_g[(long)(r0 + s) * H + n] = a;
        }
        lds_barrier();
    }

    TT(4);
    // The first W1 rows of the delta-pool product fly during the head (which needs no memory).
    constexpr int NG = (H + 4 * VW - 1) / (4 * VW);   // vectors per lane along n
    const int ntp = (pout + 15) >> 4;                 // 16-feature tiles of the product
    auto w1_tile = [&](int pt, V(&a)[NG]) {           // rows pt*16 + col of W1 (clamped, unconditional), k order permuted as below
        const int prow = min(pt * 16 + col, pout - 1);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int n = VW * kq + 4 * VW * g;
            a[g] = reinterpret_cast<const V *>(W1 + (long)prow * H)[min(n, H - VW) / VW];
            if (n >= H) a[g] = V{};
        }
    };
    // this wave's tiles: wave, wave + nwv, .. in an order that starts at a different tile per workgroup
    const int prot = (int)((blockIdx.x * 3u) % (unsigned)ntp);
    auto tile_of = [&](int i) {   // i-th tile slot (may be past the end: clamped, never multiplied)
        const int t = min(i, ntp - 1) + prot;
        return t >= ntp ? t - ntp : t;
    };
    constexpr int NPF = sizeof(T) == 4 && NWV < 16 ? 4 : 2;   // tiles in flight per wave
    V wq[NPF][NG];
#pragma unroll
    for (int h = 0; h < NPF; h++) w1_tile(tile_of(wave + h * nwv), wq[h]);

    TT(5);
    // ---- output layer, softmax, loss, delta2, delta1: one wave per sample -----------------------
    if (mine) {
        const T *ar = a1s + wave * L.hld;
        const int o = lane & 15, pt4 = lane >> 4;   // output o, a quarter of the hidden units each
        const int oc = min(o, no - 1);
        T mid = 0;
#pragma unroll
        for (int i = 0; i < 25; i++) mid = mid + w2s[oc + (pt4 * 25 + i) * no] * ar[pt4 * 25 + i];
        mid += __shfl_xor(mid, 16);
        mid += __shfl_xor(mid, 32);
        const T e = rexp(mid + b2s[oc]);
        if (lane < no) es[wave * 16 + lane] = e;
        wave_sync();
        const T p = e / tree_n(es + wave * 16, no);
        T lt_a = 0, lt_b = 0;   // this lane's class: the loss terms
        if (lane < no) {
            if (probs_g) probs_g[row * no + lane] = p;
            const T dv = (p - lab_g) / no;   // gcn_mbgd_fp64.cu:352
            d2s[wave * 16 + lane] = dv;
            d2_g[row * no + lane] = dv;
            loss_terms(p, lab_g, &lt_a, &lt_b);
        }
        wave_sync();
        const double loss = loss_sum_lanes(lt_a, lt_b, lab_g, no);
        if (lane == 0) loss_values[row] = loss;
        T dd[OMAX];   // entries past the class count are never read by the tree
#pragma unroll
        for (int oo = 0; oo < OMAX; oo++) dd[oo] = d2s[wave * 16 + oo];
        for (int n = lane; n < H; n += 64) {
            // bp_fully_connect(out -> hidden), gcn_mbgd_fp64.cu:416-434: class-wide tree
            T prod[OMAX];
#pragma unroll
            for (int oo = 0; oo < OMAX; oo++) prod[oo] = w2s[min(oo, no - 1) + n * no] * dd[oo];
            const T a = ar[n];
            const T dv = tree_n(prod, no) * (1 + a) * (1 - a);
            d1s[wave * L.hld + n] = dv;
            d1_g[row * H + n] = dv;
        }
    }
    lds_barrier();

    TT(6);
    // ---- delta at the pooled features: dp[s][p] = (sum_n W1[n + 100 p] d1[s][n]) (1 + pool)(1 - pool) ----
    // Rows of the MFMA tile = 16 pooled features (A = 16 rows of W1), columns = samples (B = rows of
    // d1).  Both operands are contiguous along n, so each lane fetches 16-byte vectors and the
    // product's k order is permuted to match: k-step s of lane group kq takes
    // n = VW kq + (s mod VW) + 4 VW (s / VW) - the same permutation on both sides.
    // dp replaces pool in LDS; dp x tanh' (the reference's second tanh' of the conv-bias path,
    // gcn_mbgd_fp64.cu:487-488; at a winner the conv output is the pooled value) goes to dsm.
    T *dsm = rb;   // [ms][pout] (the FC1 partial sums are dead)
    {
        V w[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int n = VW * kq + 4 * VW * g;
            w[g] = *reinterpret_cast<const V *>(d1s + min(col, ms - 1) * L.hld + min(n, H - VW));   // lanes past n = 100 meet a zero in a[]
        }
        auto product = [&](int pt, const V(&a)[NG]) {
            typename Mma<T>::acc_t acc = {0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < NG; g++)
#pragma unroll
                for (int j = 0; j < VW; j++) acc = Mma<T>::mma(a[g][j], w[g][j], acc);
            // lane holds sample col, features pt*16 + Mma::row(lane, 0..3)
            if (col < nb) {
                if (sizeof(T) == 4) {
                    // f32 result map: rows 4 kq .. 4 kq + 3 are consecutive features: one 16-byte access each way
                    const int p = pt * 16 + 4 * kq;
                    if (p < pout) {
                        V *pp = reinterpret_cast<V *>(pl + col * L.pld + p);
                        const V pv = *pp;
                        V o, o2;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const T t1 = (1 + pv[r % VW]) * (1 - pv[r % VW]);
                            o[r % VW] = acc[r] * t1;
                            o2[r % VW] = o[r % VW] * t1;
                        }
                        *pp = o;
                        *reinterpret_cast<V *>(dsm + col * pout + p) = o2;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int p = pt * 16 + Mma<T>::row(lane, r);
                        if (p < pout) {
                            const T pv = pl[col * L.pld + p];
                            const T t1 = (1 + pv) * (1 - pv);
                            const T o = acc[r] * t1;
                            pl[col * L.pld + p] = o;
                            dsm[col * pout + p] = o * t1;
                        }
                    }
                }
            }
        };
        // NPF tiles ahead: a slot's register set is refilled as soon as it has been multiplied, and
        // has the other slots' products to arrive
        for (int i = wave; i < ntp; i += NPF * nwv) {
#pragma unroll
            for (int h = 0; h < NPF; h++) {
                if (i + h * nwv < ntp) {
                    product(tile_of(i + h * nwv), wq[h]);
                    if (i + (h + NPF) * nwv < ntp) w1_tile(tile_of(i + (h + NPF) * nwv), wq[h]);
                }
            }
        }
    }
    lds_barrier();

    TT(7);
    // ---- conv-bias path: per sample and filter, the sum of dp x tanh' over the pooled positions ----
    if (mine) {
        const int f = lane / 3, third = lane - 3 * f;
        T s = 0;
        if (f < F)
            for (int m = third; m < mre; m += 3) s = s + dsm[wave * pout + f * mre + m];
        const T s1 = __shfl_down(s, 1), s2 = __shfl_down(s, 2);
        if (third == 0 && f < F) bs[wave * 32 + f] = s + s1 + s2;
    }
    lds_barrier();   // dsm is dead: its region becomes the delta images

    TT(8);
    // ---- bp_maxpooling scatter + conv gradient of the tile -----------------------------------------
    // dW0[f][tap] = sum over the tile's samples and positions t of delta[f][t] cube[18 t + tap]:
    // A = a sample's position-major delta tiles [4 kc][16] (two: filters 0..15 and 16..31), B = its
    // cube, 11 tap tiles.  Every wave builds the delta image of its own sample; an image covers kc
    // k-steps of positions at a time (nch chunks) where a whole one would not fit.
    T *out = part + (long)blockIdx.x * (F * K + F);
    if (ms > TILE_ACC_MS) {
        // Full tiles.  Filters 0..15 as 16 x 16 x 4 products against the 11 tap tiles; filters 16..19
        // (fp32) on the 16-block 4 x 4 x 1 instruction instead of a second 16-wide tile with twelve
        // empty rows: rows = the four filters, block b of tap group g = taps 64 g + 4 b .. + 3, one
        // position per instruction - lane l supplies delta[16 + l % 4][t] and cube[18 t + 64 g + l] and
        // receives dW0[16 + i][64 g + l] in register i.  Per four positions that is 11 x 32 + 12 x 8
        // matrix-pipe cycles instead of 22 x 32 (the conv phases run at the pipe's rate, so fewer
        // instructions is what shortens them).
        // Eight (or four) waves: every wave multiplies its own sample.  Sixteen waves: waves s and
        // s + 8 share sample s - they build its delta image together and split the work (tap tiles
        // 0..4 and tap groups 0, 1 | tap tiles 5..10 and tap group 2).  The samples' accumulators are
        // then added in sample order through LDS, four tap tiles and one tap group a round (the
        // round's buffer runs over the delta images into the cubes: both are dead).
        constexpr bool X4 = sizeof(T) == 4;         // filters 16..19 on the 4 x 4 x 1 instruction
        constexpr bool PAIR = NWV == 16;
        constexpr int NC0 = PAIR ? 6 : 11;          // tap tiles per wave
        constexpr int NXG = PAIR ? 2 : 3;           // tap groups per wave
        const int sw = PAIR ? (wave & 7) : wave, hf = PAIR ? (wave >> 3) : 0;
        const bool part = sw < nb;
        const int j0 = hf ? 5 : 0, g0 = hf ? 2 : 0;   // first tap tile and tap group of this wave
        typename Mma<T>::acc_t c0[NC0], c1[X4 ? 1 : NC0], xd[X4 ? NXG : 1];
#pragma unroll
        for (int j = 0; j < NC0; j++) c0[j] = typename Mma<T>::acc_t{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < (X4 ? 1 : NC0); j++) c1[j] = typename Mma<T>::acc_t{0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < (X4 ? NXG : 1); g++) xd[g] = typename Mma<T>::acc_t{0, 0, 0, 0};
        const int ks = (re + 3) >> 2;
        T *dl = rb + (part ? sw : 0) * 2 * L.ldd;
        const T *cube_s = base + L.cube + (part ? sw : 0) * L.cube_pad;
        const T *pls = pl + (part ? sw : 0) * L.pld;
        const unsigned char *pxs = px + (part ? sw : 0) * pout;
        constexpr int NBW = PAIR ? 2 : 1;           // waves that build one delta image
        for (int ch = 0; ch < nch; ch++) {
            const int t_lo = 4 * L.kc * ch;
            const int nks = min(L.kc, ks - L.kc * ch);
            if (nks <= 0) break;
            if (part)
                for (int e = lane + 64 * hf; e < 2 * L.ldd; e += 64 * NBW) dl[e] = 0;
            if (PAIR) lds_barrier();
            else wave_sync();
            if (part) {
                // consecutive lanes take consecutive FILTERS of one pooled position: their stores fall on
                // consecutive words of the position-major image (lanes that walk the positions of one
                // filter store 32 or 48 words apart - two banks), and the split is by a constant
                for (int i = lane + 64 * hf; i < pout; i += 64 * NBW) {
                    const int m = i / F, f = i - m * F, p = f * mre + m;
                    const int tt = 2 * m + (int)pxs[p] - t_lo;
                    if (tt >= 0 && tt < 4 * L.kc) dl[(f >> 4) * L.ldd + tt * LDWK + (f & 15)] = pls[p];
                }
            }
            if (PAIR) lds_barrier();
            else wave_sync();
            if (part) {
                const T *ap0 = dl + kq * LDWK + col, *ap1 = ap0 + L.ldd;
                const T *bp = cube_s + WSTEP * (t_lo + kq) + 16 * j0 + col;
                const T *xa = dl + L.ldd + (lane & 3);                  // delta[16 + lane % 4][.] of the image's second tile
                const T *xb = cube_s + WSTEP * t_lo + 64 * g0 + lane;   // cube[18 t + 64 g + lane]
                // NJ tap tiles and NG tap groups, compile-time: a count known only at run time would
                // put a branch around every matrix instruction
                auto products = [&](auto nj_c, auto ng_c) {
                    constexpr int NJ = decltype(nj_c)::value, NG = decltype(ng_c)::value;
                    for (int k = 0; k < nks; k++) {
                        const T a0 = ap0[4 * k * LDWK];
                        T a1 = 0;
                        if (!X4) a1 = ap1[4 * k * LDWK];
                        T b[NJ];
#pragma unroll
                        for (int j = 0; j < NJ; j++) b[j] = bp[4 * k * WSTEP + 16 * j];
                        T xav[4], xbv[4][NG];
                        if (X4) {
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const int tq = min(t_lo + 4 * k + q, re - 1) - t_lo;   // clamped: positions past the end are skipped below
                                xav[q] = xa[tq * LDWK];
#pragma unroll
                                for (int g = 0; g < NG; g++) xbv[q][g] = xb[tq * WSTEP + 64 * g];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < NJ; j++) {
                            c0[j] = Mma<T>::mma(a0, b[j], c0[j]);
                            if (!X4) c1[j] = Mma<T>::mma(a1, b[j], c1[j]);
                        }
                        if constexpr (X4) {
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                if (t_lo + 4 * k + q < re) {
#pragma unroll
                                    for (int g = 0; g < NG; g++)
                                        xd[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(xav[q], xbv[q][g], xd[g], 0, 0, 0);
                                }
                            }
                        }
                    }
                };
                if constexpr (!PAIR) {
                    products(std::integral_constant<int, 11>{}, std::integral_constant<int, 3>{});
                } else {
                    if (hf) products(std::integral_constant<int, 6>{}, std::integral_constant<int, 1>{});
                    else products(std::integral_constant<int, 5>{}, std::integral_constant<int, 2>{});
                }
            }
            if (ch + 1 < nch) {   // the image is zeroed again next round
                if (PAIR) lds_barrier();
                else wave_sync();
            }
        }
        lds_barrier();
        TT(9);
        // rounds: tap tiles 4 g .. 4 g + 3 of filters 0..15 (and of 16..31 without X4), tap group g of filters 16..19
#pragma unroll
        for (int g = 0; g < 3; g++) {
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = 4 * g + jj;
                if (j < 11) {
                    const int owner = PAIR && j >= 5 ? 1 : 0, lj = j - 5 * owner;   // compile-time after unrolling
                    if (hf == owner) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            rb[sw * TILE_RED + (jj * 2 + 0) * 256 + r * 64 + lane] = c0[lj][r];
                            if (!X4) rb[sw * TILE_RED + (jj * 2 + 1) * 256 + r * 64 + lane] = c1[X4 ? 0 : lj][r];
                        }
                    }
                }
            }
            if (X4) {
                const int owner = PAIR && g == 2 ? 1 : 0, lg = PAIR ? (g == 2 ? 0 : g) : g;
                if (hf == owner) {
#pragma unroll
                    for (int i = 0; i < 4; i++) rb[sw * TILE_RED + 256 + i * 64 + lane] = xd[X4 ? lg : 0][i];   // the slot of tile jj = 0, filter tile 1
                }
            }
            lds_barrier();
            for (int e = tid; e < TILE_RED; e += nt) {
                const int jj = e >> 9, ft = (e >> 8) & 1, r = (e >> 6) & 3, ln = e & 63;
                const int j = 4 * g + jj;
                int tap = 16 * j + (ln & 15), f = 16 * ft + Mma<T>::row(ln, r);
                bool live = j < 11 && tap < K && f < F;
                if (X4 && ft == 1) {   // only slot (jj = 0, filter tile 1) is in use: filter 16 + r, tap 64 g + ln
                    tap = 64 * g + ln;
                    f = 16 + r;
                    live = jj == 0 && tap < K;
                }
                if (live) {
                    T s = 0;
#pragma unroll
                    for (int w = 0; w < NS; w++) s = s + rb[w * TILE_RED + e];
                    out[f * K + tap] = s / re;   // gcn_mbgd_fp64.cu:491
                }
            }
            if (g < 2) lds_barrier();
        }
    } else {
        // Small tiles (fewer samples than waves): the 22 (tap tile, filter tile) products are dealt to
        // the waves and each wave runs its products over ALL the tile's samples, so the sum over the
        // samples happens in the accumulators and every wave stores its part of the result itself.
        constexpr int NJ = (22 + nwv - 1) / nwv;   // products per wave: ids wave, wave + nwv, .. (same filter tile: nwv is even)
        const int ft = wave & 1;
        typename Mma<T>::acc_t c[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) c[jj] = typename Mma<T>::acc_t{0, 0, 0, 0};
        int joff[NJ];   // cube offset of the j-th product's tap tile (a valid one where the wave has no j-th product)
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) joff[jj] = 16 * (wave + jj * nwv < 22 ? (wave + jj * nwv) >> 1 : 0);
        const int ks = (re + 3) >> 2;
        T *dl = rb + wave * 2 * L.ldd;
        for (int ch = 0; ch < nch; ch++) {
            const int t_lo = 4 * L.kc * ch;
            const int nks = min(L.kc, ks - L.kc * ch);
            if (nks <= 0) break;
            if (mine) {
                for (int e = lane; e < 2 * L.ldd; e += 64) dl[e] = 0;
                wave_sync();
                for (int i = lane; i < pout; i += 64) {   // consecutive lanes: consecutive filters of one pooled position (as above)
                    const int m = i / F, f = i - m * F, p = f * mre + m;
                    const int tt = 2 * m + (int)pxr[p] - t_lo;
                    if (tt >= 0 && tt < 4 * L.kc) dl[(f >> 4) * L.ldd + tt * LDWK + (f & 15)] = plr[p];
                }
            }
            lds_barrier();   // every sample's image is complete
            for (int sidx = 0; sidx < nb; sidx++) {
                const T *ap = rb + sidx * 2 * L.ldd + ft * L.ldd + kq * LDWK + col;
                const T *bp = base + L.cube + sidx * L.cube_pad + WSTEP * (t_lo + kq) + col;
                for (int k0 = 0; k0 < nks; k0 += 4) {   // operands of four k-steps, then their MFMAs
                    T a[4], bv[4][NJ];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int kk = min(k0 + u, nks - 1);   // clamped, unconditional
                        a[u] = ap[4 * kk * LDWK];
#pragma unroll
                        for (int jj = 0; jj < NJ; jj++) bv[u][jj] = bp[4 * kk * WSTEP + joff[jj]];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (k0 + u < nks) {
#pragma unroll
                            for (int jj = 0; jj < NJ; jj++)
                                if (wave + jj * nwv < 22) c[jj] = Mma<T>::mma(a[u], bv[u][jj], c[jj]);
                        }
                    }
                }
            }
            if (ch + 1 < nch) lds_barrier();   // the images are zeroed again next round
        }
        TT(9);
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) {
            const int id = wave + jj * nwv;
            const int tap = 16 * (id >> 1) + col;
            if (id < 22 && tap < K) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int f = 16 * ft + Mma<T>::row(lane, r);
                    if (f < F) out[f * K + tap] = c[jj][r] / re;   // gcn_mbgd_fp64.cu:491
                }
            }
        }
    }
    if (tid < F) {
        T s = 0;
        for (int w = 0; w < nb; w++) s = s + bs[w * 32 + tid];
        out[F * K + tid] = s / re;
    }
    TT(10);
    TWG(1 + (threadIdx.x >> 6));
}

template <typename T, bool FLIP>
int tile_train(const T *params, const T *cubes, const int *onehot, long first, int B, const ccnn_dims_t &d,
               const TilePlan &plan, T *pool, T *a1, T *d1, T *d2, T *probs, double *loss_values, T *part,
               hipStream_t st)
{
    const bool c13 = d.n_classes == CCNN_CLASSES;
    auto kern = plan.nwv == 8 ? (c13 ? tile_fwd_bwd<T, FLIP, CCNN_CLASSES, 8> : tile_fwd_bwd<T, FLIP, 0, 8>)
                              : (c13 ? tile_fwd_bwd<T, FLIP, CCNN_CLASSES, 4> : tile_fwd_bwd<T, FLIP, 0, 4>);
    if constexpr (sizeof(T) == 4) {   // (sixteen waves leave 128 registers each: not enough for the fp64 accumulators)
        if (plan.nwv == 16) kern = c13 ? tile_fwd_bwd<T, FLIP, CCNN_CLASSES, 16> : tile_fwd_bwd<T, FLIP, 0, 16>;
    }
    if (int e = allow_smem(kern, plan.smem)) return e;
    launch("tile_fwd_bwd", kern, dim3(plan.tiles), dim3(64 * plan.nwv), plan.smem, st, params, cubes, onehot, first, B,
           plan.ms, plan.nch, d, pool, a1, d1, d2, probs, loss_values, part);
    CCNN_LAUNCH_CHECK();
    return 0;
}
template int tile_train<float, true>(const float *, const float *, const int *, long, int, const ccnn_dims_t &,
                                     const TilePlan &, float *, float *, float *, float *, float *, double *, float *,
                                     hipStream_t);
template int tile_train<double, false>(const double *, const double *, const int *, long, int, const ccnn_dims_t &,
                                       const TilePlan &, double *, double *, double *, double *, double *, double *,
                                       double *, hipStream_t);

}  // namespace ccnn
