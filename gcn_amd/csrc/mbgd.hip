// Minibatch gradient (and fused update) of the Cube CNN in two launches per step.
//
// The reference runs 9 launches per sample on 100 streams and keeps per-sample
// gradient slabs [B x params] (gcn_mbgd_fp64.cu:931-1064).  Here:
//
//   sample_fwd_bwd  one workgroup per sample - two, with ten filters each, while the whole
//                   launch fits the GPU at once - does the whole per-sample chain with
//                   the cube resident in LDS: conv (MFMA) -> tanh -> pool -> FC1 ->
//                   FC2 -> softmax/loss -> delta2 -> delta1 -> delta at the pooled
//                   features -> bp_maxpooling scatter -> conv gradient (MFMA).
//                   It leaves pool, a1, delta1, delta2 and its conv-gradient tile
//                   [F*K + F] in the workspace.
//   grad_reduce     sums over the batch in the flat parameter layout:
//                   dW1 = pool^T delta1 and dW2 = a1^T delta2 as MFMA products with
//                   K = batch (a 16 x 16 tile per workgroup), the rest as column sums;
//                   optionally applies the update in the same pass.
//
// Only what differs between samples is per-sample; nothing of size [B x |W1|] exists.
// A launch boundary costs about as much as one of these kernels' whole run, so the
// per-sample chain is fused rather than batched layer by layer; FC1 (100 x Pout) is a
// matrix-vector product per sample here, read through L2.
//
// That is the step at the reference's batch sizes.  Batches of several hundred samples and more
// take tile_fwd_bwd (tile.hip: a tile of samples per workgroup, FC1 as a matrix product that
// reads W1 once per tile) or, where no tile fits the LDS, the layer kernels below
// (conv_pool_fwd<TRAIN> + fc_head_fwd<TRAIN> of forward.hip, dpool_bwd, conv_bwd_batch); grad_reduce
// sums the batch behind each of them; step_path() chooses.
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "ccnn_device.h"
#include "comm.h"
#include "profile.h"
#include "train_fwd.h"

// Diagnostic builds only (-DCCNN_FB_TIMING): workgroup 5 of sample_fwd_bwd stamps the wall clock
// (100 MHz) at FB_TICK(k); ccnn_debug_fb_ticks() returns the stamps of the last launch.
#ifdef CCNN_FB_TIMING
__device__ unsigned long long g_fb_ticks[32];
#define FB_TICK(k)                                                             \
    do {                                                                       \
        if (blockIdx.x == 5 && threadIdx.x == 0) g_fb_ticks[k] = wall_clock64(); \
    } while (0)
// ... and every workgroup its first instruction and the end of each of its waves ([block][0], [block][1 + wave])
__device__ unsigned long long g_fb_wg[1024][9];
#define FB_WG(slot)                                                                                      \
    do {                                                                                                 \
        if (blockIdx.x < 1024 && (threadIdx.x & 63) == 0) g_fb_wg[blockIdx.x][slot] = wall_clock64();    \
    } while (0)
// ... and every workgroup of grad_reduce its first instruction, the end of its sums and its end
__device__ unsigned long long g_gr_wg[1024][3];
#define GR_WG(slot)                                                                                   \
    do {                                                                                              \
        if (lb < 1024 && threadIdx.x == 0) g_gr_wg[lb][slot] = wall_clock64();                        \
    } while (0)
#else
#define FB_TICK(k) do { } while (0)
#define FB_WG(slot) do { } while (0)
#define GR_WG(slot) do { } while (0)
#endif

#ifndef CCNN_FB_UFF
#define CCNN_FB_UFF 16   // W1 rows (16-byte vectors) a thread keeps in flight in the forward walk
#endif
#ifndef CCNN_FB_UFB
#define CCNN_FB_UFB 1    // rounds of 16 W1 rows in flight in the backward walk
#endif
#ifndef CCNN_FB_PREW
#define CCNN_FB_PREW 1   // the backward walk's first round of W1 rows is fetched before the head
#endif
#ifndef CCNN_FB_PREW2
#define CCNN_FB_PREW2 1  // ... and a second round in fp32
#endif
#ifndef CCNN_STOP_PHASE
#define CCNN_STOP_PHASE 0   // diagnostic builds only: end sample_fwd_bwd after phase N
#endif
// The forward convolution of sample_fwd_bwd on the packed filter image (filters 16..19 as one chain of
// 57 k-steps per 64 positions instead of a second 16-wide tile with twelve empty columns).
#ifndef CCNN_FB_PACKED
#define CCNN_FB_PACKED(T) (sizeof(T) == 8)
#endif

namespace ccnn {

__host__ __device__ inline int up4(int v) { return (v + 3) & ~3; }

// LDS carve of sample_fwd_bwd, in elements of T (every region a multiple of 4
// elements so 16-byte accesses stay aligned), then the int region.
struct FusedLayout {
    int cube, wl, a0, pl, dp, a1, d1, es, ps, d2, total;   // element offsets / count
    int ldd, ks;
    size_t px_off, lab_off, bytes;
};

template <typename T>
__host__ __device__ inline FusedLayout fused_layout(int re, int mre)
{
    FusedLayout L;
    const int pout = F * mre;
    L.ks = (re + 3) >> 2;
    L.ldd = 4 * L.ks * LDWK;   // one position-major delta tile [4 ks][16 filters]
    const int wl_len = 2 * L.ldd > WP_LEN ? 2 * L.ldd : WP_LEN;   // filters (tile of 16 + packed tile), later the scattered deltas
    const int a0_len = 32 * H;   // FC1 partial sums [G][100], G <= 32 row groups
    int o = 0;
    L.cube = o; o += up4(conv_cube_pad(re));
    L.wl = o;   o += up4(wl_len);
    L.a0 = o;   o += up4(a0_len);
    L.pl = o;   o += up4(pout);
    L.dp = o;   o += up4(pout);
    L.a1 = o;   o += up4(H);
    L.d1 = o;   o += up4(H);
    L.es = o;   o += 16;
    L.ps = o;   o += 16;
    L.d2 = o;   o += 16;
    L.total = o;
    L.px_off = (size_t)o * sizeof(T);
    L.lab_off = L.px_off + (size_t)up4(pout) * sizeof(int);   // 16 ints: the sample's one-hot labels
    L.bytes = L.lab_off + 16 * sizeof(int);
    return L;
}

// Conv weight gradient of one sample as an MFMA product:
// dW0[f][tap] = sum_t delta[f][t] cube[18 t + tap] / re  ([20 x re] x [re x 171]),
// 2 filter tiles x 11 tap tiles, wave w of NW owns tap tiles w, w+NW, ..
// delta is two position-major tiles [4 ks][16] (tile_len apart; filters 20..31 and positions
// >= re zero) so that, like the forward filter tiles, A-fragment reads are bank-conflict free;
// cube zero-padded; out = [F][K].  ONE: only the first delta tile exists and its columns are filters
// fbase .. fbase + nf - 1 (a workgroup that owns half of the filters).
template <typename T, int NW, bool ONE = false>
__device__ __forceinline__ void conv_bwd_tiles(const T *cube, const T *delta, int tile_len, int ks, int re,
                                               T *__restrict__ out, int lane, int wave, int fbase = 0, int nf = F)
{
    constexpr int NTT = (11 + NW - 1) / NW;   // tap tiles per wave: wave w owns w, w+NW, ..  (171 taps = 11 tiles)
    const int col = lane & 15, kq = lane >> 4;
    const T *ap0 = delta + kq * LDWK + col;
    const T *ap1 = ap0 + tile_len;
    const T *bp[NTT];
    bool live[NTT];
#pragma unroll
    for (int j = 0; j < NTT; j++) {
        live[j] = wave + j * NW < 11;
        bp[j] = cube + WSTEP * kq + (live[j] ? wave + j * NW : wave) * 16 + col;
    }
    typename Mma<T>::acc_t c0[NTT], c1[NTT];
#pragma unroll
    for (int j = 0; j < NTT; j++) c0[j] = c1[j] = typename Mma<T>::acc_t{0, 0, 0, 0};
    auto kstep2 = [&](int s, int n) {   // n (1 or 2) k-steps: all operands first, then the MFMAs
        T a0[2], a1[2], b[2][NTT];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (u < n) {
                a0[u] = ap0[4 * (s + u) * LDWK];
                if (!ONE) a1[u] = ap1[4 * (s + u) * LDWK];
#pragma unroll
                for (int j = 0; j < NTT; j++) b[u][j] = bp[j][4 * (s + u) * WSTEP];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (u < n) {
#pragma unroll
                for (int j = 0; j < NTT; j++) {
                    if (live[j]) {
                        c0[j] = Mma<T>::mma(a0[u], b[u][j], c0[j]);
                        if (!ONE) c1[j] = Mma<T>::mma(a1[u], b[u][j], c1[j]);
                    }
                }
            }
        }
    };
    int s = 0;
    for (; s + 2 <= ks; s += 2) kstep2(s, 2);
    if (s < ks) kstep2(s, 1);
    auto emit = [&](const typename Mma<T>::acc_t &acc, int ftile, int ttile) {
        const int tap = ttile * 16 + col;
        if (tap >= K) return;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = ftile * 16 + Mma<T>::row(lane, r);
            if (f < nf) out[(fbase + f) * K + tap] = acc[r] / re;   // gcn_mbgd_fp64.cu:491
        }
    };
#pragma unroll
    for (int j = 0; j < NTT; j++) {
        if (live[j]) {
            emit(c0[j], 0, wave + j * NW);
            if (!ONE) emit(c1[j], 1, wave + j * NW);
        }
    }
}

// ---------------------------------------------------------------------------
// Trade of one value per thread between the two workgroups of a sample (sample_fwd_bwd, FS): every
// 32 bits of the value travel in an 8-byte word together with the launch's tag, so the reader knows
// from the word itself that it is this launch's (no fence, no flag, no ordering between stores); the
// accesses are relaxed agent-scope atomics (performed at the L2 / memory side, never served from a
// CU's cache).  The reader clears the words it has consumed - the partner wrote them before the
// reader could see them and does not touch them again in this launch - so a launch always finds the
// words of the launch before it empty, whatever its tag (a captured launch replays with the tag it
// was captured with).  `tag` is never 0.  The wait is bounded (2 s of the 100 MHz clock; a partner
// is dispatched right behind its workgroup): when it runs out the sum becomes a NaN, which reaches
// the loss and every parameter - a step that went wrong cannot look like one that did not.
constexpr unsigned long long PAIR_WAIT_TICKS = 200000000ull;
template <typename T>
struct PairWords {
    static constexpr int N = sizeof(T) / 4;   // tagged words per value
};
__device__ __forceinline__ unsigned long long pair_word(unsigned bits, unsigned tag)
{
    return ((unsigned long long)bits << 32) | tag;
}
// Publishes `own` at out[0 .. N), takes the partner's value from in[0 .. N) and returns the sum of
// the two in the order (first workgroup's) + (second workgroup's): the same bits in both.
__device__ __forceinline__ float pair_sum(float own, unsigned long long *out, unsigned long long *in, unsigned tag, bool first,
                                          bool publish = true)
{
    if (publish) __hip_atomic_store(out, pair_word(__float_as_uint(own), tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    unsigned long long v;
    unsigned spins = 0;
    while ((unsigned)(v = __hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != tag) {
        if ((++spins & 255) == 0 && wall_clock64() - t0 > PAIR_WAIT_TICKS) {
            v = pair_word(0x7fc00000u, tag);
            break;
        }
    }
    __hip_atomic_store(in, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float other = __uint_as_float((unsigned)(v >> 32));
    return first ? own + other : other + own;
}
__device__ __forceinline__ double pair_sum(double own, unsigned long long *out, unsigned long long *in, unsigned tag, bool first,
                                           bool publish = true)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(own);
    if (publish) {
        __hip_atomic_store(out, pair_word((unsigned)(bits >> 32), tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(out + 1, pair_word((unsigned)bits, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t0 = wall_clock64();
    unsigned long long hi, lo;
    unsigned spins = 0;
    while (true) {
        hi = __hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lo = __hip_atomic_load(in + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)hi == tag && (unsigned)lo == tag) break;
        if ((++spins & 255) == 0 && wall_clock64() - t0 > PAIR_WAIT_TICKS) {
            hi = pair_word(0x7ff80000u, tag);
            lo = pair_word(0u, tag);
            break;
        }
    }
    __hip_atomic_store(in, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(in + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double other = __longlong_as_double((long long)((hi >> 32 << 32) | (lo >> 32)));
    return first ? own + other : other + own;
}

// ---------------------------------------------------------------------------
// NT threads per workgroup (one workgroup per sample, one per CU): the per-sample chain is
// latency-bound, and its two walks over W1 are limited by the bytes a CU keeps in flight, so
// more waves shorten them even though most other phases cannot use the extra threads.
// NO: the class count when it is the reference's 13 (a compile-time constant then: the head's
// loops and its 13-wide tree cost what they cost before the count became a run-time property),
// 0 for any other count (read from d).
// SPLIT = 2 (small batches, which leave half of the chip idle): two workgroups per sample, on one
// XCD (blocks i and i + 8), each with the backward pass of ten filters: its half of the second W1
// walk, its filters' conv gradient.  FS = false: both run the whole forward pass, no workgroup waits
// for another, and every value is computed exactly as with one workgroup.  FS = true (the host asks
// for it when all workgroups of the launch fit the chip at once): the forward pass is split as well -
// a workgroup convolves and pools its ten filters only and walks only their rows of W1; the hundred
// partial sums of the hidden layer are traded through `xch` (pair_sum: 0.6-0.8 us), after which both
// hold the same z1 bit for bit and carry on as before.  The forward walk is bound by what the L2s
// deliver to all workgroups together, so half the rows are worth 0.9 us of 2.9 at 103 bands in fp32
// and 3.0 of 8.4 at 176 bands in fp64, the conv phase another 0.5 - 2 us.
// WPE: waves per SIMD the register allocation must leave room for (2: one workgroup per CU, all
// the registers the chain wants; 4: two workgroups per CU, for batches of more samples than CUs,
// whose second round then runs beside the first instead of after it).
template <typename T, bool FLIP, int NT, int NO, int SPLIT, int WPE = 2, bool FS = false>
__global__ __launch_bounds__(NT, WPE) void sample_fwd_bwd(const T *__restrict__ params,
                                                      const T *__restrict__ cubes,
                                                      const int *__restrict__ onehot, long first,
                                                      ccnn_dims_t d, T *__restrict__ pool_g,
                                                      T *__restrict__ a1_g, T *__restrict__ d1_g,
                                                      T *__restrict__ d2_g, T *__restrict__ probs_g,
                                                      double *__restrict__ loss_values,
                                                      T *__restrict__ slab, int d_rows,
                                                      unsigned long long *__restrict__ xch, unsigned xtag, int xmute)
{
    static_assert(!FS || SPLIT == 2, "the forward pass splits over the two workgroups of a sample");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool PACKED = CCNN_FB_PACKED(T) && !FS;   // a half's ten filters are one plain tile
    constexpr int NFO = FS ? F / 2 : F;                // filters this workgroup convolves
    const FusedLayout L = fused_layout<T>(d.re, d.mre);
    T *base = reinterpret_cast<T *>(smem);
    T *cube = base + L.cube, *wl = base + L.wl, *a0 = base + L.a0, *pl = base + L.pl, *dp = base + L.dp;
    T *a1s = base + L.a1, *d1s = base + L.d1, *es = base + L.es, *ps = base + L.ps, *d2s = base + L.d2;
    int *px = reinterpret_cast<int *>(smem + L.px_off);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15;
    const int re = d.re, mre = d.mre, pout = d.pout, no = NO ? NO : d.n_classes;
    const int half = SPLIT == 2 ? (blockIdx.x >> 3) & 1 : 0;
    const long row = SPLIT == 2 ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 4) : blockIdx.x;
    if (SPLIT == 2 && row >= d_rows) return;   // the grid is rounded up to whole groups of 8 samples
    const bool lead = half == 0;               // the workgroup that stores the sample's shared outputs
    FB_TICK(0);
    if (tid == 0) FB_WG(0);
    int *lab = reinterpret_cast<int *>(smem + L.lab_off);
    // the labels are needed deep in the dependent chain: fetch them with the first loads
    const int lab_g = onehot[(first + row) * no + min(tid, no - 1)];

    // ---- forward: conv + tanh + pool ------------------------------------------------
    // Issue every global load this phase needs (filters, cube, and the small output-layer
    // operands used much later) before touching LDS, so one memory round trip covers them.
    constexpr int NW = NT / 64;
    static_assert(NW >= 2, "the conv-bias path and the loss have a wave each");
    constexpr int NCREG = (NB * CCNN_MAX_BANDS + NT - 1) / NT;   // the cube comes from HBM, cold: first in the queue
    T creg[NCREG];
    {
        const T *src = cubes + (first + row) * (long)d.cube;
#pragma unroll
        for (int u = 0; u < NCREG; u++) creg[u] = src[min(u * NT + tid, d.cube - 1)];   // clamped, unconditional
    }
    // The conv filters.  PACKED: as conv_pool_chain wants them (ccnn_device.h: filters 0..15 tap-major,
    // then the packed tile of filters 16..19) - thread t owns elements t, t + NT, .. of the IMAGE and
    // fetches the filter value (or a zero) that belongs there, so there is no zeroing pass and no
    // barrier between one and a scatter.  Otherwise: two plain tap-major tiles, every thread fetches
    // filter elements and scatters them behind a zeroing pass and a barrier.  (The fp32 reference walks
    // the filter backwards in the forward pass only, gcn_mbgd_fp32.cu:241-243.)
    static_assert(LDWK == 16, "the image element -> (tap, filter) split below");
    constexpr int NWREG = PACKED ? (WP_LEN + NT - 1) / NT : (NFO * K + NT - 1) / NT;
    const int fown = FS ? half * NFO : 0;   // the first of them
    T wreg[NWREG];
    unsigned wok = 0;   // PACKED, bit u: element u holds a filter value (else a zero)
#pragma unroll
    for (int u = 0; u < NWREG; u++) {
        const int e = u * NT + tid;
        if constexpr (PACKED) {
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
            wreg[u] = params[d.off_w0 + (ok ? f * K + (FLIP ? K - 1 - kk : kk) : 0)];   // clamped: loads stay unconditional
            wok |= (unsigned)ok << u;
        } else {
            wreg[u] = params[d.off_w0 + fown * K + (e < NFO * K ? e : NFO * K - 1)];   // clamped: loads stay unconditional
        }
    }
    const T *W2 = params + d.off_w2;
    constexpr int a0_half = 16 * H;          // the FC1 partial sums [G][100] use at most 16 row groups of the 32 the region holds
    const int o8 = tid >> 3, j8 = tid & 7;   // output layer: one output per 8 lanes, 8 partial sums each
    constexpr int NP8 = (H + 7) / 8;         // products per partial sum
    // The output layer's weights go through LDS (the upper half of the FC1 partial-sum region, which
    // the partial sums never reach): three loads per thread.  Held in registers by the threads that
    // use them they were 29 loads per thread for ALL threads, and a load instruction costs a wave its
    // slot in the memory pipeline whatever its lanes carry - 0.35 us of the prologue's issue time.
    constexpr int NW2 = (H * OMAX + NT - 1) / NT;
    T w2r[NW2];
#pragma unroll
    for (int u = 0; u < NW2; u++) w2r[u] = W2[min(u * NT + tid, H * no - 1)];
    T *w2s = a0 + a0_half;
    const T b1v = params[d.off_b1 + min(tid, H - 1)];
    const T b2v = params[d.off_b2 + min(o8, no - 1)];
    const T bias0 = params[d.off_b0 + (FS ? fown + min(col, NFO - 1) : col)];
    const T biasx = params[d.off_b0 + (PACKED ? 16 + (col & 3) : min(16 + col, F - 1))];   // second tile's columns
    // The two walks over W1 (FC1 forward, delta at the pooled features): 16-byte vectors along n; QL
    // lanes per W1 row (25 or 50 of them active), G row groups.
    typedef typename Vec16<T>::type V;
    constexpr int VW = Vec16<T>::W;
    constexpr int NQ = H / VW;
    constexpr int QL = NQ <= 32 ? 32 : 64;
    constexpr int G = NT / QL;
    static_assert(G * H <= a0_half && a0_half + H * OMAX <= 32 * H, "FC1 partial sums and the output layer's weights share a0");
    const int q = tid & (QL - 1), g = tid / QL;
    const bool qv = q < NQ;
    const int qc = qv ? q : NQ - 1;
    const V *W1v = reinterpret_cast<const V *>(params + d.off_w1);
    T *z1p = a0;   // [G][H] partial sums (the conv outputs are dead)
    // Every workgroup reads all of W1; start each at a different row so that at any moment
    // the 128 workgroups pull from different L2 channels instead of queueing on the same lines.
    const int rot = (int)((row * 37) % pout);
    constexpr int UF = 16;    // W1 rows per reduce-scatter round of the backward walk
    constexpr int UFF = CCNN_FB_UFF;   // W1 rows in flight per thread in the forward walk
    // Both walks visit the rows in the same order: (g + rot) mod pout, then steps of G (mod pout).
    // A running row number costs three instructions per row; rows past the end wrap onto valid
    // ones, so the loads stay unconditional without a clamp.
    unsigned wbase = 0, wlen = (unsigned)pout;   // the rows a walk covers: wbase + (0 .. wlen - 1)
    if (FS) {
        wlen = (unsigned)(pout / 2);
        wbase = half * wlen;
    }
    unsigned p_run = (unsigned)((g + rot % (int)wlen) % (int)wlen);
    auto next_row = [&]() {
        const unsigned p = p_run, t = p_run + G;
        p_run = min(t, t - wlen);   // t < wlen: the difference wraps to a huge value
        return wbase + p;
    };
    auto w1_rows = [&](V(&w)[UF]) {
#pragma unroll
        for (int u = 0; u < UF; u++) w[u] = W1v[next_row() * NQ + qc];
    };
    // The first round of W1 rows (UF per thread: at 103 bands a half's 210 rows are ONE round) depends
    // on nothing computed here: it is fetched now, behind the cube and the filters, arrives under the
    // staging and the conv phase, and stays in registers for BOTH walks where they cover the same rows
    // in the same order (always, except with two workgroups per sample that both run the whole forward
    // pass).  Only where a workgroup has a CU's registers to itself; every barrier from here to the
    // backward walk orders LDS only, so that none of them waits for these loads.
    constexpr bool PREW = WPE == 2 && CCNN_FB_PREW;
    constexpr bool SAME_WALK = FS || SPLIT == 1;
    static_assert(!PREW || UFF == UF, "the prefetched round serves both walks");
    const unsigned p_start = p_run;
    V wk[UF];
    // 16-byte loads go through the CU's address path at 64 B/clk: eight waves' rounds are 0.9 us of it,
    // and a wave that cannot issue does nothing else.  The waves with a conv job (one per position
    // tile) fetch now, under the latency of the cube; the others at the start of the conv phase,
    // which they would spend waiting.
    // fp32 keeps a second round where the walk has one (from 128 bands with two workgroups per
    // sample): 128 of a thread's 256 registers then hold W1.
    constexpr bool PREW2 = PREW && SAME_WALK && sizeof(T) == 4 && CCNN_FB_PREW2;
    V wk2[UF];
    const bool two = PREW2 && (int)wlen > G * UF;
    const int conv_waves = FS ? conv_mtiles(re) : NW;
    if constexpr (PREW) {
        if (wave < conv_waves) {
            w1_rows(wk);
            if constexpr (PREW2)
                if (two) w1_rows(wk2);
        }
    }
    auto sync = [&]() {
        if constexpr (PREW) lds_barrier();
        else __syncthreads();
    };
    FB_TICK(11);
    for (int e = d.cube + tid; e < conv_cube_pad(re); e += NT) cube[e] = 0;
    if constexpr (FS) {
        if (tid < LDWK) wl[WL_TILE - LDWK + tid] = 0;   // tap 171 (columns 10..15 of the tile: see conv_pool_own)
    } else if constexpr (!PACKED) {
        // zeros the filter scatter below does not overwrite: the second tile (filters 20..31) and tap 171 of the first
        for (int e = WL_TILE - LDWK + tid; e < WL_LEN; e += NT) wl[e] = 0;
    }
    FB_TICK(12);
    if (tid < no) lab[tid] = lab_g;   // the first use of a loaded value: after the zeroing, which needs none
#pragma unroll
    for (int u = 0; u < NW2; u++)
        if (u * NT + tid < H * no) w2s[u * NT + tid] = w2r[u];
    FB_TICK(13);
#pragma unroll
    for (int u = 0; u < NCREG; u++) {
        const int e = u * NT + tid;
        if (e < d.cube) cube[e] = creg[u];
    }
    FB_TICK(14);
    if constexpr (PACKED) {
#pragma unroll
        for (int u = 0; u < NWREG; u++) {
            const int e = u * NT + tid;
            if (e < WP_LEN) wl[e] = (wok >> u & 1) ? wreg[u] : T(0);
        }
        FB_TICK(15);
    } else {
        if constexpr (!FS) sync();   // FS: the zeroed row (tap 171) and the scattered taps are disjoint
        FB_TICK(15);
#pragma unroll
        for (int u = 0; u < NWREG; u++) {
            const int e = u * NT + tid;
            if (e < NFO * K) {
                const int f = e / K, k = e - f * K;   // FS: column f of the one tile holds filter fown + f
                wl[wl_index(f, FLIP ? K - 1 - k : k)] = wreg[u];
            }
        }
    }
    sync();
    FB_TICK(1);
#if CCNN_STOP_PHASE == 1
    return;
#endif
    if constexpr (PREW) {
        if (wave >= conv_waves) {
            w1_rows(wk);
            if constexpr (PREW2)
                if (two) w1_rows(wk2);
        }
    }
    if constexpr (FS) {
        conv_pool_own<T>(cube, wl, re, bias0, fown, NFO, lane, wave, NW, [&](int f, int m, T v, int bt) {
            pl[f * mre + m] = v;
            px[f * mre + m] = bt;
        });
    } else if constexpr (PACKED) {
        // one chain per wave: the position tiles of filters 0..15, then the packed tile's chains
        for (int job = wave; job < conv_chains(re); job += NW) {
            conv_pool_chain<T>(cube, wl, re, job, bias0, biasx, lane, [&](int f, int m, T v, int bt) {
                pl[f * mre + m] = v;
                px[f * mre + m] = bt;
            });
        }
    } else {
        conv_pool_tiles<T>(cube, 0, 1, wl, re, bias0, biasx, lane, wave, [&](int, int f, int m, T v, int bt) {
            pl[f * mre + m] = v;
            px[f * mre + m] = bt;
        }, NW, NW > 4);
    }
    sync();
    FB_TICK(2);
#if CCNN_STOP_PHASE == 2
    return;
#endif
    if constexpr (FS) {
        const int o0 = half * (pout / 2);   // F is even: a half is ten whole filters
        for (int o = tid; o < pout / 2; o += NT) pool_g[row * pout + o0 + o] = pl[o0 + o];
    } else if (lead) {
        for (int o = tid; o < pout; o += NT) pool_g[row * pout + o] = pl[o];
    }

    // ---- FC1 forward: z1[n] = sum_p pool[p] W1[n + 100 p] ---------------------------
    {
        V acc = {};
        p_run = p_start;
        auto fc1_round = [&](int pb, const V(&w)[UFF], const unsigned(&prow)[UFF]) {
#pragma unroll
            for (int u = 0; u < UFF; u++) {
                const T x = (qv && pb + u * G + g < (int)wlen) ? pl[prow[u]] : T(0);
                acc += w[u] * x;
            }
        };
        int pb0 = 0;
        if constexpr (PREW) {
            unsigned prow[UFF];
#pragma unroll
            for (int u = 0; u < UFF; u++) prow[u] = next_row();
            fc1_round(0, wk, prow);
            pb0 = G * UFF;
            if constexpr (PREW2) {
                if (two) {
#pragma unroll
                    for (int u = 0; u < UFF; u++) prow[u] = next_row();
                    fc1_round(pb0, wk2, prow);
                    pb0 += G * UFF;
                }
            }
        }
        for (int pb = pb0; pb < (int)wlen; pb += G * UFF) {
            V w[UFF];
            unsigned prow[UFF];
#pragma unroll
            for (int u = 0; u < UFF; u++) {
                prow[u] = next_row();
                w[u] = W1v[prow[u] * NQ + qc];
            }
            fc1_round(pb, w, prow);
        }
        if (qv) reinterpret_cast<V *>(z1p + g * H)[q] = acc;
    }
    // The backward walk (delta at the pooled features, below): this workgroup's rows of W1.
    if (SPLIT == 2) {
        wlen = (unsigned)(pout / 2);   // F is even: a half is ten whole filters
        wbase = half * wlen;
    }
    const int wrot = rot % (int)wlen;
    p_run = (unsigned)((g + wrot) % (int)wlen);
    if constexpr (PREW && !SAME_WALK) w1_rows(wk);   // other rows than the forward walk's: fetched under the head
    sync();
    FB_TICK(3);
#if CCNN_STOP_PHASE == 3
    return;
#endif
    T a = 0;
    if (tid < H) {
        T zsum = 0;
#pragma unroll
        for (int gg = 0; gg < G; gg++) zsum = zsum + z1p[gg * H + tid];
        if constexpr (FS) {
            // zsum covers this workgroup's ten filters: add the partner's
            constexpr int XW = PairWords<T>::N;
            unsigned long long *xrow = xch + (size_t)row * (2 * H * XW);
            // (xmute: the test of the bounded wait - the second workgroup of every pair publishes nothing)
            zsum = pair_sum(zsum, xrow + (half * H + tid) * XW, xrow + ((half ^ 1) * H + tid) * XW, xtag, lead,
                            !(xmute && half == 1));
        }
        a = act(zsum + b1v);   // gcn_mbgd_fp64.cu:302
        a1s[tid] = a;
        if (lead) a1_g[row * H + tid] = a;
    }
    sync();

    FB_TICK(4);
    // ---- output layer, softmax, loss, delta2, delta1 ---------------------------------
    {
        T mid = 0;
#pragma unroll
        for (int k = 0; k < NP8; k++) {
            const int i = j8 + 8 * k;
            if (i < H) mid = mid + w2s[min(o8, no - 1) + i * no] * a1s[i];
        }
        mid += __shfl_xor(mid, 4);
        mid += __shfl_xor(mid, 2);
        mid += __shfl_xor(mid, 1);
        if (o8 < no && j8 == 0) es[o8] = rexp(mid + b2v);
    }
    sync();
    FB_TICK(5);
    if (tid < no) {
        const T p = es[tid] / tree_n(es, no);
        ps[tid] = p;
        if (probs_g && lead) probs_g[row * no + tid] = p;
        const T dv = (p - lab[tid]) / no;   // gcn_mbgd_fp64.cu:352
        d2s[tid] = dv;
        if (lead) d2_g[row * no + tid] = dv;
    }
    // the filters are dead: clear their region for the scattered deltas
    for (int e = tid; e < 2 * L.ldd; e += NT) wl[e] = 0;
    sync();
    FB_TICK(6);
    if (tid < H) {
        // bp_fully_connect(out -> hidden), gcn_mbgd_fp64.cu:416-434: class-wide tree
        T prod[OMAX];
#pragma unroll
        for (int o = 0; o < OMAX; o++) prod[o] = w2s[min(o, no - 1) + tid * no] * d2s[o];   // past the class count: never used
        const T dv = tree_n(prod, no) * (1 + a) * (1 - a);
        d1s[tid] = dv;
        if (lead) d1_g[row * H + tid] = dv;
    }
    sync();
    FB_TICK(7);
#if CCNN_STOP_PHASE == 4
    return;
#endif
    // ---- delta at the pooled features + bp_maxpooling scatter -----------------------
    // dpool[p] = (sum_n W1[n + 100 p] d1[n]) (1 + pool)(1 - pool), gcn_mbgd_fp64.cu:416-434, 450-452
    {
        T *delta = wl;
        V d1v = {};
        if (qv) d1v = reinterpret_cast<const V *>(d1s)[q];
        static_assert(UF == 16, "reduce_scatter16 handles 16 rows per round");
        const int ru = reduce_scatter_index<QL>(q);          // the row of the round this lane ends up owning
        const bool writer = (q & (QL / 16 - 1)) == 0;        // one lane per owned row stores it
        auto dpool_rows = [&](int pb, const V(&w)[UF]) {
            T part[UF];   // d1v is zero in the lanes past the row's end
#pragma unroll
            for (int u = 0; u < UF; u++) {
                part[u] = 0;
#pragma unroll
                for (int j = 0; j < VW; j++) part[u] = part[u] + w[u][j] * d1v[j];
            }
            const T tot = reduce_scatter16<QL>(part, q);
            const int pi = pb + ru * G + g;
            if (writer && pi < (int)wlen) {
                int q = pi + wrot;
                q = q >= (int)wlen ? q - (int)wlen : q;
                const int p = (int)wbase + q;
                const T pv = pl[p];
                const T dv = tot * (1 + pv) * (1 - pv);
                dp[p] = dv;
                const int f = p / mre;
                if (SPLIT == 2)
                    delta[px[p] * LDWK + (f - half * (F / 2))] = dv;   // own ten filters: columns 0..9 of the first tile
                else
                    delta[(f >> 4) * L.ldd + px[p] * LDWK + (f & 15)] = dv;
            }
        };
        int pb0 = 0;
        if constexpr (PREW) {
            if constexpr (SAME_WALK) {   // the forward walk's first round, still in registers
#pragma unroll
                for (int u = 0; u < UF; u++) next_row();
            }
            dpool_rows(0, wk);
            pb0 = G * UF;
            if constexpr (PREW2) {
                if (two) {
#pragma unroll
                    for (int u = 0; u < UF; u++) next_row();
                    dpool_rows(pb0, wk2);
                    pb0 += G * UF;
                }
            }
        }
        for (int pb = pb0; pb < (int)wlen; pb += CCNN_FB_UFB * G * UF) {
            V w[CCNN_FB_UFB][UF];   // rows of several rounds in flight before the first is reduced
#pragma unroll
            for (int h = 0; h < CCNN_FB_UFB; h++) w1_rows(w[h]);
#pragma unroll
            for (int h = 0; h < CCNN_FB_UFB; h++) dpool_rows(pb + h * G * UF, w[h]);
        }
    }
    __syncthreads();

    // ---- conv gradient tile of this sample -------------------------------------------
    T *out = slab + row * (long)(F * K + F);
    // Two small serial jobs that are nobody's input ride on waves with one tap tile of the conv
    // gradient, before their tile, instead of behind everybody's: the loss (a log and an exp per class,
    // then the ordered sum; lane o the terms of class o; ps and lab are not reused by the phases in
    // between) on the last wave, the conv-bias path on the wave that shares its SIMD (with eight waves
    // and eleven tap tiles the other three SIMDs each hold a wave with two tiles).
    constexpr int BIAS_WAVE = NW >= 8 ? NW - 1 - 4 : NW - 2;
    if (wave == BIAS_WAVE) {
        // bias path: at a winner the conv output equals the pooled value, so the
        // reference's second tanh' (gcn_mbgd_fp64.cu:487-488) needs only pl.  LPF lanes per
        // filter take every LPF-th pooled position; the first of them adds their sums in lane order.
        constexpr int NF = F / SPLIT, LPF = 64 / NF;
        const int fl = min(lane / LPF, NF - 1), j = lane - (lane / LPF) * LPF;
        const int f = fl + half * (F / 2);
        T part = 0;
        for (int m = j; m < mre; m += LPF) {
            const T pv = pl[f * mre + m];
            part = part + dp[f * mre + m] * (1 + pv) * (1 - pv);
        }
        T mid1 = part;
#pragma unroll
        for (int k = 1; k < LPF; k++) mid1 = mid1 + __shfl_down(part, k);
        if (j == 0 && lane < NF * LPF) out[F * K + f] = mid1 / re;
    }
    if (lead && wave == NW - 1) {
        T ta = 0, tb = 0;
        int lb = 0;
        if (lane < no) {
            lb = lab[lane];
            loss_terms(ps[lane], lb, &ta, &tb);
        }
        const double loss = loss_sum_lanes(ta, tb, lb, no);
        if (lane == 0) loss_values[row] = loss;
    }
    FB_TICK(8);
#if CCNN_STOP_PHASE == 5
    return;
#endif
    if (SPLIT == 2)
        conv_bwd_tiles<T, NW, true>(cube, wl, L.ldd, L.ks, re, out, lane, wave, half * (F / 2), F / 2);
    else
        conv_bwd_tiles<T, NW>(cube, wl, L.ldd, L.ks, re, out, lane, wave);
    FB_TICK(9);
    FB_TICK(10);
    FB_WG(1 + wave);
}

// ---------------------------------------------------------------------------
// Batch sums of the per-sample gradients, written in the flat parameter layout
// (update_params_row/col, gcn_mbgd_fp64.cu:513-517, 546-550).  With an UpdateRule
// the update is applied in the same pass instead of storing the gradient.
//   blocks [0, nb_w1)  one 16 x 16 tile per block, K = batch (MFMA) split over the eight waves:
//                      dW1[n + 100 p] = sum_b pool[b][p] d1[b][n] (p tile, neuron tile), then the 7
//                      tiles of dW2[o + 13 n] = sum_b a1[b][n] d2[b][o] (neuron tile x classes)
//   then column sums, 64 columns x 8 row groups per block, groups added in order:
//     nb_w0 blocks     dW0 | db0 = sum_b slab[b][.]
//     2 blocks         db1 = sum_b d1        1 block   db2 = sum_b d2
#ifndef CCNN_FB_THREADS
#define CCNN_FB_THREADS 512
#endif
constexpr int FB_THREADS = CCNN_FB_THREADS;   // threads per sample_fwd_bwd workgroup
constexpr int GR_THREADS = 512;
constexpr int GR_NT = (H + 15) / 16;         // neuron tiles of dW1
constexpr int GR_TILE_VALUES = 256;          // values a dW1 workgroup produces (one 16 x 16 tile)

template <typename T>
struct UpdateRule {
    T *params;       // null: store the gradient
    T lr;
    int divisor;     // fp64: w -= lr*g/divisor (gcn_mbgd_fp64.cu:519); 0: w -= lr*g (gcn_mbgd_fp32.cu:529)
    int frozen_lo, frozen_hi;
    __device__ __forceinline__ void apply(T *grad, int k, T g) const
    {
        if (!params) {
            grad[k] = g;
        } else if (!(k >= frozen_lo && k < frozen_hi)) {
            params[k] = divisor ? params[k] - lr * g / divisor : params[k] - lr * g;
        }
    }
};

// Sum over rows b = rg, rg+8, ... of term(b), U loads in flight; then over the 8 row groups.
template <typename T, int U = 16, typename Term>
__device__ __forceinline__ T colsum8(int B, int rg, int c, T *red, Term term)
{
    T acc = 0;
    for (int b0 = rg; b0 < B; b0 += 8 * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int b = b0 + 8 * u;
            const T x = term(b < B ? b : (B > 0 ? B - 1 : 0));   // clamped load, masked value (B == 0: row 0 exists, unused)
            v[u] = b < B ? x : T(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc = acc + v[u];
    }
    red[rg * 64 + c] = acc;
    __syncthreads();
    T tot = 0;
    if (rg == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) tot = tot + red[r * 64 + c];
    }
    return tot;
}

// Batch sums of the per-sample factors, one output element per thread, then either stored /
// applied directly (one GPU) or first summed over the ranks: with
// xc.world > 0 a workgroup writes its outputs into its slot of every peer's exchange buffer,
// raises its flag there, waits for the same workgroup of every rank and adds the world
// contributions in rank order before `upd` sees them (comm.h).  No grid-wide barrier: a
// workgroup only depends on its namesakes on the other ranks.
// dW1: a workgroup per 16 x 16 tile (p tile blk / 7, neuron tile blk mod 7) whose eight waves each
// take an eighth of the batch - 4 k-steps of a batch of 128: eight loads per lane, all in flight, and
// four matrix instructions - and meet in LDS, where the first 256 threads add the eight partial
// tiles in wave order.  (One workgroup per p tile with a wave per neuron tile and the whole batch
// per wave - 64 loads per lane, 32 dependent matrix instructions, two waves per SIMD - took 3.0 +
// 0.7 us per workgroup where the column sums of the same launch take 1.3 + 0.3: it WAS the launch.)
// dW2 is a matrix product too (it was column sums of a1 x d2 products, two loads per row: once the
// tiles no longer kept the kernel's register count up, the compiler fetched those rows a few at a
// time - 1.9 -> 4.8 us).  The kernel keeps to 64 registers, four workgroups per CU: ranks that share
// one GPU (tests, rehearsals) wait for each other inside it, and a rank whose workgroups found no
// room beside the others' would never answer.
template <typename T, int NO>
__global__ __launch_bounds__(GR_THREADS) void grad_reduce(int B, int pout, int off_w1, int off_b1,
                                                          int off_w2, int off_b2,
                                                          const T *__restrict__ pool,
                                                          const T *__restrict__ a1,
                                                          const T *__restrict__ d1,
                                                          const T *__restrict__ d2,
                                                          const T *__restrict__ slab, int slab_rows, int nb_w1,
                                                          int nb_w0, T *__restrict__ grad,
                                                          UpdateRule<T> upd, CommDev xc,
                                                          unsigned xstep)
{
    __shared__ T red[8 * 256];
    const int tid = threadIdx.x;
    const int no = NO ? NO : (off_b2 - off_w2) / H;   // classes: W2 is [100 x classes]; NO as in sample_fwd_bwd
    // the column-sum workgroups are dispatched first (their rounds of loads are the longer chain),
    // the dW1 tiles after them; lb is the workgroup's number in the order of the list above
    const int nsmall = (int)gridDim.x - nb_w1;
    const int lb = (int)blockIdx.x < nsmall ? nb_w1 + (int)blockIdx.x : (int)blockIdx.x - nsmall;
    int blk = lb;
    GR_WG(0);
    int ok[4] = {-1, -1, -1, -1};   // parameter index of this thread's outputs (-1: none)
    T og[4] = {0, 0, 0, 0};         // and their batch sums
    if (blk < nb_w1) {
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int col = lane & 15, kq = lane >> 4;
        // rows x columns of the tile: 16 pooled features x 16 neurons of dW1 (operands pool, d1), or, in
        // the last GR_NT tile workgroups, 16 neurons x the classes of dW2 (operands a1, d2)
        // consecutive workgroups go round the eight XCDs: the GR_NT tiles that share a row tile's operand
        // (a p tile's 16 columns of pool) are dealt to ONE XCD, so that it reaches one L2 only - groups
        // of 8 row tiles x GR_NT; the row tiles past the last whole group in plain order
        const int nrt = nb_w1 / GR_NT, nrt8 = nrt & ~7;
        int pt, nt;
        if (blk < nrt8 * GR_NT) {
            const int grp = blk / (8 * GR_NT), in = blk - grp * (8 * GR_NT);
            pt = 8 * grp + (in & 7);
            nt = in >> 3;
        } else {
            const int rest = blk - nrt8 * GR_NT;
            pt = nrt8 + rest / GR_NT;
            nt = rest - (rest / GR_NT) * GR_NT;
        }
        const bool w2 = pt == nrt - 1;
        const int nrow = w2 ? H : pout, ncol = w2 ? no : H;   // = the operands' row strides
        const int p0 = (w2 ? nt : pt) * 16, n0 = w2 ? 0 : nt * 16;
        const int obase = w2 ? off_w2 : off_w1;
        const bool pv = p0 + col < nrow, nv = n0 + col < ncol;
        const T *ap = (w2 ? a1 : pool) + min(p0 + col, nrow - 1);   // clamped: loads stay in bounds, masked below
        const T *bp = (w2 ? d2 : d1) + min(n0 + col, ncol - 1);
        typename Mma<T>::acc_t acc = {0, 0, 0, 0};
        const int ks = (B + 3) >> 2, kpw = (ks + 7) >> 3;   // k-steps of four samples: all, per wave
        const int s_lo = wave * kpw, s_hi = min(ks, s_lo + kpw);
        constexpr int KW = 8;   // k-steps of operands in flight: a batch of up to 256 in one round (16: no faster at 1024 .. 4096)
        for (int s = s_lo; s < s_hi; s += KW) {
            T a[KW], w[KW];
#pragma unroll
            for (int u = 0; u < KW; u++) {
                const int b = min(4 * (s + u) + kq, B - 1);
                a[u] = ap[(long)b * nrow];
                w[u] = bp[(long)b * ncol];
            }
#pragma unroll
            for (int u = 0; u < KW; u++) {
                const bool bv = s + u < s_hi && 4 * (s + u) + kq < B;
                acc = Mma<T>::mma((bv && pv) ? a[u] : T(0), (bv && nv) ? w[u] : T(0), acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave * 256 + r * 64 + lane] = acc[r];
        __syncthreads();
        if (tid < 256) {
            T tot = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) tot = tot + red[w * 256 + tid];
            const int r = tid >> 6, l = tid & 63;   // accumulator register r of lane l
            const int p = p0 + Mma<T>::row(l, r), n = n0 + (l & 15);
            if (n < ncol && p < nrow) ok[0] = obase + n + ncol * p;   // dW1[n + 100 p], dW2[o + classes n]
            og[0] = tot;
        }
    } else {
        blk -= nb_w1;
        const int c = tid & 63, rg = tid >> 6;
        if (blk < nb_w0) {
            const int k = min(blk * 64 + c, F * K + F - 1);
            const T tot = colsum8<T>(slab_rows, rg, c, red, [&](int b) { return slab[(long)b * (F * K + F) + k]; });
            if (rg == 0 && blk * 64 + c < F * K + F) {   // slab layout = W0 | b0 = params[0..)
                ok[0] = k;
                og[0] = tot;
            }
        } else if (blk < nb_w0 + 2) {
            const int q = blk - nb_w0;
            const int n = min(q * 64 + c, H - 1);
            const T tot = colsum8<T>(B, rg, c, red, [&](int b) { return d1[(long)b * H + n]; });
            if (rg == 0 && q * 64 + c < H) {
                ok[0] = off_b1 + n;
                og[0] = tot;
            }
        } else {
            const int o = min(c, no - 1);
            const T tot = colsum8<T>(B, rg, c, red, [&](int b) { return d2[(long)b * no + o]; });
            if (rg == 0 && c < no) {
                ok[0] = off_b2 + o;
                og[0] = tot;
            }
        }
    }
    GR_WG(1);
    int gave_up = 0;
    if (xc.world > 0) {
        // The outputs travel in launch geometry, not in parameter order (a dW1 workgroup's slot
        // region holds one value per thread of its first four waves, the others one per lane of their first wave),
        // each 32-bit half of a value in its own sequence-tagged word (comm.h): a thread trades
        // with its namesakes on the other ranks without any fence, flag or barrier.  (A
        // system-scope fence here would have to write back the megabytes of dirty L2 lines the
        // per-sample kernel has just left behind: measured +14 us per step.)
        constexpr int WPV = sizeof(T) / 4;   // tagged words per value
        const int par = xstep & 1;
        const bool tile = lb < nb_w1;
        // (threads whose value is padding - a tile's rows and columns past the matrix, a column-sum
        // workgroup's lanes past its last column - trade nothing: the same threads on every rank, and
        // an eighth of the bytes on the links)
        const bool sender = (tile ? tid < GR_TILE_VALUES : tid < 64) && ok[0] >= 0;
        const int nv = 1;                    // values this thread trades
        const long word0 = (tile ? (long)lb * GR_TILE_VALUES + tid
                                 : (long)nb_w1 * GR_TILE_VALUES + (long)(lb - nb_w1) * 64 + tid) * WPV;
        if (sender) {
            unsigned bits[4 * WPV];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (WPV == 1) {
                    bits[j] = __float_as_uint((float)og[j]);
                } else {
                    const unsigned long long q = (unsigned long long)__double_as_longlong((double)og[j]);
                    bits[2 * j] = (unsigned)q;
                    bits[2 * j + 1] = (unsigned)(q >> 32);
                }
            }
            auto send = [&](int to, int slot_of, const unsigned *b) {   // this thread's words into `to`'s buffer, slot [par][slot_of]
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(comm_slot(xc, to, par, slot_of)) + word0;
#pragma unroll
                for (int h = 0; h < WPV; h++) comm_store_tagged(dst + 4 * h, b + 4 * h, max(min(nv * WPV - 4 * h, 4), 0), xstep);
            };
            auto receive = [&](int slot_of, unsigned *in) {              // from the local buffer, slot [par][slot_of]
                const unsigned long long *src =
                    reinterpret_cast<const unsigned long long *>(comm_slot(xc, xc.rank, par, slot_of)) + word0;
#pragma unroll
                for (int h = 0; h < WPV; h++)
                    if (!comm_load_tagged(src + 4 * h, in + 4 * h, max(min(nv * WPV - 4 * h, 4), 0), xstep, xc)) gave_up = 1;
            };
            auto value = [&](const unsigned *in, int j) -> T {
                if (WPV == 1) return (T)__uint_as_float(in[j]);
                const unsigned long long q = (unsigned long long)in[2 * j] | ((unsigned long long)in[2 * j + 1] << 32);
                return (T)__longlong_as_double((long long)q);
            };
            // via_owner: the workgroup's sums of all ranks meet at ONE rank (workgroup b: rank b mod
            // world), which adds them in rank order and sends the total to everybody: 2 (world-1)/world
            // of a vector per rank over the links instead of world-1 vectors, for a second hop.  A
            // workgroup's slots [par][r] are used in one direction only per call (contributions at the
            // owner, the total elsewhere), so both hops share the buffers and the sequence number.
            const int owner = xc.via_owner ? lb % xc.world : -1;
            unsigned in[4 * WPV];
            if (owner >= 0 && owner != xc.rank) {
                send(owner, xc.rank, bits);
                receive(owner, in);
                if (!gave_up) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (j < nv) og[j] = value(in, j);
                }
            } else {
                if (owner < 0) {
                    for (int k = 0; k < xc.world; k++)
                        send((xc.rank + 1 + k) % xc.world, xc.rank, bits);   // start at the next rank: spreads the writes over the links
                }
                T mine[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    mine[j] = og[j];
                    og[j] = 0;
                }
                for (int r = 0; r < xc.world; r++) {   // rank order: the same bits on every rank
                    const bool own = owner >= 0 && r == xc.rank;   // the owner's own sums never leave its registers
                    if (!own) receive(r, in);
                    if (gave_up) break;   // in[] is not set: nothing of this step may reach the parameters
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (j < nv) og[j] = og[j] + (own ? mine[j] : value(in, j));
                }
                if (owner >= 0 && !gave_up) {
                    unsigned tot[4 * WPV];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (WPV == 1) {
                            tot[j] = __float_as_uint((float)og[j]);
                        } else {
                            const unsigned long long q = (unsigned long long)__double_as_longlong((double)og[j]);
                            tot[2 * j] = (unsigned)q;
                            tot[2 * j + 1] = (unsigned)(q >> 32);
                        }
                    }
                    for (int k = 0; k + 1 < xc.world; k++) send((xc.rank + 1 + k) % xc.world, xc.rank, tot);
                }
            }
        }
        // A wait that ran out (peer gone or stalled; the status word is set): the whole workgroup
        // leaves its parameters at the last completed step.  Once one workgroup has given up the
        // others see the status word within a thousand polls and give up too, so a peer that never
        // launched leaves every parameter untouched.
        if (__syncthreads_or(gave_up)) return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (ok[j] >= 0) upd.apply(grad, ok[j], og[j]);
    GR_WG(2);
}


// ===========================================================================
// Batched path (large minibatches).  The per-sample kernel above re-reads all of W1 twice per
// sample; here every layer runs over the whole batch:
//   conv_pool_fwd<TRAIN> + fc_head_fwd<TRAIN>   forward, loss, delta2, delta1        (forward.hip)
//   dpool_bwd        dpool = (delta1 W1^T)(1 + pool)(1 - pool): one [B x 100] x [100 x Pout] product
//   conv_bwd_batch   bp_maxpooling scatter + conv gradient, accumulated over a workgroup's samples
//   grad_reduce      dW1 = pool^T delta1, dW2, db1, db2, the column sums of conv_bwd_batch's partials, the update
// W1 is read once per 16-row tile of the products instead of twice per sample.

// dpool[b][p] = (sum_n W1[n + 100 p] d1[b][n]) (1 + pool[b][p]) (1 - pool[b][p])
// (bp_fully_connect hidden -> pool, gcn_mbgd_fp64.cu:416-434, batched).  Rows of the MFMA tile =
// 16 pooled features (A = 16 rows of W1, held in registers for the whole workgroup), columns =
// 16 samples (B = 16 rows of d1).  Both operands are contiguous along n, so each lane fetches
// 16-byte vectors and the product's k order is permuted to match: k-step s of lane group kq
// takes n = VW kq + (s mod VW) + 4 VW (s / VW) - the same permutation on both sides.
constexpr int DP_THREADS = 256;
template <typename T>
__global__ __launch_bounds__(DP_THREADS) void dpool_bwd(const T *__restrict__ W1, const T *__restrict__ d1,
                                                        const T *__restrict__ pool, T *__restrict__ dp, int B,
                                                        int pout, int tiles_per_wave)
{
    typedef typename Vec16<T>::type V;
    constexpr int VW = Vec16<T>::W;
    constexpr int NG = (H + 4 * VW - 1) / (4 * VW);   // vectors per lane along n
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, kq = lane >> 4;
    const int p0 = blockIdx.x * 16;
    const int prow = min(p0 + col, pout - 1);          // clamped: loads stay in bounds, stores masked
    V a[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int n = VW * kq + 4 * VW * g;
        a[g] = reinterpret_cast<const V *>(W1 + (long)prow * H)[min(n, H - VW) / VW];
        if (n >= H) a[g] = V{};
    }
    const int btiles = (B + 15) >> 4;
    const int bt0 = (blockIdx.y * (DP_THREADS / 64) + wave) * tiles_per_wave;
    for (int bt = bt0; bt < min(bt0 + tiles_per_wave, btiles); bt++) {
        const int b0 = bt * 16;
        const int brow = min(b0 + col, B - 1);
        V w[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int n = VW * kq + 4 * VW * g;
            w[g] = reinterpret_cast<const V *>(d1 + (long)brow * H)[min(n, H - VW) / VW];   // lanes past n = 100 meet a zero in a[]
        }
        typename Mma<T>::acc_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int j = 0; j < VW; j++) acc = Mma<T>::mma(a[g][j], w[g][j], acc);
        // lane holds sample b0 + col, features p0 + Mma::row(lane, 0..3)
        const int b = b0 + col;
        if (b < B) {
            if (sizeof(T) == 4) {
                // f32 result map: rows 4 q .. 4 q + 3 are consecutive features: one 16-byte access
                // each way (pout is a multiple of 4, so a vector is wholly inside or outside)
                const int p = p0 + 4 * kq;
                if (p < pout) {
                    const V pv = *reinterpret_cast<const V *>(pool + (long)b * pout + p);
                    V o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r % VW] = acc[r] * (1 + pv[r % VW]) * (1 - pv[r % VW]);
                    *reinterpret_cast<V *>(dp + (long)b * pout + p) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int p = p0 + Mma<T>::row(lane, r);
                    if (p < pout) {
                        const T pv = pool[(long)b * pout + p];
                        dp[(long)b * pout + p] = acc[r] * (1 + pv) * (1 - pv);
                    }
                }
            }
        }
    }
}

// bp_maxpooling + bp_convolution over a workgroup's share of the batch
// (gcn_mbgd_fp64.cu:439-496): per sample the delta of each pooled feature goes to the position
// that won its pair, and dW0[f][tap] += sum_t delta[f][t] cube[18 t + tap] is the [20 x re] x
// [re x 171] MFMA product of conv_bwd_tiles - but the accumulators live across the workgroup's
// samples, so what reaches memory is one [F K + F] partial per WORKGROUP, not per sample.
// 11 waves, one per 16-tap tile (both filter tiles each): no wave shares an output with another,
// so nothing is reduced across waves.  Cube and delta image are double-buffered in LDS (NBUF = 2,
// or 1 where two do not fit): the next sample is staged while the current one is multiplied.
// part[g][f K + tap] = sum / re, part[g][F K + f] = sum_m dp (1 + pool)(1 - pool) / re (the
// reference's second tanh' on the bias path, :487-488; at a winner the conv output is the pooled value).
constexpr int CB_WAVES = 11;
constexpr int CB_THREADS = 64 * CB_WAVES;
template <typename T>
struct ConvBwdLds {
    int cube_len, delta_len, dps_len, per_buf;
    __host__ __device__ ConvBwdLds(int re, int mre)
    {
        cube_len = up4(conv_cube_pad(re));
        delta_len = 2 * 4 * ((re + 3) >> 2) * LDWK;
        dps_len = up4(F * mre);
        per_buf = cube_len + delta_len + dps_len;
    }
    size_t bytes(int nbuf) const { return (size_t)nbuf * per_buf * sizeof(T); }
};

template <typename T, int NBUF>
__global__ __launch_bounds__(CB_THREADS) void conv_bwd_batch(const T *__restrict__ cubes, long first,
                                                             const T *__restrict__ dp, const T *__restrict__ pool,
                                                             const unsigned char *__restrict__ px, int B,
                                                             int cube_elems, int re, int mre, T *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvBwdLds<T> L(re, mre);
    T *base = reinterpret_cast<T *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kq = lane >> 4;
    const int pout = F * mre, ks = (re + 3) >> 2, ldd = 4 * ks * LDWK;
    const int G = gridDim.x;
    auto buf_cube = [&](int i) { return base + (NBUF == 2 ? i : 0) * L.per_buf; };
    auto buf_delta = [&](int i) { return buf_cube(i) + L.cube_len; };
    auto buf_dps = [&](int i) { return buf_delta(i) + L.delta_len; };

    // stage sample b into buffer i: cube (zero-padded), zeroed delta image, then the scatter
    constexpr int NCR = (NB * CCNN_MAX_BANDS + CB_THREADS - 1) / CB_THREADS;                      // cube elements per thread
    constexpr int NPU = (F * ((((CCNN_MAX_BANDS - 20) / 2 + 1) + 1) / 2) + CB_THREADS - 1) / CB_THREADS;   // pooled features per thread
    auto stage_loads = [&](int b, T(&cr)[NCR], T(&dv)[NPU], T(&pv)[NPU], unsigned(&bit)[NPU]) {
        const T *src = cubes + (first + b) * (long)cube_elems;
#pragma unroll
        for (int u = 0; u < NCR; u++) cr[u] = src[min(u * CB_THREADS + tid, cube_elems - 1)];
#pragma unroll
        for (int u = 0; u < NPU; u++) {
            const int p = min(u * CB_THREADS + tid, pout - 1);
            dv[u] = dp[(long)b * pout + p];
            pv[u] = pool[(long)b * pout + p];
            bit[u] = px[(long)b * pout + p];
        }
    };
    auto stage_zero = [&](int i) {
        T *cube = buf_cube(i), *dl = buf_delta(i);
        for (int e = cube_elems + tid; e < L.cube_len; e += CB_THREADS) cube[e] = 0;
        for (int e = tid; e < L.delta_len; e += CB_THREADS) dl[e] = 0;
    };
    auto stage_cube = [&](int i, const T(&cr)[NCR]) {
        T *cube = buf_cube(i);
#pragma unroll
        for (int u = 0; u < NCR; u++) {
            const int e = u * CB_THREADS + tid;
            if (e < cube_elems) cube[e] = cr[u];
        }
    };
    auto stage_scatter = [&](int i, const T(&dv)[NPU], const T(&pv)[NPU], const unsigned(&bit)[NPU]) {
        T *dl = buf_delta(i), *dps = buf_dps(i);
#pragma unroll
        for (int u = 0; u < NPU; u++) {
            const int p = u * CB_THREADS + tid;
            if (p < pout) {
                const int f = p / mre, m = p - f * mre;
                dl[(f >> 4) * ldd + (2 * m + (int)bit[u]) * LDWK + (f & 15)] = dv[u];
                dps[p] = dv[u] * (1 + pv[u]) * (1 - pv[u]);
            }
        }
    };

    typename Mma<T>::acc_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    T bsum = 0;   // lanes 0..19 of wave 0: the bias path of filter `lane`
    int b = blockIdx.x;
    if (b >= B) {   // no sample for this workgroup: its partial is zero
        for (int e = tid; e < F * K + F; e += CB_THREADS) part[(long)blockIdx.x * (F * K + F) + e] = 0;
        return;
    }
    // Two samples ahead: set A holds the sample that is staged into LDS next, set B receives the
    // loads of the one after it, so a load has a whole multiply phase plus a staging phase to land.
    struct Regs {
        T cr[NCR], dv[NPU], pv[NPU];
        unsigned bit[NPU];
    } ra, rb;
    stage_loads(b, ra.cr, ra.dv, ra.pv, ra.bit);
    stage_zero(0);
    stage_cube(0, ra.cr);
    __syncthreads();   // delta image zeroed before anybody scatters into it
    stage_scatter(0, ra.dv, ra.pv, ra.bit);
    if (b + G < B) stage_loads(b + G, ra.cr, ra.dv, ra.pv, ra.bit);
    __syncthreads();
    int cur = 0;
    // one sample: multiply buffer `cur`, stage the sample held in `nx` (if any) for the next
    // round, load the sample after that into `ld`
    auto round = [&](int bb, Regs &nx, Regs &ld) {
        const bool more = bb + G < B;
        if (bb + 2 * G < B) stage_loads(bb + 2 * G, ld.cr, ld.dv, ld.pv, ld.bit);
        if (NBUF == 2 && more) stage_zero(cur ^ 1);
        {
            const T *cube = buf_cube(cur), *dl = buf_delta(cur);
            const T *ap0 = dl + kq * LDWK + col, *ap1 = ap0 + ldd;
            const T *bp = cube + WSTEP * kq + wave * 16 + col;
            int s = 0;
            for (; s + 2 <= ks; s += 2) {
                const T a00 = ap0[4 * s * LDWK], a10 = ap1[4 * s * LDWK], b0 = bp[4 * s * WSTEP];
                const T a01 = ap0[4 * (s + 1) * LDWK], a11 = ap1[4 * (s + 1) * LDWK], b1 = bp[4 * (s + 1) * WSTEP];
                c0 = Mma<T>::mma(a00, b0, c0);
                c1 = Mma<T>::mma(a10, b0, c1);
                c0 = Mma<T>::mma(a01, b1, c0);
                c1 = Mma<T>::mma(a11, b1, c1);
            }
            if (s < ks) {
                const T a00 = ap0[4 * s * LDWK], a10 = ap1[4 * s * LDWK], b0 = bp[4 * s * WSTEP];
                c0 = Mma<T>::mma(a00, b0, c0);
                c1 = Mma<T>::mma(a10, b0, c1);
            }
            if (wave == 0 && lane < F) {
                const T *dps = buf_dps(cur);
                T mid1 = 0;
                for (int m = 0; m < mre; m++) mid1 = mid1 + dps[lane * mre + m];
                bsum = bsum + mid1;
            }
        }
        if (NBUF == 2) {
            if (more) stage_cube(cur ^ 1, nx.cr);
            __syncthreads();   // next image zeroed and its cube in place; current buffers no longer read
            if (more) stage_scatter(cur ^ 1, nx.dv, nx.pv, nx.bit);
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();   // everyone is done reading the only buffer
            if (more) {
                stage_zero(0);
                stage_cube(0, nx.cr);
            }
            __syncthreads();
            if (more) stage_scatter(0, nx.dv, nx.pv, nx.bit);
            __syncthreads();
        }
    };
    for (; b < B; b += 2 * G) {
        round(b, ra, rb);
        if (b + G < B) round(b + G, rb, ra);
    }
    T *out = part + (long)blockIdx.x * (F * K + F);
    const int tap = wave * 16 + col;
    if (tap < K) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f0 = Mma<T>::row(lane, r), f1 = 16 + f0;
            out[f0 * K + tap] = c0[r] / re;   // gcn_mbgd_fp64.cu:491
            if (f1 < F) out[f1 * K + tap] = c1[r] / re;
        }
    }
    if (wave == 0 && lane < F) out[F * K + lane] = bsum / re;
}

// ---------------------------------------------------------------------------
// Which path a batch of B takes.  The fused per-sample kernel is a chain of latency-bound phases
// that costs the same while all its workgroups fit the chip at once and then grows by the full
// chain per round of workgroups (every workgroup walks W1).  The sample-tile kernel's chain is longer
// for one sample but takes up to eight samples per workgroup at little more, and reads W1 twice per
// TILE.  Measured crossovers fused -> tile: step_path() below.  Where a tile holds so few samples
// that the batch needs more than one round of workgroups (fp64, or fp32 beyond 200 bands: the LDS
// holds four samples) the layer kernels win again from about 2000 samples (z = 176, fp64: 181
// against 192 us at 2048).
constexpr int BATCHED_MIN_B = 768;          // fused -> layer kernels where no tile fits
constexpr int BATCHED_MAX_PARTIALS = 256;   // workgroups (= partial gradient rows) of conv_bwd_batch
// FUSED: sample_fwd_bwd.  TILE: tile_fwd_bwd.  LAYERS: the four layer kernels (shapes whose tile does
// not fit the LDS).  grad_reduce sums the batch and updates behind each of them.
// CCNN_MBGD_PATH=fused|batched|tile|layers overrides the choice by batch size (tests run every
// path on the same inputs; "batched" = whatever a large batch would take).
enum class StepPath { FUSED, TILE, LAYERS };
// The two test switches, read ONCE per entry-point call (tests flip them inside one process, so they
// cannot be latched at load time; a host that changes its environment while a call is in progress
// on another thread gets either value).
struct StepEnv {
    int path;    // -1: the library chooses; else a StepPath, or 3 = "whatever a large batch would take"
    int waves;   // CCNN_TILE_WAVES (0: default)
    bool operator==(const StepEnv &o) const { return path == o.path && waves == o.waves; }
};
inline StepEnv step_env()
{
    StepEnv e = {-1, tile_waves_env()};
    if (const char *s = getenv("CCNN_MBGD_PATH")) {
        if (!strcmp(s, "fused")) e.path = (int)StepPath::FUSED;
        else if (!strcmp(s, "layers")) e.path = (int)StepPath::LAYERS;
        else if (!strcmp(s, "batched") || !strcmp(s, "tile")) e.path = 3;
    }
    return e;
}
template <typename T>
inline StepPath step_path(int B, const ccnn_dims_t &d, TilePlan *plan, const StepEnv &env)
{
    TilePlan tp = {};
    const bool fits = tile_plan_waves<T>(B, d, env.waves, &tp);
    if (plan && fits) *plan = tp;
    if (env.path == (int)StepPath::FUSED) return StepPath::FUSED;
    if (env.path == (int)StepPath::LAYERS) return StepPath::LAYERS;
    if (env.path == 3) return fits ? StepPath::TILE : StepPath::LAYERS;
    // The fused kernel's time steps up where its workgroups stop fitting the chip at once: past 512
    // samples in fp32 (two workgroups per CU; 103 bands: 29.4 us at 512 and 41.4 at 640 against the
    // tile kernel's 33.5 and 36; 200 bands: 43.3 and 59.7 against 53.5 and 60.3), past 256 in fp64
    // (176 bands: 46.4 us at 256 and 77.1 at 384 against 64.6 and 73.5)
    const int tile_min_b = sizeof(T) == 8 ? 257 : 513;
    // one round of workgroups, or several rounds of FULL tiles at small band counts (fp32, B = 4096:
    // 110 us against the layer kernels' 120 at z = 103, but 210 against 198 at z = 224)
    const bool rounds_ok = tp.tiles <= 256 || (sizeof(T) == 4 && tp.ms == 8 && d.z <= 128);
    if (fits && B >= tile_min_b && rounds_ok) return StepPath::TILE;
    return B >= BATCHED_MIN_B ? StepPath::LAYERS : StepPath::FUSED;
}
// What a batch of B takes, worked out once per (B, geometry, T, switches) and remembered: a training
// loop asks on every step, and the search over tile shapes is a few hundred layout computations.
struct StepChoice {
    StepPath path;
    TilePlan plan;   // path == TILE
};
template <typename T>
inline StepChoice step_choice(int B, const ccnn_dims_t &d, const StepEnv &env)
{
    struct Entry {
        int B, z, classes;
        StepEnv env;
        StepChoice c;
    };
    constexpr int N = 8;
    thread_local Entry cache[N];
    thread_local int used = 0, next = 0;
    for (int i = 0; i < used; i++) {
        const Entry &e = cache[i];
        if (e.B == B && e.z == d.z && e.classes == d.n_classes && e.env == env) return e.c;
    }
    StepChoice c = {};
    c.path = step_path<T>(B, d, &c.plan, env);
    cache[next] = Entry{B, d.z, d.n_classes, env, c};
    next = (next + 1) % N;
    if (used < N) used++;
    return c;
}
// Two workgroups per sample in the fused kernel: while both halves of every sample fit the device
// at once, one workgroup per CU (CCNN_FB_SPLIT=0|1 overrides: diagnostics and tests).
inline int fb_cus()   // compute units of the current device (looked up once per device)
{
    static std::mutex mu;
    static int known[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lk(mu);
        if (known[dev] > 0) return known[dev];
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lk(mu);
        known[dev] = cus;
    }
    return cus;
}
inline bool split_samples(int B)
{
    static const int forced = [] {
        const char *s = getenv("CCNN_FB_SPLIT");
        return s ? atoi(s) : -1;
    }();
    if (forced >= 0) return forced != 0;
    return 16 * ((B + 7) / 8) <= fb_cus();
}
// The forward pass splits over a sample's two workgroups as well when every workgroup of the launch
// has a compute unit to itself from the start: partners wait for each other, and a partner that
// could only start after its workgroup had finished would never come.  CCNN_FB_FSPLIT=0 (tests,
// measurements) keeps the whole forward pass in both.
inline bool split_forward(int B)
{
    static const bool off = [] {
        const char *s = getenv("CCNN_FB_FSPLIT");
        return s && atoi(s) == 0;
    }();
    return !off && split_samples(B) && 16 * ((B + 7) / 8) <= fb_cus();
}
// The tag of a launch's pair_sum words: a process-wide counter that starts at a value derived from
// the clock and the process id (what an earlier process left in memory does not carry this one's
// tags) and skips 0, the value of a consumed word.
inline unsigned next_pair_tag()
{
    static std::atomic<unsigned> ctr{(unsigned)std::chrono::steady_clock::now().time_since_epoch().count() * 2654435761u ^
                                     ((unsigned)getpid() << 16)};
    unsigned t;
    do t = ctr.fetch_add(1, std::memory_order_relaxed);
    while (t == 0);
    return t;
}
template <typename T>
struct MbgdWs {
    T *pool, *a1, *d1, *d2, *slab, *probs;
    unsigned long long *xch;   // fused path, forward pass split: [rows][2][H][PairWords] tagged words
    static constexpr int XCH_MAX_B = 256;   // no device fits two workgroups per sample for more
    static size_t xch_words(int B) { return (size_t)8 * ((B + 7) / 8) * 2 * H * PairWords<T>::N; }
    // batched path only
    T *dp;
    unsigned char *px;
    int *ctr;
    StepPath path;
    TilePlan plan;   // path == TILE
    static int partial_rows(int B, StepPath path, const TilePlan &tp)   // rows of the conv-gradient partials
    {
        return path == StepPath::TILE ? tp.tiles : B < BATCHED_MAX_PARTIALS ? B : BATCHED_MAX_PARTIALS;
    }
    static size_t bytes(int B, const ccnn_dims_t &d, const StepChoice &sc)
    {
        size_t b = 0;
        b += align_up((size_t)B * d.pout * sizeof(T));          // pool
        b += 2 * align_up((size_t)B * H * sizeof(T));           // a1, d1
        b += 2 * align_up((size_t)B * d.n_classes * sizeof(T));   // d2, probs
        const StepPath path = sc.path;
        const TilePlan &tp = sc.plan;
        if (path == StepPath::FUSED) {
            b += align_up((size_t)B * (F * K + F) * sizeof(T));     // per-sample conv gradient tiles
            if (B <= XCH_MAX_B) b += align_up(xch_words(B) * 8);
        } else {
            const int G = partial_rows(B, path, tp);
            b += align_up((size_t)G * (F * K + F) * sizeof(T));     // per-workgroup conv gradient partials
            if (path == StepPath::LAYERS) {
                b += align_up((size_t)B * d.pout * sizeof(T));      // dpool
                b += align_up((size_t)B * d.pout);                  // pool winners, one byte each
            }
            b += align_up(TRAIN_CTR_INTS * sizeof(int));
        }
        return b;
    }
    MbgdWs(void *ws, int B, const ccnn_dims_t &d, const StepChoice &sc)
    {
        Carver c(ws);
        pool = c.take<T>((size_t)B * d.pout);
        a1 = c.take<T>((size_t)B * H);
        d1 = c.take<T>((size_t)B * H);
        d2 = c.take<T>((size_t)B * d.n_classes);
        probs = c.take<T>((size_t)B * d.n_classes);
        dp = nullptr;
        xch = nullptr;
        px = nullptr;
        ctr = nullptr;
        path = sc.path;
        plan = sc.plan;
        if (path == StepPath::FUSED) {
            slab = c.take<T>((size_t)B * (F * K + F));
            if (B <= XCH_MAX_B) xch = c.take<unsigned long long>(xch_words(B));
            if (!split_forward(B)) xch = nullptr;   // the region is there whatever the device (the size must not depend on one)
        } else {
            const int G = partial_rows(B, path, plan);
            slab = c.take<T>((size_t)G * (F * K + F));
            if (path == StepPath::LAYERS) {
                dp = c.take<T>((size_t)B * d.pout);
                px = c.take<unsigned char>((size_t)B * d.pout);
            }
            ctr = c.take<int>(TRAIN_CTR_INTS);
        }
    }
};

// The two launches every path ends with share this: batch sums (+ update / exchange).
template <typename T>
int launch_grad_reduce(int B, const ccnn_dims_t &d, const MbgdWs<T> &w, int slab_rows, T *grad, UpdateRule<T> upd,
                       ccnn_comm *cm, hipStream_t st)
{
    const int nb_w1 = GR_NT * ((d.pout + 15) / 16 + 1);   // tiles of dW1, then of dW2
    const int nb_w0 = (F * K + F + 63) / 64;
    const int nblocks = nb_w1 + nb_w0 + 3;
    CommDev xc = {};
    unsigned xstep = 0;
    if (cm) {
        // exchange layout: one 8-byte tagged word per 32 payload bits
        const long xvalues = (long)nb_w1 * GR_TILE_VALUES + (long)(nblocks - nb_w1) * 64;
        if ((int)sizeof(T) != cm->elem || xvalues * 2 * (long)sizeof(T) > cm->dev.slot_bytes || nblocks > cm->dev.nflags)
            return CCNN_E_RANGE;
        for (int r = 0; r < cm->dev.world; r++)
            if (!cm->dev.peer[r]) return CCNN_E_INVALID;   // not connected
        if (cm->any_shared) {
            // Ranks that share a GPU: a batch-sum kernel occupies compute units while it waits for
            // its peers, and a peer whose per-sample kernel finds no free unit would never get to
            // answer.  A one-workgroup exchange of one value first: no rank's batch sums start
            // before every rank's per-sample kernel has finished.
            void *rdv = cm->dev.peer[cm->dev.rank] + cm->dev.status_off + 64;
            const int e = cm->elem == 4 ? ccnn_allreduce_f32(cm, static_cast<float *>(rdv), 1, st)
                                        : ccnn_allreduce_f64(cm, static_cast<double *>(rdv), 1, st);
            if (e) return e;
        }
        xc = cm->dev;
        xstep = cm->next_step();
    }
    const bool c13 = d.n_classes == CCNN_CLASSES;
    launch("grad_reduce", c13 ? grad_reduce<T, CCNN_CLASSES> : grad_reduce<T, 0>,
           dim3(nblocks), dim3(GR_THREADS), 0, st,
           B, d.pout, d.off_w1, d.off_b1, d.off_w2, d.off_b2, w.pool, w.a1, w.d1, w.d2, w.slab, slab_rows, nb_w1, nb_w0,
           grad, upd, xc, xstep);
    CCNN_LAUNCH_CHECK();
    return 0;
}

template <typename T, bool FLIP>
int mbgd_batched(const T *params, const T *cubes, const int *onehot, long first, int B, int z, const ccnn_dims_t &d,
                 T *grad, UpdateRule<T> upd, double *loss_values, T *probs, MbgdWs<T> &w, hipStream_t st, ccnn_comm *cm)
{
    const int G = MbgdWs<T>::partial_rows(B, w.path, w.plan);
    if (w.path == StepPath::TILE) {
        if (int e = tile_train<T, FLIP>(params, cubes, onehot, first, B, d, w.plan, w.pool, w.a1, w.d1, w.d2, probs,
                                        loss_values, w.slab, st))
            return e;
    } else {
    if (int e = train_forward<T, FLIP>(params, cubes, onehot, first, B, z, w.pool, w.px, w.a1, w.d1, w.d2, probs,
                                       loss_values, w.ctr, st))
        return e;
    {
        const int btiles = (B + 15) / 16, waves = DP_THREADS / 64;
        int gy = (btiles + waves - 1) / waves;
        if (gy > 8) gy = 8;
        const int tpw = (btiles + gy * waves - 1) / (gy * waves);
        launch("dpool_bwd", dpool_bwd<T>, dim3((d.pout + 15) / 16, gy), dim3(DP_THREADS), 0, st, params + d.off_w1,
               (const T *)w.d1, (const T *)w.pool, w.dp, B, d.pout, tpw);
        CCNN_LAUNCH_CHECK();
    }
    {
        const ConvBwdLds<T> L(d.re, d.mre);
        const bool two = L.bytes(2) <= 150 * 1024;
        auto kern = two ? conv_bwd_batch<T, 2> : conv_bwd_batch<T, 1>;
        const size_t smem = L.bytes(two ? 2 : 1);
        if (int e = allow_smem(kern, smem)) return e;
        launch("conv_bwd_batch", kern, dim3(G), dim3(CB_THREADS), smem, st, cubes, first, (const T *)w.dp,
               (const T *)w.pool, (const unsigned char *)w.px, B, d.cube, d.re, d.mre, w.slab);
        CCNN_LAUNCH_CHECK();
    }
    }
    return launch_grad_reduce<T>(B, d, w, G, grad, upd, cm, st);
}

// grad != null: store the batch-summed gradient.  grad == null: apply `upd` in place.
template <typename T, bool FLIP>
int mbgd_impl(const T *params, const T *cubes, const int *onehot, long first, int B, int z, T *grad,
              UpdateRule<T> upd, double *loss_values, T *probs, void *ws, size_t ws_bytes,
              hipStream_t st, ccnn_comm *cm = nullptr)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    if (!params || !cubes || !onehot || !loss_values || !ws || B <= 0 || first < 0) return CCNN_E_INVALID;
    if (!grad && !upd.params) return CCNN_E_INVALID;
    const StepChoice sc = step_choice<T>(B, d, step_env());
    if (ws_bytes < MbgdWs<T>::bytes(B, d, sc)) return CCNN_E_WORKSPACE;
    MbgdWs<T> w(ws, B, d, sc);
    if (w.path != StepPath::FUSED) return mbgd_batched<T, FLIP>(params, cubes, onehot, first, B, z, d, grad, upd, loss_values, probs, w, st, cm);
    const FusedLayout L = fused_layout<T>(d.re, d.mre);
    // two workgroups per sample while that still fits the chip in one round (one workgroup per CU)
    const bool split = split_samples(B);
    const bool c13 = d.n_classes == CCNN_CLASSES;
    static const int dense_env = getenv("CCNN_FB_DENSE") ? atoi(getenv("CCNN_FB_DENSE")) : -1;   // diagnostics
    const bool dense = dense_env >= 0 ? dense_env != 0 : (!split && sizeof(T) == 4 && B > fb_cus());
    const bool fs = w.xch != nullptr;
    const auto fb = fs      ? (c13 ? sample_fwd_bwd<T, FLIP, FB_THREADS, CCNN_CLASSES, 2, 2, true> : sample_fwd_bwd<T, FLIP, FB_THREADS, 0, 2, 2, true>)
                    : split ? (c13 ? sample_fwd_bwd<T, FLIP, FB_THREADS, CCNN_CLASSES, 2> : sample_fwd_bwd<T, FLIP, FB_THREADS, 0, 2>)
                    : dense ? (c13 ? sample_fwd_bwd<T, FLIP, FB_THREADS, CCNN_CLASSES, 1, 4> : sample_fwd_bwd<T, FLIP, FB_THREADS, 0, 1, 4>)
                            : (c13 ? sample_fwd_bwd<T, FLIP, FB_THREADS, CCNN_CLASSES, 1> : sample_fwd_bwd<T, FLIP, FB_THREADS, 0, 1>);
    if (int e = allow_smem(fb, L.bytes)) return e;
    {
        launch("sample_fwd_bwd", fb, dim3(split ? 16 * ((B + 7) / 8) : B), dim3(FB_THREADS), L.bytes, st, params, cubes,
               onehot, first, d, w.pool, w.a1, w.d1, w.d2, probs, loss_values, w.slab, B, w.xch, fs ? next_pair_tag() : 0u,
               fs && getenv("CCNN_FB_TEST_MUTE_PARTNER") ? 1 : 0);
    }
    CCNN_LAUNCH_CHECK();
    return launch_grad_reduce<T>(B, d, w, B, grad, upd, cm, st);
}

}  // namespace ccnn

using namespace ccnn;

// ---------------------------------------------------------------------------
// Scratch of the entry point that allocates nothing (ccnn_mbgd_step_noalloc_f32): the
// per-sample rows the two training kernels hand to each other (pool, a1, delta1, delta2,
// conv-gradient tiles), which every other entry point takes from a caller-owned workspace.
// Every device that loads this library holds its own copy of the array, at its own address,
// until the process ends.  Host code must never pass the array's NAME to a kernel or a memset
// (on the host it is a stand-in, not a device address): scratch_on_current_device() resolves
// the address once per device and everything else uses that pointer.
static __device__ __align__(256) unsigned char scratch[CCNN_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int SCRATCH_MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[SCRATCH_MAX_DEVICES];   // [device] -> address of that device's copy of `scratch`

int scratch_on_current_device(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) return CCNN_E_RANGE;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));   // the current device's copy
        if (e != hipSuccess) return (int)e;
        if (!p) return (int)hipErrorInvalidSymbol;
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

#ifdef CCNN_FB_TIMING
extern "C" int ccnn_debug_fb_ticks(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fb_ticks), sizeof(unsigned long long) * 32);
}
extern "C" int ccnn_debug_gr_wg(unsigned long long *out)   // [1024][3]
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gr_wg), sizeof(unsigned long long) * 1024 * 3);
}
extern "C" int ccnn_debug_fb_wg(unsigned long long *out)   // [1024][9]
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fb_wg), sizeof(unsigned long long) * 1024 * 9);
}
#endif

extern "C" {

size_t ccnn_mbgd_workspace_bytes(int B, int z, int elem_size)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || B <= 0) return 0;
    const StepEnv env = step_env();
    return elem_size == 8 ? MbgdWs<double>::bytes(B, d, step_choice<double>(B, d, env))
                          : MbgdWs<float>::bytes(B, d, step_choice<float>(B, d, env));
}

int ccnn_mbgd_path(int B, int z, int elem_size)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || B <= 0 || (elem_size != 4 && elem_size != 8)) return CCNN_E_INVALID;
    const StepEnv env = step_env();
    return (int)(elem_size == 8 ? step_choice<double>(B, d, env).path : step_choice<float>(B, d, env).path);
}

int ccnn_mbgd_grad_f32(const float *params, const float *cubes, const int *onehot, long first, int B,
                       int z, float *grad, double *loss_values, float *probs, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    if (!grad) return CCNN_E_INVALID;
    return mbgd_impl<float, true>(params, cubes, onehot, first, B, z, grad, UpdateRule<float>{}, loss_values,
                                  probs, workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_mbgd_grad_f64(const double *params, const double *cubes, const int *onehot, long first, int B,
                       int z, double *grad, double *loss_values, double *probs, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    if (!grad) return CCNN_E_INVALID;
    return mbgd_impl<double, false>(params, cubes, onehot, first, B, z, grad, UpdateRule<double>{},
                                    loss_values, probs, workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_mbgd_step_f32(float *params, const float *cubes, const int *onehot, long first, int B, int z,
                       float lr, double *loss_values, void *workspace, size_t workspace_bytes, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params) return CCNN_E_INVALID;
    // plain-sum update, conv bias frozen (gcn_mbgd_fp32.cu:529, 561, 1052-1059)
    UpdateRule<float> upd{params, lr, 0, d.off_b0, d.off_w1};
    return mbgd_impl<float, true>(params, cubes, onehot, first, B, z, nullptr, upd, loss_values, nullptr,
                                  workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_mbgd_step_noalloc_f32(float *params, const float *cubes, const int *onehot, long first, int B, int z,
                               float lr, double *loss_values, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params || B <= 0) return CCNN_E_INVALID;
    if (MbgdWs<float>::bytes(B, d, step_choice<float>(B, d, step_env())) > (size_t)CCNN_NOALLOC_SCRATCH_BYTES) return CCNN_E_SCRATCH;
    void *ws = nullptr;
    if (int e = scratch_on_current_device(&ws)) return e;
    // Every row of the scratch a step reads is written by the same step first, on every path, so
    // nothing has to be cleared between calls and what an earlier call left elsewhere is never
    // read: the fused kernel writes pool, a1, d1, d2 and one conv-gradient row for exactly the B
    // samples grad_reduce then sums (the words its workgroup pairs trade are cleared by their readers,
    // and the array starts out zeroed); the sample-tile kernel writes the same per-sample rows and one
    // conv-gradient partial per TILE (a ragged last tile included), and grad_reduce sums exactly those
    // `tiles` rows; the layer kernels zero their work counters inside their own launches.
    // The regions' offsets move with B (MbgdWs carves them in order), which is harmless for the
    // same reason.
    UpdateRule<float> upd{params, lr, 0, d.off_b0, d.off_w1};
    return mbgd_impl<float, true>(params, cubes, onehot, first, B, z, nullptr, upd, loss_values, nullptr, ws,
                                  (size_t)CCNN_NOALLOC_SCRATCH_BYTES, (hipStream_t)stream);
}

int ccnn_mbgd_step_f64(double *params, const double *cubes, const int *onehot, long first, int B, int z,
                       double lr, double *loss_values, void *workspace, size_t workspace_bytes, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params) return CCNN_E_INVALID;
    UpdateRule<double> upd{params, lr, B, 0, 0};   // mean over the batch (gcn_mbgd_fp64.cu:519)
    return mbgd_impl<double, false>(params, cubes, onehot, first, B, z, nullptr, upd, loss_values, nullptr,
                                    workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_mbgd_step_dp_f32(ccnn_comm_t *comm, float *params, const float *cubes, const int *onehot, long first,
                          int B, int z, float lr, double *loss_values, void *workspace, size_t workspace_bytes,
                          void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params || !comm) return CCNN_E_INVALID;
    UpdateRule<float> upd{params, lr, 0, d.off_b0, d.off_w1};
    return mbgd_impl<float, true>(params, cubes, onehot, first, B, z, nullptr, upd, loss_values, nullptr,
                                  workspace, workspace_bytes, (hipStream_t)stream, comm);
}

int ccnn_mbgd_step_dp_f64(ccnn_comm_t *comm, double *params, const double *cubes, const int *onehot, long first,
                          int B, int z, double lr, int divisor, double *loss_values, void *workspace,
                          size_t workspace_bytes, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params || !comm || divisor <= 0) return CCNN_E_INVALID;
    UpdateRule<double> upd{params, lr, divisor, 0, 0};   // mean over the GLOBAL batch (hyper_version_mGPU_MBGD.cu:518, 550)
    return mbgd_impl<double, false>(params, cubes, onehot, first, B, z, nullptr, upd, loss_values, nullptr,
                                    workspace, workspace_bytes, (hipStream_t)stream, comm);
}

}  // extern "C"
