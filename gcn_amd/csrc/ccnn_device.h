// Device-side building blocks shared by the forward (inference) and minibatch
// training kernels: the 16x16x4 MFMA wrapper, LDS staging helpers, the conv
// forward/backward tile loops and the reference's small reductions.
#pragma once

#include "ccnn_common.h"

namespace ccnn {

// ---------------------------------------------------------------------------
// 16x16x4 MFMA wrapper.  Lane l supplies A[row l&15][k l>>4] and B[k l>>4][col l&15];
// it receives four results in column l&15 at rows Mma<T>::row(l, 0..3).
template <typename T>
struct Mma;
template <>
struct Mma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return ((lane >> 4) << 2) + r; }
};
template <>
struct Mma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + (r << 2); }
};

// LDS row strides.  16 (mod 32) elements lets the 16 lanes that walk a row's columns and the
// 16 lanes of the next row hit distinct banks; an odd stride does it for lanes that walk rows.
constexpr int LDWK = 16;    // conv filters: two tap-major tiles [172 taps][16 filters].  Lanes (col, kq)
                            // read tap 4s+kq, filter col: 16 kq + col covers all 32 banks once per
                            // half-wave (a filter-major [f][175] tile costs a 2-way conflict per read).
constexpr int WL_TILE = 172 * LDWK;  // tap 171 and filters 20..31 are zero
constexpr int WL_LEN = 2 * WL_TILE;
// Within a tap's row the 16 filters are stored at column (f ^ tap) & 15: the conv's reads (16
// lanes = 16 filters of one tap) stay a permutation of the row's banks, and the staging writes
// (consecutive lanes = consecutive taps of one filter, 16 words apart) no longer all land in two
// banks (a 32-way conflict) but spread over all of them.
__host__ __device__ inline int wl_index(int f, int k) { return (f >> 4) * WL_TILE + k * LDWK + ((f ^ k) & 15); }
constexpr int LDW1 = 112;   // W1 chunk       [64 p][112], fc_head_fwd (lanes walk n)
constexpr int LDP = 79;     // pool tile      [16 b][79]
constexpr int KSTEPS0 = (K + 3) / 4;   // 43 k-steps cover the 171 taps

__host__ __device__ inline int conv_mtiles(int re) { return (re + 15) >> 4; }
// zero-padded cube length: the windows of a padded 16-position tile end at 288 mt + 156, the
// shifted-filter tile below reads up to 18 (re - 1) + 227
__host__ __device__ inline int conv_cube_pad(int re) { return WSTEP * 16 * conv_mtiles(re) + 224; }

// Packed tile for the four filters that do not fill a second 16-wide MFMA tile.  Column 4 g + f'
// holds filter 16 + f' shifted down by 18 g taps (zeros elsewhere), so against a row that starts
// at cube offset 72 j the product yields filter 16 + f' at position 4 j + g: one MFMA chain of 57
// k-steps covers 64 positions x 4 filters, where a plain filter tile spends 4 x 43 k-steps with
// 12 of its 16 columns empty.
constexpr int WX_TAPS = 228;                    // 171 + 3 x 18 = 225 taps, whole k-steps
constexpr int KSTEPSX = WX_TAPS / 4;
constexpr int WP_LEN = WL_TILE + WX_TAPS * LDWK;   // filters 0..15 tap-major, then the packed tile


// Cooperative global -> LDS staging with U loads in flight per thread: all loads of
// a round are issued before the first LDS store waits on them (a plain
// load/store loop is compiled to one exposed memory round trip per element).
// put(e, v) stores element e wherever the layout wants it.
template <int U, int NT, typename T, typename Put>
__device__ __forceinline__ void stage(const T *__restrict__ src, int n, int tid, Put put)
{
    for (int base = 0; base < n; base += U * NT) {
        T tmp[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            // clamped, unconditional: a guarded load becomes a branch plus a full wait per element
            const int e = base + u * NT + tid;
            tmp[u] = src[e < n ? e : n - 1];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = base + u * NT + tid;
            if (e < n) put(e, tmp[u]);
        }
    }
}

// 16-byte vectors of T (float4 / double2) for aligned staging.
template <typename T>
struct Vec16 {
    static constexpr int W = 16 / sizeof(T);
    typedef T type __attribute__((ext_vector_type(16 / sizeof(T))));
};
// The reference's ceil-halving tree over n values (gcn_mbgd_fp64.cu:337-346: offset = ceil(n/2);
// t[i] += t[i + offset] while i + offset < n; then n = offset ..; 13 -> 7 -> 4 -> 2 -> 1), in
// registers.  The pairing depends on n, so each class count has its own unrolled tree and the
// run-time count picks one (a wave-uniform branch).
template <typename T, int N>
__device__ __forceinline__ T tree_fixed(const T *v)
{
    T t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = v[i];
    int len = N;
#pragma unroll
    for (int lvl = 0; lvl < 4; lvl++) {
        const int off = (len - 1) / 2 + 1;
        if (len >= 2) {
#pragma unroll
            for (int i = 0; i < N; i++)
                if (i + off < len) t[i] = t[i] + t[i + off];
        }
        len = off;
    }
    return t[0];
}
template <typename T>
__device__ __forceinline__ T tree_n(const T *v, int n)
{
    switch (n) {
    case 2: return tree_fixed<T, 2>(v);
    case 3: return tree_fixed<T, 3>(v);
    case 4: return tree_fixed<T, 4>(v);
    case 5: return tree_fixed<T, 5>(v);
    case 6: return tree_fixed<T, 6>(v);
    case 7: return tree_fixed<T, 7>(v);
    case 8: return tree_fixed<T, 8>(v);
    case 9: return tree_fixed<T, 9>(v);
    case 10: return tree_fixed<T, 10>(v);
    case 11: return tree_fixed<T, 11>(v);
    case 12: return tree_fixed<T, 12>(v);
    case 14: return tree_fixed<T, 14>(v);
    case 15: return tree_fixed<T, 15>(v);
    case 16: return tree_fixed<T, 16>(v);
    default: return tree_fixed<T, 13>(v);
    }
}

// Per-sample loss as the two MBGD references define it, split so that the 13
// per-class terms (the exp/log work) are computed by 13 lanes and only the cheap
// ordered accumulation is serial.
// fp64: -sum(y log p + (1-y) log(1-p))/13          (gcn_mbgd_fp64.cu:572-594)
__device__ __forceinline__ void loss_terms(double p, int lab, double *ta, double *tb)
{
    *ta = lab * log(p);
    *tb = (1 - lab) * log(1 - p);
}
__device__ __forceinline__ double loss_sum(const double *ta, const double *tb, const int *lab, int no)
{
    double tmp = 0.0;
    for (int i = 0; i < no; i++) tmp = tmp + ta[i] + tb[i];
    return -tmp / no;
}
// fp32: class-balanced logistic expression on the probabilities, float inner
// arithmetic, double accumulators                   (gcn_mbgd_fp32.cu:578-604)
__device__ __forceinline__ void loss_terms(float o, int lab, float *ta, float *tb)
{
    const int ge = (o >= 0);
    *ta = o * (lab - ge) - logf(1 + expf(o - 2 * o * ge));
    *tb = 0;
}
__device__ __forceinline__ double loss_sum(const float *ta, const float *tb, const int *lab, int no)
{
    int count_pos = 0, count_neg = 0;
    double pos = 0, neg = 0;
    for (int i = 0; i < no; i++) {
        if (lab[i] == 1) {
            count_pos++;
            pos -= ta[i];
        } else if (lab[i] == 0) {
            count_neg++;
            neg -= ta[i];
        }
    }
    return (pos * count_neg / no) * 1 + (neg * count_pos / no);
}
// The same ordered sums with class i's terms in the registers of lane i of the calling wave (all 64
// lanes active): the class index is wave-uniform, so a term arrives through v_readlane and the
// fp32 form's branches are scalar.  The form above pays two dependent LDS round trips per class -
// 1.2 us for 13 classes, on one lane, with the rest of its wave waiting.
__device__ __forceinline__ float lane_value(float v, int i)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
}
__device__ __forceinline__ double lane_value(double v, int i)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), i),
                            __builtin_amdgcn_readlane(__double2loint(v), i));
}
__device__ __forceinline__ double loss_sum_lanes(double ta, double tb, int lab, int no)
{
    double tmp = 0.0;
    for (int i = 0; i < no; i++) tmp = tmp + lane_value(ta, i) + lane_value(tb, i);
    return -tmp / no;
}
__device__ __forceinline__ double loss_sum_lanes(float ta, float tb, int lab, int no)
{
    int count_pos = 0, count_neg = 0;
    double pos = 0, neg = 0;
    for (int i = 0; i < no; i++) {
        const int lb = __builtin_amdgcn_readlane(lab, i);
        const float t = lane_value(ta, i);
        if (lb == 1) {
            count_pos++;
            pos -= t;
        } else if (lb == 0) {
            count_neg++;
            neg -= t;
        }
    }
    return (pos * count_neg / no) * 1 + (neg * count_pos / no);
}

// Sum of 16 per-lane values over the QL (32 or 64) lanes of a group with ~1 shuffle
// per value instead of log2(QL): at each step a lane keeps half of its values and
// trades the other half with its partner, so the sums end up scattered: the lane
// returns the complete sum of value index reduce_scatter_index<QL>(lane) (lanes that
// share an index hold the same sum).
template <int QL>
__device__ __forceinline__ int reduce_scatter_index(int lane)
{
    return QL == 32 ? (lane >> 1) & 15 : (lane >> 2) & 15;
}
template <int QL, typename T>
__device__ __forceinline__ T reduce_scatter16(T (&v)[16], int lane)
{
    static_assert(QL == 32 || QL == 64, "group of 32 or 64 lanes");
    {
        const bool hi = lane & (QL / 2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const T keep = hi ? v[i + 8] : v[i], send = hi ? v[i] : v[i + 8];
            v[i] = keep + __shfl_xor(send, QL / 2);
        }
    }
    {
        const bool hi = lane & (QL / 4);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const T keep = hi ? v[i + 4] : v[i], send = hi ? v[i] : v[i + 4];
            v[i] = keep + __shfl_xor(send, QL / 4);
        }
    }
    {
        const bool hi = lane & (QL / 8);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const T keep = hi ? v[i + 2] : v[i], send = hi ? v[i] : v[i + 2];
            v[i] = keep + __shfl_xor(send, QL / 8);
        }
    }
    {
        const bool hi = lane & (QL / 16);
        const T keep = hi ? v[1] : v[0], send = hi ? v[0] : v[1];
        v[0] = keep + __shfl_xor(send, QL / 16);
    }
#pragma unroll
    for (int off = QL / 32; off > 0; off >>= 1) v[0] += __shfl_xor(v[0], off);
    return v[0];
}

// ---------------------------------------------------------------------------
// tanh + max-pool of one 16-position conv tile straight from the MFMA accumulators
// (window 2, clipped at re, strict '<' so the first maximum wins,
// gcn_mbgd_fp64.cu:237-268).  Position pairs (2m, 2m+1) never straddle a tile (tiles
// start at multiples of 16).  In the f32 C/D map a lane holds rows 4q..4q+3, so both
// members of a pair are in the same lane; in the f64 map it holds rows q, q+4, q+8,
// q+12 and the partner row lives in lane^16.  put(f, m, value, winning position).
// WIN = false (callers that drop the winning position: inference): the larger conv output is picked
// first and activated once.  Bit-identical values - the fp32 act() is monotone non-decreasing over
// all finite floats (tools/probes/act_monotone.hip checks every pair of neighbouring floats on the
// hardware), and so is the rounded bias add - for half the exp / rcp work, which on this part runs
// instead of, not beside, the matrix instructions.
template <bool WIN = true, typename Put>
__device__ __forceinline__ void pool_tile(const Mma<float>::acc_t &acc, float bias, int f, bool fvalid, int t0,
                                          int re, int lane, Put put)
{
    const int q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const int t = t0 + 4 * q + r;
        if (!WIN) {
            if (fvalid && t < re) {
                float x = acc[r] + bias;
                if (t + 1 < re) x = fmaxf(x, acc[r + 1] + bias);
                put(f, t >> 1, act(x), t);
            }
            continue;
        }
        if (fvalid && t < re) {
            float best = act(acc[r] + bias);
            int bt = t;
            if (t + 1 < re) {
                const float vn = act(acc[r + 1] + bias);
                if (best < vn) {
                    best = vn;
                    bt = t + 1;
                }
            }
            put(f, t >> 1, best, bt);
        }
    }
}
template <bool WIN = true, typename Put>
__device__ __forceinline__ void pool_tile(const Mma<double>::acc_t &acc, double bias, int f, bool fvalid, int t0,
                                          int re, int lane, Put put)
{
    const int q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int t = t0 + q + 4 * r;
        const double v = act(acc[r] + bias);
        const double vn = __shfl_xor(v, 16);     // row q^1 of the same column
        if (fvalid && !(q & 1) && t < re) {
            double best = v;
            int bt = t;
            if (t + 1 < re && best < vn) {
                best = vn;
                bt = t + 1;
            }
            put(f, t >> 1, best, bt);
        }
    }
}

// Conv forward + tanh + pool for LDS-resident samples: per sample the
// [re x 171] x [171 x 20] product is ceil(re/16) x 2 MFMA tiles.  The nsamp x ceil(re/16)
// position tiles are dealt round-robin to the 4 waves; each tile runs both filter halves so
// an A fragment feeds two MFMAs.  A comes straight from the cube (window t starts at 18 t:
// no im2col).  wl holds the two tap-major filter tiles [172][LDWK]; every cube is
// zero-padded to conv_cube_pad(re) and they are cube_stride apart.
// put(sample, f, m, value, winning position).
template <typename T, typename Put>
__device__ __forceinline__ void conv_pool_tiles(const T *cube, int cube_stride, int nsamp, const T *wl, int re,
                                                T bias0, T bias1, int lane, int wave, Put put, int nwaves = 4,
                                                bool split = false)
{
    const int col = lane & 15, kq = lane >> 4;
    const int mt = conv_mtiles(re);
    // B operand of k-step s: tap 4 s + kq, filter col, stored at column (col ^ tap) & 15
    // (wl_index) = (col ^ kq) ^ (4 s & 12): one pointer per s & 3
    const T *bq[4];
#pragma unroll
    for (int u = 0; u < 4; u++) bq[u] = wl + kq * LDWK + ((col ^ kq) ^ (4 * u));
    constexpr int KMAIN = KSTEPS0 & ~3;
    if (split) {
        // one filter half per job: twice the jobs for workgroups with more waves than position tiles
        for (int job = wave; job < 2 * nsamp * mt; job += nwaves) {
            const int half = job & 1, pj = job >> 1;
            const int sidx = pj / mt, tile = pj - sidx * mt;
            const T *ap = cube + sidx * cube_stride + WSTEP * (tile * 16 + col) + kq;
            const int hoff = half ? WL_TILE : 0;
            typename Mma<T>::acc_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};   // two chains: even / odd k-steps
            for (int s = 0; s < KMAIN; s += 4) {
                T a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    a[u] = ap[4 * (s + u)];
                    b[u] = bq[u][hoff + 4 * (s + u) * LDWK];
                }
                acc = Mma<T>::mma(a[0], b[0], acc);
                acc2 = Mma<T>::mma(a[1], b[1], acc2);
                acc = Mma<T>::mma(a[2], b[2], acc);
                acc2 = Mma<T>::mma(a[3], b[3], acc2);
            }
#pragma unroll
            for (int u = 0; u < KSTEPS0 - KMAIN; u++)
                acc = Mma<T>::mma(ap[4 * (KMAIN + u)], bq[u][hoff + 4 * (KMAIN + u) * LDWK], acc);
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] += acc2[r];
            auto put1 = [&](int f, int m, T v, int bt) { put(sidx, f, m, v, bt); };
            if (half) pool_tile(acc, bias1, 16 + col, 16 + col < F, tile * 16, re, lane, put1);
            else pool_tile(acc, bias0, col, true, tile * 16, re, lane, put1);
        }
        return;
    }
    for (int job = wave; job < nsamp * mt; job += nwaves) {
        const int sidx = job / mt, tile = job - sidx * mt;
        const T *ap = cube + sidx * cube_stride + WSTEP * (tile * 16 + col) + kq;
        typename Mma<T>::acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int s = 0; s < KMAIN; s += 4) {   // operands of four k-steps in flight
            T a[4], b0[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = ap[4 * (s + u)];
                b0[u] = bq[u][4 * (s + u) * LDWK];
                b1[u] = bq[u][WL_TILE + 4 * (s + u) * LDWK];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                acc0 = Mma<T>::mma(a[u], b0[u], acc0);
                acc1 = Mma<T>::mma(a[u], b1[u], acc1);
            }
        }
#pragma unroll
        for (int u = 0; u < KSTEPS0 - KMAIN; u++) {
            const T a = ap[4 * (KMAIN + u)];
            acc0 = Mma<T>::mma(a, bq[u][4 * (KMAIN + u) * LDWK], acc0);
            acc1 = Mma<T>::mma(a, bq[u][WL_TILE + 4 * (KMAIN + u) * LDWK], acc1);
        }
        auto put1 = [&](int f, int m, T v, int bt) { put(sidx, f, m, v, bt); };
        pool_tile(acc0, bias0, col, true, tile * 16, re, lane, put1);
        pool_tile(acc1, bias1, 16 + col, 16 + col < F, tile * 16, re, lane, put1);
    }
}

// The same for a workgroup that owns the filters fbase .. fbase + nf - 1 (nf <= 16) of ONE sample:
// only the first tile of wl exists, its columns 0 .. nf - 1 hold those filters (columns past them
// may hold anything: a column's values never leave the column, and nothing is put for it), tap 171
// is zero.  One job per position tile.  put(filter, m, value, winning position).
template <typename T, typename Put>
__device__ __forceinline__ void conv_pool_own(const T *cube, const T *wl, int re, T bias, int fbase, int nf, int lane,
                                              int wave, int nwaves, Put put)
{
    const int col = lane & 15, kq = lane >> 4;
    const int mt = conv_mtiles(re);
    const T *bq[4];
#pragma unroll
    for (int u = 0; u < 4; u++) bq[u] = wl + kq * LDWK + ((col ^ kq) ^ (4 * u));
    constexpr int KMAIN = KSTEPS0 & ~3;
    for (int tile = wave; tile < mt; tile += nwaves) {
        const T *ap = cube + WSTEP * (tile * 16 + col) + kq;
        typename Mma<T>::acc_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};   // two chains: even / odd k-steps
        for (int s = 0; s < KMAIN; s += 4) {
            T a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = ap[4 * (s + u)];
                b[u] = bq[u][4 * (s + u) * LDWK];
            }
            acc = Mma<T>::mma(a[0], b[0], acc);
            acc2 = Mma<T>::mma(a[1], b[1], acc2);
            acc = Mma<T>::mma(a[2], b[2], acc);
            acc2 = Mma<T>::mma(a[3], b[3], acc2);
        }
#pragma unroll
        for (int u = 0; u < KSTEPS0 - KMAIN; u++) acc = Mma<T>::mma(ap[4 * (KMAIN + u)], bq[u][4 * (KMAIN + u) * LDWK], acc);
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] += acc2[r];
        pool_tile(acc, bias, fbase + col, col < nf, tile * 16, re, lane, put);
    }
}

// One MFMA chain of KS k-steps as two interleaved accumulators (even / odd k-steps).  The
// operands of the next four k-steps are fetched before the MFMAs of the current four are
// issued, so a wave always has LDS reads in flight behind its matrix instructions.
// ap walks A along k (stride 1), bp walks B (stride LDWK).
template <typename T, int KS>
__device__ __forceinline__ typename Mma<T>::acc_t mma_chain(const T *ap, const T *bp)
{
    constexpr int NB4 = (KS + 3) / 4;   // batches of up to four k-steps
    typename Mma<T>::acc_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    T a[2][4], b[2][4];
    auto load = [&](int bi, int set) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (4 * bi + u < KS) {
                a[set][u] = ap[4 * (4 * bi + u)];
                b[set][u] = bp[4 * (4 * bi + u) * LDWK];
            }
        }
    };
    load(0, 0);
#pragma unroll
    for (int bi = 0; bi < NB4; bi++) {
        const int set = bi & 1;
        if (bi + 1 < NB4) load(bi + 1, set ^ 1);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (4 * bi + u < KS) {
                if (u & 1) acc2 = Mma<T>::mma(a[set][u], b[set][u], acc2);
                else acc = Mma<T>::mma(a[set][u], b[set][u], acc);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] += acc2[r];
    return acc;
}

// NT chains that share their B operand (NT position tiles against one filter tile), run together:
// a k-step's B fragment is read once and feeds NT matrix instructions, so the LDS serves
// (NT + 1) / NT operand reads per instruction instead of two, and the NT accumulators give the
// matrix pipe independent work without splitting a chain in two.  ap[t] walks tile t's A along k.
template <typename T, int KS, int NT>
__device__ __forceinline__ void mma_chain_n(const T *const (&ap)[NT], const T *bp, typename Mma<T>::acc_t (&acc)[NT])
{
    constexpr int NB4 = (KS + 3) / 4;   // batches of up to four k-steps
    T a[2][4][NT], b[2][4];
    auto load = [&](int bi, int set) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (4 * bi + u < KS) {
                b[set][u] = bp[4 * (4 * bi + u) * LDWK];
#pragma unroll
                for (int t = 0; t < NT; t++) a[set][u][t] = ap[t][4 * (4 * bi + u)];
            }
        }
    };
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = typename Mma<T>::acc_t{0, 0, 0, 0};
    load(0, 0);
#pragma unroll
    for (int bi = 0; bi < NB4; bi++) {
        const int set = bi & 1;
        if (bi + 1 < NB4) load(bi + 1, set ^ 1);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (4 * bi + u < KS) {
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = Mma<T>::mma(a[set][u][t], b[set][u], acc[t]);
            }
        }
    }
}

// Orders this wave's LDS traffic: everything its lanes wrote before is visible to all its
// lanes after (the LDS serves one wave's requests in order; this keeps the compiler from
// moving accesses across and waits for outstanding ones).  No other wave is involved.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Workgroup barrier that orders LDS traffic only: global stores still in flight (results nobody in
// the workgroup reads back) are not waited for, where __syncthreads() stalls every wave until the
// last store has been acknowledged by memory.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Conv forward + tanh + pool of ONE LDS-resident sample by ONE wave: ceil(re/16) chains of 43
// k-steps for filters 0..15 (rows = 16 positions) and ceil(ceil(re/4)/16) chains of 57 k-steps
// for filters 16..19 on the packed tile (rows = 16 consecutive j; columns = 4 shifts x 4
// filters, so both members of a pool pair sit in one row, 4 columns apart).
// wl = [172][LDWK] filters 0..15 | [228][LDWK] packed tile; cube zero-padded to
// conv_cube_pad(re).  put(f, m, value, winning position).
// Chains per sample: ceil(re/16) over filters 0..15, then the packed-tile chains.
__host__ __device__ inline int conv_chains(int re) { return conv_mtiles(re) + (((re - 1) >> 2) >> 4) + 1; }

// Chain c of a sample (c < conv_chains(re)) by one wave.
template <typename T, bool WIN = true, typename Put>
__device__ __forceinline__ void conv_pool_chain(const T *cube, const T *wl, int re, int c, T bias0, T biasx, int lane,
                                                Put put)
{
    const int col = lane & 15, kq = lane >> 4;
    const int mt = conv_mtiles(re);
    const T *bp0 = wl + kq * LDWK + col;
    if (c < mt) {
        const T *ap = cube + WSTEP * (c * 16 + col) + kq;
        const typename Mma<T>::acc_t acc = mma_chain<T, KSTEPS0>(ap, bp0);
        pool_tile<WIN>(acc, bias0, col, true, c * 16, re, lane, put);
        return;
    }
    const int tau = c - mt, jmax = (re - 1) >> 2;
    const int g = col >> 2, fl = col & 3;
    const int j = min(tau * 16 + col, jmax);   // clamped: rows past the end are discarded below
    const T *ap = cube + 4 * WSTEP * j + kq;
    const typename Mma<T>::acc_t acc = mma_chain<T, KSTEPSX>(ap, bp0 + WL_TILE);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int t = 4 * (tau * 16 + Mma<T>::row(lane, r)) + g;
        const T v = act(acc[r] + biasx);
        const T vn = __shfl_xor(v, 4);           // same row, shift g ^ 1: the pool partner
        if (!(g & 1) && t < re) {
            T best = v;
            int bt = t;
            if (t + 1 < re && best < vn) {       // strict '<': the first maximum wins
                best = vn;
                bt = t + 1;
            }
            put(16 + fl, t >> 1, best, bt);
        }
    }
}

// Position tiles t0 .. t0 + NT - 1 of filters 0..15 as one group of chains (mma_chain_n).
template <typename T, int NT, bool WIN = true, typename Put>
__device__ __forceinline__ void conv_pool_group(const T *cube, const T *wl, int re, int t0, T bias0, int lane, Put put)
{
    const int col = lane & 15, kq = lane >> 4;
    const T *ap[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) ap[t] = cube + WSTEP * ((t0 + t) * 16 + col) + kq;
    typename Mma<T>::acc_t acc[NT];
    mma_chain_n<T, KSTEPS0, NT>(ap, wl + kq * LDWK + col, acc);
#pragma unroll
    for (int t = 0; t < NT; t++) pool_tile<WIN>(acc[t], bias0, col, true, (t0 + t) * 16, re, lane, put);
}

// All chains of one sample by one wave: the position tiles of filters 0..15 in groups of three
// (or two) that share the filter fragments, then the packed-tile chains.
template <typename T, bool WIN = true, typename Put>
__device__ __forceinline__ void conv_pool_wave(const T *cube, const T *wl, int re, T bias0, T biasx, int lane,
                                               Put put)
{
    const int mt = conv_mtiles(re), nc = conv_chains(re);
    int t = 0;
    while (mt - t >= 3 && mt - t != 4) {
        conv_pool_group<T, 3, WIN>(cube, wl, re, t, bias0, lane, put);
        t += 3;
    }
    while (mt - t >= 2) {
        conv_pool_group<T, 2, WIN>(cube, wl, re, t, bias0, lane, put);
        t += 2;
    }
    for (int c = t; c < nc; c++) conv_pool_chain<T, WIN>(cube, wl, re, c, bias0, biasx, lane, put);
}

// Stages the 20 x 171 filters as conv_pool_wave wants them (zero taps and columns included), by a
// workgroup of nt threads.  The fp32 reference walks the filter backwards in the forward pass only
// (gcn_mbgd_fp32.cu:241-243): FLIP stores tap k at 170-k.  Needs a barrier after.
template <typename T, bool FLIP>
__device__ __forceinline__ void stage_filters_packed(const T *__restrict__ w0, T *wl, int tid, int nt)
{
    for (int e = tid; e < WP_LEN; e += nt) wl[e] = 0;
    __syncthreads();
    for (int base = 0; base < F * K; base += 8 * nt) {
        T tmp[8];
#pragma unroll
        for (int u = 0; u < 8; u++) tmp[u] = w0[min(base + u * nt + tid, F * K - 1)];   // clamped, unconditional
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = base + u * nt + tid;
            if (e < F * K) {
                const int f = e / K, k = e - f * K;
                const int kk = FLIP ? K - 1 - k : k;
                if (f < 16) {
                    wl[kk * LDWK + f] = tmp[u];
                } else {
#pragma unroll
                    for (int g = 0; g < 4; g++) wl[WL_TILE + (kk + WSTEP * g) * LDWK + 4 * g + (f - 16)] = tmp[u];
                }
            }
        }
    }
}

}  // namespace ccnn
