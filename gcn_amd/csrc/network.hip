// Batched forward / minibatch-gradient / update kernels of the Cube CNN and their
// C entry points.  One launch per layer over the whole batch replaces the
// reference's nine launches per sample (gcn_mbgd_fp64.cu:931-1014), and every
// contraction runs on the matrix cores: v_mfma_f32_16x16x4_f32 for the fp32
// variant, v_mfma_f64_16x16x4_f64 for fp64 (same A/B lane maps, different C/D map).
//
// Kernel map (reference kernel -> here)
//   convolution + maxpooling            -> conv_pool_fwd  (implicit GEMM: positions x taps x filters)
//   fully_connect                       -> fc1_partial + head (K split in 64-wide chunks)
//   output_and_dvalue, loss_function,
//   bp_fully_connect(out->hidden)       -> head
//   bp_fully_connect(hidden->pool)      -> fc1_bwd_delta (delta) and grad_reduce (dW1 = pool^T d1)
//   bp_maxpooling + bp_convolution      -> conv_bwd       (GEMM: filters x positions x taps)
//   update_params_row/col               -> grad_reduce (batch sums, optional fused update) + update
#include <map>
#include <mutex>
#include <utility>

#include "ccnn_common.h"
#include "profile.h"

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

// LDS row strides.  A stride of 15 (mod 32) elements lets 16 lanes that walk rows
// and the 16 lanes one element further hit distinct banks (one pair excepted); 16
// (mod 32) does the same for lanes that walk columns of two adjacent rows.
constexpr int LDW0 = 175;   // conv filters   [F][175], 171 taps + zero pad
constexpr int LDW1 = 112;   // W1 chunk       [64 p][112], fc1_partial (lanes walk n)
constexpr int LDP = 79;     // pool tile      [16 b][79]
constexpr int LDW2 = 111;   // W1 chunk       [64 p][111], fc1_bwd_delta (lanes walk p)
constexpr int KSTEPS0 = (K + 3) / 4;   // 43 k-steps cover the 171 taps

__host__ __device__ inline int conv_mtiles(int re) { return (re + 15) >> 4; }
__host__ __device__ inline int conv_cube_pad(int re) { return WSTEP * 16 * conv_mtiles(re) + 160; }
__host__ __device__ inline int odd_up(int v) { return v | 1; }
// smallest stride >= n that is 15 (mod 32)
__host__ __device__ inline int stride15(int n) { return n + ((15 - n) % 32 + 32) % 32; }

// ---------------------------------------------------------------------------
// convolution + tanh + max-pool.  One sample at a time per workgroup (grid-stride
// over samples); per sample the [re x 171] x [171 x 20] product is tiled as
// ceil(re/16) x 2 MFMA tiles, one 16-position tile (both filter halves) per wave.
// A comes straight from the LDS-resident cube (window t starts at 18 t: no im2col).
// pool[row][f*mre + m], pidx[row][f*mre + m] = winning position t in [0, re).
template <typename T, bool FLIP>
__global__ __launch_bounds__(256) void conv_pool_fwd(const T *__restrict__ params,
                                                     const T *__restrict__ cubes, long first, int nrows,
                                                     int cube_len, int re, int mre, int off_b0,
                                                     T *__restrict__ pool, int *__restrict__ pidx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int mt = conv_mtiles(re);
    const int cube_pad = conv_cube_pad(re);
    const int lda = odd_up(re);
    T *cube = reinterpret_cast<T *>(smem);
    T *wl = cube + cube_pad;
    T *a0 = wl + F * LDW0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int pout = F * mre;

    for (int e = tid; e < F * K; e += 256) {
        const int f = e / K, k = e - f * K;
        // the fp32 reference walks the filter backwards (gcn_mbgd_fp32.cu:241-243)
        wl[f * LDW0 + (FLIP ? K - 1 - k : k)] = params[e];
    }
    if (tid < F * 4) wl[(tid >> 2) * LDW0 + K + (tid & 3)] = 0;
    for (int e = cube_len + tid; e < cube_pad; e += 256) cube[e] = 0;
    const int f1 = 16 + col;
    const bool v1 = f1 < F;
    const T bias0 = params[off_b0 + col];
    const T bias1 = v1 ? params[off_b0 + f1] : T(0);

    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        __syncthreads();   // everyone is done with the previous sample's cube and a0
        const T *src = cubes + (first + row) * (long)cube_len;
        for (int e = tid; e < cube_len; e += 256) cube[e] = src[e];
        __syncthreads();
        for (int tile = wave; tile < mt; tile += 4) {
            const T *ap = cube + WSTEP * (tile * 16 + col) + kq;
            const T *bp0 = wl + col * LDW0 + kq;
            const T *bp1 = wl + (v1 ? f1 : F - 1) * LDW0 + kq;
            typename Mma<T>::acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll 4
            for (int s = 0; s < KSTEPS0; s++) {
                const T a = ap[4 * s];
                const T b0 = bp0[4 * s];
                const T b1 = v1 ? bp1[4 * s] : T(0);
                acc0 = Mma<T>::mma(a, b0, acc0);
                acc1 = Mma<T>::mma(a, b1, acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int t = tile * 16 + Mma<T>::row(lane, r);
                if (t < re) {
                    a0[col * lda + t] = act(acc0[r] + bias0);
                    if (v1) a0[f1 * lda + t] = act(acc1[r] + bias1);
                }
            }
        }
        __syncthreads();
        for (int o = tid; o < pout; o += 256) {
            const int f = o / mre, m = o - f * mre;
            const int t = 2 * m;
            T best = a0[f * lda + t];
            int bt = t;
            if (t + 1 < re) {
                const T vn = a0[f * lda + t + 1];
                if (best < vn) {   // strict '<': first maximum wins (gcn_mbgd_fp64.cu:257)
                    best = vn;
                    bt = t + 1;
                }
            }
            pool[row * pout + o] = best;
            if (pidx) pidx[row * pout + o] = bt;
        }
    }
}

template <typename T>
size_t conv_smem(const ccnn_dims_t &d)
{
    return (size_t)(conv_cube_pad(d.re) + F * LDW0 + F * odd_up(d.re)) * sizeof(T);
}

// ---------------------------------------------------------------------------
// FC1 partial products: z1part[c][row][n] = sum_{p in chunk c} pool[row][p] * W1[n + 100 p].
// grid (chunks, ceil(nrows/16)); a [16 x 64] x [64 x 100] product per workgroup,
// the 7 neuron tiles spread over the 4 waves.
template <typename T>
__global__ __launch_bounds__(256) void fc1_partial(const T *__restrict__ W1,
                                                   const T *__restrict__ pool, int nrows, int pout,
                                                   T *__restrict__ z1part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *wl = reinterpret_cast<T *>(smem);   // [64][LDW1]
    T *pt = wl + PCHUNK * LDW1;            // [16][LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int c = blockIdx.x;
    const int r0 = blockIdx.y * 16;
    const int p0 = c * PCHUNK;
    const int pmax = min(PCHUNK, pout - p0);
    for (int e = tid; e < PCHUNK * LDW1; e += 256) {
        const int p = e / LDW1, n = e - p * LDW1;
        wl[e] = (p < pmax && n < H) ? W1[(long)(p0 + p) * H + n] : T(0);
    }
    for (int e = tid; e < 16 * PCHUNK; e += 256) {
        const int b = e >> 6, p = e & 63;
        pt[b * LDP + p] = (r0 + b < nrows && p < pmax) ? pool[(long)(r0 + b) * pout + p0 + p] : T(0);
    }
    __syncthreads();
    const int n0 = wave * 16, n1 = (wave + 4) * 16;
    const bool two = n1 < H;   // waves 0..2 own two neuron tiles, wave 3 one
    const T *ap = pt + col * LDP + kq;
    const T *bp0 = wl + kq * LDW1 + n0 + col;
    const T *bp1 = wl + kq * LDW1 + (two ? n1 : n0) + col;
    typename Mma<T>::acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll 4
    for (int s = 0; s < PCHUNK / 4; s++) {
        const T a = ap[4 * s];
        acc0 = Mma<T>::mma(a, bp0[4 * s * LDW1], acc0);
        if (two) acc1 = Mma<T>::mma(a, bp1[4 * s * LDW1], acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int b = r0 + Mma<T>::row(lane, r);
        if (b < nrows) {
            T *dst = z1part + ((long)c * nrows + b) * H;
            if (n0 + col < H) dst[n0 + col] = acc0[r];
            if (two && n1 + col < H) dst[n1 + col] = acc1[r];
        }
    }
}

template <typename T>
constexpr size_t fc1_smem()
{
    return (size_t)(PCHUNK * LDW1 + 16 * LDP) * sizeof(T);
}

// The reference's ceil-halving tree over 13 values (13 -> 7 -> 4 -> 2 -> 1), in
// registers: t[i] += t[i+7] (i<6); t[i] += t[i+4] (i<3); t[i] += t[i+2] (i<2); t[0] += t[1].
template <typename T>
__device__ __forceinline__ T tree13(const T *v)
{
    T t[O];
#pragma unroll
    for (int i = 0; i < O; i++) t[i] = v[i];
#pragma unroll
    for (int i = 0; i < 6; i++) t[i] = t[i] + t[i + 7];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = t[i] + t[i + 4];
    t[0] = t[0] + t[2];
    t[1] = t[1] + t[3];
    return t[0] + t[1];
}

// Per-sample loss as the two MBGD references define it.
// fp64: -sum(y log p + (1-y) log(1-p))/13          (gcn_mbgd_fp64.cu:572-594)
__device__ __forceinline__ double sample_loss(const double *p, const int *lab)
{
    double tmp = 0.0;
    for (int i = 0; i < O; i++) tmp = tmp + lab[i] * log(p[i]) + (1 - lab[i]) * log(1 - p[i]);
    return -tmp / O;
}
// fp32: class-balanced logistic expression on the probabilities, float inner
// arithmetic, double accumulators                   (gcn_mbgd_fp32.cu:578-604)
__device__ __forceinline__ double sample_loss(const float *p, const int *lab)
{
    int count_pos = 0, count_neg = 0;
    double pos = 0, neg = 0;
    for (int i = 0; i < O; i++) {
        const float o = p[i];
        const int ge = (o >= 0);
        const float term = o * (lab[i] - ge) - logf(1 + expf(o - 2 * o * ge));
        if (lab[i] == 1) {
            count_pos++;
            pos -= term;
        } else if (lab[i] == 0) {
            count_neg++;
            neg -= term;
        }
    }
    return (pos * count_neg / O) * 1 + (neg * count_pos / O);
}

// ---------------------------------------------------------------------------
// FC1 epilogue + output layer + softmax (+ loss, delta2, delta1 when onehot given)
// + count_err argmax.  128 threads, grid-stride over samples; W2/b2/b1 are staged
// in LDS once per workgroup.
template <typename T>
__global__ __launch_bounds__(128) void head(const T *__restrict__ params, int off_b1, int off_w2,
                                            int off_b2, const T *__restrict__ z1part, int nch,
                                            int nrows, const int *__restrict__ onehot,
                                            T *__restrict__ a1out, T *__restrict__ d1out,
                                            T *__restrict__ d2out, T *__restrict__ probs,
                                            int *__restrict__ argmax,
                                            double *__restrict__ loss_values)
{
    __shared__ T w2s[H * O];
    __shared__ T a1s[H];
    __shared__ T es[16];
    __shared__ T ps[16];
    __shared__ T d2s[16];
    const int n = threadIdx.x;
    for (int e = n; e < H * O; e += 128) w2s[e] = params[off_w2 + e];
    const T b1 = n < H ? params[off_b1 + n] : T(0);
    // output unit o = n >> 3 (13 x 8 threads), each summing every 8th hidden unit
    const int o8 = n >> 3, j8 = n & 7;
    const T b2 = o8 < O ? params[off_b2 + o8] : T(0);

    for (long b = blockIdx.x; b < nrows; b += gridDim.x) {
        __syncthreads();   // previous sample's a1s/es/ps/d2s readers are done (and w2s is staged)
        T a = 0;
        if (n < H) {
            T zsum = 0;
            for (int c = 0; c < nch; c++) zsum = zsum + z1part[((long)c * nrows + b) * H + n];
            a = 2 / (1 + 1 / rexp((zsum + b1) * 2)) - 1;   // gcn_mbgd_fp64.cu:302
            a1s[n] = a;
            if (a1out) a1out[b * H + n] = a;
        }
        __syncthreads();
        {
            T mid = 0;
            if (o8 < O)
                for (int i = j8; i < H; i += 8) mid = mid + w2s[o8 + i * O] * a1s[i];
            mid += __shfl_xor(mid, 4);
            mid += __shfl_xor(mid, 2);
            mid += __shfl_xor(mid, 1);
            if (o8 < O && j8 == 0) es[o8] = rexp(mid + b2);
        }
        __syncthreads();
        if (n < O) {
            const T p = es[n] / tree13(es);
            ps[n] = p;
            if (probs) probs[b * O + n] = p;
            if (onehot) {
                const T d = (p - onehot[b * O + n]) / O;   // gcn_mbgd_fp64.cu:352
                d2s[n] = d;
                d2out[b * O + n] = d;
            }
        }
        __syncthreads();
        if (n == 127) {
            if (argmax) {   // count_err: running max starts at 0, strict '>' (gcn_mbgd_fp64.cu:636-647)
                T mx = 0;
                int idx = 0;
                for (int i = 0; i < O; i++)
                    if (ps[i] > mx) {
                        mx = ps[i];
                        idx = i;
                    }
                argmax[b] = idx;
            }
            if (onehot && loss_values) loss_values[b] = sample_loss(ps, onehot + b * O);
        }
        if (onehot && n < H) {
            // bp_fully_connect(out -> hidden), gcn_mbgd_fp64.cu:416-434: 13-wide tree
            T prod[O];
#pragma unroll
            for (int o = 0; o < O; o++) prod[o] = w2s[o + n * O] * d2s[o];
            d1out[b * H + n] = tree13(prod) * (1 + a) * (1 - a);
        }
    }
}

// ---------------------------------------------------------------------------
// delta at the pooled features: dpool[row][p] = (sum_n d1[row][n] W1[n + 100 p]) (1+pool)(1-pool)
// (bp_fully_connect(hidden -> pool), gcn_mbgd_fp64.cu:416-434).
// grid (chunks, ceil(nrows/16)); [16 x 100] x [100 x 64] per workgroup, one 16-wide
// p tile per wave, 25 k-steps.
template <typename T>
__global__ __launch_bounds__(256) void fc1_bwd_delta(const T *__restrict__ W1,
                                                     const T *__restrict__ d1,
                                                     const T *__restrict__ pool, int nrows,
                                                     int pout, T *__restrict__ dpool)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *wl = reinterpret_cast<T *>(smem);   // [64][LDW2]
    T *dl = wl + PCHUNK * LDW2;            // [16][LDW2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int p0 = blockIdx.x * PCHUNK;
    const int r0 = blockIdx.y * 16;
    const int pmax = min(PCHUNK, pout - p0);
    for (int e = tid; e < PCHUNK * H; e += 256) {
        const int p = e / H, n = e - p * H;
        wl[p * LDW2 + n] = (p < pmax) ? W1[(long)(p0 + p) * H + n] : T(0);
    }
    for (int e = tid; e < 16 * H; e += 256) {
        const int s = e / H, n = e - s * H;
        dl[s * LDW2 + n] = (r0 + s < nrows) ? d1[(long)(r0 + s) * H + n] : T(0);
    }
    __syncthreads();
    const T *ap = dl + col * LDW2 + kq;
    const T *bp = wl + (wave * 16 + col) * LDW2 + kq;
    typename Mma<T>::acc_t acc = {0, 0, 0, 0};
#pragma unroll 5
    for (int s = 0; s < H / 4; s++) acc = Mma<T>::mma(ap[4 * s], bp[4 * s], acc);
    const int p = p0 + wave * 16 + col;
    if (p < pout) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int b = r0 + Mma<T>::row(lane, r);
            if (b < nrows) {
                const T pv = pool[(long)b * pout + p];
                dpool[(long)b * pout + p] = acc[r] * (1 + pv) * (1 - pv);
            }
        }
    }
}

template <typename T>
constexpr size_t fc1bwd_smem()
{
    return (size_t)((PCHUNK + 16) * LDW2) * sizeof(T);
}

// ---------------------------------------------------------------------------
// bp_maxpooling + bp_convolution for one sample per workgroup
// (gcn_mbgd_fp64.cu:439-496): the pooled deltas are scattered into a zeroed
// [F][re] LDS image (as the reference does in global memory), then
// dW0[f][tap] = sum_t Delta[f][t] cube[18 t + tap] / re is a [20 x re] x [re x 171]
// MFMA product (2 filter tiles x 11 tap tiles, three tap tiles per wave).
// At a winner the conv output equals the pooled value, which gives the bias path
// its second tanh' without keeping the conv output.  slab[row] = [dW0 (F*K) | db0 (F)].
template <typename T>
__global__ __launch_bounds__(256) void conv_bwd(const T *__restrict__ cubes, long first,
                                                int cube_len, int re, int mre,
                                                const T *__restrict__ dpool,
                                                const T *__restrict__ pool,
                                                const int *__restrict__ pidx,
                                                T *__restrict__ slab)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ks = (re + 3) >> 2;
    const int ldd = stride15(4 * ks);
    const int cube_pad = WSTEP * 4 * ks + 160;
    T *cube = reinterpret_cast<T *>(smem);
    T *delta = cube + cube_pad;   // [F][ldd]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int pout = F * mre;
    const long row = blockIdx.x;
    const T *src = cubes + (first + row) * (long)cube_len;
    for (int e = tid; e < cube_pad; e += 256) cube[e] = e < cube_len ? src[e] : T(0);
    for (int e = tid; e < F * ldd; e += 256) delta[e] = 0;
    __syncthreads();
    for (int o = tid; o < pout; o += 256) {
        const int f = o / mre;
        delta[f * ldd + pidx[row * pout + o]] = dpool[row * pout + o];
    }
    __syncthreads();

    const int f1 = 16 + col;
    const bool v1 = f1 < F;
    const T *ap0 = delta + col * ldd + kq;
    const T *ap1 = delta + (v1 ? f1 : F - 1) * ldd + kq;
    const bool t2 = wave + 8 < 11;   // tap tiles wave, wave+4, (wave+8): 171 taps = 11 tiles
    const T *bp0 = cube + WSTEP * kq + wave * 16 + col;
    const T *bp1 = bp0 + 64;
    const T *bp2 = bp0 + (t2 ? 128 : 0);
    typename Mma<T>::acc_t c00 = {0, 0, 0, 0}, c01 = c00, c02 = c00, c10 = c00, c11 = c00, c12 = c00;
#pragma unroll 2
    for (int s = 0; s < ks; s++) {
        const T a0 = ap0[4 * s];
        const T a1 = v1 ? ap1[4 * s] : T(0);
        const T b0 = bp0[4 * s * WSTEP], b1 = bp1[4 * s * WSTEP];
        c00 = Mma<T>::mma(a0, b0, c00);
        c10 = Mma<T>::mma(a1, b0, c10);
        c01 = Mma<T>::mma(a0, b1, c01);
        c11 = Mma<T>::mma(a1, b1, c11);
        if (t2) {
            const T b2 = bp2[4 * s * WSTEP];
            c02 = Mma<T>::mma(a0, b2, c02);
            c12 = Mma<T>::mma(a1, b2, c12);
        }
    }
    T *out = slab + row * (long)(F * K + F);
    auto emit = [&](const typename Mma<T>::acc_t &acc, int ftile, int ttile) {
        const int tap = ttile * 16 + col;
        if (tap >= K) return;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = ftile * 16 + Mma<T>::row(lane, r);
            if (f < F) out[f * K + tap] = acc[r] / re;
        }
    };
    emit(c00, 0, wave);
    emit(c10, 1, wave);
    emit(c01, 0, wave + 4);
    emit(c11, 1, wave + 4);
    if (t2) {
        emit(c02, 0, wave + 8);
        emit(c12, 1, wave + 8);
    }
    if (tid < F) {
        const T *d = dpool + row * pout + tid * mre;
        const T *pv = pool + row * pout + tid * mre;
        T mid1 = 0;
        for (int m = 0; m < mre; m++) mid1 = mid1 + d[m] * (1 + pv[m]) * (1 - pv[m]);
        out[F * K + tid] = mid1 / re;
    }
}

template <typename T>
size_t convbwd_smem(const ccnn_dims_t &d)
{
    const int ks = (d.re + 3) >> 2;
    return (size_t)(WSTEP * 4 * ks + 160 + F * stride15(4 * ks)) * sizeof(T);
}

// ---------------------------------------------------------------------------
// Batch sums of the per-sample gradients, written in the flat parameter layout
// (update_params_row/col, gcn_mbgd_fp64.cu:513-517, 546-550).  With UPD the
// update is applied in the same pass instead of storing the gradient.
//   blocks [0, nb_w1)   dW1[n + 100 p] = sum_b pool[b][p] d1[b][n]: one 16-wide p
//                       tile per block, one neuron tile per wave, K = batch (MFMA)
//   next nb_w0          dW0 | db0 = sum_b slab[b][.]
//   next nb_w2          dW2[o + 13 n] = sum_b a1[b][n] d2[b][o]
//   last                db1 = sum_b d1, db2 = sum_b d2
constexpr int GR_THREADS = 448;   // 7 waves = the 7 neuron tiles of FC1

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

template <typename T>
__global__ __launch_bounds__(GR_THREADS) void grad_reduce(int B, int pout, int off_w1, int off_b1,
                                                          int off_w2, int off_b2,
                                                          const T *__restrict__ pool,
                                                          const T *__restrict__ a1,
                                                          const T *__restrict__ d1,
                                                          const T *__restrict__ d2,
                                                          const T *__restrict__ slab, int nb_w1,
                                                          int nb_w0, int nb_w2, T *__restrict__ grad,
                                                          UpdateRule<T> upd)
{
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    if (blk < nb_w1) {
        const int lane = tid & 63, wave = tid >> 6;
        const int col = lane & 15, kq = lane >> 4;
        const int p0 = blk * 16, n0 = wave * 16;
        const bool pv = p0 + col < pout, nv = n0 + col < H;
        const T *ap = pool + p0 + col;
        const T *bp = d1 + n0 + col;
        typename Mma<T>::acc_t acc = {0, 0, 0, 0};
        const int ks = (B + 3) >> 2;
#pragma unroll 8
        for (int s = 0; s < ks; s++) {
            const int b = 4 * s + kq;
            const bool bv = b < B;
            const T a = (bv && pv) ? ap[(long)b * pout] : T(0);
            const T w = (bv && nv) ? bp[(long)b * H] : T(0);
            acc = Mma<T>::mma(a, w, acc);
        }
        if (nv) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int p = p0 + Mma<T>::row(lane, r);
                if (p < pout) upd.apply(grad, off_w1 + n0 + col + H * p, acc[r]);
            }
        }
        return;
    }
    blk -= nb_w1;
    if (blk < nb_w0) {
        const int k = blk * GR_THREADS + tid;
        if (k >= F * K + F) return;
        T acc = 0;
#pragma unroll 8
        for (int b = 0; b < B; b++) acc = acc + slab[(long)b * (F * K + F) + k];
        upd.apply(grad, k, acc);   // off_w0 = 0 and b0 follows W0, exactly the slab layout
        return;
    }
    blk -= nb_w0;
    if (blk < nb_w2) {
        const int k = blk * GR_THREADS + tid;   // k = o + 13 n
        if (k >= H * O) return;
        const int nn = k / O, o = k - nn * O;
        T acc = 0;
#pragma unroll 8
        for (int b = 0; b < B; b++) acc = acc + a1[(long)b * H + nn] * d2[(long)b * O + o];
        upd.apply(grad, off_w2 + k, acc);
        return;
    }
    if (tid < H) {
        T acc = 0;
#pragma unroll 8
        for (int b = 0; b < B; b++) acc = acc + d1[(long)b * H + tid];
        upd.apply(grad, off_b1 + tid, acc);
    } else if (tid - H < O) {
        const int o = tid - H;
        T acc = 0;
#pragma unroll 8
        for (int b = 0; b < B; b++) acc = acc + d2[(long)b * O + o];
        upd.apply(grad, off_b2 + o, acc);
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

constexpr long FWD_CHUNK = 8192;   // samples per forward pass through the workspace
constexpr int PERSIST_GRID = 1024; // workgroups for the grid-stride kernels

// Kernels whose dynamic LDS can exceed the 64 KB default need the limit raised once.
template <typename Kern>
int allow_smem(Kern k, size_t bytes)
{
    if (bytes <= 64 * 1024) return 0;
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> raised;   // (kernel, device) -> limit already set
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const void *fn = reinterpret_cast<const void *>(k);
    std::lock_guard<std::mutex> lk(mu);
    size_t &have = raised[{fn, dev}];
    if (have >= bytes) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return (int)e;
}

template <typename T>
struct FwdWs {
    T *pool, *z1part;
    static size_t bytes(long rows, const ccnn_dims_t &d)
    {
        return align_up((size_t)rows * d.pout * sizeof(T)) +
               align_up((size_t)nchunks(d.pout) * rows * H * sizeof(T));
    }
    FwdWs(void *ws, long rows, const ccnn_dims_t &d)
    {
        Carver c(ws);
        pool = c.take<T>((size_t)rows * d.pout);
        z1part = c.take<T>((size_t)nchunks(d.pout) * rows * H);
    }
};

template <typename T, bool FLIP>
int forward_impl(const T *params, const T *cubes, long n, int z, T *probs, int *argmax, void *ws,
                 size_t ws_bytes, hipStream_t st)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    if (!params || !cubes || n < 0 || (n > 0 && !ws)) return CCNN_E_INVALID;
    if (n == 0) return 0;
    const long rows = n < FWD_CHUNK ? n : FWD_CHUNK;
    if (ws_bytes < FwdWs<T>::bytes(rows, d)) return CCNN_E_WORKSPACE;
    FwdWs<T> w(ws, rows, d);
    const int nch = nchunks(d.pout);
    if (int e = allow_smem(conv_pool_fwd<T, FLIP>, conv_smem<T>(d))) return e;
    if (int e = allow_smem(fc1_partial<T>, fc1_smem<T>())) return e;
    for (long s0 = 0; s0 < n; s0 += rows) {
        const int nr = (int)((n - s0) < rows ? (n - s0) : rows);
        const int pg = nr < PERSIST_GRID ? nr : PERSIST_GRID;
        {
            ProfileScope prof_("conv_pool_fwd", st);
            conv_pool_fwd<T, FLIP><<<pg, 256, conv_smem<T>(d), st>>>(params, cubes, s0, nr, d.cube, d.re,
                                                                    d.mre, d.off_b0, w.pool, nullptr);
        }
        CCNN_LAUNCH_CHECK();
        {
            ProfileScope prof_("fc1_partial", st);
            fc1_partial<T><<<dim3(nch, (nr + 15) / 16), 256, fc1_smem<T>(), st>>>(params + d.off_w1, w.pool,
                                                                                nr, d.pout, w.z1part);
        }
        CCNN_LAUNCH_CHECK();
        {
            ProfileScope prof_("head", st);
            head<T><<<pg, 128, 0, st>>>(params, d.off_b1, d.off_w2, d.off_b2, w.z1part, nch, nr, nullptr,
                                        nullptr, nullptr, nullptr, probs ? probs + s0 * O : nullptr,
                                        argmax ? argmax + s0 : nullptr, nullptr);
        }
        CCNN_LAUNCH_CHECK();
    }
    return 0;
}

template <typename T>
struct MbgdWs {
    T *pool, *z1part, *a1, *d1, *d2, *dpool, *slab, *probs;
    int *pidx;
    static size_t bytes(int B, const ccnn_dims_t &d)
    {
        size_t b = 0;
        b += 2 * align_up((size_t)B * d.pout * sizeof(T));              // pool, dpool
        b += align_up((size_t)B * d.pout * sizeof(int));                // pidx
        b += align_up((size_t)nchunks(d.pout) * B * H * sizeof(T));     // z1part
        b += 2 * align_up((size_t)B * H * sizeof(T));                   // a1, d1
        b += 2 * align_up((size_t)B * O * sizeof(T));                   // d2, probs
        b += align_up((size_t)B * (F * K + F) * sizeof(T));             // conv slab
        return b;
    }
    MbgdWs(void *ws, int B, const ccnn_dims_t &d)
    {
        Carver c(ws);
        pool = c.take<T>((size_t)B * d.pout);
        dpool = c.take<T>((size_t)B * d.pout);
        pidx = c.take<int>((size_t)B * d.pout);
        z1part = c.take<T>((size_t)nchunks(d.pout) * B * H);
        a1 = c.take<T>((size_t)B * H);
        d1 = c.take<T>((size_t)B * H);
        d2 = c.take<T>((size_t)B * O);
        probs = c.take<T>((size_t)B * O);
        slab = c.take<T>((size_t)B * (F * K + F));
    }
};

// grad != null: store the batch-summed gradient.  grad == null: apply `upd` in place.
template <typename T, bool FLIP>
int mbgd_impl(const T *params, const T *cubes, const int *onehot, long first, int B, int z, T *grad,
              UpdateRule<T> upd, double *loss_values, T *probs, void *ws, size_t ws_bytes,
              hipStream_t st)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    if (!params || !cubes || !onehot || !loss_values || !ws || B <= 0 || first < 0) return CCNN_E_INVALID;
    if (!grad && !upd.params) return CCNN_E_INVALID;
    if (ws_bytes < MbgdWs<T>::bytes(B, d)) return CCNN_E_WORKSPACE;
    MbgdWs<T> w(ws, B, d);
    const int nch = nchunks(d.pout);
    const dim3 tiles(nch, (B + 15) / 16);
    if (int e = allow_smem(conv_pool_fwd<T, FLIP>, conv_smem<T>(d))) return e;
    if (int e = allow_smem(fc1_partial<T>, fc1_smem<T>())) return e;
    if (int e = allow_smem(fc1_bwd_delta<T>, fc1bwd_smem<T>())) return e;
    if (int e = allow_smem(conv_bwd<T>, convbwd_smem<T>(d))) return e;
    {
        ProfileScope prof_("conv_pool_fwd", st);
        conv_pool_fwd<T, FLIP><<<B, 256, conv_smem<T>(d), st>>>(params, cubes, first, B, d.cube, d.re, d.mre,
                                                               d.off_b0, w.pool, w.pidx);
    }
    CCNN_LAUNCH_CHECK();
    {
        ProfileScope prof_("fc1_partial", st);
        fc1_partial<T><<<tiles, 256, fc1_smem<T>(), st>>>(params + d.off_w1, w.pool, B, d.pout, w.z1part);
    }
    CCNN_LAUNCH_CHECK();
    {
        ProfileScope prof_("head", st);
        head<T><<<B, 128, 0, st>>>(params, d.off_b1, d.off_w2, d.off_b2, w.z1part, nch, B,
                                   onehot + first * O, w.a1, w.d1, w.d2, probs ? probs : w.probs, nullptr,
                                   loss_values);
    }
    CCNN_LAUNCH_CHECK();
    {
        ProfileScope prof_("fc1_bwd_delta", st);
        fc1_bwd_delta<T><<<tiles, 256, fc1bwd_smem<T>(), st>>>(params + d.off_w1, w.d1, w.pool, B, d.pout,
                                                             w.dpool);
    }
    CCNN_LAUNCH_CHECK();
    {
        ProfileScope prof_("conv_bwd", st);
        conv_bwd<T><<<B, 256, convbwd_smem<T>(d), st>>>(cubes, first, d.cube, d.re, d.mre, w.dpool, w.pool,
                                                       w.pidx, w.slab);
    }
    CCNN_LAUNCH_CHECK();
    const int nb_w1 = (d.pout + 15) / 16;
    const int nb_w0 = (F * K + F + GR_THREADS - 1) / GR_THREADS;
    const int nb_w2 = (H * O + GR_THREADS - 1) / GR_THREADS;
    {
        ProfileScope prof_("grad_reduce", st);
        grad_reduce<T><<<nb_w1 + nb_w0 + nb_w2 + 1, GR_THREADS, 0, st>>>(B, d.pout, d.off_w1, d.off_b1,
                                                                         d.off_w2, d.off_b2, w.pool, w.a1,
                                                                         w.d1, w.d2, w.slab, nb_w1, nb_w0,
                                                                         nb_w2, grad, upd);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace ccnn

using namespace ccnn;

extern "C" {

size_t ccnn_forward_workspace_bytes(long n, int z, int elem_size)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || n <= 0) return 0;
    const long rows = n < FWD_CHUNK ? n : FWD_CHUNK;
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
        ProfileScope prof_("count_correct_kernel", st);
        count_correct_kernel<<<(int)blocks, 256, 0, st>>>(argmax, labels, n, count);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

size_t ccnn_mbgd_workspace_bytes(int B, int z, int elem_size)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || B <= 0) return 0;
    return elem_size == 8 ? MbgdWs<double>::bytes(B, d) : MbgdWs<float>::bytes(B, d);
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

int ccnn_mbgd_step_f64(double *params, const double *cubes, const int *onehot, long first, int B, int z,
                       double lr, double *loss_values, void *workspace, size_t workspace_bytes, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params) return CCNN_E_INVALID;
    UpdateRule<double> upd{params, lr, B, 0, 0};   // mean over the batch (gcn_mbgd_fp64.cu:519)
    return mbgd_impl<double, false>(params, cubes, onehot, first, B, z, nullptr, upd, loss_values, nullptr,
                                    workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_update_f64(double *params, const double *grad, double lr, int divisor, long n, void *stream)
{
    if (!params || !grad || divisor <= 0 || n < 0) return CCNN_E_INVALID;
    if (n == 0) return 0;
    {
        ProfileScope prof_("update_f64_kernel", (hipStream_t)stream);
        update_f64_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(params, grad, lr, divisor, n);
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
        ProfileScope prof_("update_f32_kernel", (hipStream_t)stream);
        update_f32_kernel<<<(d.nparams + 255) / 256, 256, 0, (hipStream_t)stream>>>(params, grad, lr, d.nparams,
                                                                                   lo, hi);
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
        ProfileScope prof_("update_adagrad_f64_kernel", (hipStream_t)stream);
        update_adagrad_f64_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(params, rms, grad, lr,
                                                                                          iter, n);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
