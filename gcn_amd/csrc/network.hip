// Batched forward / minibatch-gradient / update kernels of the Cube CNN and their
// C entry points.  One launch per layer over the whole batch replaces the
// reference's nine launches per sample (gcn_mbgd_fp64.cu:931-1014).
//
// Kernel map (reference kernel -> here)
//   convolution + maxpooling            -> conv_pool_fwd      (one workgroup per sample)
//   fully_connect                       -> fc1_partial + head (K split in 64-wide chunks)
//   output_and_dvalue, loss_function,
//   bp_fully_connect(out->hidden)       -> head
//   bp_fully_connect(hidden->pool)      -> fc1_bwd_delta (delta) and grad_reduce (dW1)
//   bp_maxpooling + bp_convolution      -> conv_bwd           (sparse over pooled maxima)
//   update_params_row/col               -> grad_reduce (batch sums) + update
#include "ccnn_common.h"
#include "profile.h"

namespace ccnn {

// ---------------------------------------------------------------------------
// convolution + tanh + max-pool for one sample per workgroup.
// Lane <-> conv position t (t fastest) so the LDS cube reads are stride-18 (at
// most 2-way bank conflicts) and a pool pair (2m, 2m+1) sits in adjacent lanes.
// pool[row][f*mre + m], pidx[row][f*mre + m] = winning position t in [0, re).
template <typename T, bool FLIP>
__global__ __launch_bounds__(256) void conv_pool_fwd(const T *__restrict__ params,
                                                     const T *__restrict__ cubes, long first,
                                                     int cube_len, int re, int mre, int off_b0,
                                                     T *__restrict__ pool, int *__restrict__ pidx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *cube = reinterpret_cast<T *>(smem);
    T *w0 = cube + ((cube_len + 3) & ~3);
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    const T *src = cubes + (first + row) * (long)cube_len;
    for (int e = tid; e < cube_len; e += 256) cube[e] = src[e];
    for (int e = tid; e < F * K; e += 256) {
        int f = e / K, i = e - f * K;
        // the fp32 reference walks the filter backwards (gcn_mbgd_fp32.cu:241-243)
        w0[e] = FLIP ? params[f * K + (K - 1 - i)] : params[e];
    }
    __syncthreads();

    const int re_e = 2 * mre;
    const int total = F * re_e;
    const int pout = F * mre;
    for (int base = 0; base < total; base += 256) {
        const int o = base + tid;
        const int f = o / re_e;
        const int t = o - f * re_e;
        const bool valid = (o < total) && (t < re);
        T v = 0;
        if (valid) {
            const T *w = w0 + f * K;
            const T *c = cube + t * WSTEP;
            T acc = 0;
#pragma unroll 9
            for (int i = 0; i < K; i++) acc = acc + w[i] * c[i];
            v = act(acc + params[off_b0 + f]);
        }
        const T vn = __shfl_down(v, 1);
        if (valid && !(t & 1)) {
            T best = v;
            int bt = t;
            if (t + 1 < re && best < vn) {   // strict '<': first maximum wins (gcn_mbgd_fp64.cu:257)
                best = vn;
                bt = t + 1;
            }
            const int m = t >> 1;
            pool[row * pout + f * mre + m] = best;
            if (pidx) pidx[row * pout + f * mre + m] = bt;
        }
    }
}

// ---------------------------------------------------------------------------
// FC1 partial products: z1part[c][row][n] = sum_{p in chunk c} pool[row][p] * W1[n + 100 p].
// grid (chunks, ceil(nrows/16)); thread <-> neuron n, 16 samples per workgroup.
template <typename T>
__global__ __launch_bounds__(128) void fc1_partial(const T *__restrict__ W1,
                                                   const T *__restrict__ pool, int nrows, int pout,
                                                   T *__restrict__ z1part)
{
    __shared__ T tile[PCHUNK][16];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    const int r0 = blockIdx.y * 16;
    const int p0 = c * PCHUNK;
    for (int e = tid; e < 16 * PCHUNK; e += 128) {
        int s = e / PCHUNK, p = e - s * PCHUNK;
        T v = 0;
        if (r0 + s < nrows && p0 + p < pout) v = pool[(long)(r0 + s) * pout + p0 + p];
        tile[p][s] = v;
    }
    __syncthreads();
    const int n = tid;
    if (n >= H) return;
    T acc[16];
#pragma unroll
    for (int s = 0; s < 16; s++) acc[s] = 0;
    const int pmax = min(PCHUNK, pout - p0);
    for (int p = 0; p < pmax; p++) {
        const T w = W1[n + H * (p0 + p)];
#pragma unroll
        for (int s = 0; s < 16; s++) acc[s] = acc[s] + tile[p][s] * w;
    }
#pragma unroll
    for (int s = 0; s < 16; s++)
        if (r0 + s < nrows) z1part[((long)c * nrows + r0 + s) * H + n] = acc[s];
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
// + count_err argmax.  One workgroup of 128 threads per sample.
template <typename T>
__global__ __launch_bounds__(128) void head(const T *__restrict__ params, int off_b1, int off_w2,
                                            int off_b2, const T *__restrict__ z1part, int nch,
                                            int nrows, const int *__restrict__ onehot,
                                            T *__restrict__ a1out, T *__restrict__ d1out,
                                            T *__restrict__ d2out, T *__restrict__ probs,
                                            int *__restrict__ argmax,
                                            double *__restrict__ loss_values)
{
    __shared__ T a1s[H];
    __shared__ T es[16];
    __shared__ T ps[16];
    __shared__ T d2s[16];
    const int n = threadIdx.x;
    const long b = blockIdx.x;
    const T *W2 = params + off_w2;
    T a = 0;
    if (n < H) {
        T zsum = 0;
        for (int c = 0; c < nch; c++) zsum = zsum + z1part[((long)c * nrows + b) * H + n];
        a = 2 / (1 + 1 / rexp((zsum + params[off_b1 + n]) * 2)) - 1;   // gcn_mbgd_fp64.cu:302
        a1s[n] = a;
        if (a1out) a1out[b * H + n] = a;
    }
    __syncthreads();
    if (n < O) {
        T mid = 0;
        for (int i = 0; i < H; i++) mid = mid + W2[n + i * O] * a1s[i];
        es[n] = rexp(mid + params[off_b2 + n]);
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
    if (n == 0) {
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
        for (int o = 0; o < O; o++) prod[o] = W2[o + n * O] * d2s[o];
        d1out[b * H + n] = tree13(prod) * (1 + a) * (1 - a);
    }
}

// ---------------------------------------------------------------------------
// delta at the pooled features: dpool[row][p] = (sum_n W1[n + 100 p] d1[row][n]) (1+pool)(1-pool)
// (bp_fully_connect(hidden -> pool), gcn_mbgd_fp64.cu:416-434).
// grid (chunks, ceil(nrows/16)), 256 threads: lane <-> p in chunk, wave <-> 4 samples.
template <typename T>
__global__ __launch_bounds__(256) void fc1_bwd_delta(const T *__restrict__ W1,
                                                     const T *__restrict__ d1,
                                                     const T *__restrict__ pool, int nrows,
                                                     int pout, T *__restrict__ dpool)
{
    __shared__ T wl[PCHUNK][H + 1];
    __shared__ T dl[16][H];
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * PCHUNK;
    const int r0 = blockIdx.y * 16;
    for (int e = tid; e < PCHUNK * H; e += 256) {
        int p = e / H, n = e - p * H;
        wl[p][n] = (p0 + p < pout) ? W1[(long)(p0 + p) * H + n] : T(0);
    }
    for (int e = tid; e < 16 * H; e += 256) {
        int s = e / H, n = e - s * H;
        dl[s][n] = (r0 + s < nrows) ? d1[(long)(r0 + s) * H + n] : T(0);
    }
    __syncthreads();
    const int p = tid & 63;
    const int sq = tid >> 6;
    T acc[4] = {0, 0, 0, 0};
    for (int n = 0; n < H; n++) {
        const T w = wl[p][n];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = acc[j] + w * dl[sq * 4 + j][n];
    }
    if (p0 + p < pout) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = r0 + sq * 4 + j;
            if (r < nrows) {
                const T pv = pool[(long)r * pout + p0 + p];
                dpool[(long)r * pout + p0 + p] = acc[j] * (1 + pv) * (1 - pv);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// bp_maxpooling + bp_convolution for one sample per workgroup
// (gcn_mbgd_fp64.cu:439-496).  The reference scatters dpool into a zeroed [F][re]
// buffer and sums densely over all re positions; only the pooled maxima are
// non-zero, so the same sum runs over the mre winners in increasing position.
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
    T *cube = reinterpret_cast<T *>(smem);
    T *dp = cube + ((cube_len + 3) & ~3);
    const int pout = F * mre;
    int *px = reinterpret_cast<int *>(dp + ((pout + 3) & ~3));
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    const T *src = cubes + (first + row) * (long)cube_len;
    for (int e = tid; e < cube_len; e += 256) cube[e] = src[e];
    for (int e = tid; e < pout; e += 256) {
        dp[e] = dpool[row * pout + e];
        px[e] = pidx[row * pout + e] * WSTEP;
    }
    __syncthreads();
    T *out = slab + row * (long)(F * K + F);
    for (int o = tid; o < F * K; o += 256) {
        const int f = o / K, tap = o - f * K;
        const T *d = dp + f * mre;
        const int *x = px + f * mre;
        T mid0 = 0;
        for (int m = 0; m < mre; m++) mid0 = mid0 + d[m] * cube[x[m] + tap];
        out[o] = mid0 / re;
    }
    if (tid < F) {
        const T *d = dp + tid * mre;
        const T *pv = pool + row * pout + tid * mre;
        T mid1 = 0;
        for (int m = 0; m < mre; m++) mid1 = mid1 + d[m] * (1 + pv[m]) * (1 - pv[m]);
        out[F * K + tid] = mid1 / re;
    }
}

// ---------------------------------------------------------------------------
// Batch sums of the per-sample gradients, in increasing sample order like
// update_params_row/col (gcn_mbgd_fp64.cu:513-517, 546-550), written in the flat
// parameter layout.  Block ranges:
//   [0, nb_w1)            dW1[n + 100 p] = sum_b pool[b][p] d1[b][n]   (8 p per block)
//   [nb_w1, nb_w1+nb_w0)  dW0 | db0 = sum_b slab[b][.]                 (128 per block)
//   next 11               dW2[o + 13 n] = sum_b a1[b][n] d2[b][o]
//   last                  db1 = sum_b d1, db2 = sum_b d2
template <typename T>
__global__ __launch_bounds__(128) void grad_reduce(int B, int pout, int off_w1, int off_b1,
                                                   int off_w2, int off_b2,
                                                   const T *__restrict__ pool,
                                                   const T *__restrict__ a1,
                                                   const T *__restrict__ d1,
                                                   const T *__restrict__ d2,
                                                   const T *__restrict__ slab, int nb_w1,
                                                   int nb_w0, T *__restrict__ grad)
{
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    if (blk < nb_w1) {
        const int n = tid;
        const int p0 = blk * 8;
        if (n >= H) return;
        T acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0;
        const int np = min(8, pout - p0);
        for (int b = 0; b < B; b++) {
            const T d = d1[(long)b * H + n];
            const T *pl = pool + (long)b * pout + p0;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j < np) acc[j] = acc[j] + pl[j] * d;
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < np) grad[off_w1 + n + H * (p0 + j)] = acc[j];
        return;
    }
    blk -= nb_w1;
    if (blk < nb_w0) {
        const int k = blk * 128 + tid;
        if (k >= F * K + F) return;
        T acc = 0;
        for (int b = 0; b < B; b++) acc = acc + slab[(long)b * (F * K + F) + k];
        grad[k] = acc;   // off_w0 = 0 and b0 follows W0, exactly the slab layout
        return;
    }
    blk -= nb_w0;
    if (blk < 11) {
        const int k = blk * 128 + tid;   // k = o + 13 n
        if (k >= H * O) return;
        const int nn = k / O, o = k - nn * O;
        T acc = 0;
        for (int b = 0; b < B; b++) acc = acc + a1[(long)b * H + nn] * d2[(long)b * O + o];
        grad[off_w2 + k] = acc;
        return;
    }
    if (tid < H) {
        T acc = 0;
        for (int b = 0; b < B; b++) acc = acc + d1[(long)b * H + tid];
        grad[off_b1 + tid] = acc;
    } else if (tid - H < O) {
        const int o = tid - H;
        T acc = 0;
        for (int b = 0; b < B; b++) acc = acc + d2[(long)b * O + o];
        grad[off_b2 + o] = acc;
    }
}

// ---------------------------------------------------------------------------
// Parameter updates.
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

template <typename T>
size_t conv_smem(const ccnn_dims_t &d)
{
    return (size_t)(((d.cube + 3) & ~3) + F * K) * sizeof(T);
}

template <typename T>
size_t convbwd_smem(const ccnn_dims_t &d)
{
    return (size_t)(((d.cube + 3) & ~3) + ((d.pout + 3) & ~3)) * sizeof(T) + (size_t)d.pout * sizeof(int);
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
    for (long s0 = 0; s0 < n; s0 += rows) {
        const int nr = (int)((n - s0) < rows ? (n - s0) : rows);
        {
            ProfileScope prof_("conv_pool_fwd", st);
            conv_pool_fwd<T, FLIP><<<nr, 256, conv_smem<T>(d), st>>>(params, cubes, s0, d.cube, d.re, d.mre,
                                                                d.off_b0, w.pool, nullptr);
        }
        CCNN_LAUNCH_CHECK();
        {
            ProfileScope prof_("fc1_partial", st);
            fc1_partial<T><<<dim3(nch, (nr + 15) / 16), 128, 0, st>>>(params + d.off_w1, w.pool, nr, d.pout,
                                                                  w.z1part);
        }
        CCNN_LAUNCH_CHECK();
        {
            ProfileScope prof_("head", st);
            head<T><<<nr, 128, 0, st>>>(params, d.off_b1, d.off_w2, d.off_b2, w.z1part, nch, nr, nullptr,
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

template <typename T, bool FLIP>
int mbgd_grad_impl(const T *params, const T *cubes, const int *onehot, long first, int B, int z,
                   T *grad, double *loss_values, T *probs, void *ws, size_t ws_bytes,
                   hipStream_t st)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    if (!params || !cubes || !onehot || !grad || !loss_values || !ws || B <= 0 || first < 0)
        return CCNN_E_INVALID;
    if (ws_bytes < MbgdWs<T>::bytes(B, d)) return CCNN_E_WORKSPACE;
    MbgdWs<T> w(ws, B, d);
    const int nch = nchunks(d.pout);
    const dim3 tiles(nch, (B + 15) / 16);
    {
        ProfileScope prof_("conv_pool_fwd", st);
        conv_pool_fwd<T, FLIP><<<B, 256, conv_smem<T>(d), st>>>(params, cubes, first, d.cube, d.re, d.mre,
                                                           d.off_b0, w.pool, w.pidx);
    }
    CCNN_LAUNCH_CHECK();
    {
        ProfileScope prof_("fc1_partial", st);
        fc1_partial<T><<<tiles, 128, 0, st>>>(params + d.off_w1, w.pool, B, d.pout, w.z1part);
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
        fc1_bwd_delta<T><<<tiles, 256, 0, st>>>(params + d.off_w1, w.d1, w.pool, B, d.pout, w.dpool);
    }
    CCNN_LAUNCH_CHECK();
    {
        ProfileScope prof_("conv_bwd", st);
        conv_bwd<T><<<B, 256, convbwd_smem<T>(d), st>>>(cubes, first, d.cube, d.re, d.mre, w.dpool, w.pool,
                                                   w.pidx, w.slab);
    }
    CCNN_LAUNCH_CHECK();
    const int nb_w1 = (d.pout + 7) / 8;
    const int nb_w0 = (F * K + F + 127) / 128;
    {
        ProfileScope prof_("grad_reduce", st);
        grad_reduce<T><<<nb_w1 + nb_w0 + 11 + 1, 128, 0, st>>>(B, d.pout, d.off_w1, d.off_b1, d.off_w2,
                                                           d.off_b2, w.pool, w.a1, w.d1, w.d2, w.slab,
                                                           nb_w1, nb_w0, grad);
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
    return mbgd_grad_impl<float, true>(params, cubes, onehot, first, B, z, grad, loss_values, probs,
                                       workspace, workspace_bytes, (hipStream_t)stream);
}

int ccnn_mbgd_grad_f64(const double *params, const double *cubes, const int *onehot, long first, int B,
                       int z, double *grad, double *loss_values, double *probs, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    return mbgd_grad_impl<double, false>(params, cubes, onehot, first, B, z, grad, loss_values, probs,
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
