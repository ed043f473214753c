// Forward (inference) path of the Cube CNN, batched over pixels, plus the
// stand-alone parameter updates and the accuracy counter.
//
// Kernel map (reference kernel -> here)
//   convolution + maxpooling   -> conv_pool_fwd  (implicit GEMM on MFMA: positions x taps x filters)
//   fully_connect              -> fc1_partial (MFMA, K split in 64-wide chunks) + head
//   output_and_dvalue(false)   -> head
//   count_err                  -> head (device argmax) + count_correct_kernel
//   update_params_row/col      -> update_*_kernel (after an all-reduce; single-GPU training
//                                 fuses the update into grad_reduce, see mbgd.hip)
#include <map>
#include <mutex>
#include <utility>

#include "ccnn_device.h"
#include "profile.h"

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
// convolution + tanh + max-pool, one sample at a time per workgroup (grid-stride
// over samples; the filters are staged once per workgroup).
// pool[row][f*mre + m] = pooled value.
template <typename T, bool FLIP>
__global__ __launch_bounds__(256) void conv_pool_fwd(const T *__restrict__ params,
                                                     const T *__restrict__ cubes, long first, int nrows,
                                                     int cube_len, int re, int mre, int off_b0,
                                                     T *__restrict__ pool)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cube_pad = conv_cube_pad(re);
    const int lda = odd_up(re);
    T *cube = reinterpret_cast<T *>(smem);
    T *wl = cube + cube_pad;
    T *a0 = wl + 32 * LDW0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15;
    const int pout = F * mre;

    for (int e = cube_len + tid; e < cube_pad; e += 256) cube[e] = 0;
    stage_filters<T, FLIP>(params, wl, tid);
    const T bias0 = params[off_b0 + col];
    const T bias1 = 16 + col < F ? params[off_b0 + 16 + col] : T(0);

    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        __syncthreads();   // everyone is done with the previous sample's cube and a0
        stage<8, 256>(cubes + (first + row) * (long)cube_len, cube_len, tid, [&](int e, T v) { cube[e] = v; });
        __syncthreads();
        conv_fwd_tiles<T>(cube, wl, a0, lda, re, bias0, bias1, lane, wave);
        __syncthreads();
        T *dst = pool + row * pout;
        pool_rows<T>(a0, lda, re, mre, tid, [&](int o, T v, int) { dst[o] = v; });
    }
}

template <typename T>
size_t conv_smem(const ccnn_dims_t &d)
{
    return (size_t)(conv_cube_pad(d.re) + 32 * LDW0 + F * odd_up(d.re)) * sizeof(T);
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
    // zero what the loads below do not cover: neuron columns 100..111, rows past the
    // chunk's end, and the whole pool tile (ragged batch / last chunk)
    for (int e = tid; e < PCHUNK * LDW1; e += 256) {
        const int p = e / LDW1, n = e - p * LDW1;
        if (n >= H || p >= pmax) wl[e] = 0;
    }
    for (int e = tid; e < 16 * LDP; e += 256) pt[e] = 0;
    __syncthreads();
    // the chunk is contiguous in W1 ([p][100]) and 16-byte aligned (100 p0 and off_w1 are multiples of 4)
    constexpr int VW = Vec16<T>::W;
    stage_vec<7, 256>(W1 + (long)p0 * H, pmax * H / VW, tid, [&](int e, typename Vec16<T>::type v) {
        const int p = (e * VW) / H, n = e * VW - p * H;
        *reinterpret_cast<typename Vec16<T>::type *>(wl + p * LDW1 + n) = v;
    });
    {
        const int nb = min(16, nrows - r0);
        T tmp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = u * 256 + tid, b = e >> 6, p = e & 63;
            const T x = pool[(long)(r0 + min(b, nb - 1)) * pout + p0 + min(p, pmax - 1)];   // clamped load
            tmp[u] = (b < nb && p < pmax) ? x : T(0);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = u * 256 + tid, b = e >> 6, p = e & 63;
            pt[b * LDP + p] = tmp[u];
        }
    }
    __syncthreads();
    const int n0 = wave * 16, n1 = (wave + 4) * 16;
    const bool two = n1 < H;   // waves 0..2 own two neuron tiles, wave 3 one
    const T *ap = pt + col * LDP + kq;
    const T *bp0 = wl + kq * LDW1 + n0 + col;
    const T *bp1 = wl + kq * LDW1 + (two ? n1 : n0) + col;
    typename Mma<T>::acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int s = 0; s < PCHUNK / 4; s += 4) {
        T a[4], b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a[u] = ap[4 * (s + u)];
            b0[u] = bp0[4 * (s + u) * LDW1];
            b1[u] = bp1[4 * (s + u) * LDW1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc0 = Mma<T>::mma(a[u], b0[u], acc0);
            acc1 = Mma<T>::mma(a[u], b1[u], acc1);   // wave 3 repeats its first tile; discarded below
        }
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
    stage<11, 128>(params + off_w2, H * O, n, [&](int e, T v) { w2s[e] = v; });
    const T b1 = n < H ? params[off_b1 + n] : T(0);
    // output unit o = n >> 3 (13 x 8 threads), each summing every 8th hidden unit
    const int o8 = n >> 3, j8 = n & 7;
    const T b2 = o8 < O ? params[off_b2 + o8] : T(0);

    for (long b = blockIdx.x; b < nrows; b += gridDim.x) {
        __syncthreads();   // previous sample's a1s/es/ps/d2s readers are done (and w2s is staged)
        T a = 0;
        if (n < H) {
            T zsum = 0;
            const T *zp = z1part + b * H + n;
            const long cs = (long)nrows * H;
            int c = 0;
            for (; c + 4 <= nch; c += 4) {
                const T z0 = zp[c * cs], z1 = zp[(c + 1) * cs], z2 = zp[(c + 2) * cs], z3 = zp[(c + 3) * cs];
                zsum = (((zsum + z0) + z1) + z2) + z3;
            }
            for (; c < nch; c++) zsum = zsum + zp[c * cs];
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
            if (onehot && loss_values) {
                T ta[O], tb[O];
                for (int i = 0; i < O; i++) loss_terms(ps[i], onehot[b * O + i], &ta[i], &tb[i]);
                loss_values[b] = loss_sum(ta, tb, onehot + b * O);
            }
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
                                                                    d.mre, d.off_b0, w.pool);
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
