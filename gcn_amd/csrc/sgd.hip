// fp64 per-sample SGD (gcn_sgd_fp64.cu:776-855) as two launches per sample.
//
// The sample sequence is serial (sample k+1 sees the weights sample k wrote), so
// the parallelism is inside one sample: workgroup f owns filter f, the pooled
// features it produces and the FC1 columns those features feed.  The only
// cross-workgroup exchange per sample is the 20 x 100 table of FC1 partial sums,
// which is the boundary between the two kernels.
//
//   sgd_fwd  (20 WGs): apply the previous sample's pending W2/b2/b1 update,
//                      conv + tanh + pool for filter f, FC1 partial sums.
//   sgd_bwd  (20 WGs): every WG recomputes the (tiny) head: a1, softmax, loss,
//                      delta2, delta1; then its own slice: delta at its pooled
//                      features, in-place W1 slice update, conv gradient and
//                      in-place filter/bias update.  WG 0 parks a1/delta1/delta2
//                      as the pending head update (the other WGs are still
//                      reading W2/b2/b1 in this launch).
#include "ccnn_common.h"
#include "profile.h"

namespace ccnn {

struct SgdWs {
    double *z1part;   // [F][H]
    double *pool;     // [pout]
    int *pidx;        // [pout] winning position * 18
    double *a1, *d1, *d2;   // pending head update
    static size_t bytes(const ccnn_dims_t &d)
    {
        return align_up((size_t)F * H * 8) + align_up((size_t)d.pout * 8) + align_up((size_t)d.pout * 4) +
               2 * align_up((size_t)H * 8) + align_up(16 * 8);
    }
    SgdWs(void *ws, const ccnn_dims_t &d)
    {
        Carver c(ws);
        z1part = c.take<double>((size_t)F * H);
        pool = c.take<double>(d.pout);
        pidx = c.take<int>(d.pout);
        a1 = c.take<double>(H);
        d1 = c.take<double>(H);
        d2 = c.take<double>(16);
    }
};

__device__ __forceinline__ double tree13d(const double *v)
{
    double t[O];
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

__global__ __launch_bounds__(256) void sgd_fwd(double *__restrict__ params, const double *__restrict__ cube_g,
                                               int cube_len, int re, int mre, int off_b0, int off_w1,
                                               int off_b1, int off_w2, int off_b2, double lr, int pending,
                                               SgdWs ws)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *cube = reinterpret_cast<double *>(smem);
    double *w = cube + cube_len;          // filter f, 171 taps
    double *a0 = w + K + 1;               // conv outputs of filter f, re values
    double *pl = a0 + re + 1;             // pooled values, mre
    const int tid = threadIdx.x;
    const int f = blockIdx.x;

    if (pending) {
        // bp_fully_connect(out) in-place update, gcn_sgd_fp64.cu:389, 403-404, and the
        // bias half of bp_fully_connect(fc), applied one launch late
        const int g = f * 256 + tid;
        if (g < H * O) {
            const int n = g / O, o = g - n * O;
            params[off_w2 + g] = params[off_w2 + g] - lr * ws.a1[n] * ws.d2[o];
        } else if (g < H * O + O) {
            const int o = g - H * O;
            params[off_b2 + o] = params[off_b2 + o] - lr * ws.d2[o];
        } else if (g < H * O + O + H) {
            const int n = g - H * O - O;
            params[off_b1 + n] = params[off_b1 + n] - lr * ws.d1[n];
        }
    }
    if (!cube_g) return;   // flush-only launch

    for (int e = tid; e < cube_len; e += 256) cube[e] = cube_g[e];
    if (tid < K) w[tid] = params[f * K + tid];
    __syncthreads();
    if (tid < re) {
        const double *c = cube + tid * WSTEP;
        double mid = 0;
        for (int i = 0; i < K; i++) mid = mid + w[i] * c[i];   // serial tap order, :217-220
        mid = mid + params[off_b0 + f];
        a0[tid] = act(mid);
    }
    __syncthreads();
    const int pout = F * mre;
    if (tid < mre) {
        const int t = 2 * tid;
        double best = a0[t];
        int bt = t;
        if (t + 1 < re && best < a0[t + 1]) {
            best = a0[t + 1];
            bt = t + 1;
        }
        pl[tid] = best;
        ws.pool[f * mre + tid] = best;
        ws.pidx[f * mre + tid] = bt * WSTEP;
    }
    __syncthreads();
    if (tid < H) {
        const double *W1 = params + off_w1 + (long)f * mre * H;
        double acc = 0;
        for (int m = 0; m < mre; m++) acc = acc + W1[tid + H * m] * pl[m];
        ws.z1part[f * H + tid] = acc;
    }
    (void)pout;
}

__global__ __launch_bounds__(256) void sgd_bwd(double *__restrict__ params, const double *__restrict__ cube_g,
                                               const int *__restrict__ lab, int cube_len, int re, int mre,
                                               int off_b0, int off_w1, int off_b1, int off_w2, int off_b2,
                                               double lr, int sample_is_first, double *__restrict__ loss,
                                               SgdWs ws)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *cube = reinterpret_cast<double *>(smem);
    double *a1s = cube + cube_len;   // 100
    double *d1s = a1s + H;           // 100
    double *es = d1s + H;            // 16
    double *ps = es + 16;            // 16
    double *d2s = ps + 16;           // 16
    double *pl = d2s + 16;           // mre pooled values
    double *dp = pl + mre + 1;       // mre deltas
    int *px = reinterpret_cast<int *>(dp + mre + 1);
    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    const double *W2 = params + off_w2;

    for (int e = tid; e < cube_len; e += 256) cube[e] = cube_g[e];
    if (tid < mre) {
        pl[tid] = ws.pool[f * mre + tid];
        px[tid] = ws.pidx[f * mre + tid];
    }
    double a = 0;
    if (tid < H) {
        double zsum = 0;
        for (int g = 0; g < F; g++) zsum = zsum + ws.z1part[g * H + tid];
        a = 2 / (1 + 1 / exp((zsum + params[off_b1 + tid]) * 2)) - 1;   // :289
        a1s[tid] = a;
    }
    __syncthreads();
    if (tid < O) {
        double mid = 0;
        for (int i = 0; i < H; i++) mid = mid + W2[tid + i * O] * a1s[i];
        es[tid] = exp(mid + params[off_b2 + tid]);
    }
    __syncthreads();
    if (tid < O) {
        const double p = es[tid] / tree13d(es);
        ps[tid] = p;
        d2s[tid] = (p - lab[tid]) / O;   // :341
    }
    __syncthreads();
    if (tid < H) {
        double prod[O];
#pragma unroll
        for (int o = 0; o < O; o++) prod[o] = W2[o + tid * O] * d2s[o];
        d1s[tid] = tree13d(prod) * (1 + a) * (1 - a);   // :407
    }
    if (f == 0 && tid == 0) {
        // output_and_loss, :343-365
        double terms[O];
#pragma unroll
        for (int i = 0; i < O; i++) terms[i] = lab[i] * log(ps[i]) + (1 - lab[i]) * log(1 - ps[i]);
        double l = sample_is_first ? 0.0 : *loss;
        *loss = l - tree13d(terms) / O;
    }
    __syncthreads();
    if (f == 0) {
        if (tid < H) {
            ws.a1[tid] = a1s[tid];
            ws.d1[tid] = d1s[tid];
        }
        if (tid < O) ws.d2[tid] = d2s[tid];
    }

    // delta at this filter's pooled features (bp_fully_connect(fc), :386-407):
    // one wave per feature, lanes over the 100 hidden units
    double *W1 = params + off_w1 + (long)f * mre * H;
    const int lane = tid & 63, wave = tid >> 6;
    for (int m = wave; m < mre; m += 4) {
        double s = W1[lane + H * m] * d1s[lane];
        if (lane + 64 < H) s = s + W1[lane + 64 + H * m] * d1s[lane + 64];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) dp[m] = s * (1 + pl[m]) * (1 - pl[m]);
    }
    __syncthreads();
    // in-place FC1 slice update, :389 (after every product above has been read)
    for (int e = tid; e < mre * H; e += 256) {
        const int m = e / H, n = e - m * H;
        W1[e] = W1[e] - lr * pl[m] * d1s[n];
    }
    // bp_maxpooling + bp_convolution with in-place update, :411-468
    if (tid < K) {
        double mid0 = 0;
        for (int m = 0; m < mre; m++) mid0 = mid0 + dp[m] * cube[px[m] + tid];
        params[f * K + tid] = params[f * K + tid] - lr * mid0 / re;
    } else if (tid == K) {
        double mid1 = 0;
        for (int m = 0; m < mre; m++) mid1 = mid1 + dp[m] * (1 + pl[m]) * (1 - pl[m]);
        params[off_b0 + f] = params[off_b0 + f] - lr * (mid1 / re);
    }
}

}  // namespace ccnn

using namespace ccnn;

extern "C" {

size_t ccnn_sgd_workspace_bytes(int z)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return 0;
    return SgdWs::bytes(d);
}

int ccnn_sgd_f64(double *params, const double *cubes, const int *onehot, long first, long count, int z,
                 double lr, double *loss, void *workspace, size_t workspace_bytes, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    if (!params || !cubes || !onehot || !loss || !workspace || first < 0 || count < 0) return CCNN_E_INVALID;
    if (workspace_bytes < SgdWs::bytes(d)) return CCNN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    SgdWs ws(workspace, d);
    const size_t smem_f = (size_t)(d.cube + K + 1 + d.re + 1 + d.mre + 1) * 8;
    const size_t smem_b = (size_t)(d.cube + 2 * H + 48 + 2 * (d.mre + 1)) * 8 + (size_t)d.mre * 4;
    for (long i = 0; i < count; i++) {
        const long s = first + i;
        const double *cube = cubes + s * (long)d.cube;
        {
            launch("sgd_fwd", sgd_fwd, dim3(F), dim3(256), smem_f, st, params, cube, d.cube, d.re, d.mre, d.off_b0,
                   d.off_w1, d.off_b1, d.off_w2, d.off_b2, lr, i > 0, ws);
        }
        CCNN_LAUNCH_CHECK();
        {
            launch("sgd_bwd", sgd_bwd, dim3(F), dim3(256), smem_b, st, params, cube, onehot + s * O, d.cube, d.re, d.mre,
                   d.off_b0, d.off_w1, d.off_b1, d.off_w2, d.off_b2, lr, s == 0, loss, ws);
        }
        CCNN_LAUNCH_CHECK();
    }
    if (count > 0) {
        // flush the last sample's pending head update
        {
            launch("sgd_fwd", sgd_fwd, dim3(F), dim3(256), smem_f, st, params, (const double *)nullptr, d.cube, d.re,
                   d.mre, d.off_b0, d.off_w1, d.off_b1, d.off_w2, d.off_b2, lr, 1, ws);
        }
        CCNN_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
