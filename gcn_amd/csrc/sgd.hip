// fp64 per-sample SGD (gcn_sgd_fp64.cu:776-855) as ONE persistent launch per call.
//
// The sample sequence is serial (sample k+1 sees the weights sample k wrote), so the
// parallelism is inside one sample and the cost that matters is latency.  Twenty
// workgroups stay resident for the whole call; workgroup f owns filter f, the pooled
// features it produces and the FC1 rows those features feed, and keeps all of that in
// LDS from the first sample to the last: its 171 taps, its [mre x 100] slice of W1, and
// a private replica of the small head (W2, b2, b1), which every workgroup updates with
// identical arithmetic so the replicas never diverge.  Per sample a workgroup
//   conv + tanh + pool for its filter -> FC1 partial sums over its features
//   -> publishes the 100 partial sums, collects the other 19 workgroups' (the ONLY
//      cross-workgroup exchange of a sample)
//   -> head (a1, softmax, loss, delta2, delta1), redundantly in every workgroup
//   -> delta at its pooled features, in-place update of its W1 slice, conv gradient and
//      in-place update of its filter and bias, update of its head replica.
// The next sample's cube is fetched into registers while the current one is processed.
//
// Exchange: each partial sum travels as two 8-byte words {32 data bits, 32-bit sample
// sequence number} written and polled with relaxed agent-scope atomics, so a reader knows
// a word is current from the word itself and no fence or grid barrier is needed.  Slots are
// double-buffered by sample parity: a workgroup can only reach sample s+2 after every
// workgroup has published s+1, which each did after it finished reading s.
#include "ccnn_common.h"
#include "ccnn_device.h"
#include "profile.h"

namespace ccnn {

constexpr int SGD_XCD_STRIDE = 8;    // consecutive workgroup ids go to different XCDs: every 8th shares an L2
constexpr int SGD_THREADS = 512, SGD_WAVES = SGD_THREADS / 64;
constexpr int SGD_NREG = 8;          // cube elements a thread prefetches (covers 9 * 448 / 512)
constexpr unsigned long long SGD_SPIN_LIMIT = 400000000ull;   // 4 s of the 100 MHz wall clock

struct SgdWs {
    unsigned long long *xch;   // [2][F][H][2] exchange words
    int *status;               // [0] != 0: a workgroup gave up waiting (never expected)
    static size_t xch_words() { return (size_t)2 * F * H * 2; }
    static size_t bytes() { return align_up(xch_words() * 8) + align_up(64 + 16 * 8); }
    explicit SgdWs(void *ws)
    {
        Carver c(ws);
        xch = c.take<unsigned long long>(xch_words());
        status = c.take<int>(16);
    }
};

// LDS carve (doubles, then ints)
struct SgdLds {
    int cube, w, a0, pl, dp, w1, w2, b1, b2, z1, a1, d1, es, ps, d2, b0, total;
    size_t px_off, bytes;
};
__host__ __device__ inline SgdLds sgd_lds(const ccnn_dims_t &d)
{
    SgdLds L;
    int o = 0;
    L.cube = o; o += d.cube + 32;   // zero pad: the conv reads 192 taps per position
    L.w = o;    o += 192;
    L.a0 = o;   o += d.re + 1;
    L.pl = o;   o += d.mre + 1;
    L.dp = o;   o += d.mre + 1;
    L.w1 = o;   o += d.mre * H;
    L.w2 = o;   o += H * OMAX;
    L.b1 = o;   o += H;
    L.b2 = o;   o += 16;
    L.z1 = o;   o += H;
    L.a1 = o;   o += H;
    L.d1 = o;   o += H;
    L.es = o;   o += 16;
    L.ps = o;   o += 16;
    L.d2 = o;   o += 16;
    L.b0 = o;   o += 2;
    L.total = o;
    L.px_off = (size_t)o * 8;
    L.bytes = L.px_off + (size_t)(d.mre + 1) * 4;
    return L;
}

// NO: 13 when the class count is the reference's (compile-time then), 0 for any other (d.n_classes).
template <int NO>
__global__ __launch_bounds__(SGD_THREADS) void sgd_epoch(double *__restrict__ params,
                                                         const double *__restrict__ cubes,
                                                         const int *__restrict__ onehot, long first, long count,
                                                         ccnn_dims_t d, double lr, double *__restrict__ loss,
                                                         SgdWs ws)
{
    if (blockIdx.x % SGD_XCD_STRIDE) return;   // the 20 workers are the workgroups that share XCD 0's L2
    const int f = blockIdx.x / SGD_XCD_STRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SgdLds L = sgd_lds(d);
    double *base = reinterpret_cast<double *>(smem);
    double *cube = base + L.cube, *w = base + L.w, *a0 = base + L.a0, *pl = base + L.pl, *dp = base + L.dp;
    double *w1 = base + L.w1, *w2 = base + L.w2, *b1 = base + L.b1, *b2 = base + L.b2, *z1 = base + L.z1;
    double *a1s = base + L.a1, *d1s = base + L.d1, *ps = base + L.ps, *d2s = base + L.d2, *lt = base + L.es;
    double *b0s = base + L.b0;
    int *px = reinterpret_cast<int *>(smem + L.px_off);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int re = d.re, mre = d.mre, cube_len = d.cube, no = NO ? NO : d.n_classes;
    double *W1g = params + d.off_w1 + (long)f * mre * H;

    // ---- residents: filter, W1 slice, head replica -----------------------------------------
    if (tid < K) w[tid] = params[f * K + tid];
    if (tid >= K && tid < 192) w[tid] = 0;   // taps 171..191 read as zero by the 3-taps-per-lane conv
    for (int e = tid; e < mre * H; e += SGD_THREADS) w1[e] = W1g[e];
    for (int e = tid; e < H * no; e += SGD_THREADS) w2[e] = params[d.off_w2 + e];
    if (tid < H) b1[tid] = params[d.off_b1 + tid];
    if (tid < no) b2[tid] = params[d.off_b2 + tid];
    if (tid == 0) b0s[0] = params[d.off_b0 + f];
    for (int e = cube_len + tid; e < cube_len + 32; e += SGD_THREADS) cube[e] = 0;   // pad read by taps 171..191
    double loss_acc = (first == 0) ? 0.0 : *loss;   // used by one thread of workgroup 0

    double creg[SGD_NREG];
    auto issue = [&](long s) {   // cube of sample s -> registers (clamped, unconditional)
        const double *src = cubes + s * (long)cube_len;
#pragma unroll
        for (int u = 0; u < SGD_NREG; u++) creg[u] = src[min(u * SGD_THREADS + tid, cube_len - 1)];
    };
    if (count > 0) issue(first);
    bool aborted = false;
#ifdef CCNN_SGD_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = wall_clock64();
#define SGD_TICK(k) do { unsigned long long t_ = wall_clock64(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define SGD_TICK(k) do { } while (0)
#endif

    for (long i = 0; i < count; i++) {
        const long s = first + i;
        const int *lab = onehot + s * no;
        __syncthreads();   // the previous sample's readers of cube / pl / dp / d1s are done
#pragma unroll
        for (int u = 0; u < SGD_NREG; u++) {
            const int e = u * SGD_THREADS + tid;
            if (e < cube_len) cube[e] = creg[u];
        }
        __syncthreads();
        issue(i + 1 < count ? s + 1 : s);
        SGD_TICK(0);

        // ---- convolution + tanh (gcn_sgd_fp64.cu:217-224): wave w takes positions w, w+8, ..;
        // lanes split the 171 taps (3 each, consecutive lanes on consecutive taps: conflict-free
        // LDS reads) and 16 positions are reduced together
        {
            const double wt0 = w[lane], wt1 = w[lane + 64], wt2 = w[lane + 128];
            const double bias = b0s[0];
            const int ru = reduce_scatter_index<64>(lane);
            for (int tb = wave; tb < re; tb += 16 * SGD_WAVES) {
                double part[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int t = min(tb + u * SGD_WAVES, re - 1);
                    const double *c = cube + t * WSTEP + lane;
                    part[u] = wt0 * c[0] + wt1 * c[64] + wt2 * c[128];
                }
                const double tot = reduce_scatter16<64>(part, lane);
                const int t = tb + ru * SGD_WAVES;
                if ((lane & 3) == 0 && t < re) a0[t] = act(tot + bias);
            }
        }
        __syncthreads();
        // ---- max-pool (:237-255)
        if (tid < mre) {
            const int t = 2 * tid;
            double best = a0[t];
            int bt = t;
            if (t + 1 < re && best < a0[t + 1]) {
                best = a0[t + 1];
                bt = t + 1;
            }
            pl[tid] = best;
            px[tid] = bt * WSTEP;
        }
        __syncthreads();
        SGD_TICK(1);

        // ---- FC1 partial sums over this filter's features; publish, then collect all 20
        const unsigned seq = (unsigned)(i + 1);
        unsigned long long *xslot = ws.xch + (size_t)(i & 1) * F * H * 2;
        if (tid < H) {
            double acc = 0;
            for (int m = 0; m < mre; m++) acc = acc + w1[tid + H * m] * pl[m];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
            unsigned long long *dst = xslot + ((size_t)f * H + tid) * 2;
            __hip_atomic_store(dst, (bits & 0xffffffffull) | ((unsigned long long)seq << 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, (bits >> 32) | ((unsigned long long)seq << 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long lo[F], hi[F];
            unsigned pending = (1u << F) - 1, spins = 0;
            const unsigned long long t0 = wall_clock64();
            while (pending) {
#pragma unroll
                for (int g = 0; g < F; g++) {
                    if (pending >> g & 1) {
                        const unsigned long long *src = xslot + ((size_t)g * H + tid) * 2;
                        lo[g] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hi[g] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int g = 0; g < F; g++)
                    if ((unsigned)(lo[g] >> 32) == seq && (unsigned)(hi[g] >> 32) == seq) pending &= ~(1u << g);
                // every workgroup is resident, so this wait always ends; the limit only bounds the
                // damage if one of them died
                if (pending && (++spins & 255) == 0 &&
                    (wall_clock64() - t0 > SGD_SPIN_LIMIT ||
                     __hip_atomic_load(ws.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(ws.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    aborted = true;
                    break;
                }
            }
            double zsum = 0;
#pragma unroll
            for (int g = 0; g < F; g++)   // filter order, as the reference's tree leaves are laid out
                zsum = zsum + __longlong_as_double((long long)((lo[g] & 0xffffffffull) | (hi[g] << 32)));
            z1[tid] = zsum;
        }
        if (__syncthreads_or(aborted)) {        // every wave of the workgroup leaves together
            aborted = true;
            break;
        }
        SGD_TICK(2);

        // ---- head, redundantly per workgroup: a1 (:289), softmax (:318-340), delta2 (:341), delta1 (:407)
        double a = 0;
        if (tid < H) {
            a = 2 / (1 + 1 / exp((z1[tid] + b1[tid]) * 2)) - 1;
            a1s[tid] = a;
        }
        __syncthreads();
        if (wave == 0) {
            // one wave, no workgroup barrier inside: output o = lane >> 2 sums its 100-term dot
            // product in 4 interleaved parts, then every lane rebuilds the class-wide tree
            const int o4 = min(lane >> 2, no - 1), j4 = lane & 3;
            double mid = 0;
            for (int n = j4; n < H; n += 4) mid = mid + w2[o4 + n * no] * a1s[n];
            mid += __shfl_xor(mid, 2);
            mid += __shfl_xor(mid, 1);
            const double e = exp(mid + b2[o4]);
            double ev[OMAX];
#pragma unroll
            for (int k = 0; k < OMAX; k++) ev[k] = __shfl(e, 4 * k);
            const double p = e / tree_n(ev, no);
            if (lane < 4 * no && j4 == 0) {
                ps[o4] = p;
                d2s[o4] = (p - lab[o4]) / no;
            }
        }
        __syncthreads();
        if (tid < H) {
            double prod[OMAX];
#pragma unroll
            for (int o = 0; o < OMAX; o++) prod[o] = w2[min(o, no - 1) + tid * no] * d2s[o];
            d1s[tid] = tree_n(prod, no) * (1 + a) * (1 - a);
        }
        if (f == 0 && wave == SGD_WAVES - 1 && lane < no)   // loss terms in parallel (output_and_loss, :343-365)
            lt[lane] = lab[lane] * log(ps[lane]) + (1 - lab[lane]) * log(1 - ps[lane]);
        __syncthreads();
        SGD_TICK(3);

        // ---- delta at this filter's pooled features (bp_fully_connect(fc), :386-407): wave w takes
        // features w, w+8, ..; lanes split the 100 hidden units; 16 features reduced together
        {
            const double dl0 = d1s[lane], dl1 = lane + 64 < H ? d1s[lane + 64] : 0.0;
            const int l1 = min(lane + 64, H - 1);
            const int ru = reduce_scatter_index<64>(lane);
            for (int mb = wave; mb < mre; mb += 16 * SGD_WAVES) {
                double part[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int m = min(mb + u * SGD_WAVES, mre - 1);
                    part[u] = w1[lane + H * m] * dl0 + w1[l1 + H * m] * dl1;
                }
                const double tot = reduce_scatter16<64>(part, lane);
                const int m = mb + ru * SGD_WAVES;
                if ((lane & 3) == 0 && m < mre) dp[m] = tot * (1 + pl[m]) * (1 - pl[m]);
            }
        }
        __syncthreads();
        SGD_TICK(4);
        // ---- in-place updates, all on LDS residents, split by wave role
        if (wave < 3) {
            // bp_maxpooling + bp_convolution with in-place update, :411-468
            if (tid < K) {
                double mid0 = 0;
                for (int m = 0; m < mre; m++) mid0 = mid0 + dp[m] * cube[px[m] + tid];
                w[tid] = w[tid] - lr * mid0 / re;
            }
        } else if (wave == 3) {
            // conv bias (:463-466), FC biases (:403-404), loss accumulation
            if (lane == 0) {
                double mid1 = 0;
                for (int m = 0; m < mre; m++) mid1 = mid1 + dp[m] * (1 + pl[m]) * (1 - pl[m]);
                b0s[0] = b0s[0] - lr * (mid1 / re);
            } else if (lane == 1) {
                if (f == 0) loss_acc = loss_acc - tree_n(lt, no) / no;
            } else if (lane >= 16 && lane < 16 + no) {
                b2[lane - 16] = b2[lane - 16] - lr * d2s[lane - 16];
            }
            b1[lane] = b1[lane] - lr * d1s[lane];
            if (lane + 64 < H) b1[lane + 64] = b1[lane + 64] - lr * d1s[lane + 64];
        } else {
            // W1 slice (:389) and W2 replica (:389): element e = m*100 + n walked without divisions
            const int t4 = tid - 4 * 64;   // 0..255
            int m = t4 / H, n = t4 - m * H;
            for (int e = t4; e < mre * H; e += 256) {
                w1[e] = w1[e] - lr * pl[m] * d1s[n];
                n += 56;
                m += 2;
                if (n >= H) {
                    n -= H;
                    m++;
                }
            }
            int hn = t4 / no, ho = t4 - hn * no;
            const int step_n = 256 / no, step_o = 256 - step_n * no;   // 256 = 19 * 13 + 9 for 13 classes
            for (int e = t4; e < H * no; e += 256) {
                w2[e] = w2[e] - lr * a1s[hn] * d2s[ho];
                ho += step_o;
                hn += step_n;
                if (ho >= no) {
                    ho -= no;
                    hn++;
                }
            }
        }
        SGD_TICK(5);
    }
#ifdef CCNN_SGD_TIMING
    if (tid == 0 && f < 2)
        for (int k = 0; k < 8; k++) reinterpret_cast<unsigned long long *>(ws.status + 4)[f * 8 + k] = tacc[k];
#endif

    // ---- write the residents back -------------------------------------------------------------
    __syncthreads();
    if (tid < K) params[f * K + tid] = w[tid];
    for (int e = tid; e < mre * H; e += SGD_THREADS) W1g[e] = w1[e];
    if (tid == 0) params[d.off_b0 + f] = b0s[0];
    if (f == 0) {
        for (int e = tid; e < H * no; e += SGD_THREADS) params[d.off_w2 + e] = w2[e];
        if (tid < H) params[d.off_b1 + tid] = b1[tid];
        if (tid < no) params[d.off_b2 + tid] = b2[tid];
        // a workgroup that gave up waiting (never expected: see the residency check on the host)
        // leaves a partly trained model behind: the loss says so instead of looking like a result
        if (tid == 3 * 64 + 1) *loss = aborted ? __longlong_as_double(0x7ff8000000000000ll) : loss_acc;
    }
}

}  // namespace ccnn

using namespace ccnn;

extern "C" {

size_t ccnn_sgd_workspace_bytes(int z)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return 0;
    return SgdWs::bytes();
}

int ccnn_sgd_f64(double *params, const double *cubes, const int *onehot, long first, long count, int z,
                 double lr, double *loss, void *workspace, size_t workspace_bytes, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    if (!params || !cubes || !onehot || !loss || !workspace || first < 0 || count < 0) return CCNN_E_INVALID;
    if (workspace_bytes < SgdWs::bytes()) return CCNN_E_WORKSPACE;
    if (count >= 0xffffffffL) return CCNN_E_RANGE;   // 32-bit sequence numbers in the exchange words
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    SgdWs ws(workspace);
    // sequence numbers restart at 1 every call: stale words of an earlier call must not match
    hipError_t e = hipMemsetAsync(workspace, 0, SgdWs::bytes(), st);
    if (e != hipSuccess) return (int)e;
    const SgdLds L = sgd_lds(d);
    const auto kern = d.n_classes == CCNN_CLASSES ? sgd_epoch<CCNN_CLASSES> : sgd_epoch<0>;
    int rc = allow_smem(kern, L.bytes);
    if (rc) return rc;
    // the 20 worker workgroups wait for each other inside the launch: every workgroup of the grid
    // must be resident at once (a device with masked or too few CUs cannot run this kernel)
    if (resident_blocks(reinterpret_cast<const void *>(kern), SGD_THREADS, L.bytes) < F * SGD_XCD_STRIDE) return CCNN_E_RANGE;
    launch("sgd_epoch", kern, dim3(F * SGD_XCD_STRIDE), dim3(SGD_THREADS), L.bytes, st, params, cubes, onehot, first,
           count, d, lr, loss, ws);
    CCNN_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
