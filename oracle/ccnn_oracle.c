/*
 * TEST INFRASTRUCTURE ONLY -- see ccnn_oracle.h.  Parity status: UNPINNED (the
 * reference has no golden vectors and cannot be built in this image).
 *
 * CPU restatement of the reference's Cube-CNN hot path.  Host-side logic is
 * restated from gcn_sgd_fp64.cu:526-547 (shuffle), :560-685 (index build),
 * :870-884 (LR schedule); device kernels live in ccnn_kernels.inc.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * Contraction is off so the arithmetic is plain IEEE mul/add on every host.
 */
#include "ccnn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_dims(int z, orc_dims_t *d)
{
    d->z = z;
    d->cube = ORC_PER_LAYER * z;
    int re = 0;
    for (int i = 0; i + ORC_COV_LEN < z; i += ORC_STRIDE) /* gcn_sgd_fp64.cu:726-729 */
        re++;
    d->re = re;
    d->mre = (re - 1) / ORC_POOL_LEN + 1;
    d->pin = re * ORC_FILTERS;
    d->pout = d->mre * ORC_FILTERS;
    d->off_w0 = 0;
    d->off_b0 = d->off_w0 + ORC_FILTER_SIZE * ORC_FILTERS;
    d->off_w1 = d->off_b0 + ORC_FILTERS;
    d->off_b1 = d->off_w1 + d->pout * ORC_HIDDEN;
    d->off_w2 = d->off_b1 + ORC_HIDDEN;
    d->off_b2 = d->off_w2 + ORC_HIDDEN * ORC_CLASSES;
    d->nparams = d->off_b2 + ORC_CLASSES;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* host logic                                                                 */

int orc_scan_labels(const double *labels, int x, int y, int *data_index)
{
    int data_size = 0;
    for (int i = 0; i < x * y; i++) {
        if (labels[i] != 0) {
            data_index[data_size] = i;
            data_size++;
        }
    }
    return data_size;
}

void orc_split_sizes(int n, int *train, int *test)
{
    *test = (n - 1) / 5 + 1;
    *train = n - *test;
}

/* The NEIGHBOR==8 branch of gcn_sgd_fp64.cu:601-626 for one pixel. */
static void neighbours9(int p, int x, int y, int *row)
{
    row[4] = p;
    row[3] = p - 1;
    row[5] = p + 1;
    for (int j0 = 0; j0 < 3; j0++) {
        row[j0] = p - 1 - x + j0;
        row[j0 + 6] = p - 1 + x + j0;
    }
    if ((p % x) == 0)
        for (int j = 0; j < 3; j++)
            row[j * 3] = row[j * 3 + 2];
    if ((p % x) == (x - 1))
        for (int j = 0; j < 3; j++)
            row[j * 3 + 2] = row[j * 3];
    if ((p / x) == 0)
        for (int j = 0; j < 3; j++)
            row[j] = row[j + 6];
    if ((p / x) == (y - 1))
        for (int j = 0; j < 3; j++)
            row[j + 6] = row[j];
}

void orc_build_index(const double *labels, const int *data_index, int n, int x, int y,
                     int *train_idx9, int *test_idx9, int *onehot, int *test_labels)
{
    int train, test;
    orc_split_sizes(n, &train, &test);
    memset(onehot, 0, sizeof(int) * (size_t)train * ORC_CLASSES);
    int tr = 0, te = 0;
    for (int i = 0; i < n; i++) {
        if (i % 5 != 0) {
            neighbours9(data_index[i], x, y, train_idx9 + (size_t)tr * ORC_PER_LAYER);
            int mid = (int)labels[data_index[i]] - 1 + tr * ORC_CLASSES;
            onehot[mid] = 1;
            tr++;
        } else {
            neighbours9(data_index[i], x, y, test_idx9 + (size_t)te * ORC_PER_LAYER);
            test_labels[te] = (int)labels[data_index[i]];
            te++;
        }
    }
}

void orc_shuffle(int *data, int *labels, int width, unsigned seed)
{
    const int dim_row = ORC_PER_LAYER;
    srand(seed);
    for (int i = 0; i < width; i++) {
        int index = rand() % (width - i) + i;
        if (index != i) {
            for (int j = 0; j < dim_row; j++) {
                int temp = data[j + i * dim_row];
                data[j + i * dim_row] = data[j + index * dim_row];
                data[j + index * dim_row] = temp;
            }
            for (int j = 0; j < ORC_CLASSES; j++) {
                int tmp = labels[j + i * ORC_CLASSES];
                labels[j + i * ORC_CLASSES] = labels[j + index * ORC_CLASSES];
                labels[j + index * ORC_CLASSES] = tmp;
            }
        }
    }
}

void orc_init_params_f64(double *params, int z, long pool_bias_draws)
{
    orc_dims_t d;
    orc_dims(z, &d);
#define DRAW64() ((2 * (rand() / (double)RAND_MAX) - 1) / 50)
    for (int i = 0; i < ORC_FILTER_SIZE * ORC_FILTERS; i++) params[d.off_w0 + i] = DRAW64();
    for (int i = 0; i < ORC_FILTERS; i++) params[d.off_b0 + i] = DRAW64();
    /* the pooling Layer draws weights (pin) and bias it never uses */
    for (long i = 0; i < d.pin; i++) (void)rand();
    for (long i = 0; i < pool_bias_draws; i++) (void)rand();
    for (int i = 0; i < d.pout * ORC_HIDDEN; i++) params[d.off_w1 + i] = DRAW64();
    for (int i = 0; i < ORC_HIDDEN; i++) params[d.off_b1 + i] = DRAW64();
    for (int i = 0; i < ORC_HIDDEN * ORC_CLASSES; i++) params[d.off_w2 + i] = DRAW64();
    for (int i = 0; i < ORC_CLASSES; i++) params[d.off_b2 + i] = DRAW64();
#undef DRAW64
}

void orc_init_params_f32(float *params, int z)
{
    orc_dims_t d;
    orc_dims(z, &d);
    /* gcn_mbgd_fp32.cu:135: weights /10, biases zero without consuming rand() */
#define DRAW32() ((2 * (rand() / (float)RAND_MAX) - 1) / 10)
    memset(params, 0, sizeof(float) * (size_t)d.nparams);
    for (int i = 0; i < ORC_FILTER_SIZE * ORC_FILTERS; i++) params[d.off_w0 + i] = DRAW32();
    for (long i = 0; i < d.pin; i++) (void)rand();
    for (int i = 0; i < d.pout * ORC_HIDDEN; i++) params[d.off_w1 + i] = DRAW32();
    for (int i = 0; i < ORC_HIDDEN * ORC_CLASSES; i++) params[d.off_w2 + i] = DRAW32();
#undef DRAW32
}

void orc_sched_init_f64(orc_sched_f64 *s, double init)
{
    for (int i = 0; i < ORC_VALID_BATCH; i++) s->fifo[i] = init;
    s->cur_min = init;
    s->count = 1;
}

/* returns 1 when the learning rate was decayed this epoch */
int orc_sched_step_f64(orc_sched_f64 *s, double loss, double *lr)
{
    for (int i = 1; i < ORC_VALID_BATCH; i++) s->fifo[i - 1] = s->fifo[i];
    s->fifo[ORC_VALID_BATCH - 1] = loss;
    double new_min = s->fifo[0];
    for (int i = 1; i < ORC_VALID_BATCH; i++)
        if (s->fifo[i] < new_min) new_min = s->fifo[i];
    if (s->cur_min > new_min) {
        s->cur_min = new_min;
        s->count = 1;
    } else {
        s->count++;
    }
    if (s->count >= ORC_VALID_BATCH) {
        *lr = *lr * 0.9;
        s->count = 1;
        s->cur_min = new_min;
        return 1;
    }
    return 0;
}

void orc_sched_init_f32(orc_sched_f32 *s, float init)
{
    for (int i = 0; i < ORC_VALID_BATCH; i++) s->fifo[i] = init;
    s->cur_min = init;
    s->count = 1;
}

/* gcn_mbgd_fp32.cu:1075-1090: the FIFO and lr are float, the loss double */
int orc_sched_step_f32(orc_sched_f32 *s, double loss, float *lr)
{
    for (int i = 1; i < ORC_VALID_BATCH; i++) s->fifo[i - 1] = s->fifo[i];
    s->fifo[ORC_VALID_BATCH - 1] = (float)loss;
    float new_min = s->fifo[0];
    for (int i = 1; i < ORC_VALID_BATCH; i++)
        if (s->fifo[i] < new_min) new_min = s->fifo[i];
    if (s->cur_min > new_min) {
        s->cur_min = new_min;
        s->count = 1;
    } else {
        s->count++;
    }
    if (s->count >= ORC_VALID_BATCH) {
        *lr = (float)(*lr * 0.9);
        s->count = 1;
        s->cur_min = new_min;
        return 1;
    }
    return 0;
}

int orc_argmax_f64(const double *output)
{
    double max = 0;
    int idx = 0;
    for (int i = 0; i < ORC_CLASSES; i++)
        if (output[i] > max) {
            max = output[i];
            idx = i;
        }
    return idx;
}

int orc_argmax_f32(const float *output)
{
    float max = 0;
    int idx = 0;
    for (int i = 0; i < ORC_CLASSES; i++)
        if (output[i] > max) {
            max = output[i];
            idx = i;
        }
    return idx;
}

/* ------------------------------------------------------------------------- */
/* device kernels, both precisions                                            */

#define REAL double
#define SUF(x) x##_f64
#define R_EXP exp
#define R_LOG log
#include "ccnn_kernels.inc"
#undef REAL
#undef SUF
#undef R_EXP
#undef R_LOG

#define REAL float
#define SUF(x) x##_f32
#define R_EXP expf
#define R_LOG logf
#include "ccnn_kernels.inc"
#undef REAL
#undef SUF
#undef R_EXP
#undef R_LOG

void orc_gather_f64(const double *scene, const int *idx9, double *out, int x, int y, int z, long n)
{
    k_gather_f64(scene, idx9, out, x, y, z, n);
}

void orc_gather_f32(const float *scene, const int *idx9, float *out, int x, int y, int z, long n)
{
    k_gather_f32(scene, idx9, out, x, y, z, n);
}

static size_t fwd_ws_len(const orc_dims_t *d)
{
    return (size_t)d->pin + 2 * d->pout + ORC_HIDDEN + (d->pout > ORC_HIDDEN ? d->pout : ORC_HIDDEN);
}

void orc_forward_f64(const double *params, const double *cubes, long n, int z, double *probs, int *argmax)
{
    orc_dims_t d;
    orc_dims(z, &d);
#pragma omp parallel
    {
        double *ws = (double *)malloc(sizeof(double) * fwd_ws_len(&d));
#pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            double p[ORC_CLASSES];
            forward_one_f64(params, cubes + (size_t)i * d.cube, z, 0, p, ws);
            if (probs) memcpy(probs + i * ORC_CLASSES, p, sizeof(p));
            if (argmax) argmax[i] = orc_argmax_f64(p);
        }
        free(ws);
    }
}

void orc_forward_f32(const float *params, const float *cubes, long n, int z, float *probs, int *argmax)
{
    orc_dims_t d;
    orc_dims(z, &d);
#pragma omp parallel
    {
        float *ws = (float *)malloc(sizeof(float) * fwd_ws_len(&d));
#pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            float p[ORC_CLASSES];
            forward_one_f32(params, cubes + (size_t)i * d.cube, z, 1, p, ws);
            if (probs) memcpy(probs + i * ORC_CLASSES, p, sizeof(p));
            if (argmax) argmax[i] = orc_argmax_f32(p);
        }
        free(ws);
    }
}

void orc_forward_trace_f64(const double *params, const double *cube, int z, double *a0, double *pool,
                           int *pool_index, double *a1, double *probs)
{
    orc_dims_t d;
    orc_dims(z, &d);
    double *ws = (double *)malloc(sizeof(double) * fwd_ws_len(&d));
    forward_one_f64(params, cube, z, 0, probs, ws);
    memcpy(a0, ws, sizeof(double) * d.pin);
    memcpy(pool, ws + d.pin, sizeof(double) * d.pout);
    for (int i = 0; i < d.pout; i++) pool_index[i] = (int)ws[d.pin + d.pout + i];
    memcpy(a1, ws + d.pin + 2 * d.pout, sizeof(double) * ORC_HIDDEN);
    free(ws);
}

/* ------------------------------------------------------------------------- */
/* fp64 per-sample SGD                                                        */

/* bp_fully_connect with the in-place update, gcn_sgd_fp64.cu:372-409 */
static void sgd_bp_fully_connect(double *weights, double *bias, const double *data,
                                 const double *delta_b, double *fol_delta, int n_in, int n_out,
                                 double lr, double *scratch)
{
    for (int bid = 0; bid < n_in; bid++) {
        for (int tid = 0; tid < n_out; tid++)
            scratch[tid] = weights[tid + bid * n_out] * delta_b[tid];
        for (int tid = 0; tid < n_out; tid++)
            weights[tid + bid * n_out] = weights[tid + bid * n_out] - lr * data[bid] * delta_b[tid];
        tree_sum_f64(scratch, n_out);
        fol_delta[bid] = scratch[0] * (1 + data[bid]) * (1 - data[bid]);
    }
    for (int tid = 0; tid < n_out; tid++) /* block 0 only, :403-404 */
        bias[tid] = bias[tid] - lr * delta_b[tid];
}

/* bp_convolution with the in-place update, gcn_sgd_fp64.cu:426-468 */
static void sgd_bp_convolution(const double *pre_delta, double *filters, double *bias,
                               const double *cube, const double *conv_out, int re, double lr)
{
    const int fs = ORC_FILTER_SIZE;
    for (int bid = 0; bid < ORC_FILTERS; bid++) {
        double mid1 = 0;
        for (int i = 0; i < re; i++)
            mid1 = mid1 + pre_delta[i + bid * re] * (1 + conv_out[i + bid * re]) * (1 - conv_out[i + bid * re]);
        for (int tid = 0; tid < fs; tid++) {
            double mid0 = 0;
            for (int i = 0; i < re; i++)
                mid0 = mid0 + pre_delta[i + bid * re] * cube[tid + i * ORC_PER_LAYER * ORC_STRIDE];
            filters[tid + bid * fs] = filters[tid + bid * fs] - lr * mid0 / re;
        }
        bias[bid] = bias[bid] - lr * (mid1 / re);
    }
}

void orc_sgd_f64(double *params, const double *cubes, const int *onehot, long first, long count,
                 int z, double lr, double *loss)
{
    orc_dims_t d;
    orc_dims(z, &d);
    double *W0 = params + d.off_w0, *b0 = params + d.off_b0;
    double *W1 = params + d.off_w1, *b1 = params + d.off_b1;
    double *W2 = params + d.off_w2, *b2 = params + d.off_b2;
    size_t n = (size_t)2 * d.pin + 3 * d.pout + 2 * ORC_HIDDEN + (d.pout > ORC_HIDDEN ? d.pout : ORC_HIDDEN);
    double *ws = (double *)malloc(sizeof(double) * n);
    double *a0 = ws, *Delta = a0 + d.pin, *pool = Delta + d.pin, *pidx = pool + d.pout,
           *dpool = pidx + d.pout, *a1 = dpool + d.pout, *d1 = a1 + ORC_HIDDEN, *scratch = d1 + ORC_HIDDEN;
    double p[ORC_CLASSES], d2[ORC_CLASSES], tmp[ORC_CLASSES];

    for (long s = first; s < first + count; s++) {
        const double *cube = cubes + (size_t)s * d.cube;
        const int *lab = onehot + (size_t)s * ORC_CLASSES;
        memset(Delta, 0, sizeof(double) * d.pin); /* cudaMemset, :778 */
        k_convolution_f64(cube, W0, b0, a0, z, 0);
        k_maxpooling_f64(a0, pool, pidx, d.re, 0);
        k_fully_connect_f64(pool, W1, b1, a1, d.pout, ORC_HIDDEN, scratch);
        /* output_and_loss, :293-368 */
        for (int id = 0; id < ORC_CLASSES; id++) {
            double mid = 0;
            for (int i = 0; i < ORC_HIDDEN; i++)
                mid = mid + W2[id + i * ORC_CLASSES] * a1[i];
            p[id] = exp(mid + b2[id]);
            tmp[id] = p[id];
        }
        tree_sum_f64(tmp, ORC_CLASSES);
        for (int id = 0; id < ORC_CLASSES; id++) {
            p[id] = p[id] / tmp[0];
            d2[id] = (p[id] - lab[id]) / ORC_CLASSES;
        }
        if (s == 0) *loss = 0;
        for (int id = 0; id < ORC_CLASSES; id++)
            tmp[id] = lab[id] * log(p[id]) + (1 - lab[id]) * log(1 - p[id]);
        tree_sum_f64(tmp, ORC_CLASSES);
        *loss = *loss - tmp[0] / ORC_CLASSES;

        sgd_bp_fully_connect(W2, b2, a1, d2, d1, ORC_HIDDEN, ORC_CLASSES, lr, scratch);
        sgd_bp_fully_connect(W1, b1, pool, d1, dpool, d.pout, ORC_HIDDEN, lr, scratch);
        k_bp_maxpooling_f64(pidx, dpool, Delta, d.pout, 0);
        sgd_bp_convolution(Delta, W0, b0, cube, a0, d.re, lr);
    }
    free(ws);
}

/* ------------------------------------------------------------------------- */
/* minibatch steps                                                            */

/* fp64 loss_function, gcn_mbgd_fp64.cu:559-596 */
static double loss_sample_f64(const double *p, const int *lab)
{
    double tmp = 0.0;
    for (int i = 0; i < ORC_CLASSES; i++)
        tmp = tmp + lab[i] * log(p[i]) + (1 - lab[i]) * log(1 - p[i]);
    return -tmp / ORC_CLASSES;
}

/* fp32 loss_function, gcn_mbgd_fp32.cu:569-607: the inner expression is float
 * arithmetic (exp/log resolve to the float overloads), the accumulators double. */
static double loss_sample_f32(const float *p, const int *lab)
{
    int count_pos = 0, count_neg = 0;
    double temp_loss_pos = 0, temp_loss_neg = 0;
    for (int i = 0; i < ORC_CLASSES; i++) {
        float o = p[i];
        int ge = (o >= 0);
        if (lab[i] == 1) {
            count_pos = count_pos + 1;
            temp_loss_pos -= o * (lab[i] - ge) - logf(1 + expf(o - 2 * o * ge));
        } else if (lab[i] == 0) {
            count_neg++;
            temp_loss_neg -= o * (lab[i] - ge) - logf(1 + expf(o - 2 * o * ge));
        }
    }
    return (temp_loss_pos * count_neg / ORC_CLASSES) * 1 + (temp_loss_neg * count_pos / ORC_CLASSES);
}

#define MBGD_SAMPLE(T, SFX, FLIP)                                                                   \
    static void mbgd_sample_##SFX(const T *params, const T *cube, const int *lab, int z,            \
                                  const orc_dims_t *d, T *probs, T *d2, T *d1, T *dW0, T *dB0,      \
                                  T *dW1, T *dW2, T *ws, const int *flip, T *a0_out)                \
    {                                                                                               \
        const T *W0 = params + d->off_w0, *b0 = params + d->off_b0;                                 \
        const T *W1 = params + d->off_w1, *b1 = params + d->off_b1;                                 \
        const T *W2 = params + d->off_w2, *b2 = params + d->off_b2;                                 \
        T *a0 = ws, *Delta = a0 + d->pin, *pool = Delta + d->pin, *pidx = pool + d->pout,           \
          *dpool = pidx + d->pout, *a1 = dpool + d->pout, *scratch = a1 + ORC_HIDDEN;               \
        memset(Delta, 0, sizeof(T) * d->pin);                                                       \
        k_convolution_##SFX(cube, W0, b0, a0, z, FLIP);                                             \
        k_maxpooling_##SFX(a0, pool, pidx, d->re, 0);                                               \
        if (a0_out) memcpy(a0_out, a0, sizeof(T) * d->pin);                                         \
        if (flip) /* test hook: pooled feature p takes the OTHER member of its pair */              \
            for (int p = 0; p < d->pout; p++) {                                                     \
                const int f = p / d->mre, head = (p % d->mre) * ORC_POOL_LEN + f * d->re;           \
                if (flip[p] && head + 1 < (f + 1) * d->re) {                                        \
                    const int other = ((int)pidx[p] == head) ? head + 1 : head;                     \
                    pool[p] = a0[other];                                                            \
                    pidx[p] = (T)other;                                                             \
                }                                                                                   \
            }                                                                                       \
        k_fully_connect_##SFX(pool, W1, b1, a1, d->pout, ORC_HIDDEN, scratch);                      \
        k_output_and_dvalue_##SFX(a1, W2, b2, probs, lab, d2);                                      \
        k_bp_fully_connect_mbgd_##SFX(W2, d2, dW2, a1, d1, ORC_HIDDEN, ORC_CLASSES, scratch);       \
        k_bp_fully_connect_mbgd_##SFX(W1, d1, dW1, pool, dpool, d->pout, ORC_HIDDEN, scratch);      \
        k_bp_maxpooling_##SFX(pidx, dpool, Delta, d->pout, 0);                                      \
        k_bp_convolution_mbgd_##SFX(Delta, dW0, dB0, cube, a0, d->re);                              \
    }

MBGD_SAMPLE(double, f64, 0)
MBGD_SAMPLE(float, f32, 1)

static size_t mbgd_ws_len(const orc_dims_t *d)
{
    return (size_t)2 * d->pin + 3 * d->pout + ORC_HIDDEN + (d->pout > ORC_HIDDEN ? d->pout : ORC_HIDDEN);
}

/* Per-sample gradient slabs for samples [first, first+nb): the reference's
 * deltaW/deltaB buffers, slab[i] laid out like the flat parameter vector. */
static void mbgd_slabs_f64(const double *params, const double *cubes, const int *onehot, long first,
                           int nb, int z, const orc_dims_t *d, double *slabs, double *probs)
{
#pragma omp parallel
    {
        double *ws = (double *)malloc(sizeof(double) * mbgd_ws_len(d));
#pragma omp for schedule(static)
        for (int i1 = 0; i1 < nb; i1++) {
            long s = first + i1;
            double *g = slabs + (size_t)i1 * d->nparams;
            mbgd_sample_f64(params, cubes + (size_t)s * d->cube, onehot + (size_t)s * ORC_CLASSES, z, d,
                            probs + (size_t)i1 * ORC_CLASSES, g + d->off_b2, g + d->off_b1,
                            g + d->off_w0, g + d->off_b0, g + d->off_w1, g + d->off_w2, ws, NULL, NULL);
        }
        free(ws);
    }
}

void orc_mbgd_step_f64(double *params, const double *cubes, const int *onehot, long first, int B,
                       int z, double lr, double *loss_values, double *probs_out, double *grad_sum)
{
    orc_dims_t d;
    orc_dims(z, &d);
    double *slabs = (double *)malloc(sizeof(double) * (size_t)B * d.nparams);
    double *probs = (double *)malloc(sizeof(double) * (size_t)B * ORC_CLASSES);
    mbgd_slabs_f64(params, cubes, onehot, first, B, z, &d, slabs, probs);
    for (int i = 0; i < B; i++)
        loss_values[i] = loss_sample_f64(probs + (size_t)i * ORC_CLASSES,
                                         onehot + (size_t)(first + i) * ORC_CLASSES);
    if (probs_out) memcpy(probs_out, probs, sizeof(double) * (size_t)B * ORC_CLASSES);
    /* update_params_row x2, update_params_col: gcn_mbgd_fp64.cu:500-557.  Every
     * element is the same expression: w - lr * (sum over samples in order) / B. */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < d.nparams; k++) {
        double tmp0 = 0;
        for (int i = 0; i < B; i++)
            tmp0 = tmp0 + slabs[k + (size_t)i * d.nparams];
        if (grad_sum) grad_sum[k] = tmp0;
        params[k] = params[k] - lr * tmp0 / B;
    }
    free(slabs);
    free(probs);
}

void orc_mbgd_step_f32(float *params, const float *cubes, const int *onehot, long first, int B,
                       int z, float lr, double *loss_values, float *probs_out, float *grad_sum)
{
    orc_mbgd_step_f32_ex(params, cubes, onehot, first, B, z, lr, loss_values, probs_out, grad_sum, NULL, NULL);
}

/* orc_mbgd_step_f32 with two test hooks: a0_out [B][F*re] receives every sample's conv outputs
 * (what maxpooling compares), and flip [B][F*mre] != 0 makes that pooled feature take the other
 * member of its pair -- the step the reference would have taken had the two conv outputs, equal
 * to within rounding, compared the other way (maxpooling's strict '<', gcn_mbgd_fp32.cu:251-285). */
void orc_mbgd_step_f32_ex(float *params, const float *cubes, const int *onehot, long first, int B,
                          int z, float lr, double *loss_values, float *probs_out, float *grad_sum,
                          const int *flip, float *a0_out)
{
    orc_dims_t d;
    orc_dims(z, &d);
    float *slabs = (float *)malloc(sizeof(float) * (size_t)B * d.nparams);
    float *probs = (float *)malloc(sizeof(float) * (size_t)B * ORC_CLASSES);
#pragma omp parallel
    {
        float *ws = (float *)malloc(sizeof(float) * mbgd_ws_len(&d));
#pragma omp for schedule(static)
        for (int i1 = 0; i1 < B; i1++) {
            long s = first + i1;
            float *g = slabs + (size_t)i1 * d.nparams;
            mbgd_sample_f32(params, cubes + (size_t)s * d.cube, onehot + (size_t)s * ORC_CLASSES, z, &d,
                            probs + (size_t)i1 * ORC_CLASSES, g + d.off_b2, g + d.off_b1,
                            g + d.off_w0, g + d.off_b0, g + d.off_w1, g + d.off_w2, ws,
                            flip ? flip + (size_t)i1 * d.pout : NULL,
                            a0_out ? a0_out + (size_t)i1 * d.pin : NULL);
        }
        free(ws);
    }
    for (int i = 0; i < B; i++)
        loss_values[i] = loss_sample_f32(probs + (size_t)i * ORC_CLASSES,
                                         onehot + (size_t)(first + i) * ORC_CLASSES);
    if (probs_out) memcpy(probs_out, probs, sizeof(float) * (size_t)B * ORC_CLASSES);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < d.nparams; k++) {
        if (k < d.off_b0) {
            /* update_params_col, gcn_mbgd_fp32.cu:542-567: float accumulator, no 1/B */
            float mid0 = 0;
            for (int i = 0; i < B; i++)
                mid0 = mid0 + slabs[k + (size_t)i * d.nparams];
            if (grad_sum) grad_sum[k] = mid0;
            params[k] = params[k] - lr * mid0;
        } else if (k < d.off_w1) {
            /* conv bias: the call site passes deltaB as the bias pointer
             * (gcn_mbgd_fp32.cu:1052-1059), so the real bias never moves */
            float mid1 = 0;
            for (int i = 0; i < B; i++)
                mid1 = mid1 + slabs[k + (size_t)i * d.nparams];
            if (grad_sum) grad_sum[k] = mid1;
        } else {
            /* update_params_row, gcn_mbgd_fp32.cu:510-538: double accumulator, no 1/B */
            double tmp0 = 0;
            for (int i = 0; i < B; i++)
                tmp0 = tmp0 + slabs[k + (size_t)i * d.nparams];
            if (grad_sum) grad_sum[k] = (float)tmp0;
            params[k] = params[k] - lr * tmp0;
        }
    }
    free(slabs);
    free(probs);
}

/* ------------------------------------------------------------------------- */
/* multi-GPU master/worker step                                               */

long orc_mgpu_first_sample(long train_size, int n_dev, int B, int slot, long i0)
{
    long sub_b = B / n_dev;                  /* hyper_version_mGPU_MBGD.cu:903 */
    long slice = train_size / n_dev;         /* :934-940 */
    return slot * slice + i0 * sub_b;        /* :990 */
}

double orc_mgpu_step_f64(double *params, const double *cubes, const int *onehot, long train_size,
                         long i0, int B, int n_dev, int z, double lr)
{
    orc_dims_t d;
    orc_dims(z, &d);
    const int sub_b = B / n_dev;
    double *slabs = (double *)calloc((size_t)B * d.nparams, sizeof(double));
    double *probs = (double *)calloc((size_t)B * ORC_CLASSES, sizeof(double));
    /* every device runs its sub-batch with the broadcast weights (:962-1091);
     * slabs land on the master at slot*sub_b (:1093-1124) */
    for (int slot = 0; slot < n_dev; slot++) {
        long first = orc_mgpu_first_sample(train_size, n_dev, B, slot, i0);
        mbgd_slabs_f64(params, cubes, onehot, first, sub_b, z, &d,
                       slabs + (size_t)slot * sub_b * d.nparams, probs + (size_t)slot * sub_b * ORC_CLASSES);
    }
    /* loss_function, :558-612: thread tid sums its own sample on every device,
     * then a tree over sub_b threads */
    double *tmp = (double *)calloc(sub_b, sizeof(double));
    for (int tid = 0; tid < sub_b; tid++) {
        for (int slot = 0; slot < n_dev; slot++) {
            long s = orc_mgpu_first_sample(train_size, n_dev, B, slot, i0) + tid;
            const int *lab = onehot + (size_t)s * ORC_CLASSES;
            const double *p = probs + (size_t)(tid + slot * sub_b) * ORC_CLASSES;
            for (int i = 0; i < ORC_CLASSES; i++)
                tmp[tid] += lab[i] * log(p[i]) + (1 - lab[i]) * log(1 - p[i]);
        }
    }
    tree_sum_f64(tmp, sub_b);
    double loss = -tmp[0] / ORC_CLASSES;
    free(tmp);
    /* update_fully_connect / update_convolution over all B slabs, :499-556 */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < d.nparams; k++) {
        double tmp0 = 0;
        for (int i = 0; i < B; i++)
            tmp0 = tmp0 + slabs[k + (size_t)i * d.nparams];
        params[k] = params[k] - lr * tmp0 / B;
    }
    free(slabs);
    free(probs);
    return loss;
}

void orc_adagrad_update_f64(double *params, double *rms, const double *grad_sum, long n, double lr,
                            long iter)
{
    for (long k = 0; k < n; k++) {
        double g = grad_sum[k];
        rms[k] = rms[k] + g * g;
        params[k] = params[k] - lr * g / sqrt(rms[k] / iter + 1);
    }
}
