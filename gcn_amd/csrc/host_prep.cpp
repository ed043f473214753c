// Host-side sample preparation: labelled-pixel scan, train/test split, 3x3
// neighbour lists, shuffle, weight initialisation, LR schedule, multi-GPU
// partition.  Plain C++ on host memory, exported through the C ABI so the CLI and
// the Python host share one implementation.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "ccnn_common.h"

using namespace ccnn;

extern "C" {

int ccnn_host_scan_labels(const double *labels, int x, int y, int *data_index, int *n_labelled)
{
    if (!labels || !data_index || !n_labelled || x <= 0 || y <= 0) return CCNN_E_INVALID;
    if ((long)x * y > 0x7fffffffL) return CCNN_E_RANGE;
    const int total = x * y;
    int n = 0;
    for (int i = 0; i < total; i++)
        if (labels[i] != 0) data_index[n++] = i;
    *n_labelled = n;
    return 0;
}

int ccnn_host_split_sizes(int n_labelled, int *n_train, int *n_test)
{
    if (n_labelled <= 0 || !n_train || !n_test) return CCNN_E_INVALID;
    *n_test = (n_labelled - 1) / 5 + 1;
    *n_train = n_labelled - *n_test;
    return 0;
}

// Mirrored 3x3 neighbourhood of pixel p = r + c*x.  The reference builds the raw
// offsets and then copies slot columns/rows over the out-of-scene ones
// (gcn_sgd_fp64.cu:603-625); for x, y >= 2 that is a reflection about the centre.
static inline void neighbourhood(int p, int x, int y, int *row)
{
    const int r = p % x, c = p / x;
    for (int dc = -1; dc <= 1; dc++) {
        const int cc = (c + dc < 0 || c + dc >= y) ? c - dc : c + dc;
        for (int dr = -1; dr <= 1; dr++) {
            const int rr = (r + dr < 0 || r + dr >= x) ? r - dr : r + dr;
            row[(dc + 1) * 3 + (dr + 1)] = rr + cc * x;
        }
    }
}

int ccnn_host_build_index(const double *labels, const int *data_index, int n, int x, int y,
                          int *train_idx9, int *test_idx9, int *onehot, int *test_labels)
{
    return ccnn_host_build_index_classes(labels, data_index, n, x, y, CCNN_CLASSES, train_idx9, test_idx9, onehot,
                                         test_labels);
}

int ccnn_host_build_index_classes(const double *labels, const int *data_index, int n, int x, int y, int n_classes,
                                  int *train_idx9, int *test_idx9, int *onehot, int *test_labels)
{
    if (!labels || !data_index || !train_idx9 || !test_idx9 || !onehot || !test_labels) return CCNN_E_INVALID;
    if (n <= 0 || x < 2 || y < 2) return CCNN_E_INVALID;
    if (n_classes < CCNN_MIN_CLASSES || n_classes > CCNN_MAX_CLASSES) return CCNN_E_INVALID;
    const int O = n_classes;
    int n_train, n_test;
    ccnn_host_split_sizes(n, &n_train, &n_test);
    std::memset(onehot, 0, sizeof(int) * (size_t)n_train * O);
    int tr = 0, te = 0;
    for (int k = 0; k < n; k++) {
        const int p = data_index[k];
        const int label = (int)labels[p];
        if (label < 1 || label > O) return CCNN_E_INVALID;
        if (k % 5 == 0) {
            neighbourhood(p, x, y, test_idx9 + (size_t)te * NB);
            test_labels[te++] = label;
        } else {
            neighbourhood(p, x, y, train_idx9 + (size_t)tr * NB);
            onehot[(size_t)tr * O + label - 1] = 1;
            tr++;
        }
    }
    return 0;
}

int ccnn_host_shuffle(int *idx9, int *onehot, int n, unsigned seed)
{
    return ccnn_host_shuffle_classes(idx9, onehot, n, CCNN_CLASSES, seed);
}

int ccnn_host_shuffle_classes(int *idx9, int *onehot, int n, int n_classes, unsigned seed)
{
    if (!idx9 || !onehot || n < 0) return CCNN_E_INVALID;
    if (n_classes < CCNN_MIN_CLASSES || n_classes > CCNN_MAX_CLASSES) return CCNN_E_INVALID;
    const int O = n_classes;
    srand(seed);
    for (int i = 0; i < n; i++) {
        const int j = rand() % (n - i) + i;
        if (j == i) continue;
        std::swap_ranges(idx9 + (size_t)i * NB, idx9 + (size_t)(i + 1) * NB, idx9 + (size_t)j * NB);
        std::swap_ranges(onehot + (size_t)i * O, onehot + (size_t)(i + 1) * O, onehot + (size_t)j * O);
    }
    return 0;
}

int ccnn_host_init_params_f64(double *params, int z, long pool_bias_draws)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params || pool_bias_draws < 0) return CCNN_E_INVALID;
    auto draw = [] { return (2 * (rand() / double(RAND_MAX)) - 1) / 50; };
    auto fill = [&](int off, int len) { for (int i = 0; i < len; i++) params[off + i] = draw(); };
    fill(d.off_w0, F * K);
    fill(d.off_b0, F);
    for (long i = 0; i < d.pin + pool_bias_draws; i++) (void)rand();   // pooling layer's dead tensors
    fill(d.off_w1, d.pout * H);
    fill(d.off_b1, H);
    fill(d.off_w2, H * d.n_classes);
    fill(d.off_b2, d.n_classes);
    return 0;
}

int ccnn_host_init_params_f32(float *params, int z)
{
    ccnn_dims_t d;
    if (make_dims(z, &d) || !params) return CCNN_E_INVALID;
    auto draw = [] { return (2 * (rand() / float(RAND_MAX)) - 1) / 10; };
    auto fill = [&](int off, int len) { for (int i = 0; i < len; i++) params[off + i] = draw(); };
    std::fill(params, params + d.nparams, 0.0f);
    fill(d.off_w0, F * K);
    for (long i = 0; i < d.pin; i++) (void)rand();
    fill(d.off_w1, d.pout * H);
    fill(d.off_w2, H * d.n_classes);
    return 0;
}

int ccnn_host_sched_init(ccnn_lr_sched_t *s, double fifo_init, int use_float)
{
    if (!s) return CCNN_E_INVALID;
    for (double &v : s->fifo) v = fifo_init;
    s->cur_min = fifo_init;
    s->count = 1;
    s->use_float = use_float;
    return 0;
}

int ccnn_host_sched_step(ccnn_lr_sched_t *s, double epoch_loss, double *lr)
{
    if (!s || !lr) return CCNN_E_INVALID;
    // the fp32 trainer keeps the FIFO, the minimum and lr in float
    const double v = s->use_float ? (double)(float)epoch_loss : epoch_loss;
    std::rotate(s->fifo, s->fifo + 1, s->fifo + 5);
    s->fifo[4] = v;
    const double new_min = *std::min_element(s->fifo, s->fifo + 5);
    if (s->cur_min > new_min) {
        s->cur_min = new_min;
        s->count = 1;
    } else {
        s->count++;
    }
    if (s->count < 5) return 0;
    *lr = s->use_float ? (double)(float)((float)*lr * 0.9) : *lr * 0.9;
    s->count = 1;
    s->cur_min = new_min;
    return 1;
}

int ccnn_host_partition(long n_train, int batch, int n_ranks, int rank, long step, long *first_sample,
                        int *sub_batch)
{
    if (n_train <= 0 || batch <= 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks || step < 0 ||
        !first_sample || !sub_batch)
        return CCNN_E_INVALID;
    const int sub = batch / n_ranks;
    const long slice = n_train / n_ranks;
    if (sub <= 0 || (step + 1) * (long)sub > slice) return CCNN_E_INVALID;
    *sub_batch = sub;
    *first_sample = rank * slice + step * sub;
    return 0;
}

}  // extern "C"
