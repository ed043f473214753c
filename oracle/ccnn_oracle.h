/*
 * TEST INFRASTRUCTURE ONLY.  CPU oracle for the Cube-CNN hot path: a plain-C
 * restatement of the reference's CUDA kernels and host driver logic, used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg to check the
 * HIP library.  Nothing in the product path may include or link this.
 *
 * Parity status: UNPINNED.  The reference ships no tests, golden vectors or
 * dataset, and cannot be built here (CUDA-only, needs nvcc and MATLAB libmat).
 * The oracle is cross-checked against an independent numpy restatement on the
 * reference's own trained checkpoint (model/KSCmodel.mat); see
 * tests/golden/make_golden.py.
 */
#ifndef CCNN_ORACLE_H
#define CCNN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Compile-time constants of the reference (gcn_sgd_fp64.cu:13-19). */
#define ORC_FILTERS 20
#define ORC_COV_LEN 19
#define ORC_STRIDE 2
#define ORC_POOL_LEN 2
#define ORC_HIDDEN 100
#ifndef ORC_CLASSES   /* NEU_NUM2: a reference user with another class count rebuilds with another value; so does the Makefile (-DORC_CLASSES=16) */
#define ORC_CLASSES 13
#endif
#define ORC_NEIGHBOR 8
#define ORC_PER_LAYER (ORC_NEIGHBOR + 1)
#define ORC_FILTER_SIZE (ORC_PER_LAYER * ORC_COV_LEN)
#define ORC_VALID_BATCH 5

typedef struct {
    int z, cube, re, mre, pin, pout;
    int off_w0, off_b0, off_w1, off_b1, off_w2, off_b2, nparams;
} orc_dims_t;

/* Derived sizes (gcn_sgd_fp64.cu:726-737) and the flat parameter layout
 * W0[F][171] | b0[F] | W1[n + 100 p] | b1[100] | W2[o + 13 n] | b2[13]. */
void orc_dims(int z, orc_dims_t *d);

/* ---- host logic ---------------------------------------------------------- */
/* gcn_sgd_fp64.cu:560-569: linear indices of labelled pixels; returns count. */
int orc_scan_labels(const double *labels, int x, int y, int *data_index);
/* gcn_sgd_fp64.cu:568-569 */
void orc_split_sizes(int n, int *train, int *test);
/* gcn_sgd_fp64.cu:576-685: neighbour lists, one-hot rows, test labels. */
void orc_build_index(const double *labels, const int *data_index, int n, int x, int y,
                     int *train_idx9, int *test_idx9, int *onehot, int *test_labels);
/* gcn_sgd_fp64.cu:526-547 with srand(seed) instead of srand(time(NULL)). */
void orc_shuffle(int *idx9, int *onehot, int n, unsigned seed);
/* Layer constructors' rand() draws, continuing the current rand() stream
 * (gcn_sgd_fp64.cu:112-145, 765-768; gcn_mbgd_fp64.cu:128-149, 904-910;
 * gcn_mbgd_fp32.cu:111-166, 901-907).  pool_bias_draws = Pout (SGD) or
 * Pout*B (fp64 MBGD). */
void orc_init_params_f64(double *params, int z, long pool_bias_draws);
void orc_init_params_f32(float *params, int z);
/* LR schedule + early stop (gcn_sgd_fp64.cu:870-884, gcn_mbgd_fp64.cu:1102-1118). */
typedef struct { double fifo[ORC_VALID_BATCH]; double cur_min; int count; } orc_sched_f64;
typedef struct { float fifo[ORC_VALID_BATCH]; float cur_min; int count; } orc_sched_f32;
void orc_sched_init_f64(orc_sched_f64 *s, double init);
int orc_sched_step_f64(orc_sched_f64 *s, double loss, double *lr);
void orc_sched_init_f32(orc_sched_f32 *s, float init);
int orc_sched_step_f32(orc_sched_f32 *s, double loss, float *lr);
/* count_err (gcn_sgd_fp64.cu:499-515): argmax with max=0, strict '>'. */
int orc_argmax_f64(const double *probs);
int orc_argmax_f32(const float *probs);

/* ---- device-kernel restatements ----------------------------------------- */
void orc_gather_f64(const double *scene, const int *idx9, double *out, int x, int y, int z, long n);
void orc_gather_f32(const float *scene, const int *idx9, float *out, int x, int y, int z, long n);

/* Test loop: probabilities [n][13] and count_err argmax [n]; flip=1 for the
 * fp32 variant's reversed filter. */
void orc_forward_f64(const double *params, const double *cubes, long n, int z, double *probs, int *argmax);
void orc_forward_f32(const float *params, const float *cubes, long n, int z, float *probs, int *argmax);
/* Intermediate activations of one sample (fixtures): a0[pin], pool[pout],
 * pool_index[pout] (as int), a1[100], probs[13]. */
void orc_forward_trace_f64(const double *params, const double *cube, int z, double *a0, double *pool,
                           int *pool_index, double *a1, double *probs);

/* fp64 per-sample SGD (gcn_sgd_fp64.cu:776-855): samples [first, first+count),
 * weights updated in place after every sample; *loss follows the device scalar
 * (reset when the sample index is 0, :343-344). */
void orc_sgd_f64(double *params, const double *cubes, const int *onehot, long first, long count,
                 int z, double lr, double *loss);

/* One minibatch step (gcn_mbgd_fp64.cu:931-1064 / gcn_mbgd_fp32.cu:925-1061).
 * loss_values[B] as loss_function writes them; probs[B][13] and grad_sum[nparams]
 * (sequential-in-sample sum of the per-sample slabs, before lr or 1/B) are optional. */
void orc_mbgd_step_f64(double *params, const double *cubes, const int *onehot, long first, int B,
                       int z, double lr, double *loss_values, double *probs, double *grad_sum);
void orc_mbgd_step_f32(float *params, const float *cubes, const int *onehot, long first, int B,
                       int z, float lr, double *loss_values, float *probs, float *grad_sum);
/* The same step with test hooks: conv outputs of every sample out (a0_out [B][F*re]), and
 * pooled features whose pair is to be decided the other way (flip [B][F*mre], non-zero). */
void orc_mbgd_step_f32_ex(float *params, const float *cubes, const int *onehot, long first, int B,
                          int z, float lr, double *loss_values, float *probs_out, float *grad_sum,
                          const int *flip, float *a0_out);

/* Multi-GPU master/worker step (hyper_version_mGPU_MBGD.cu:903-904, 923-952,
 * 962-1173) replayed on one host: n_dev devices, sub-batch B/n_dev, device slices
 * of train_size/n_dev samples, gradient slabs gathered on the master, mean over B.
 * Slab slots beyond n_dev*(B/n_dev) are defined as zero.  Returns the scalar the
 * mGPU loss_function writes (:558-612). */
double orc_mgpu_step_f64(double *params, const double *cubes, const int *onehot, long train_size,
                         long i0, int B, int n_dev, int z, double lr);
/* First sample of device slot `slot` at step i0 (slot 0 = master). */
long orc_mgpu_first_sample(long train_size, int n_dev, int B, int slot, long i0);

/* AdaGrad-like updater of hyper_version_streamwithalda.cu:325-436:
 * rms += g^2; w -= lr*g/sqrt(rms/iter + 1), g = batch sum. */
void orc_adagrad_update_f64(double *params, double *rms, const double *grad_sum, long n, double lr,
                            long iter);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
