/*
 * libccnn_hip -- C ABI of the MI355X (gfx950) Cube-CNN hot path.
 *
 * The reference (jdjd1114/GCN) has no library or FFI surface: every kernel is a
 * `__global__ static` function launched from one monolithic training() per .cu
 * file.  Each entry point below replaces one group of those launches; the
 * comment on each names the reference launches (file:line) it stands in for.
 *
 * Conventions
 *   - plain C, no C++ or torch types; every pointer is caller-owned DEVICE
 *     memory unless its comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     every call only enqueues work on it and returns;
 *   - return value: 0 on success, a positive hipError_t from the runtime, or a
 *     negative CCNN_E_* code for arguments the kernels cannot accept.  Nothing
 *     throws, prints or exits;
 *   - no hidden state: calls on different (device, stream) pairs are independent;
 *   - `_f32` / `_f64` pairs compute in float / double.  The two precisions are
 *     not the same algorithm: `_f32` reproduces gcn_mbgd_fp32.cu (reversed
 *     filter in the forward convolution, its own loss expression), `_f64`
 *     reproduces gcn_sgd_fp64.cu / gcn_mbgd_fp64.cu.
 *
 * Layouts (all as in the reference)
 *   scene   [z][y][x], x fastest: element (r, c, b) at r + c*x + b*x*y
 *           (MATLAB column-major x-by-y-by-z array, gcn_sgd_fp64.cu:481)
 *   idx9    [n][9] int32 linear pixel indices r + c*x (gcn_sgd_fp64.cu:601-626)
 *   cubes   [n][z][9]: sample i, band b, neighbour j at i*9z + b*9 + j
 *   onehot  [n][13] int32 (n_classes wide: see CCNN_GEOMETRY)
 *   params  flat vector, ccnn_dims_t gives the offsets:
 *           W0[f*171 + tap] | b0[20] | W1[n + 100*p] | b1[100] | W2[o + 13*n] | b2[13]
 *           (the order hyper_version_mGPU_MBGD.cu:966-976 broadcasts them in;
 *           the same column-major matrices model.mat stores)
 *   grad    same layout as params
 */
#ifndef CCNN_H
#define CCNN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCNN_FILTERS 20      /* FILTER_NUM   gcn_mbgd_fp64.cu:13 */
#define CCNN_COV_LEN 19      /* COV_LEN      :14 */
#define CCNN_STRIDE 2        /* STRIDE       :15 */
#define CCNN_POOL_LEN 2      /* POOLONG_LEN  :16 */
#define CCNN_HIDDEN 100      /* NEU_NUM1     :17 */
#define CCNN_CLASSES 13      /* NEU_NUM2     :18: the class count unless the caller names another */
#define CCNN_MIN_CLASSES 2
#define CCNN_MAX_CLASSES 16  /* the output layer stays inside one 16-wide tile */
#define CCNN_PER_LAYER 9     /* NEIGHBOR + 1 :19 */
#define CCNN_FILTER_SIZE 171 /* (NEIGHBOR+1)*COV_LEN */

#define CCNN_E_INVALID (-1)   /* null pointer, non-positive size, z out of range */
#define CCNN_E_WORKSPACE (-2) /* workspace smaller than ccnn_*_workspace_bytes() */
#define CCNN_E_RANGE (-3)     /* a size exceeds what the kernels index with int32 */
#define CCNN_E_SCRATCH (-4)   /* the batch needs more than CCNN_NOALLOC_SCRATCH_BYTES of scratch */

#define CCNN_MIN_BANDS 21    /* at least one convolution output */
#define CCNN_MAX_BANDS 448     /* whole cube + filters stay under 64 KB of LDS in fp64 */

/* The class count is a compile-time constant of the reference (NEU_NUM2 = 13, one build per
 * data set).  Here it is a run-time property of the network, 2..16: wherever an entry point
 * takes the band count `z`, a caller with a class count other than 13 passes
 * CCNN_GEOMETRY(z, n_classes) instead (a plain band count means 13 classes, as in the
 * reference).  onehot / probs rows are then n_classes wide and W2 / b2 sized to match. */
#define CCNN_GEOMETRY(z, n_classes) ((z) | ((n_classes) << 16))

typedef struct ccnn_dims {
    int z, cube, re, mre, pin, pout;
    int off_w0, off_b0, off_w1, off_b1, off_w2, off_b2, nparams;
    int n_classes;
} ccnn_dims_t;

/* Derived sizes of the network for z bands (gcn_sgd_fp64.cu:726-737). Host only. */
int ccnn_get_dims(int z, ccnn_dims_t *dims);

/* Text for any code this library returns (CCNN_E_* or hipError_t). Host only. */
const char *ccnn_error_string(int code);

/* Cube gather.  Replaces preprocessing<<<64,512>>> (gcn_mbgd_fp64.cu:599-615,
 * 846-847; gcn_sgd_fp64.cu:470-486, 709-710) and processing<<<64,1024>>>
 * (preprocess048.cu:50-65, 229-232).  out[i*9z + b*9 + j] =
 * scene[idx9[i*9+j] + b*x*y] for every i < n (the trainers' single iter=0 launch
 * stops at 32768 / 65536 samples; this gathers all n like preprocess048 does).
 * workspace is optional (NULL, 0): with ccnn_gather_workspace_bytes(n) bytes the
 * samples are visited in centre-pixel order, which lets neighbouring pixels share
 * their reads; the result is the same either way. */
size_t ccnn_gather_workspace_bytes(long n);
int ccnn_gather_f32(const float *scene, const int *idx9, long n, int x, int y, int z,
                    float *out, void *workspace, size_t workspace_bytes, void *stream);
int ccnn_gather_f64(const double *scene, const int *idx9, long n, int x, int y, int z,
                    double *out, void *workspace, size_t workspace_bytes, void *stream);

/* The same gather for a sample list that is gathered more than once (several scenes with one
 * label mask, repeated passes).  ccnn_gather_plan_create looks at the list once (it synchronises
 * the stream) and keeps, in the caller's workspace, what later calls need: for lists of at least
 * 65536 plain 3x3 neighbourhoods (as ccnn_host_build_index makes them) that fill three quarters
 * of the 32 x 8 pixel tiles they touch, a pixel -> sample table and the list of those tiles - the
 * gather then stages each tile with its halo once in LDS and needs no sort (fully or densely
 * labelled scenes); for every other list, the centre-pixel order ccnn_gather_* would compute on
 * each call.  ccnn_gather_planned_* gathers the planned list from a scene; results are identical
 * to ccnn_gather_*.  idx9 and the workspace (ccnn_gather_plan_workspace_bytes(n, x, y) bytes)
 * must stay alive and unchanged while the plan is used.  ccnn_gather_plan_kind: 1 = tiles,
 * 0 = ordered list.  The launches replaced are those of ccnn_gather_*. */
typedef struct ccnn_gather_plan ccnn_gather_plan_t;
size_t ccnn_gather_plan_workspace_bytes(long n, int x, int y);
int ccnn_gather_plan_create(const int *idx9, long n, int x, int y, void *workspace,
                            size_t workspace_bytes, void *stream, ccnn_gather_plan_t **plan);
int ccnn_gather_plan_kind(const ccnn_gather_plan_t *plan);
int ccnn_gather_plan_destroy(ccnn_gather_plan_t *plan);
int ccnn_gather_planned_f32(const ccnn_gather_plan_t *plan, const float *scene, int z, float *out,
                            void *stream);
int ccnn_gather_planned_f64(const ccnn_gather_plan_t *plan, const double *scene, int z, double *out,
                            void *stream);

/* Forward pass over n cubes: the test loop's convolution / maxpooling /
 * fully_connect / output_and_dvalue(false) launches plus the host count_err
 * argmax (gcn_mbgd_fp64.cu:1182-1231, 628-652; gcn_sgd_fp64.cu:941-983, 499-515).
 * probs [n][13] and argmax [n] are each optional (NULL to skip).  argmax follows
 * count_err: running maximum starts at 0, strict '>', so all-non-positive rows
 * give 0.  workspace: ccnn_forward_workspace_bytes(n, z, elem_size) bytes. */
size_t ccnn_forward_workspace_bytes(long n, int z, int elem_size);
int ccnn_forward_f32(const float *params, const float *cubes, long n, int z, float *probs,
                     int *argmax, void *workspace, size_t workspace_bytes, void *stream);
int ccnn_forward_f64(const double *params, const double *cubes, long n, int z, double *probs,
                     int *argmax, void *workspace, size_t workspace_bytes, void *stream);

/* ccnn_forward_* fed from the scene instead of pre-gathered cubes: sample i is the
 * neighbourhood idx9[i][0..9) and its cube element b*9+j is read as
 * scene[idx9[i][j] + b*x*y] inside the convolution kernel (preprocessing,
 * gcn_mbgd_fp64.cu:599-615, folded into the test loop :1160-1209), so full-scene
 * inference never writes 9z values per pixel to memory.  Results are identical to
 * ccnn_gather_* followed by ccnn_forward_*.  Same workspace as ccnn_forward_*. */
int ccnn_forward_scene_f32(const float *params, const float *scene, const int *idx9, long n, int x,
                           int y, int z, float *probs, int *argmax, void *workspace,
                           size_t workspace_bytes, void *stream);
int ccnn_forward_scene_f64(const double *params, const double *scene, const int *idx9, long n, int x,
                           int y, int z, double *probs, int *argmax, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Number of i < n with argmax[i] + 1 == labels[i], written to *count (device int).
 * Replaces the per-pixel D2H + count_err accumulation (gcn_mbgd_fp64.cu:1226-1230). */
int ccnn_count_correct(const int *argmax, const int *labels, long n, int *count, void *stream);

/* Per-sample SGD over samples [first, first+count) of cubes/onehot, weights
 * updated in place after every sample: the per-sample launch sequence
 * cudaMemset + convolution .. bp_convolution of gcn_sgd_fp64.cu:776-855, with
 * the in-place updates of :389, :403-404, :462-465.  *loss follows the device
 * scalar of output_and_loss (:343-365): zeroed when sample 0 is processed, then
 * decreased by sum(y log p + (1-y) log(1-p))/13 per sample.
 * workspace: ccnn_sgd_workspace_bytes(z) bytes; its contents carry state between
 * calls of one training run only in params and loss.  The kernel's 20 worker workgroups
 * trade partial sums inside one launch, so the whole grid must be resident at once:
 * CCNN_E_RANGE if the device cannot hold it; should a workgroup ever give up waiting
 * (bounded wait), *loss comes back NaN and params are not a valid model. */
size_t ccnn_sgd_workspace_bytes(int z);
int ccnn_sgd_f64(double *params, const double *cubes, const int *onehot, long first, long count,
                 int z, double lr, double *loss, void *workspace, size_t workspace_bytes,
                 void *stream);

/* Minibatch gradient of samples [first, first+B): everything the reference does
 * between the start of a batch and its update kernels -- B x (convolution,
 * maxpooling, fully_connect, output_and_dvalue, bp_fully_connect x2,
 * bp_maxpooling, bp_convolution) and loss_function (gcn_mbgd_fp64.cu:931-1018;
 * gcn_mbgd_fp32.cu:925-1017) -- but with the per-sample gradient slabs already
 * summed over the batch: grad[k] = sum_i slab_i[k] (no lr, no 1/B).
 * loss_values [B] double, as loss_function writes them; probs [B][13] optional.
 * workspace: ccnn_mbgd_workspace_bytes(B, z, elem_size) bytes (a function of its arguments only,
 * never of a device); its contents need not be initialised and nothing in it has to survive from
 * one call to the next, but calls that share a workspace must be ordered by one stream.  At batch
 * sizes that leave the GPU half empty two workgroups share a sample and trade partial sums through
 * the workspace; should one of them wait for its partner in vain (2 s: never seen), the sample's
 * loss and every parameter of the step become NaN - a step that went wrong cannot pass for one
 * that did not.
 * ccnn_mbgd_path: which implementation a batch of B takes (the results are the same to the
 * tolerances above; the choice is by speed): 0 = one workgroup per sample (the reference's
 * batch sizes), 1 = one workgroup per tile of up to eight samples (large batches), 2 = one launch
 * per layer (large batches of shapes whose tile does not fit the LDS); negative: invalid
 * arguments.  The environment variable CCNN_MBGD_PATH=fused|tile|layers overrides it. */
size_t ccnn_mbgd_workspace_bytes(int B, int z, int elem_size);
int ccnn_mbgd_path(int B, int z, int elem_size);
int ccnn_mbgd_grad_f32(const float *params, const float *cubes, const int *onehot, long first,
                       int B, int z, float *grad, double *loss_values, float *probs,
                       void *workspace, size_t workspace_bytes, void *stream);
int ccnn_mbgd_grad_f64(const double *params, const double *cubes, const int *onehot, long first,
                       int B, int z, double *grad, double *loss_values, double *probs,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ccnn_mbgd_grad + the variant's own update fused into one call (single-GPU
 * training, where nothing has to happen between gradient and update): the batch
 * sums are applied to params in place and no gradient is stored.
 *   _f32: params[k] -= lr * sum, conv bias untouched   (gcn_mbgd_fp32.cu:1031-1059)
 *   _f64: params[k] -= lr * sum / B                    (gcn_mbgd_fp64.cu:1034-1060) */
int ccnn_mbgd_step_f32(float *params, const float *cubes, const int *onehot, long first, int B,
                       int z, float lr, double *loss_values, void *workspace,
                       size_t workspace_bytes, void *stream);
int ccnn_mbgd_step_f64(double *params, const double *cubes, const int *onehot, long first, int B,
                       int z, double lr, double *loss_values, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ccnn_mbgd_step_f32 without a workspace argument and without any allocation.  No
 * counterpart in the reference: it is for callers that cannot allocate device memory or
 * pass a workspace along -- code under HIP graph capture, tight serving/training loops,
 * harnesses that forbid hipMalloc.  The per-sample rows the step's two kernels hand to
 * each other live in a fixed device array inside the library, CCNN_NOALLOC_SCRATCH_BYTES
 * per GPU, held until the process ends; a batch whose rows
 * (ccnn_mbgd_workspace_bytes(B, z, 4)) do not fit is rejected with CCNN_E_SCRATCH: 8 MiB
 * holds every B up to 1946 at z = 103, up to 457 at z = 200, up to 359 at z = 448 (larger
 * batches that take the sample-tile path keep fewer rows and can fit again, e.g. 513..1142 at
 * z = 200: the size function tells).  The call makes no hipMalloc /
 * hipFree and uses no allocating library routine (keep ccnn_profile_enable off: the timing
 * hooks create events).  Results are bit-identical to ccnn_mbgd_step_f32.  All calls on one
 * GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two
 * threads) race; calls ordered by one stream are safe, and nothing has to be cleared
 * between them. */
#define CCNN_NOALLOC_SCRATCH_BYTES (8 << 20)
int ccnn_mbgd_step_noalloc_f32(float *params, const float *cubes, const int *onehot, long first,
                               int B, int z, float lr, double *loss_values, void *stream);

/* Parameter update from a summed gradient.  Replaces update_params_row x2 +
 * update_params_col (gcn_mbgd_fp64.cu:500-557, 1034-1060) and, on several GPUs,
 * the master's update over all gathered slabs
 * (hyper_version_mGPU_MBGD.cu:1144-1171) once grad has been all-reduced.
 *   _f64: params[k] -= lr * grad[k] / divisor            (divisor = batch size B)
 *   _f32: params[k] -= lr * grad[k], except k in [off_b0, off_w1): the fp32
 *         reference never updates its convolution bias (gcn_mbgd_fp32.cu:1052-1059),
 *         so those 20 entries are left alone when freeze_conv_bias != 0. */
int ccnn_update_f64(double *params, const double *grad, double lr, int divisor, long n,
                    void *stream);
int ccnn_update_f32(float *params, const float *grad, float lr, int z, int freeze_conv_bias,
                    void *stream);

/* AdaGrad-like update of hyper_version_streamwithalda.cu:325-436:
 * rms[k] += g^2; params[k] -= lr * g / sqrt(rms[k]/iter + 1), g = grad[k]
 * (batch sum), iter = epoch*batch_num + batch + 1 (:948-957). */
int ccnn_update_adagrad_f64(double *params, double *rms, const double *grad, double lr,
                            long iter, long n, void *stream);


/* ---- host-side sample preparation (all pointers are HOST memory) ----------
 * These restate the serial prologue of every reference training() so that a
 * caller gets identical sample lists, split, shuffle and initial weights. */

/* Labelled-pixel scan: data_index[k] = k-th linear index i (i = r + c*x) with
 * labels[i] != 0; *n_labelled = count (gcn_sgd_fp64.cu:560-567).  data_index
 * must hold x*y ints. */
int ccnn_host_scan_labels(const double *labels, int x, int y, int *data_index, int *n_labelled);

/* 80/20 split sizes: test = (n-1)/5 + 1, train = n - test (gcn_sgd_fp64.cu:568-569). */
int ccnn_host_split_sizes(int n_labelled, int *n_train, int *n_test);

/* Neighbour lists and labels (gcn_sgd_fp64.cu:576-685): labelled pixel k goes to
 * the test set iff k % 5 == 0.  Each idx9 row holds the 3x3 neighbourhood in the
 * order (r-1,c-1) (r,c-1) (r+1,c-1) (r-1,c) ... (r+1,c+1) with out-of-scene
 * neighbours mirrored about the centre pixel.  onehot [n_train][13] (zeroed
 * here), test_labels [n_test].  Needs x >= 2 and y >= 2 (the reference indexes
 * outside the scene otherwise) and labels in 0..13 (CCNN_E_INVALID otherwise). */
int ccnn_host_build_index(const double *labels, const int *data_index, int n_labelled, int x, int y,
                          int *train_idx9, int *test_idx9, int *onehot, int *test_labels);
/* The same with n_classes (2..16) instead of 13: onehot [n_train][n_classes], labels in
 * 1..n_classes.  The reference fixes the count at compile time (NEU_NUM2, gcn_sgd_fp64.cu:18). */
int ccnn_host_build_index_classes(const double *labels, const int *data_index, int n_labelled, int x,
                                  int y, int n_classes, int *train_idx9, int *test_idx9, int *onehot,
                                  int *test_labels);

/* Fisher-Yates over the training rows with libc rand(): srand(seed), then for
 * i: j = rand() % (n-i) + i, swap rows i and j of idx9 and onehot
 * (gcn_sgd_fp64.cu:526-547, whose seed is time(NULL)).  Leaves the rand() stream
 * where the reference's weight initialisation continues it. */
int ccnn_host_shuffle(int *idx9, int *onehot, int n_train, unsigned seed);
/* The same for onehot rows n_classes wide. */
int ccnn_host_shuffle_classes(int *idx9, int *onehot, int n_train, int n_classes, unsigned seed);

/* Initial weights from the current libc rand() stream, in the reference's draw
 * order (Layer constructors, gcn_sgd_fp64.cu:765-768, gcn_mbgd_fp64.cu:904-910,
 * gcn_mbgd_fp32.cu:901-907):
 *   _f64: every weight and bias (2*rand()/RAND_MAX - 1)/50; the pooling layer's
 *         unused weights (pin draws) and bias (pool_bias_draws: pout for SGD,
 *         pout*B for MBGD) are drawn and discarded;
 *   _f32: weights (2*rand()/float(RAND_MAX) - 1)/10, all biases zero and drawing
 *         nothing. */
int ccnn_host_init_params_f64(double *params, int z, long pool_bias_draws);
int ccnn_host_init_params_f32(float *params, int z);

/* Learning-rate schedule and early stop of every trainer's epoch loop
 * (gcn_sgd_fp64.cu:870-884, gcn_mbgd_fp64.cu:1102-1118): FIFO of the last five
 * epoch losses; lr *= 0.9 when the FIFO minimum has not improved for five
 * epochs.  fifo_init is 1 (fp64 trainers) or 100 (gcn_mbgd_fp32.cu:885-890);
 * use_float selects the fp32 trainer's float FIFO/lr arithmetic. */
typedef struct ccnn_lr_sched {
    double fifo[5];
    double cur_min;
    int count;
    int use_float;
} ccnn_lr_sched_t;
int ccnn_host_sched_init(ccnn_lr_sched_t *s, double fifo_init, int use_float);
/* Returns 1 if lr was decayed (and *lr updated), 0 if not, <0 on error. */
int ccnn_host_sched_step(ccnn_lr_sched_t *s, double epoch_loss, double *lr);

/* Data-parallel partition of hyper_version_mGPU_MBGD.cu:903-904, 934-940, 990:
 * rank r of n_ranks owns samples [r*(T/n), (r+1)*(T/n)) and at step i0 consumes
 * its local samples [i0*(B/n), (i0+1)*(B/n)).  Writes the first global sample
 * index and the per-rank batch size. */
int ccnn_host_partition(long n_train, int batch, int n_ranks, int rank, long step,
                        long *first_sample, int *sub_batch);


/* ---- multi-GPU exchange ------------------------------------------------------
 * Replaces the weight broadcast, per-device slab gather and master-side update of
 * hyper_version_mGPU_MBGD.cu:962-978, 1093-1171 by one kernel per step: every rank
 * writes its flat gradient straight into the other ranks' exchange buffers over
 * xGMI, then sums the n contributions in rank order (bit-identical on all ranks)
 * and applies the update to its own copy of the parameters.
 *
 * One ccnn_comm_t per rank and device.  ccnn_comm_create (device = the current one)
 * allocates the rank's exchange buffer for vectors of up to max_elems elements and
 * returns its IPC handle (CCNN_COMM_HANDLE_BYTES bytes); the caller passes all
 * ranks' handles, rank-major, to ccnn_comm_connect (one process per GPU), or the
 * other ranks' comm objects to ccnn_comm_connect_local (all GPUs in one process).
 * Collective calls must be issued by all ranks in the same order; on one rank they
 * are ordered by the stream.  At most 16 ranks.  ccnn_comm_status synchronises and
 * reports 1 if a rank ever gave up waiting for a peer (peer process gone or stalled).  A
 * call whose wait ran out applies NOTHING: the parameters (or the buffer) stay as the last
 * completed call left them, and every later call on that comm gives up at once. */
#define CCNN_COMM_HANDLE_BYTES 64
typedef struct ccnn_comm ccnn_comm_t;
int ccnn_comm_create(int world, int rank, long max_elems, int elem_size, ccnn_comm_t **comm,
                     void *handle);
int ccnn_comm_connect(ccnn_comm_t *comm, const void *handles);
int ccnn_comm_connect_local(ccnn_comm_t *comm, ccnn_comm_t *const *peers);
int ccnn_comm_status(ccnn_comm_t *comm, int *status);
/* *shared = 1 if, after connect, some peer rank turned out to run on this rank's own GPU
 * (several processes on one card).  Kernels that wait for their peers inside a launch then
 * compete with those peers for the same compute units: a caller should keep such ranks'
 * launches close together or use ccnn_mbgd_grad_* + ccnn_allreduce_update_* (whose waiting
 * kernel leaves room for the peers' work).  No counterpart in the reference. */
int ccnn_comm_shares_device(ccnn_comm_t *comm, int *shared);
/* How ccnn_mbgd_step_dp_* trades its sums: via_owner = 0, every rank writes each workgroup's
 * sums to every rank (one hop, world-1 vectors per rank over the links); via_owner = 1, they meet
 * at one owner rank per workgroup, which adds them and sends the total round (two hops,
 * 2 (world-1)/world of a vector per rank).  The sums are added in rank order either way: the
 * results are bit-identical.  Default: 0 (the owner form has not been timed over xGMI yet;
 * CCNN_COMM_EXCHANGE=direct|owner overrides at creation).  Every rank must use the same setting for the same call.  No
 * counterpart in the reference. */
int ccnn_comm_set_exchange(ccnn_comm_t *comm, int via_owner);
/* Tear-down is two-phase across ranks: every rank first unmaps its peers' buffers
 * (ccnn_comm_disconnect; returns the first unmap error, if any), and only when ALL ranks
 * have done so may any rank destroy its own comm (ccnn_comm_destroy; it disconnects first if
 * that was not done).  The exchange buffer itself is not given back to the allocator: the
 * library keeps it for the process's next ccnn_comm_create on that device (uncached memory that is
 * freed and handed out again proved unsafe for its next user on this runtime). */
int ccnn_comm_disconnect(ccnn_comm_t *comm);
int ccnn_comm_destroy(ccnn_comm_t *comm);
/* buf[k] = sum over ranks of buf[k], in place. */
int ccnn_allreduce_f32(ccnn_comm_t *comm, float *buf, long n, void *stream);
int ccnn_allreduce_f64(ccnn_comm_t *comm, double *buf, long n, void *stream);
/* all-reduce of grad fused with ccnn_update_f32 / ccnn_update_f64 (same arguments). */
int ccnn_allreduce_update_f32(ccnn_comm_t *comm, float *params, const float *grad, float lr, int z,
                              int freeze_conv_bias, void *stream);
int ccnn_allreduce_update_f64(ccnn_comm_t *comm, double *params, const double *grad, double lr,
                              int divisor, long n, void *stream);
/* The whole data-parallel minibatch step in two launches: ccnn_mbgd_step_* on this rank's B
 * samples with the exchange inside the batch-reduction kernel (each of its workgroups trades
 * its outputs with the same workgroup on the other ranks), so no gradient vector is stored and
 * no third kernel runs.  _f64 divides by `divisor` = the GLOBAL batch
 * (hyper_version_mGPU_MBGD.cu:518, 550); _f32 applies the plain sum with the conv bias frozen. */
int ccnn_mbgd_step_dp_f32(ccnn_comm_t *comm, float *params, const float *cubes, const int *onehot,
                          long first, int B, int z, float lr, double *loss_values, void *workspace,
                          size_t workspace_bytes, void *stream);
int ccnn_mbgd_step_dp_f64(ccnn_comm_t *comm, double *params, const double *cubes, const int *onehot,
                          long first, int B, int z, double lr, int divisor, double *loss_values,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Measurement hooks (no counterpart in the reference).  While enabled, every
 * kernel launch of this library carries a start/stop HIP event pair on its own
 * stream (attached to the dispatch, so the interval is the kernel's own).  ccnn_profile_report(i, ...) returns 1 and the i-th kernel's name,
 * summed milliseconds and launch count, 0 past the end; index 0 first waits for
 * and folds in all launches recorded so far.  Process-wide state; meant for
 * bench.py, leave disabled otherwise. */
int ccnn_profile_enable(int on);
int ccnn_profile_report(int index, char *name, int name_len, double *total_ms, long *launches);

#ifdef __cplusplus
}
#endif
#endif
