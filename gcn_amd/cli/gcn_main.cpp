// `gcn` -- drop-in command line of the reference trainers on top of libccnn_hip.
//
//   gcn DATASET.mat device_id                  single GPU  (gcn_sgd_fp64.cu:996-1045,
//                                               gcn_mbgd_fp64.cu:1244-1296, gcn_mbgd_fp32.cu:1222-1274)
//   gcn DATASET.mat device_num master_dev_id   multi GPU   (hyper_version_mGPU_MBGD.cu:1331-1388)
//
// The reference compiles one trainer per binary; here the trainer is picked with
// --variant=sgd_fp64|mbgd_fp64|mbgd_fp32 (default mbgd_fp32, what the reference's
// shipped bin/gcn was built from); the four-argument form is the multi-GPU minibatch
// trainer, fp64 like the reference's unless --variant=mbgd_fp32 is given.
// The DataSet variable is streamed from the file through two pinned buffers straight into
// every device's memory (never whole in host memory); each device gathers its own slice of the
// training cubes from its own copy of the scene on a side stream, and the test cubes are
// gathered there while training already runs.
// The multi-GPU form drives every device from a host thread of its own (the reference walks its
// devices round-robin from one thread, hyper_version_mGPU_MBGD.cu:980-1091: at ten microseconds of
// launch work per step and device, eight devices would wait for the host); the threads meet once
// per epoch, for the loss.
// Options (all before the positional arguments):
//   --variant=V  --seed=N (default: time, like the reference)  --epochs=N (300)
//   --batch=N (100)  --model=PATH (load weights instead of initialising)
//   --classes=N (2..16; default 13 like the reference's NEU_NUM2, or the largest label when the
//                label map holds 14..16 classes - the reference needs a rebuild for those)
//   --exchange=rccl|oneshot (multi-GPU form: gradient exchange, default rccl)
//   --tail=keep|drop (single GPU MBGD: train the last T % B samples as a short batch; default drop)
//   --preprocess (write traindata.mat / testdata.mat like pre-treatment/preprocess048 and stop)
//   --infer (skip training; needs --model)
//   --mat-info=FILE, --mat-rewrite=SRC:DST, --mat-stream=FILE:VAR[:CHUNK]   MAT-file utilities, no GPU needed
// stdout keeps the reference's lines and spacing so tools/run.sh's field cuts work.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../include/ccnn.h"
#include "matio.h"

// error_util.h:10-33 convention: message with file:line on stderr, non-zero exit
#define CK(x)                                                                                    \
    do {                                                                                         \
        int e_ = (int)(x);                                                                       \
        if (e_ != 0) {                                                                           \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, ccnn_error_string(e_));           \
            exit(EXIT_FAILURE);                                                                  \
        }                                                                                        \
    } while (0)
#define NCK(x)                                                                                   \
    do {                                                                                         \
        ncclResult_t r_ = (x);                                                                   \
        if (r_ != ncclSuccess) {                                                                 \
            fprintf(stderr, "%s:%d: RCCL: %s\n", __FILE__, __LINE__, ncclGetErrorString(r_));    \
            exit(EXIT_FAILURE);                                                                  \
        }                                                                                        \
    } while (0)

namespace {

typedef std::chrono::steady_clock Clock;
double secs(Clock::time_point a) { return std::chrono::duration<double>(Clock::now() - a).count(); }

struct Options {
    std::string variant = "mbgd_fp32", model;
    bool variant_given = false;
    long seed = -1;
    int epochs = 300, batch = 100, classes = 0;   // classes 0: 13, or the largest label if that is 14..16
    bool infer = false, preprocess = false, keep_tail = false;
    std::string exchange = "rccl";   // multi-GPU gradient exchange: rccl | oneshot
    std::vector<std::string> pos;
};

struct Variant {
    const char *tag;       // text inside "[Cube CNN training with ...]"
    bool sgd, f32;
    double lr, min_err, fifo;
    const char *names[6];  // model.mat variable names
    const char *model_path;
};
const Variant kSgd = {"SGD algo", true, false, 0.008, 0.0003, 1.0,
                      {"filters", "conv_bias", "fulconnect_weights", "fulconnect_bias", "output_weights", "output_bias"},
                      "model.mat"};                                      // gcn_sgd_fp64.cu:20-22, 901-931
const Variant kMbgd64 = {"MBGD algo", false, false, 0.5, 0.0001, 1.0,
                         {"filters", "bias0", "omega1", "bias1", "omega2", "bias2"}, "model/model.mat"};   // gcn_mbgd_fp64.cu:20-23
const Variant kMbgd32 = {"MBGD algo", false, true, 0.04, 0.001, 100.0,
                         {"filters", "bias0", "omega1", "bias1", "omega2", "bias2"}, "model/model.mat"};   // gcn_mbgd_fp32.cu:21-24
const Variant kMgpu = {"MBGD algo", false, false, 0.5, 0.001, 1.0,
                       {"filters", "bias0", "omega1", "bias1", "omega2", "bias2"}, "model/model.mat"};     // hyper_version_mGPU_MBGD.cu:20-23
// the same data-parallel driver around the fp32 trainer's arithmetic (gcn_mbgd_fp32.cu:21-24): no
// reference file combines the two, its constants are the single-GPU fp32 trainer's
const Variant kMgpu32 = {"MBGD algo", false, true, 0.04, 0.001, 100.0,
                         {"filters", "bias0", "omega1", "bias1", "omega2", "bias2"}, "model/model.mat"};

template <typename T>
T *dmalloc(size_t n)
{
    void *p = nullptr;
    CK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    return static_cast<T *>(p);
}
template <typename T>
T *to_device(const std::vector<T> &h)
{
    T *d = dmalloc<T>(h.size());
    if (!h.empty()) CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

// Accepts the reference's three naming schemes for the six tensors.
bool load_model(const std::string &path, const ccnn_dims_t &d, std::vector<double> &params)
{
    static const char *alias[6][3] = {{"filters", "kernel", nullptr},
                                      {"bias0", "conv_bias", nullptr},
                                      {"omega1", "fulconnect_weights", nullptr},
                                      {"bias1", "fulconnect_bias", nullptr},
                                      {"omega2", "output_weights", nullptr},
                                      {"bias2", "output_bias", nullptr}};
    const int off[7] = {d.off_w0, d.off_b0, d.off_w1, d.off_b1, d.off_w2, d.off_b2, d.nparams};
    std::vector<matio::Array> arrs;
    std::string err;
    if (!matio::read(path, {}, arrs, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return false;
    }
    params.assign(d.nparams, 0.0);
    for (int t = 0; t < 6; t++) {
        const matio::Array *found = nullptr;
        for (auto &a : arrs)
            for (int k = 0; alias[t][k]; k++)
                if (a.name == alias[t][k]) found = &a;
        const size_t want = (size_t)(off[t + 1] - off[t]);
        if (!found || found->numel() != want) {
            fprintf(stderr, "model '%s': tensor %s missing or of the wrong size (need %zu values)\n", path.c_str(),
                    alias[t][0], want);
            return false;
        }
        for (size_t i = 0; i < want; i++)
            params[off[t] + i] = found->is_single ? (double)found->f32[i] : found->f64[i];
    }
    return true;
}

void save_model(const Variant &v, const ccnn_dims_t &d, const std::vector<double> &params, const std::vector<double> &logloss)
{
    if (strchr(v.model_path, '/')) mkdir("model", 0777);
    std::vector<matio::DoubleMatrix> vars = {
        {v.names[0], (size_t)CCNN_FILTER_SIZE, (size_t)CCNN_FILTERS, params.data() + d.off_w0},
        {v.names[1], (size_t)CCNN_FILTERS, 1, params.data() + d.off_b0},
        {v.names[2], (size_t)CCNN_HIDDEN, (size_t)d.pout, params.data() + d.off_w1},
        {v.names[3], (size_t)CCNN_HIDDEN, 1, params.data() + d.off_b1},
        {v.names[4], (size_t)d.n_classes, (size_t)CCNN_HIDDEN, params.data() + d.off_w2},
        {v.names[5], (size_t)d.n_classes, 1, params.data() + d.off_b2},
        {"loss", logloss.size(), 1, logloss.data()},
    };
    std::string err;
    if (!matio::write(v.model_path, vars, err)) fprintf(stderr, "warning: %s\n", err.c_str());
}

struct Samples {
    int n_train = 0, n_test = 0;
    std::vector<int> train_idx9, test_idx9, onehot, test_labels;
};

Samples build_samples(const std::vector<double> &labels, int x, int y, int classes, unsigned seed, bool shuffle = true)
{
    Samples s;
    std::vector<int> data_index((size_t)x * y);
    int n = 0;
    CK(ccnn_host_scan_labels(labels.data(), x, y, data_index.data(), &n));
    if (n <= 0) {
        fprintf(stderr, "no labelled pixels in the dataset\n");
        exit(EXIT_FAILURE);
    }
    CK(ccnn_host_split_sizes(n, &s.n_train, &s.n_test));
    s.train_idx9.resize((size_t)s.n_train * 9);
    s.test_idx9.resize((size_t)s.n_test * 9);
    s.onehot.resize((size_t)s.n_train * classes);
    s.test_labels.resize(s.n_test);
    CK(ccnn_host_build_index_classes(labels.data(), data_index.data(), n, x, y, classes, s.train_idx9.data(),
                                     s.test_idx9.data(), s.onehot.data(), s.test_labels.data()));
    if (shuffle) CK(ccnn_host_shuffle_classes(s.train_idx9.data(), s.onehot.data(), s.n_train, classes, seed));
    return s;
}

// One host thread per device.  run(f) calls f(k) on worker k for every device at once and returns
// when all are done; worker k makes device k current once and issues every HIP / RCCL call for it,
// so no thread ever switches devices.  With a single device there are no workers: f(0) runs on the
// caller's thread.  An error inside f ends the process through CK (error_util.h's convention).
class Crew {
public:
    explicit Crew(const std::vector<int> &device_ids) : n_((int)device_ids.size())
    {
        if (n_ == 1) {
            CK(hipSetDevice(device_ids[0]));
            return;
        }
        for (int k = 0; k < n_; k++)
            workers_.emplace_back([this, k, id = device_ids[k]] {
                CK(hipSetDevice(id));
                long seen = 0;
                for (;;) {
                    std::function<void(int)> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        go_.wait(lk, [&] { return quit_ || gen_ != seen; });
                        if (quit_) return;
                        seen = gen_;
                        job = job_;
                    }
                    job(k);
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        if (--pending_ == 0) done_.notify_one();
                    }
                }
            });
    }
    ~Crew()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        go_.notify_all();
        for (auto &t : workers_) t.join();
    }
    Crew(const Crew &) = delete;
    Crew &operator=(const Crew &) = delete;
    int size() const { return n_; }
    void run(const std::function<void(int)> &f)
    {
        if (n_ == 1) {
            f(0);
            return;
        }
        std::unique_lock<std::mutex> lk(mu_);
        job_ = f;
        pending_ = n_;
        gen_++;
        go_.notify_all();
        done_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    int n_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable go_, done_;
    std::function<void(int)> job_;
    long gen_ = 0;
    int pending_ = 0;
    bool quit_ = false;
};

// Everything one device holds.
template <typename T>
struct Device {
    int id = 0;
    hipStream_t st = nullptr;    // training / testing
    hipStream_t side = nullptr;  // scene upload and gathers
    hipEvent_t train_ready = nullptr;
    T *scene = nullptr, *cubes = nullptr, *params = nullptr, *grad = nullptr;
    int *idx = nullptr, *onehot = nullptr;
    double *loss = nullptr;
    void *ws = nullptr, *gws = nullptr;
    size_t ws_bytes = 0, gws_bytes = 0;
    long n_local = 0;
};

// The two precisions of the minibatch trainer behind one set of names.
template <typename T>
struct Api;
template <>
struct Api<float> {
    static constexpr ncclDataType_t nccl = ncclFloat;
    static int gather(const float *s, const int *i, long n, int x, int y, int z, float *o, void *ws, size_t wsb, hipStream_t st)
    {
        return ccnn_gather_f32(s, i, n, x, y, z, o, ws, wsb, st);
    }
    static int step(float *p, const float *c, const int *oh, long first, int B, int z, double lr, double *lv, void *ws,
                    size_t wsb, hipStream_t st)
    {
        return ccnn_mbgd_step_f32(p, c, oh, first, B, z, (float)lr, lv, ws, wsb, st);
    }
    static int step_dp(ccnn_comm_t *cm, float *p, const float *c, const int *oh, long first, int B, int z, double lr,
                       int /*global batch: the fp32 update is a plain sum*/, double *lv, void *ws, size_t wsb, hipStream_t st)
    {
        return ccnn_mbgd_step_dp_f32(cm, p, c, oh, first, B, z, (float)lr, lv, ws, wsb, st);
    }
    static int grad(const float *p, const float *c, const int *oh, long first, int B, int z, float *g, double *lv, void *ws,
                    size_t wsb, hipStream_t st)
    {
        return ccnn_mbgd_grad_f32(p, c, oh, first, B, z, g, lv, nullptr, ws, wsb, st);
    }
    static int update(float *p, const float *g, double lr, int z, int, long, hipStream_t st)
    {
        return ccnn_update_f32(p, g, (float)lr, z, /*freeze_conv_bias=*/1, st);   // gcn_mbgd_fp32.cu:1052-1059
    }
    static int forward(const float *p, const float *c, long n, int z, int *am, void *ws, size_t wsb, hipStream_t st)
    {
        return ccnn_forward_f32(p, c, n, z, nullptr, am, ws, wsb, st);
    }
};
template <>
struct Api<double> {
    static constexpr ncclDataType_t nccl = ncclDouble;
    static int gather(const double *s, const int *i, long n, int x, int y, int z, double *o, void *ws, size_t wsb, hipStream_t st)
    {
        return ccnn_gather_f64(s, i, n, x, y, z, o, ws, wsb, st);
    }
    static int step(double *p, const double *c, const int *oh, long first, int B, int z, double lr, double *lv, void *ws,
                    size_t wsb, hipStream_t st)
    {
        return ccnn_mbgd_step_f64(p, c, oh, first, B, z, lr, lv, ws, wsb, st);
    }
    static int step_dp(ccnn_comm_t *cm, double *p, const double *c, const int *oh, long first, int B, int z, double lr,
                       int global_batch, double *lv, void *ws, size_t wsb, hipStream_t st)
    {
        return ccnn_mbgd_step_dp_f64(cm, p, c, oh, first, B, z, lr, global_batch, lv, ws, wsb, st);
    }
    static int grad(const double *p, const double *c, const int *oh, long first, int B, int z, double *g, double *lv,
                    void *ws, size_t wsb, hipStream_t st)
    {
        return ccnn_mbgd_grad_f64(p, c, oh, first, B, z, g, lv, nullptr, ws, wsb, st);
    }
    static int update(double *p, const double *g, double lr, int, int global_batch, long n, hipStream_t st)
    {
        return ccnn_update_f64(p, g, lr, global_batch, n, st);
    }
    static int forward(const double *p, const double *c, long n, int z, int *am, void *ws, size_t wsb, hipStream_t st)
    {
        return ccnn_forward_f64(p, c, n, z, nullptr, am, ws, wsb, st);
    }
};

// DataSet from the file into every device's scene buffer: the values are read (inflated,
// converted to T) a chunk at a time into one of two pinned host buffers while the other one's
// copies are still in flight on the devices' side streams (each device's copies are issued by its
// own thread of the crew).  Host memory: the two buffers.
template <typename T>
void stream_scene(matio::Stream &ds, Crew &crew, const std::vector<T *> &scene, const std::vector<hipStream_t> &side)
{
    const size_t total = ds.numel();
    const size_t nd = scene.size();
    size_t chunk = (size_t)(64u << 20) / sizeof(T);
    if (const char *e = getenv("CCNN_SCENE_CHUNK"))   // elements per chunk (tests cross chunk boundaries with a small value)
        if (atol(e) > 0) chunk = (size_t)atol(e);
    if (chunk > total) chunk = total ? total : 1;
    T *buf[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev[2];
    for (int b = 0; b < 2; b++) {
        CK(hipHostMalloc(reinterpret_cast<void **>(&buf[b]), chunk * sizeof(T), hipHostMallocPortable));
        ev[b].resize(nd);
    }
    crew.run([&](int k) {
        for (int b = 0; b < 2; b++) CK(hipEventCreateWithFlags(&ev[b][k], hipEventDisableTiming));
    });
    std::string err;
    size_t off = 0;
    for (int i = 0; off < total; i++) {
        const int b = i & 1;
        const size_t n = total - off < chunk ? total - off : chunk;
        if (i >= 2)
            for (auto e : ev[b]) CK(hipEventSynchronize(e));   // the copies that last read this buffer are done
        if (!ds.next(buf[b], n, err)) {
            fprintf(stderr, "%s\n", err.c_str());
            exit(EXIT_FAILURE);
        }
        crew.run([&](int k) {
            CK(hipMemcpyAsync(scene[k] + off, buf[b], n * sizeof(T), hipMemcpyHostToDevice, side[k]));
            CK(hipEventRecord(ev[b][k], side[k]));
        });
        off += n;
    }
    crew.run([&](int k) {
        for (int b = 0; b < 2; b++) {
            CK(hipEventSynchronize(ev[b][k]));
            CK(hipEventDestroy(ev[b][k]));
        }
    });
    for (int b = 0; b < 2; b++) CK(hipHostFree(buf[b]));
}

template <typename T>
double run(const Options &opt, const Variant &v, matio::Stream &ds, const std::vector<double> &labels, int x, int y, int z,
           int classes, const std::vector<int> &devices)
{
    const int zg = classes == CCNN_CLASSES ? z : CCNN_GEOMETRY(z, classes);   // what the network entry points take as z
    const int nd = (int)devices.size();
    const bool multi = opt.pos.size() == 3;
    Clock::time_point t0 = Clock::now();
    ccnn_dims_t d;
    CK(ccnn_get_dims(zg, &d));
    const unsigned seed = opt.seed >= 0 ? (unsigned)opt.seed : (unsigned)time(NULL);
    Samples s = build_samples(labels, x, y, classes, seed);

    // weights: libc rand() continues from the shuffle, in the reference's draw order
    const int B = opt.batch;
    std::vector<double> params_h(d.nparams);
    if (!opt.model.empty()) {
        if (!load_model(opt.model, d, params_h)) exit(EXIT_FAILURE);
    } else if (v.f32) {
        std::vector<float> p(d.nparams);
        CK(ccnn_host_init_params_f32(p.data(), zg));
        std::copy(p.begin(), p.end(), params_h.begin());
    } else {
        CK(ccnn_host_init_params_f64(params_h.data(), zg, v.sgd ? d.pout : (long)d.pout * B));
    }
    std::vector<T> params_t(params_h.begin(), params_h.end());

    // device k owns samples [k*floor(T/n), (k+1)*floor(T/n)) (hyper_version_mGPU_MBGD.cu:934-940)
    const long slice = nd > 1 ? s.n_train / nd : s.n_train;
    const int sub_b = v.sgd ? 1 : B / nd;
    const long batch_num = v.sgd ? 0 : s.n_train / B;
    // --tail=keep (single GPU): the last T % B samples train as a short final batch averaged over its
    // own size (hyper_version_MBGD.cu:780, 794-796); default: dropped (gcn_mbgd_fp64.cu:895)
    const int tail = (!v.sgd && !multi && opt.keep_tail) ? (int)(s.n_train % B) : 0;

    // Every device takes its own copy of the scene and gathers its own slice on its side stream
    // (the reference gathers everything on the master and hands slices out through host memory,
    // hyper_version_mGPU_MBGD.cu:856-857, 923-952: same cubes, no device-to-device copies) - and
    // every device has a host thread of its own for all of it.
    Crew crew(devices);
    std::vector<Device<T>> dev(nd);
    std::vector<T *> scene_ptrs(nd);
    std::vector<hipStream_t> sides(nd);
    crew.run([&](int k) {
        Device<T> &D = dev[k];
        D.id = devices[k];
        CK(hipStreamCreate(&D.st));
        CK(hipStreamCreate(&D.side));
        CK(hipEventCreateWithFlags(&D.train_ready, hipEventDisableTiming));
        D.n_local = slice;
        D.scene = dmalloc<T>((size_t)x * y * z);
        D.cubes = dmalloc<T>((size_t)slice * d.cube);
        std::vector<int> idx(s.train_idx9.begin() + (size_t)k * slice * 9, s.train_idx9.begin() + (size_t)(k + 1) * slice * 9);
        D.idx = to_device(idx);
        D.gws_bytes = ccnn_gather_workspace_bytes(slice);
        if (D.gws_bytes) CK(hipMalloc(&D.gws, D.gws_bytes));
        scene_ptrs[k] = D.scene;
        sides[k] = D.side;
    });
    stream_scene<T>(ds, crew, scene_ptrs, sides);
    ds.close();
    // test cubes: on the first device's side stream, behind its training gather - they are not
    // needed before the test loop, so the training steps below start without waiting for them
    CK(hipSetDevice(dev[0].id));
    int *te_idx_d = to_device(s.test_idx9);
    T *test_d = dmalloc<T>((size_t)s.n_test * d.cube);
    hipEvent_t test_ready;
    CK(hipEventCreateWithFlags(&test_ready, hipEventDisableTiming));
    // the test gather is issued when the first training step has been queued (below, by the first
    // device's thread): it then runs on the side stream beside the training kernels instead of in
    // front of them
    auto gather_test_cubes = [&]() {
        // the test list is already in pixel order: no ordering pass needed
        CK(Api<T>::gather(dev[0].scene, te_idx_d, s.n_test, x, y, z, test_d, nullptr, 0, dev[0].side));
        CK(hipEventRecord(test_ready, dev[0].side));
    };
    bool test_gather_issued = false;

    // per device: the gather of its slice, then - while that runs - its training state
    crew.run([&](int k) {
        Device<T> &D = dev[k];
        CK(Api<T>::gather(D.scene, D.idx, slice, x, y, z, D.cubes, D.gws, D.gws_bytes, D.side));
        CK(hipEventRecord(D.train_ready, D.side));
        CK(hipStreamWaitEvent(D.st, D.train_ready, 0));
        std::vector<int> oh(s.onehot.begin() + (size_t)k * slice * classes,
                            s.onehot.begin() + (size_t)(k + 1) * slice * classes);
        D.onehot = to_device(oh);
        D.params = to_device(params_t);
        D.grad = dmalloc<T>(d.nparams);
        D.loss = dmalloc<double>(std::max<long>(batch_num + 1, 1) * std::max(sub_b, 1));
        D.ws_bytes = v.sgd ? ccnn_sgd_workspace_bytes(zg)
                           : std::max(ccnn_mbgd_workspace_bytes(std::max(sub_b, 1), zg, (int)sizeof(T)),
                                      tail ? ccnn_mbgd_workspace_bytes(tail, zg, (int)sizeof(T)) : (size_t)0);
        CK(hipMalloc(&D.ws, D.ws_bytes ? D.ws_bytes : 1));
    });
    // Multi-GPU gradient exchange: RCCL all-reduce + update, or (--exchange=oneshot) the library's
    // one-kernel exchange over directly mapped peer buffers, fused with the update.
    const bool oneshot = multi && !v.sgd && opt.exchange == "oneshot";
    const bool exchange = nd > 1 || oneshot;
    std::vector<ncclComm_t> comms(nd);
    std::vector<ccnn_comm_t *> xch(nd, nullptr);
    if (oneshot) {
        unsigned char handle[CCNN_COMM_HANDLE_BYTES];
        for (int k = 0; k < nd; k++) {
            CK(hipSetDevice(dev[k].id));
            CK(ccnn_comm_create(nd, k, d.nparams, (int)sizeof(T), &xch[k], handle));
        }
        for (int k = 0; k < nd; k++) {
            CK(hipSetDevice(dev[k].id));
            CK(ccnn_comm_connect_local(xch[k], xch.data()));
        }
    } else if (nd > 1) {
        // RCCL prints a version banner on file descriptor 1 when it initialises; stdout belongs to the
        // reference's own lines (tools/run.sh cuts fields out of them), so the banner goes to stderr
        fflush(stdout);
        const int saved_stdout = dup(1);
        dup2(2, 1);
        const ncclResult_t init_rc = ncclCommInitAll(comms.data(), nd, devices.data());
        fflush(stdout);
        dup2(saved_stdout, 1);
        close(saved_stdout);
        NCK(init_rc);
    }
    crew.run([&](int k) { CK(hipEventSynchronize(dev[k].train_ready)); });
    fprintf(stdout,
            "[Samples prepared with %d Nearest-Neighbor-Pixels Strategy] Proportion of Training Samples: %d%%  Execution "
            "time: %.3f sec\n",
            8, 80, secs(t0));

    std::vector<double> logloss(300, 0.0);
    char head[96];
    if (v.sgd) snprintf(head, sizeof(head), "[Cube CNN training with SGD algo");
    else snprintf(head, sizeof(head), "[Cube CNN training with MBGD algo.  BatchSize = %d", B);
    double lr = v.lr;
    Clock::time_point t_train = Clock::now();
    if (!opt.infer) {
        if (!v.sgd && (sub_b <= 0 || batch_num * sub_b > slice)) {
            fprintf(stderr, "batch size %d does not fit %d device(s) with %d training samples\n", B, nd, s.n_train);
            exit(EXIT_FAILURE);
        }
        ccnn_lr_sched_t sched;
        CK(ccnn_host_sched_init(&sched, v.fifo, v.f32));
        fprintf(stdout, "%s] lr = %lf\n", head, lr);
        const size_t loss_n = (size_t)std::max<long>(batch_num + 1, 1) * std::max(sub_b, 1);
        std::vector<std::vector<double>> loss_h(nd, std::vector<double>(loss_n));
        std::vector<int> xch_status(nd, 0);
        for (int iter = 0; iter < opt.epochs; iter++) {
            Clock::time_point t_it = Clock::now();
            double single_rate = 0;
            if (v.sgd) {
                Device<T> &D = dev[0];
                CK(ccnn_sgd_f64((double *)D.params, (const double *)D.cubes, D.onehot, 0, s.n_train, zg, lr, D.loss,
                                D.ws, D.ws_bytes, D.st));
                double l = 0;
                CK(hipMemcpyAsync(&l, D.loss, sizeof(double), hipMemcpyDeviceToHost, D.st));
                CK(hipStreamSynchronize(D.st));
                single_rate = l / s.n_train;
                if (single_rate != single_rate) {   // NaN: the kernel's workgroups lost each other, or the run diverged
                    fprintf(stderr, "%s:%d: epoch %d: NaN loss, the parameters are not a valid model\n", __FILE__, __LINE__, iter + 1);
                    exit(EXIT_FAILURE);
                }
                if (iter < 300) logloss[iter] = single_rate;
                if (single_rate < v.min_err) break;   // before printing and scheduling (gcn_sgd_fp64.cu:860-861)
            } else {
                // One epoch of device k, on its own thread: every step is queued without waiting for
                // anything (the kernels of a step wait for their peers on the devices; RCCL orders its
                // all-reduce on the device's stream), then the epoch's per-sample losses come back in
                // one copy.  The devices' steps are therefore enqueued side by side, not one device
                // after another (hyper_version_mGPU_MBGD.cu:980-1091 walks them in turn).
                crew.run([&](int k) {
                    Device<T> &D = dev[k];
                    for (long i0 = 0; i0 < batch_num; i0++) {
                        double *lv = D.loss + i0 * sub_b;
                        if (!exchange) {
                            CK(Api<T>::step(D.params, D.cubes, D.onehot, i0 * sub_b, sub_b, zg, lr, lv, D.ws, D.ws_bytes, D.st));
                        } else if (oneshot) {
                            // gradient, exchange with the other devices and update in two launches
                            CK(Api<T>::step_dp(xch[k], D.params, D.cubes, D.onehot, i0 * sub_b, sub_b, zg, lr, B, lv, D.ws,
                                               D.ws_bytes, D.st));
                        } else {
                            // gradient, all-reduce, update: replaces weight broadcast + slab gather + master
                            // update (hyper_version_mGPU_MBGD.cu:962-978, 1093-1171)
                            CK(Api<T>::grad(D.params, D.cubes, D.onehot, i0 * sub_b, sub_b, zg, D.grad, lv, D.ws, D.ws_bytes,
                                            D.st));
                            if (nd > 1) NCK(ncclAllReduce(D.grad, D.grad, d.nparams, Api<T>::nccl, ncclSum, comms[k], D.st));
                            CK(Api<T>::update(D.params, D.grad, lr, zg, B, d.nparams, D.st));
                        }
                        if (k == 0 && !test_gather_issued) {   // the first step is queued: the test cubes are gathered beside the training
                            gather_test_cubes();
                            test_gather_issued = true;
                        }
                    }
                    if (k == 0 && tail) {
                        double *lv = D.loss + batch_num * sub_b;
                        CK(Api<T>::step(D.params, D.cubes, D.onehot, batch_num * sub_b, tail, zg, lr, lv, D.ws, D.ws_bytes, D.st));
                    }
                    // one D2H per device per epoch
                    CK(hipMemcpyAsync(loss_h[k].data(), D.loss, sizeof(double) * (batch_num * sub_b + (k == 0 ? tail : 0)),
                                      hipMemcpyDeviceToHost, D.st));
                    CK(hipStreamSynchronize(D.st));
                    // a wait that ran out inside the exchange leaves the parameters at the last completed
                    // step and raises the status word
                    if (oneshot) CK(ccnn_comm_status(xch[k], &xch_status[k]));
                });
                // the host adds the per-sample losses in device and sample order
                float loss32 = 0;
                double loss64 = 0;
                for (int k = 0; k < nd; k++) {
                    for (long j = 0; j < batch_num * sub_b + (k == 0 ? tail : 0); j++) {
                        if (v.f32) loss32 = loss32 + loss_h[k][j];   // `float loss` (gcn_mbgd_fp32.cu:882, 1027)
                        else loss64 = loss64 + loss_h[k][j];
                    }
                }
                single_rate = (v.f32 ? (double)loss32 : loss64) / s.n_train;
                if (iter < 300) logloss[iter] = single_rate;
                for (int k = 0; k < nd; k++) {
                    if (xch_status[k]) {   // stop instead of training on
                        fprintf(stderr, "%s:%d: device %d gave up waiting for a peer inside the gradient exchange (epoch %d)\n",
                                __FILE__, __LINE__, dev[k].id, iter + 1);
                        exit(EXIT_FAILURE);
                    }
                }
            }
            char str[50];
            snprintf(str, sizeof(str), "%d,", iter + 1);
            if (v.sgd)
                fprintf(stdout, "[Cube CNN training with SGD algo. Execution time: %.3f sec] Iteration %-4s loss = %lf;\n",
                        secs(t_it), str, single_rate);
            else
                fprintf(stdout, "%s  Execution time: %.3f sec] Iteration %-4s loss = %lf;\n", head, secs(t_it), str,
                        single_rate);
            if (ccnn_host_sched_step(&sched, single_rate, &lr) == 1) fprintf(stdout, "%s] lr = %lf\n", head, lr);
            if (!v.sgd && single_rate < v.min_err) break;
        }
        if (v.sgd) fprintf(stdout, "[Cube CNN training with SGD algo] Completed! Global Execution time is %.3f sec\n", secs(t_train));
        else fprintf(stdout, "%s] Completed! Global Exesution time is %.3f sec\n", head, secs(t_train));

        CK(hipSetDevice(dev[0].id));
        CK(hipMemcpy(params_t.data(), dev[0].params, sizeof(T) * d.nparams, hipMemcpyDeviceToHost));
        std::copy(params_t.begin(), params_t.end(), params_h.begin());
        if (!multi) save_model(v, d, params_h, logloss);   // the multi-GPU reference saves nothing (:1214-1252)
    }

    // test on the first device
    Clock::time_point t_test = Clock::now();
    CK(hipSetDevice(dev[0].id));
    if (!test_gather_issued) gather_test_cubes();   // inference only, or per-sample SGD
    CK(hipEventSynchronize(test_ready));   // the test cubes were gathered on the side stream meanwhile
    crew.run([&](int k) {                   // nothing reads the scene copies any more
        CK(hipFree(dev[k].scene));
        dev[k].scene = nullptr;
    });
    int *argmax_d = dmalloc<int>(s.n_test), *count_d = dmalloc<int>(1), *tl_d = to_device(s.test_labels);
    const size_t fws_bytes = ccnn_forward_workspace_bytes(s.n_test, zg, (int)sizeof(T));
    void *fws = nullptr;
    CK(hipMalloc(&fws, fws_bytes ? fws_bytes : 1));
    CK(Api<T>::forward(dev[0].params, test_d, s.n_test, zg, argmax_d, fws, fws_bytes, dev[0].st));
    CK(ccnn_count_correct(argmax_d, tl_d, s.n_test, count_d, dev[0].st));
    int right = 0;
    CK(hipMemcpyAsync(&right, count_d, sizeof(int), hipMemcpyDeviceToHost, dev[0].st));
    CK(hipStreamSynchronize(dev[0].st));
    fprintf(stdout, "[Cube CNN testing] Execution time is %.3f sec. ", secs(t_test));
    if (oneshot) {
        crew.run([&](int) { CK(hipDeviceSynchronize()); });
        for (auto c : xch) ccnn_comm_disconnect(c);
        for (auto c : xch) ccnn_comm_destroy(c);
        CK(hipSetDevice(dev[0].id));
    } else if (nd > 1) {
        for (auto c : comms) ncclCommDestroy(c);
    }
    return s.n_test ? (double)right / s.n_test : 0.0;
}

// `gcn --preprocess DATASET.mat device_id`: the stand-alone gather of pre-treatment/preprocess048.cu
// (:66-262): cubes of all labelled pixels in scan order (no shuffle), fp64, written as
// testdata.mat {data (9z x test_size), labels (test_size x 1)} and
// traindata.mat {data (9z x train_size), labels (13 x train_size, one-hot)}.
int preprocess_dump(matio::Stream &ds, const std::vector<double> &labels, int x, int y, int z, int classes, int device)
{
    Clock::time_point t0 = Clock::now();
    ccnn_dims_t d;
    CK(ccnn_get_dims(z, &d));
    Samples s = build_samples(labels, x, y, classes, 0, false);
    fprintf(stdout, "train_size:%d  test_size:%d\n", s.n_train, s.n_test);
    fprintf(stdout, "Index computing completed!\n");
    CK(hipSetDevice(device));
    double *scene_d = dmalloc<double>((size_t)x * y * z);
    hipStream_t side;
    CK(hipStreamCreate(&side));
    Crew crew({device});
    stream_scene<double>(ds, crew, {scene_d}, {side});
    ds.close();
    std::vector<double> train((size_t)s.n_train * d.cube), test((size_t)s.n_test * d.cube);
    struct Part {
        const std::vector<int> *idx;
        int n;
        std::vector<double> *out;
    } parts[2] = {{&s.train_idx9, s.n_train, &train}, {&s.test_idx9, s.n_test, &test}};
    for (auto &p : parts) {
        if (p.n == 0) continue;
        int *idx_d = to_device(*p.idx);
        double *out_d = dmalloc<double>((size_t)p.n * d.cube);
        CK(ccnn_gather_f64(scene_d, idx_d, p.n, x, y, z, out_d, nullptr, 0, nullptr));   // scan order: no ordering pass
        CK(hipMemcpy(p.out->data(), out_d, sizeof(double) * p.out->size(), hipMemcpyDeviceToHost));
        CK(hipFree(out_d));
        CK(hipFree(idx_d));
    }
    CK(hipFree(scene_d));
    if (train.size() >= 4) fprintf(stdout, "Processed train data:%lf %lf %lf %lf\n", train[0], train[1], train[2], train[3]);
    if (test.size() >= 4) fprintf(stdout, "Processed test data:%lf %lf %lf %lf\n", test[0], test[1], test[2], test[3]);
    std::vector<double> onehot(s.onehot.begin(), s.onehot.end()), tl(s.test_labels.begin(), s.test_labels.end());
    std::string err;
    if (!matio::write("testdata.mat", {{"data", (size_t)d.cube, (size_t)s.n_test, test.data()},
                                       {"labels", (size_t)s.n_test, 1, tl.data()}}, err) ||
        !matio::write("traindata.mat", {{"data", (size_t)d.cube, (size_t)s.n_train, train.data()},
                                        {"labels", (size_t)classes, (size_t)s.n_train, onehot.data()}}, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    fprintf(stdout, "Using time of preprocessing:%lfs\n", secs(t0));
    return 0;
}

// File utilities that need no GPU: list a MAT-file's numeric variables, and
// optionally write its real double 2-D matrices back out with this writer.
int mat_info(const std::string &src, const std::string &dst)
{
    std::vector<matio::Array> arrs;
    std::string err;
    if (!matio::read(src, {}, arrs, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    std::vector<matio::DoubleMatrix> vars;
    for (auto &a : arrs) {
        double sum = 0;
        for (double v : a.f64) sum += v;
        for (float v : a.f32) sum += v;
        printf("%s %s", a.name.c_str(), a.is_single ? "single" : "double");
        for (size_t dmn : a.dims) printf(" %zu", dmn);
        printf(" sum=%.17g\n", sum);
        if (!a.is_single && a.dims.size() == 2) vars.push_back({a.name, a.dims[0], a.dims[1], a.f64.data()});
    }
    if (!dst.empty() && !matio::write(dst, vars, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    return 0;
}

// `--mat-stream=FILE:VAR[:CHUNK]`: the streaming reader on one variable, CHUNK values at a time
// as float (single arrays) or double; prints class, dimensions, sum and the peak resident set.
int mat_stream(const std::string &spec)
{
    const size_t c1 = spec.find(':');
    if (c1 == std::string::npos) {
        fprintf(stderr, "--mat-stream=FILE:VAR[:CHUNK]\n");
        return 1;
    }
    const size_t c2 = spec.find(':', c1 + 1);
    const std::string file = spec.substr(0, c1), var = spec.substr(c1 + 1, c2 == std::string::npos ? c2 : c2 - c1 - 1);
    long chunk_arg = 1L << 20;
    if (c2 != std::string::npos) {
        char *end = nullptr;
        chunk_arg = strtol(spec.c_str() + c2 + 1, &end, 10);
        if (end == spec.c_str() + c2 + 1 || *end != '\0' || chunk_arg <= 0) {   // not a number, zero or negative
            fprintf(stderr, "--mat-stream=FILE:VAR[:CHUNK]: CHUNK must be a positive count of values\n");
            return 1;
        }
    }
    matio::Stream st;
    std::string err;
    if (!st.open(file, var, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    size_t chunk = (size_t)chunk_arg;
    if (chunk > st.numel()) chunk = st.numel() ? st.numel() : 1;   // never a buffer larger than the variable
    double sum = 0;
    std::vector<float> f32(st.is_single() ? chunk : 0);
    std::vector<double> f64(st.is_single() ? 0 : chunk);
    while (st.remaining()) {
        const size_t n = st.remaining() < chunk ? st.remaining() : chunk;
        if (!(st.is_single() ? st.next(f32.data(), n, err) : st.next(f64.data(), n, err))) {
            fprintf(stderr, "%s\n", err.c_str());
            return 1;
        }
        for (size_t i = 0; i < n; i++) sum += st.is_single() ? (double)f32[i] : f64[i];
    }
    long rss_kb = 0;
    if (FILE *f = fopen("/proc/self/status", "r")) {
        char line[256];
        while (fgets(line, sizeof(line), f))
            if (sscanf(line, "VmHWM: %ld kB", &rss_kb) == 1) break;
        fclose(f);
    }
    printf("%s %s", var.c_str(), st.is_single() ? "single" : "double");
    for (size_t dmn : st.dims()) printf(" %zu", dmn);
    printf(" sum=%.17g peak_rss_kb=%ld\n", sum, rss_kb);
    return 0;
}

}  // namespace

int main(int argc, char *argv[])
{
    Options opt;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a.rfind("--mat-info=", 0) == 0) return mat_info(a.substr(11), "");
        if (a.rfind("--mat-stream=", 0) == 0) return mat_stream(a.substr(13));
        if (a.rfind("--mat-rewrite=", 0) == 0) {
            const size_t c = a.find(':', 14);
            if (c == std::string::npos) {
                fprintf(stderr, "--mat-rewrite=SRC:DST\n");
                return 1;
            }
            return mat_info(a.substr(14, c - 14), a.substr(c + 1));
        }
        if (a.rfind("--variant=", 0) == 0) {
            opt.variant = a.substr(10);
            opt.variant_given = true;
        }
        else if (a.rfind("--seed=", 0) == 0) opt.seed = atol(a.c_str() + 7);
        else if (a.rfind("--epochs=", 0) == 0) opt.epochs = atoi(a.c_str() + 9);
        else if (a.rfind("--batch=", 0) == 0) opt.batch = atoi(a.c_str() + 8);
        else if (a.rfind("--classes=", 0) == 0) opt.classes = atoi(a.c_str() + 10);
        else if (a.rfind("--exchange=", 0) == 0) opt.exchange = a.substr(11);
        else if (a.rfind("--model=", 0) == 0) opt.model = a.substr(8);
        else if (a == "--infer") opt.infer = true;
        else if (a == "--preprocess") opt.preprocess = true;
        else if (a == "--tail=keep") opt.keep_tail = true;
        else if (a == "--tail=drop") opt.keep_tail = false;
        else opt.pos.push_back(a);
    }
    if (const char *e = getenv("CCNN_SEED"))
        if (opt.seed < 0) opt.seed = atol(e);
    const bool multi = opt.pos.size() == 3;
    const Variant *v = multi ? (!opt.variant_given || opt.variant == "mbgd_fp64" ? &kMgpu
                                : opt.variant == "mbgd_fp32" ? &kMgpu32 : nullptr)
                             : opt.variant == "sgd_fp64" ? &kSgd
                             : opt.variant == "mbgd_fp64" ? &kMbgd64
                             : opt.variant == "mbgd_fp32" ? &kMbgd32 : nullptr;
    if (!v && multi && opt.variant == "sgd_fp64") {
        fprintf(stderr, "per-sample SGD is sequential: it has no multi-GPU form (use `gcn DATASET.mat device_id`)\n");
        return 1;
    }
    if (!v) {
        fprintf(stderr, "unknown --variant '%s' (sgd_fp64, mbgd_fp64, mbgd_fp32)\n", opt.variant.c_str());
        return 1;
    }
    const char *algo = v->sgd ? "SGD algo" : "MBGD algo";
    fprintf(stdout, "[Cube CNN training with %s] ", algo);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        fprintf(stderr, "There is no device.\n");   // InitCUDA, gcn_mbgd_fp64.cu:26-48
        return 0;
    }
    printf("HIP initialized.\n");
    fprintf(stdout, "[Cube CNN training with %s] Available Device List: ", algo);
    for (int dv = 0; dv < count; dv++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dv) != hipSuccess) continue;
        printf(dv == 0 ? "Device %d -- %s(Default)  " : "Device %d -- %s  ", dv, prop.name);
    }
    printf("\n");
    if (opt.pos.size() != 2 && opt.pos.size() != 3) {
        // the single-GPU trainers ask for 3 arguments, the multi-GPU one for 4
        // (gcn_mbgd_fp64.cu:1269-1272, hyper_version_mGPU_MBGD.cu:1355-1358)
        fprintf(stderr, opt.pos.size() > 3 ? "4 input arguments required!\n" : "3 input arguments required!\n");
        return 0;
    }
    if (opt.infer && opt.model.empty()) {
        fprintf(stderr, "--infer needs --model=PATH\n");
        return 1;
    }
    if (opt.exchange != "rccl" && opt.exchange != "oneshot") {
        fprintf(stderr, "unknown --exchange '%s' (rccl, oneshot)\n", opt.exchange.c_str());
        return 1;
    }
    if (opt.batch <= 0 || opt.epochs < 0) {
        fprintf(stderr, "--batch and --epochs must be positive\n");
        return 1;
    }
    std::vector<int> devices;
    // Rehearsal hook for boxes with fewer GPUs than the run wants: CCNN_CLI_DEVICE_MAP=0,0 runs the
    // device slots 0 and 1 of `gcn DATASET.mat 2 0` both on physical device 0 (own streams, own
    // host threads, own parameter copies: the multi-device driver as it is, minus the second card;
    // RCCL refuses two ranks on one GPU, so this needs --exchange=oneshot).
    std::vector<int> device_map;
    if (const char *m = getenv("CCNN_CLI_DEVICE_MAP")) {
        for (const char *q = m; *q;) {
            char *end = nullptr;
            const long id = strtol(q, &end, 10);
            if (end == q || id < 0 || id >= count) {
                fprintf(stderr, "CCNN_CLI_DEVICE_MAP: '%s' is not a list of device ordinals below %d\n", m, count);
                return 1;
            }
            device_map.push_back((int)id);
            q = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') {
                fprintf(stderr, "CCNN_CLI_DEVICE_MAP: '%s' is not a comma-separated list\n", m);
                return 1;
            }
        }
    }
    const int slots = device_map.empty() ? count : (int)device_map.size();
    if (multi) {
        const int device_num = atoi(opt.pos[1].c_str()), master = atoi(opt.pos[2].c_str());
        if (device_num <= 0 || device_num > slots || master < 0 || master >= device_num) {
            fprintf(stderr, "Device number error: %d device(s) available, asked for %d with master %d\n", count, device_num, master);
            return 1;
        }
        devices.push_back(master);   // the master holds slice 0, the others follow in id order (:923-952)
        for (int k = 0; k < device_num; k++)
            if (k != master) devices.push_back(k);
        fprintf(stdout, "[Cube CNN training with %s] Training implemented on %d Device(s), master %d.\n", algo, device_num, master);
    } else {
        const int dv = atoi(opt.pos[1].c_str());
        if (dv < 0 || dv >= count) {
            fprintf(stderr, "%s:%d: invalid device ordinal %d\n", __FILE__, __LINE__, dv);
            return 1;
        }
        devices.push_back(dv);
        fprintf(stdout, "[Cube CNN training with %s] Training implemented on Device %d.\n", algo, dv);
    }
    if (multi && !device_map.empty()) {
        for (int &dv : devices) dv = device_map[dv];
        if (opt.exchange != "oneshot" && devices.size() > 1) {
            fprintf(stderr, "CCNN_CLI_DEVICE_MAP with several slots needs --exchange=oneshot (RCCL wants one GPU per rank)\n");
            return 1;
        }
    }

    // labels whole (x*y doubles); DataSet is only opened here and streamed to the devices later
    std::string err;
    matio::Stream lab, ds;
    if (!lab.open(opt.pos[0], "labels", err) || !ds.open(opt.pos[0], "DataSet", err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    if (ds.dims().size() != 3 || lab.numel() != ds.dims()[0] * ds.dims()[1]) {
        fprintf(stderr, "'%s' must hold DataSet (x-by-y-by-z) and labels (x-by-y)\n", opt.pos[0].c_str());
        return 1;
    }
    // the kernels index pixels and planes with 32-bit integers: refuse what does not fit before narrowing
    if (ds.dims()[0] > 0x7fffffffu || ds.dims()[1] > 0x7fffffffu || ds.dims()[2] > (size_t)CCNN_MAX_BANDS ||
        ds.dims()[0] * ds.dims()[1] > 0x7fffffffu || ds.dims()[0] < 2 || ds.dims()[1] < 2) {
        fprintf(stderr, "DataSet of %zu x %zu x %zu is outside what this build handles (2..2^31-1 pixels, at most %d bands)\n",
                ds.dims()[0], ds.dims()[1], ds.dims()[2], CCNN_MAX_BANDS);
        return 1;
    }
    const int x = (int)ds.dims()[0], y = (int)ds.dims()[1], z = (int)ds.dims()[2];
    std::vector<double> labels(lab.numel());
    if (!lab.next(labels.data(), labels.size(), err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    lab.close();

    // class count: the reference's 13 unless told otherwise or the label map needs more
    int classes = opt.classes;
    if (classes == 0) {
        double top = 0;
        for (double l : labels) top = std::max(top, l);
        classes = top > CCNN_CLASSES && top <= CCNN_MAX_CLASSES ? (int)top : CCNN_CLASSES;
        if (classes != CCNN_CLASSES) fprintf(stderr, "labels run up to %d: training a %d-class output layer\n", classes, classes);
    }
    if (classes < CCNN_MIN_CLASSES || classes > CCNN_MAX_CLASSES) {
        fprintf(stderr, "--classes must be %d..%d\n", CCNN_MIN_CLASSES, CCNN_MAX_CLASSES);
        return 1;
    }
    if (opt.preprocess) return preprocess_dump(ds, labels, x, y, z, classes, devices[0]);
    const double correct = v->f32 ? run<float>(opt, *v, ds, labels, x, y, z, classes, devices)
                                  : run<double>(opt, *v, ds, labels, x, y, z, classes, devices);
    fprintf(stdout, "Accuracy: %.3f%% \n", correct * 100);
    return 0;
}
