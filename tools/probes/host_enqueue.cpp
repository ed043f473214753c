// Host cost of one training step through the C ABI, without Python: how long ccnn_mbgd_step_f32
// takes to RETURN (two kernel launches enqueued), and whether several host threads - one per
// device slot, each with its own stream, parameters and workspace, as the `gcn` multi-GPU form
// drives its devices - enqueue side by side or serialise inside the runtime.
//
//   hipcc -O2 -std=c++17 -o host_enqueue host_enqueue.cpp -I../../include -L../../gcn_amd/lib -lccnn_hip \
//         -Wl,-rpath,'$ORIGIN/../../gcn_amd/lib'
//   ./host_enqueue [bands] [batch] [steps]
//
// All slots share device 0 (the probe is about the host side; the GPU is given time to drain
// between the timed enqueue bursts so that a full queue never blocks a launch).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "ccnn.h"

#define CK(x)                                                                           \
    do {                                                                                \
        int e_ = (int)(x);                                                              \
        if (e_) {                                                                       \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, ccnn_error_string(e_));  \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

struct Slot {
    hipStream_t st;
    float *params, *cubes;
    int *onehot;
    double *loss;
    void *ws;
    size_t ws_bytes;
};

int main(int argc, char **argv)
{
    const int z = argc > 1 ? atoi(argv[1]) : 103, B = argc > 2 ? atoi(argv[2]) : 128, steps = argc > 3 ? atoi(argv[3]) : 400;
    ccnn_dims_t d;
    CK(ccnn_get_dims(z, &d));
    const int nb = 8;
    std::vector<float> params(d.nparams), cubes((size_t)nb * B * d.cube);
    std::vector<int> onehot((size_t)nb * B * 13, 0);
    srand(1);
    CK(ccnn_host_init_params_f32(params.data(), z));
    for (auto &v : cubes) v = rand() / (float)RAND_MAX;
    for (int i = 0; i < nb * B; i++) onehot[(size_t)i * 13 + rand() % 13] = 1;
    const int max_threads = 8;
    std::vector<Slot> slot(max_threads);
    CK(hipSetDevice(0));
    for (auto &s : slot) {
        CK(hipStreamCreate(&s.st));
        CK(hipMalloc((void **)&s.params, params.size() * 4));
        CK(hipMalloc((void **)&s.cubes, cubes.size() * 4));
        CK(hipMalloc((void **)&s.onehot, onehot.size() * 4));
        CK(hipMalloc((void **)&s.loss, sizeof(double) * B));
        s.ws_bytes = ccnn_mbgd_workspace_bytes(B, z, 4);
        CK(hipMalloc(&s.ws, s.ws_bytes));
        CK(hipMemcpy(s.params, params.data(), params.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(s.cubes, cubes.data(), cubes.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(s.onehot, onehot.data(), onehot.size() * 4, hipMemcpyHostToDevice));
    }
    auto burst = [&](Slot &s, int n) {   // lr = 0: the weights stay put however long the probe runs
        for (int i = 0; i < n; i++)
            CK(ccnn_mbgd_step_f32(s.params, s.cubes, s.onehot, (long)(i % nb) * B, B, z, 0.0f, s.loss, s.ws, s.ws_bytes, s.st));
    };
    for (auto &s : slot) burst(s, 20);
    CK(hipDeviceSynchronize());
    printf("bands %d, batch %d, %d steps per thread, path %d\n", z, B, steps, ccnn_mbgd_path(B, z, 4));
    for (int nt : {1, 2, 4, 8}) {
        std::vector<double> us(nt);
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < nt; k++)
            th.emplace_back([&, k] {
                CK(hipSetDevice(0));
                const auto a = std::chrono::steady_clock::now();
                burst(slot[k], steps);
                us[k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / steps;
            });
        for (auto &t : th) t.join();
        const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        CK(hipDeviceSynchronize());
        const double drained = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        double worst = 0;
        for (double v : us) worst = v > worst ? v : worst;
        printf("%d host thread(s): %.2f us per step to enqueue (slowest thread), all threads done after %.0f us, GPU drained after %.0f us "
               "(%.2f us per step and slot)\n", nt, worst, wall, drained, drained / steps / nt);
    }
    return 0;
}
