// Probe: what a launch of sample_fwd_bwd's shape costs when the kernel does (almost) nothing -
// 256 workgroups x 512 threads, 60 KB of dynamic LDS, a register allocation of 240 per lane -
// by the dispatch-attached events bench.py uses, alone and behind a second small kernel as in a
// training step.  The difference between a kernel's event time and its in-kernel stamps is this.
//   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

template <int NREG>
__global__ __launch_bounds__(512, 2) void shaped(float *out, int never)
{
    extern __shared__ float lds[];
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; i++) r[i] = out[(threadIdx.x + i) & 1023];   // keeps NREG registers live
    lds[threadIdx.x] = r[0];
    __syncthreads();
    float s = lds[(threadIdx.x + 1) & 511];
    if (never) {
#pragma unroll
        for (int i = 0; i < NREG; i++) s += r[i] * r[(i + 7) % NREG];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void small(float *out) { out[blockIdx.x * 512 + threadIdx.x] += 1.0f; }

int main()
{
    float *buf;
    CK(hipMalloc(&buf, sizeof(float) * 512 * 512));
    CK(hipMemset(buf, 0, sizeof(float) * 512 * 512));
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&f0));
    CK(hipEventCreate(&f1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(shaped<200>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(shaped<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440));
    for (int big = 0; big < 2; big++) {
        double a = 0, b = 0;
        const int n = 200;
        for (int i = 0; i < n + 20; i++) {
            if (big) hipExtLaunchKernelGGL(shaped<200>, dim3(256), dim3(512), 61440, 0, e0, e1, 0, buf, 0);
            else hipExtLaunchKernelGGL(shaped<8>, dim3(256), dim3(512), 61440, 0, e0, e1, 0, buf, 0);
            hipExtLaunchKernelGGL(small, dim3(105), dim3(512), 0, 0, f0, f1, 0, buf);
            CK(hipEventSynchronize(f1));
            float ms0, ms1;
            CK(hipEventElapsedTime(&ms0, e0, e1));
            CK(hipEventElapsedTime(&ms1, f0, f1));
            if (i >= 20) {
                a += ms0;
                b += ms1;
            }
        }
        printf("%s registers: shaped kernel %.2f us, the small kernel behind it %.2f us (dispatch events)\n", big ? "~200" : "few", a / n * 1e3,
               b / n * 1e3);
    }
    // back to back, wall clock per pair
    hipEvent_t w0, w1;
    CK(hipEventCreate(&w0));
    CK(hipEventCreate(&w1));
    CK(hipEventRecord(w0));
    for (int i = 0; i < 2000; i++) {
        hipLaunchKernelGGL(shaped<200>, dim3(256), dim3(512), 61440, 0, buf, 0);
        hipLaunchKernelGGL(small, dim3(105), dim3(512), 0, 0, buf);
    }
    CK(hipEventRecord(w1));
    CK(hipEventSynchronize(w1));
    float ms;
    CK(hipEventElapsedTime(&ms, w0, w1));
    printf("2000 pairs back to back: %.2f us per pair\n", ms / 2000 * 1e3);
    // what the shape contributes: workgroups, threads, LDS of the first kernel; workgroups of the second
    struct Shape { int g1, t1, lds, g2, t2; } shapes[] = {
        {256, 512, 61440, 105, 512}, {256, 512, 0, 105, 512},  {256, 256, 61440, 105, 512}, {128, 512, 61440, 105, 512},
        {256, 512, 61440, 105, 256}, {256, 512, 61440, 32, 512}, {256, 512, 61440, 256, 512}, {256, 64, 0, 32, 64}, {1, 64, 0, 1, 64}};
    for (const Shape &sh : shapes) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(w0));
        for (int i = 0; i < 2000; i++) {
            hipLaunchKernelGGL(shaped<8>, dim3(sh.g1), dim3(sh.t1), sh.lds, 0, buf, 0);
            hipLaunchKernelGGL(small, dim3(sh.g2), dim3(sh.t2), 0, 0, buf);
        }
        CK(hipEventRecord(w1));
        CK(hipEventSynchronize(w1));
        CK(hipEventElapsedTime(&ms, w0, w1));
        printf("first %3d x %3d threads, %5d B of LDS; second %3d x %3d: %.2f us per pair\n", sh.g1, sh.t1, sh.lds, sh.g2, sh.t2, ms / 2000 * 1e3);
    }
    // the same pair without the second launch's wait for the first (any-order launch: what a kernel
    // boundary costs beyond the dispatch itself), and 100 pairs per HIP graph launch
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(w0));
    for (int i = 0; i < 2000; i++) {
        hipLaunchKernelGGL(shaped<8>, dim3(256), dim3(512), 61440, 0, buf, 0);
        hipExtLaunchKernelGGL(small, dim3(105), dim3(512), 0, 0, nullptr, nullptr, hipExtAnyOrderLaunch, buf);
    }
    CK(hipEventRecord(w1));
    CK(hipEventSynchronize(w1));
    CK(hipEventElapsedTime(&ms, w0, w1));
    printf("second launch any-order: %.2f us per pair\n", ms / 2000 * 1e3);
    for (int pairs : {1, 4, 100}) {
        hipStream_t st;
        CK(hipStreamCreate(&st));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < pairs; i++) {
            hipLaunchKernelGGL(shaped<8>, dim3(256), dim3(512), 61440, st, buf, 0);
            hipLaunchKernelGGL(small, dim3(105), dim3(512), 0, st, buf);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(w0, st));
        const int nl = 2000 / pairs;
        for (int i = 0; i < nl; i++) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(w1, st));
        CK(hipEventSynchronize(w1));
        CK(hipEventElapsedTime(&ms, w0, w1));
        printf("%d pair(s) per graph launch: %.2f us per pair\n", pairs, ms / (nl * pairs) * 1e3);
    }
    return 0;
}
