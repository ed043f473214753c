// Probe: shader cycles per v_mfma_f32_16x16x4_f32 on one SIMD, (a) operands in registers, one or two
// accumulators, (b) both operands read from LDS with 4-byte reads before every instruction, as the
// conv chains do; 1, 2 and 4 waves per SIMD (one workgroup of 256 / 512 / 1024 threads per CU, every CU busy).
//   hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096;   // MFMAs per wave

template <int MODE>
__global__ void k(const float *src, float *out, unsigned long long *cyc)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float a = src[lane], b = src[64 + lane];
    const float *ap = lds + 18 * (lane & 15) + (lane >> 4), *bp = lds + 4096 + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {          // registers, one accumulator
#pragma unroll 16
        for (int i = 0; i < N; i++) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    } else if (MODE == 1) {   // registers, two accumulators
#pragma unroll 8
        for (int i = 0; i < N; i += 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        }
    } else {                  // operands from LDS, 43-step chains like the conv
        for (int i = 0; i < N; i += 44) {
            float av[2][4], bv[2][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                av[0][u] = ap[4 * u];
                bv[0][u] = bp[64 * u];
            }
#pragma unroll
            for (int bi = 0; bi < 11; bi++) {
                const int set = bi & 1;
                if (bi + 1 < 11) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        av[set ^ 1][u] = ap[4 * (4 * (bi + 1) + u)];
                        bv[set ^ 1][u] = bp[64 * (4 * (bi + 1) + u)];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (u & 1) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][u], bv[set][u], c1, 0, 0, 0);
                    else c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][u], bv[set][u], c0, 0, 0, 0);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *what, int threads, const float *src, float *out, unsigned long long *cyc)
{
    const int blocks = 256, waves = blocks * threads / 64;
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const int per_simd = threads / 256;
    const int n = MODE == 2 ? (N + 43) / 44 * 44 : N;
    printf("%-44s %d wave(s)/SIMD: median %.1f cycles per MFMA per wave = %.1f per SIMD\n", what, per_simd,
           (double)h[waves / 2] / n, (double)h[waves / 2] / n / per_simd);
}

int main()
{
    float *src, *out;
    unsigned long long *cyc;
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    hipMalloc(&src, 8192 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int threads : {256, 512, 1024}) {
        run<0>("registers, one accumulator", threads, src, out, cyc);
        run<1>("registers, two accumulators", threads, src, out, cyc);
        run<2>("both operands from LDS (4-byte reads)", threads, src, out, cyc);
    }
    return 0;
}
