// Probe: the conv phase of tile_fwd_bwd on its own - the library's conv_pool_chain (ccnn_device.h) run
// as (sample, chain) jobs by the sixteen waves of a workgroup over eight LDS-resident cubes, one pass
// between two barriers, every CU busy.  Prints shader cycles per v_mfma_f32_16x16x4_f32 per SIMD for
// the pass and for its variants:
//   V = 0  as in the kernel (results stored to LDS: a float and a byte per pooled value)
//   V = 1  results summed into a register instead of stored
//   V = 3  no epilogue at all: the chains alone (mma_chain), accumulators summed into a register
//   V = 5  as V = 0 with the geometry (positions, pooled positions, samples, strides) in kernel arguments, as the library has it
//   V = 4  every wave runs all chains of its own sample (conv_pool_wave: position tiles in groups of three)
//   hipcc --offload-arch=gfx950 -O3 -I../../gcn_amd/csrc -I../../include conv_jobs_rate.hip -o conv_jobs_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "ccnn_device.h"
using namespace ccnn;

constexpr int RE = 43, MRE = 21, POUT = 420, PLD = 452, NSAMP = 8;

template <int V>
__global__ __launch_bounds__(1024) void k(const float *src, float *out, unsigned long long *cyc, int reps, int re_a, int mre_a, int nb_a, int pld_a, int pout_a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int cube_pad = conv_cube_pad(RE);
    float *wl = lds, *cubes = wl + WP_LEN, *pl = cubes + NSAMP * cube_pad;
    unsigned char *px = (unsigned char *)(pl + NSAMP * PLD);
    const int total = WP_LEN + NSAMP * cube_pad + NSAMP * PLD + NSAMP * POUT / 4;
    for (int i = threadIdx.x; i < total; i += blockDim.x) lds[i] = src[i & 8191];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int re = V == 5 ? re_a : RE, mre = V == 5 ? mre_a : MRE, pld = V == 5 ? pld_a : PLD, pout = V == 5 ? pout_a : POUT;
    const int nc = conv_chains(re), nb = V == 5 ? nb_a : NSAMP;
    float keep = 0;
    const float bias0 = src[lane & 15], biasx = src[16 + (lane & 3)];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; rep++) {
        asm volatile("" ::: "memory");
        if (V == 4) {
            const int s = wave & 7;
            float *ps = pl + s * PLD;
            unsigned char *pxs = px + s * POUT;
            for (int half = 0; half < (nwv == 8 ? 1 : 0); half++)
                conv_pool_wave<float>(cubes + s * cube_pad, wl, RE, bias0, biasx, lane, [&](int f, int m, float v, int bt) {
                    ps[f * MRE + m] = v;
                    pxs[f * MRE + m] = (unsigned char)(bt - 2 * m);
                });
            continue;
        }
        for (int job = wave; job < nb * nc; job += nwv) {
            const int c = job / nb, s = job - c * nb;
            if (V == 3) {
                const int col = lane & 15, kq = lane >> 4;
                const float *bp0 = wl + kq * LDWK + col, *cube = cubes + s * cube_pad;
                Mma<float>::acc_t acc;
                if (c < conv_mtiles(RE)) acc = mma_chain<float, KSTEPS0>(cube + WSTEP * (c * 16 + col) + kq, bp0);
                else acc = mma_chain<float, KSTEPSX>(cube + 4 * WSTEP * min(col, (RE - 1) >> 2) + kq, bp0 + WL_TILE);
                keep += acc[0] + acc[1] + acc[2] + acc[3];
                continue;
            }
            float *ps = pl + s * pld;
            unsigned char *pxs = px + s * pout;
            conv_pool_chain<float>(cubes + s * (V == 5 ? conv_cube_pad(re) : cube_pad), wl, re, c, bias0, biasx, lane, [&](int f, int m, float v, int bt) {
                if (V == 1) {
                    keep += v + (float)bt;
                } else {
                    ps[f * mre + m] = v;
                    pxs[f * mre + m] = (unsigned char)(bt - 2 * m);
                }
            });
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep + pl[threadIdx.x] + (float)px[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char *what, int threads, int reps, const float *src, float *out, unsigned long long *cyc)
{
    const int blocks = 256;
    const size_t smem = (size_t)(WP_LEN + NSAMP * conv_cube_pad(RE) + NSAMP * PLD + NSAMP * POUT / 4) * 4;
    hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), smem, 0, src, out, cyc, reps, RE, MRE, NSAMP, PLD, POUT);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double mfma_per_simd = reps * NSAMP * (3.0 * KSTEPS0 + KSTEPSX) / 4;
    printf("%-50s %2d waves, %d pass(es): %6.0f cycles = %.1f per MFMA per SIMD\n", what, threads / 64, reps,
           (double)h[blocks / 2], (double)h[blocks / 2] / mfma_per_simd);
}

int main()
{
    float *src, *out;
    unsigned long long *cyc;
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    hipMalloc(&src, 8192 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int threads : {512, 1024}) {
        run<0>("conv_pool_chain jobs, as in the kernel", threads, 1, src, out, cyc);
        run<1>("results kept in a register", threads, 1, src, out, cyc);
        run<0>("as in the kernel, steady state", threads, 8, src, out, cyc);
        run<1>("results kept in a register, steady state", threads, 8, src, out, cyc);
        run<5>("geometry in kernel arguments", threads, 1, src, out, cyc);
        run<3>("chains only", threads, 1, src, out, cyc);
        run<3>("chains only, steady state", threads, 8, src, out, cyc);
        if (threads == 512) {
            run<4>("a wave per sample (groups of three tiles)", threads, 1, src, out, cyc);
            run<4>("a wave per sample, steady state", threads, 8, src, out, cyc);
        }
    }
    return 0;
}
