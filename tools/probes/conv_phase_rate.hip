// Probe: what a conv phase adds to a bare MFMA chain.  Sixteen (or eight) waves per CU run, per
// "sample", the chain shapes of the conv kernels from LDS - three position tiles of 43 k-steps that
// share their filter fragments, then a 57-step chain on the packed tile - and the features below are
// switched on one at a time.  Prints shader cycles per v_mfma_f32_16x16x4_f32 per SIMD.
//   bit 0: every wave reads its own cube (else all read cube 0)
//   bit 1: the packed chain reads A rows 72 floats apart (4-way bank conflict) instead of 18
//   bit 2: tanh + pool epilogue after every chain, results written to LDS (4-byte + 1-byte stores)
//   bit 3: no packed chain at all (four position tiles instead)
//   hipcc --offload-arch=gfx950 -O3 conv_phase_rate.hip -o conv_phase_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int CUBE = 944, WL = 172 * 16, WLX = 228 * 16;

__device__ __forceinline__ float act(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * -2.885390081777927f);
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + e) - 1.0f;
}

template <int KS, int NT>
__device__ __forceinline__ void chain(const float *const (&ap)[NT], const float *bp, f4 (&acc)[NT])
{
    constexpr int NB4 = (KS + 3) / 4;
    float a[2][4][NT], b[2][4];
    auto load = [&](int bi, int set) {
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (4 * bi + u < KS) {
                b[set][u] = bp[4 * (4 * bi + u) * 16];
#pragma unroll
                for (int t = 0; t < NT; t++) a[set][u][t] = ap[t][4 * (4 * bi + u)];
            }
    };
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = f4{0, 0, 0, 0};
    load(0, 0);
#pragma unroll
    for (int bi = 0; bi < NB4; bi++) {
        const int set = bi & 1;
        if (bi + 1 < NB4) load(bi + 1, set ^ 1);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (4 * bi + u < KS) {
#pragma unroll
                for (int t = 0; t < NT; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[set][u][t], b[set][u], acc[t], 0, 0, 0);
            }
    }
}

template <int FEAT>
__global__ __launch_bounds__(1024) void k(const float *src, float *out, unsigned long long *cyc, int nwaves, int SAMPLES)
{
    extern __shared__ float lds[];
    float *wl = lds, *wlx = lds + WL, *cubes = wlx + WLX, *pl = cubes + 16 * CUBE;
    unsigned char *px = (unsigned char *)(pl + 16 * 352);
    const int total = WL + WLX + 16 * CUBE + 16 * 352 + 16 * 352 / 4;
    for (int i = threadIdx.x; i < total; i += blockDim.x) lds[i] = src[i & 8191];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
    const float *cube = cubes + ((FEAT & 1) ? wave * CUBE : 0);
    float *plw = pl + wave * 352;
    unsigned char *pxw = px + wave * 352;
    float keep = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < SAMPLES; s++) {
        asm volatile("" ::: "memory");   // the operands are read again for every sample
        {
            const float *ap[3];
#pragma unroll
            for (int t = 0; t < 3; t++) ap[t] = cube + 18 * (t * 16 + col) + kq;
            f4 acc[3];
            chain<43, 3>(ap, wl + kq * 16 + col, acc);
#pragma unroll
            for (int t = 0; t < 3; t++) {
                if (FEAT & 4) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const int pos = 16 * t + 4 * kq + r;
                        if (pos < 43) {
                            float best = act(acc[t][r] + 0.1f);
                            int bt = pos;
                            const float vn = act(acc[t][r + 1] + 0.1f);
                            if (pos + 1 < 43 && best < vn) { best = vn; bt = pos + 1; }
                            plw[col * 22 + (pos >> 1)] = best;
                            pxw[col * 22 + (pos >> 1)] = (unsigned char)bt;
                        }
                    }
                } else keep += acc[t][0] + acc[t][3];
            }
        }
        asm volatile("" ::: "memory");
        {
            const float *ap[1];
            const int stride = (FEAT & 8) ? 18 : (FEAT & 2) ? 72 : 18;
            ap[0] = cube + stride * min(col, 10) + kq;
            f4 acc[1];
            if (FEAT & 8) chain<43, 1>(ap, wl + kq * 16 + col, acc);
            else chain<57, 1>(ap, wlx + kq * 16 + col, acc);
            if (FEAT & 4) {
                const int g = col >> 2, fl = col & 3;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int pos = 4 * (4 * kq + r) + g;
                    const float v = act(acc[0][r] + 0.2f);
                    const float vn = __shfl_xor(v, 4);
                    if (!(g & 1) && pos < 43) {
                        float best = v;
                        int bt = pos;
                        if (pos + 1 < 43 && best < vn) { best = vn; bt = pos + 1; }
                        plw[(16 + fl) * 22 + (pos >> 1) - 100] = best;
                        pxw[(16 + fl) * 22 + (pos >> 1) - 100] = (unsigned char)bt;
                    }
                }
            } else keep += acc[0][1];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep + plw[lane] + (float)pxw[lane];
    if (lane == 0) cyc[blockIdx.x * nwaves + wave] = t1 - t0;
}

template <int FEAT>
void run(const char *what, int threads, const float *src, float *out, unsigned long long *cyc, int SAMPLES = 24)
{
    const int blocks = 256, nw = threads / 64, waves = blocks * nw;
    const size_t smem = (size_t)(WL + WLX + 16 * CUBE + 16 * 352 + 16 * 352 / 4) * 4;
    hipFuncSetAttribute((const void *)k<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k<FEAT>, dim3(blocks), dim3(threads), smem, 0, src, out, cyc, nw, SAMPLES);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    // a SIMD's share: the time of the slowest wave of a workgroup (all waves start together)
    std::vector<double> per_block(blocks);
    for (int b = 0; b < blocks; b++) per_block[b] = (double)*std::max_element(h.begin() + b * nw, h.begin() + (b + 1) * nw);
    std::sort(per_block.begin(), per_block.end());
    const int mfma_per_wave = SAMPLES * (3 * 43 + ((FEAT & 8) ? 43 : 57));
    const double per_simd = per_block[blocks / 2] / ((double)mfma_per_wave * nw / 4);
    printf("%-58s %2d waves, %2d sample(s) per wave: %.1f cycles per MFMA per SIMD\n", what, nw, SAMPLES, per_simd);
}

int main()
{
    float *src, *out;
    unsigned long long *cyc;
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    hipMalloc(&src, 8192 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int threads : {512, 1024}) {
        run<8>("position tiles only, one cube", threads, src, out, cyc);
        run<9>("position tiles only, a cube per wave", threads, src, out, cyc);
        run<1>("+ packed chain, rows 18 apart", threads, src, out, cyc);
        run<3>("+ packed chain, rows 72 apart (4-way conflict)", threads, src, out, cyc);
        run<13>("position tiles only + epilogue", threads, src, out, cyc);
        run<5>("packed rows 18 apart + epilogue", threads, src, out, cyc);
        run<7>("packed rows 72 apart + epilogue (the conv phase)", threads, src, out, cyc);
        // one pass from a standing start, as between two barriers of a training step
        run<7>("the conv phase", threads, src, out, cyc, 4);
        run<7>("the conv phase", threads, src, out, cyc, 2);
        run<7>("the conv phase", threads, src, out, cyc, 1);
    }
    return 0;
}
