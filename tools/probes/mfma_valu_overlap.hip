// Probe: does vector-ALU work of one wave run under the matrix instructions of another wave of the
// same SIMD?  Workgroups of eight waves, two per SIMD: waves 0..3 run a chain of
// v_mfma_f32_16x16x4_f32 (operands in registers), waves 4..7 a loop of activation-like vector work
// (v_exp_f32 / v_rcp_f32 / v_fma_f32).  Each kind is timed alone and both together.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096;

// what: bit 0 = the MFMA waves work, bit 1 = the VALU waves work; kind: 0 transcendental mix, 1 plain FMAs
__global__ __launch_bounds__(512) void k(const float *src, float *out, unsigned long long *cyc, int what, int kind, int prio, int swap, int mtype)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool mf = swap ? wave >= 4 : wave < 4;   // swap: the vector waves are the older ones
    float r = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mf && (what & 1) && mtype == 1) {          // v_mfma_f64_16x16x4_f64
        d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        const double a = src[lane], b = src[64 + lane];
#pragma unroll 8
        for (int i = 0; i < N; i += 2) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        }
        r = (float)(c0[0] + c1[1]);
    } else if (mf && (what & 1) && mtype == 2) {   // v_mfma_f32_16x16x16_f16
        f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        const _Float16 x = (_Float16)src[lane], y = (_Float16)src[64 + lane];
        const h4 a = {x, y, x, y}, b = {y, x, y, x};
#pragma unroll 8
        for (int i = 0; i < N; i += 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(b, a, c1, 0, 0, 0);
        }
        r = c0[0] + c1[1];
    } else if (mf && (what & 1)) {
        f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        const float a = src[lane], b = src[64 + lane];
#pragma unroll 8
        for (int i = 0; i < N; i += 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        }
        r = c0[0] + c1[1];
    } else if (!mf && (what & 2)) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        float v0 = src[lane], v1 = src[64 + lane], v2 = src[128 + lane], v3 = src[192 + lane];
        if (kind == 0) {
#pragma unroll 4
            for (int i = 0; i < N; i++) {   // four independent activations per round
                v0 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v0 * 2.885f))) - 1.0f;
                v1 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v1 * 2.885f))) - 1.0f;
                v2 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v2 * 2.885f))) - 1.0f;
                v3 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v3 * 2.885f))) - 1.0f;
            }
        } else {
#pragma unroll 4
            for (int i = 0; i < 6 * N; i++) {
                v0 = __builtin_fmaf(v0, 0.999f, 0.001f);
                v1 = __builtin_fmaf(v1, 0.998f, 0.002f);
                v2 = __builtin_fmaf(v2, 0.997f, 0.003f);
                v3 = __builtin_fmaf(v3, 0.996f, 0.004f);
            }
        }
        r = v0 + v1 + v2 + v3;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// One wave per SIMD: every MFMA followed by KV independent vector FMAs of the same wave.
template <int KV>
__global__ __launch_bounds__(256) void interleaved(const float *src, float *out, unsigned long long *cyc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const float a = src[lane], b = src[64 + lane];
    float v[12];
#pragma unroll
    for (int j = 0; j < 12; j++) v[j] = src[128 + lane + j];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int i = 0; i < N; i += 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < KV; j++) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < KV; j++) v[j] = __builtin_fmaf(v[j], 0.998f, 0.002f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = c0[0] + c1[1];
#pragma unroll
    for (int j = 0; j < 12; j++) r += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KV>
void run_interleaved(const float *src, float *out, unsigned long long *cyc)
{
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(interleaved<KV>, dim3(256), dim3(256), 0, 0, src, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(256 * 4);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    printf("one wave, %2d vector FMAs after every f32 MFMA: %.1f cycles per MFMA\n", KV, (double)c[c.size() / 2] / N);
}

int main()
{
    float *src, *out;
    unsigned long long *cyc;
    std::vector<float> h(256);
    for (int i = 0; i < 256; i++) h[i] = (float)(i % 17) / 17.0f - 0.5f;
    hipMalloc(&src, 256 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemcpy(src, h.data(), 256 * 4, hipMemcpyHostToDevice);
    static const char *mname[3] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f64_16x16x4_f64", "v_mfma_f32_16x16x16_f16"};
    for (int mtype = 0; mtype < 3; mtype++)
    for (int cfg = 0; cfg < (mtype ? 1 : 4); cfg++)
    for (int kind = 0; kind < 2; kind++)
        for (int what = (cfg ? 3 : 1); what <= 3; what++) {
            const int prio = cfg & 1, swap = cfg >> 1;
            if (what == 1 && kind == 0) printf("---- %s\n", mname[mtype]);
            if (what == 3) printf("[vector waves %s, s_setprio %d] ", swap ? "older" : "younger", prio ? 3 : 0);
            for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, src, out, cyc, what, kind, prio, swap, mtype);
            hipDeviceSynchronize();
            std::vector<unsigned long long> c(256 * 8);
            hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> m, v;
            for (int b = 0; b < 256; b++)
                for (int w = 0; w < 8; w++) ((swap ? w >= 4 : w < 4) ? m : v).push_back((double)c[b * 8 + w]);
            std::sort(m.begin(), m.end());
            std::sort(v.begin(), v.end());
            printf("%s, %s: MFMA wave %8.0f cycles (%.1f per MFMA), vector wave %8.0f cycles\n",
                   kind ? "plain FMAs      " : "exp / rcp / rcp ", what == 1 ? "MFMA waves alone  " : what == 2 ? "vector waves alone" : "both              ",
                   m[m.size() / 2], m[m.size() / 2] / N, v[v.size() / 2]);
        }
    run_interleaved<0>(src, out, cyc);
    run_interleaved<2>(src, out, cyc);
    run_interleaved<4>(src, out, cyc);
    run_interleaved<6>(src, out, cyc);
    run_interleaved<7>(src, out, cyc);
    run_interleaved<8>(src, out, cyc);
    run_interleaved<12>(src, out, cyc);
    return 0;
}
