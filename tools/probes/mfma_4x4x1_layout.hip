// Probe: which lane supplies which element of v_mfma_f32_4x4x1_16b_f32 and where the results land.
// Expectation checked below: A[i][block] from lane 4*block + i, B[j][block] from lane 4*block + j,
// D[i][j] of block b in VGPR i of lane 4*b + j.   hipcc --offload-arch=gfx950 -O2 mfma_4x4x1_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[i * 64 + l] = acc[i];
}
int main()
{
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; l++) {
        ha[l] = 1.0f + l;          // A: lane l
        hb[l] = 100.0f + 3 * l;    // B: lane l
    }
    float *a, *b, *d;
    hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&d, sizeof hd);
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; blk++)
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                const float want = ha[4 * blk + i] * hb[4 * blk + j];
                const float got = hd[i * 64 + 4 * blk + j];
                if (want != got) bad++;
            }
    printf("4x4x1 layout as expected: %s (%d mismatches); d[0..3] of lane 5: %g %g %g %g\n", bad ? "NO" : "yes", bad,
           hd[5], hd[64 + 5], hd[128 + 5], hd[192 + 5]);
    return bad != 0;
}
