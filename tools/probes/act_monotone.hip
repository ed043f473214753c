// Probe: is the fp32 activation of the library (ccnn_common.h: tanh through v_exp_f32 / v_rcp_f32)
// monotone non-decreasing over ALL finite floats?  If it is, max(act(a), act(b)) == act(max(a, b)) bit
// for bit, i.e. pooling may pick the larger conv output first and activate once.
// Every pair (x, next float above x) is checked; prints the number of pairs with act(x) > act(next).
//   hipcc --offload-arch=gfx950 -O3 -I../../gcn_amd/csrc -I../../include act_monotone.hip -o act_monotone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "ccnn_common.h"

// Floats in increasing order <-> integers: key k in [0, 2^32) maps to the float whose ordered
// position it is (negative floats reversed).
__device__ inline float ordered(unsigned k)
{
    const unsigned bits = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(bits);
}

__global__ void check(unsigned long long *bad, unsigned *first_bad)
{
    const unsigned long long n = 1ull << 32;
    unsigned long long local = 0;
    for (unsigned long long k = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; k + 1 < n;
         k += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = ordered((unsigned)k), y = ordered((unsigned)(k + 1));
        if (x != x || y != y || fabsf(x) == INFINITY || fabsf(y) == INFINITY) continue;   // NaNs, infinities
        if (ccnn::act(x) > ccnn::act(y)) {
            local++;
            atomicMin(first_bad, (unsigned)k);
        }
    }
    if (local) atomicAdd(bad, local);
}

int main()
{
    unsigned long long *bad, h = 0;
    unsigned *fb, hf = 0xffffffffu;
    hipMalloc(&bad, 8); hipMalloc(&fb, 4);
    hipMemcpy(bad, &h, 8, hipMemcpyHostToDevice);
    hipMemcpy(fb, &hf, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, bad, fb);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost);
    printf("pairs of neighbouring finite floats with act(x) > act(next): %llu", h);
    if (h) {
        const unsigned bits = (hf & 0x80000000u) ? (hf & 0x7fffffffu) : ~hf;
        float x; memcpy(&x, &bits, 4);
        printf(" (first at x = %.9g)", x);
    }
    printf("\n");
    return 0;
}
