// Cube gather: out[i][b][j] = scene[idx9[i][j] + b*x*y]
// (preprocessing, gcn_mbgd_fp64.cu:599-615; processing, preprocess048.cu:50-65).
//
// The reference gives one thread a whole sample, so neither its reads nor its
// writes coalesce.  Here a wave owns one sample and walks its 9z output values in
// order: the stores are one contiguous run per sample, and the nine neighbour
// reads of a band are three short runs that neighbouring samples share in L2.
#include "ccnn_common.h"
#include "profile.h"

namespace ccnn {

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ scene,
                                                     const int *__restrict__ idx9, long n, long plane,
                                                     int z, T *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int cube_len = NB * z;
    for (long i = wave; i < n; i += nwaves) {
        // lane j < 9 holds neighbour j's pixel index; broadcast by shuffle
        const int mine = lane < NB ? idx9[i * NB + lane] : 0;
        T *dst = out + i * (long)cube_len;
        for (int e = lane; e < cube_len + 63 - (cube_len + 63) % 64; e += 64) {
            const int b = e / NB;
            const int j = e - b * NB;
            const int pix = __shfl(mine, j);
            if (e < cube_len) dst[e] = scene[pix + b * plane];
        }
    }
}

template <typename T>
int gather_impl(const T *scene, const int *idx9, long n, int x, int y, int z, T *out, hipStream_t st)
{
    if (n < 0 || x <= 0 || y <= 0 || z <= 0) return CCNN_E_INVALID;
    if (n == 0) return 0;                                  // empty sample list: nothing to read
    if (!scene || !idx9 || !out) return CCNN_E_INVALID;
    if ((long)x * y > 0x7fffffffL) return CCNN_E_RANGE;   // idx9 is int32
    long blocks = (n + 3) / 4;          // 4 waves per block, one sample per wave
    if (blocks > 8192) blocks = 8192;
    {
        ProfileScope prof_("gather_kernel", st);
        gather_kernel<T><<<(int)blocks, 256, 0, st>>>(scene, idx9, n, (long)x * y, z, out);
    }
    CCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace ccnn

extern "C" {

int ccnn_gather_f32(const float *scene, const int *idx9, long n, int x, int y, int z, float *out,
                    void *stream)
{
    return ccnn::gather_impl<float>(scene, idx9, n, x, y, z, out, (hipStream_t)stream);
}

int ccnn_gather_f64(const double *scene, const int *idx9, long n, int x, int y, int z, double *out,
                    void *stream)
{
    return ccnn::gather_impl<double>(scene, idx9, n, x, y, z, out, (hipStream_t)stream);
}

int ccnn_get_dims(int z, ccnn_dims_t *dims) { return ccnn::make_dims(z, dims); }

const char *ccnn_error_string(int code)
{
    switch (code) {
        case 0: return "success";
        case CCNN_E_INVALID: return "ccnn: invalid argument (null pointer, bad size, or band count out of range)";
        case CCNN_E_WORKSPACE: return "ccnn: workspace too small";
        case CCNN_E_RANGE: return "ccnn: size exceeds 32-bit index range";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ccnn: unknown error";
    }
}

}  // extern "C"
