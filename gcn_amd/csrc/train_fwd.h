// Forward half of the batched minibatch step, implemented next to the inference kernels it
// shares (forward.hip) and called by the training driver (mbgd.hip).
#pragma once
#include "ccnn_common.h"

namespace ccnn {

constexpr int TRAIN_CTR_INTS = 8 * 32;   // work counters of the persistent forward kernels (zeroed by them per launch)

// conv + tanh + pool of samples [first, first + B) -> pool [B][pout] and px [B][pout] (which member
// of each pool pair won, one byte), then FC1 + output layer + softmax -> a1 [B][100],
// d1 [B][100], d2 [B][13], loss_values [B] and, optionally, probs [B][13].  Two launches on st.
template <typename T, bool FLIP>
int train_forward(const T *params, const T *cubes, const int *onehot, long first, int B, int z, T *pool,
                  unsigned char *px, T *a1, T *d1, T *d2, T *probs, double *loss_values, int *ctr, hipStream_t st);

// The whole per-sample part of a step in one launch, a tile of up to eight samples per workgroup
// (tile.hip): pool, a1, d1, d2 rows (+ probs, loss_values) as train_forward leaves them, and one
// conv-gradient partial [F K + F] per workgroup in part[plan.tiles][F K + F].
struct TilePlan {
    int nwv, ms, nch;   // waves per workgroup, samples per tile, position chunks of the delta image
    size_t smem;        // dynamic LDS bytes
    int tiles;          // workgroups = rows of `part`
};
// false: the shape does not fit the LDS in any configuration (the caller uses the layer kernels).
// Depends on B, the geometry and T only, never on the device, so workspace sizes can be computed anywhere.
template <typename T>
bool tile_plan(int B, const ccnn_dims_t &d, TilePlan *out);
// The same with the wave count of full fp32 tiles given by the caller (8 | 16, else the default)
// instead of read from CCNN_TILE_WAVES; tile_waves_env() reads that variable.
template <typename T>
bool tile_plan_waves(int B, const ccnn_dims_t &d, int forced_waves, TilePlan *out);
int tile_waves_env();
template <typename T, bool FLIP>
int tile_train(const T *params, const T *cubes, const int *onehot, long first, int B, const ccnn_dims_t &d,
               const TilePlan &plan, T *pool, T *a1, T *d1, T *d2, T *probs, double *loss_values, T *part,
               hipStream_t st);

}  // namespace ccnn
