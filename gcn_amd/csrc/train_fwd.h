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

}  // namespace ccnn
