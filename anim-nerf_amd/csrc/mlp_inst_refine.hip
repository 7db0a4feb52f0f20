// explicit instantiations of the fused MLP kernel for the `_refine` stage (networks frozen, poses train: train.py:433-437):
// the training forward that keeps only the ReLU sign bits; see mlp_core.h (BITS_ONLY)
#include "mlp_core.h"

namespace anr {
template int launch_mlp<ANR_MLP_F32, true, false, true, false, false, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
template int launch_mlp<ANR_MLP_BF16_W8, true, false, true, false, false, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
}  // namespace anr
