// explicit instantiations of the fused MLP kernel for a pre-embedded input (models/mlp.py:226-297); see mlp_core.h
#include "mlp_core.h"

namespace anr {
#define ANR_PRE(M, S) template int launch_mlp<M, true, S, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
ANR_PRE(ANR_MLP_F32, false) ANR_PRE(ANR_MLP_F32, true) ANR_PRE(ANR_MLP_BF16_W8, false) ANR_PRE(ANR_MLP_BF16_W8, true)
#undef ANR_PRE
// ... keeping the activations (the 256-wide feature feeds a view-dependent colour head outside the kernel)
template int launch_mlp<ANR_MLP_F32, true, false, true, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
template int launch_mlp<ANR_MLP_BF16_W8, true, false, true, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
}  // namespace anr
