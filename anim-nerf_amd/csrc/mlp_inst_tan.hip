// explicit instantiations of the fused MLP kernel in tangent mode (forward-mode normals); see mlp_core.h
#include "mlp_core.h"

namespace anr {
template int launch_mlp<ANR_MLP_F32, true, true, true, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
template int launch_mlp<ANR_MLP_BF16_W8, true, true, true, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
}  // namespace anr
