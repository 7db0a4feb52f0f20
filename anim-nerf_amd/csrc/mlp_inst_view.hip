// explicit instantiations of the fused MLP kernel with the view-dependent colour head inside (models/nerf.py:141-153); see mlp_core.h
#include "mlp_core.h"

namespace anr {
template int launch_mlp<ANR_MLP_F32, true, false, false, false, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
template int launch_mlp<ANR_MLP_BF16_W8, true, false, false, false, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
}  // namespace anr
