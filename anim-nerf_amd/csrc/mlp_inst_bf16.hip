// explicit instantiations of the fused MLP kernel (see mlp_core.h); split over files so they compile in parallel
#include "mlp_core.h"

namespace anr {
template int launch_mlp<ANR_MLP_BF16_W8, true, false, false>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
template int launch_mlp<ANR_MLP_BF16_W8, false, false, false>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
template int launch_mlp<ANR_MLP_BF16_W8, true, true, false>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
}  // namespace anr
