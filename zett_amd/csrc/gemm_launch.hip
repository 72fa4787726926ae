// gemm_launch.hip — variant dispatch of the GEMM launchers (gemm_launch.hip.h).
#include "gemm_launch.hip.h"

namespace zett {

hipError_t launch_gemm_variant(int variant, const GemmArgs<f16_t>& g, hipStream_t stream) {
    if (variant == 7 || variant == 8) return launch_gemm_4d(g, stream, variant == 8);
    return launch_gemm_x(variant, g, stream);
}
hipError_t launch_gemm_variant(int variant, const GemmArgs<bf16_t>& g, hipStream_t stream) {
    if (variant == 7 || variant == 8) return launch_gemm_4d(g, stream, variant == 8);
    return launch_gemm_x(variant, g, stream);
}
hipError_t launch_gemm_variant(int variant, const GemmArgs<float>& g, hipStream_t stream) {
    if (variant == 7 || variant == 8) return hipErrorInvalidValue;      // fp32 has no direct-to-LDS tile
    return launch_gemm_x(variant, g, stream);
}

}  // namespace zett
