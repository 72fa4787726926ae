// gemm_launch.hip.h — the GEMM launchers as the host translation unit sees them.
//
// The tile kernels are instantiated in their own translation units (gemm_x_<type>.hip: 128x128, gemm8r, gemm384;
// gemm4d_<type>.hip: the four-wave direct-to-LDS tile with its epilogue instantiations) so that hipcc compiles them in parallel (zett_amd/build.py) and an edit to one kernel
// does not rebuild the others.  zett_hip.hip sees only these non-template entry points.
//
//   variant: 1 = 128x128, 2 = 256x256 register-staged eight-wave (gemm8r), 3 = 384x256 LDS-DMA (gemm384),
//            7 = 256x256 four-wave direct-to-LDS (gemm4d), 8 = 7 with the generic epilogue drain
#pragma once

#include <hip/hip_runtime.h>

#include "gemm.hip.h"

namespace zett {

hipError_t launch_gemm_variant(int variant, const GemmArgs<f16_t>& g, hipStream_t stream);
hipError_t launch_gemm_variant(int variant, const GemmArgs<bf16_t>& g, hipStream_t stream);
hipError_t launch_gemm_variant(int variant, const GemmArgs<float>& g, hipStream_t stream);

// per-type pieces (defined in the translation units named above)
hipError_t launch_gemm_x(int variant, const GemmArgs<f16_t>& g, hipStream_t stream);
hipError_t launch_gemm_x(int variant, const GemmArgs<bf16_t>& g, hipStream_t stream);
hipError_t launch_gemm_x(int variant, const GemmArgs<float>& g, hipStream_t stream);
hipError_t launch_gemm_4d(const GemmArgs<f16_t>& g, hipStream_t stream, bool generic_epilogue);
hipError_t launch_gemm_4d(const GemmArgs<bf16_t>& g, hipStream_t stream, bool generic_epilogue);

}  // namespace zett
