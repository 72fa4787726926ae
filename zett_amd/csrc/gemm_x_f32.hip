// gemm_x_f32.hip — see gemm_x.inc
#define ZETT_GEMM_T float
#include "gemm_x.inc"
