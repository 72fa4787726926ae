// gemm_x_f16.hip — see gemm_x.inc
#define ZETT_GEMM_T f16_t
#include "gemm_x.inc"
