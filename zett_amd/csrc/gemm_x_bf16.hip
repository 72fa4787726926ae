// gemm_x_bf16.hip — see gemm_x.inc
#define ZETT_GEMM_T bf16_t
#include "gemm_x.inc"
