// gemm4d_bf16.hip — see gemm4d.inc
#define ZETT_GEMM_T bf16_t
#include "gemm4d.inc"
