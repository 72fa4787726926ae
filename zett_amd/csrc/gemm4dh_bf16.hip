// gemm4dh_bf16.hip — see gemm4dh.inc
#define ZETT_GEMM_T bf16_t
#include "gemm4dh.inc"
