// gemm4d_f16.hip — see gemm4d.inc
#define ZETT_GEMM_T f16_t
#include "gemm4d.inc"
