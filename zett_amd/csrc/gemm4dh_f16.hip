// gemm4dh_f16.hip — see gemm4dh.inc
#define ZETT_GEMM_T f16_t
#include "gemm4dh.inc"
