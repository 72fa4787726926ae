// retok.hip.h — placeholder until the retokenizer kernels land.
#pragma once
#include "common.hip.h"

struct zett_retok { int device; };

extern "C" {
int zett_retok_create(const zett_retok_model*, int, zett_retok**) { return zett::fail(ZETT_E_NOT_IMPLEMENTED, "retokenizer not built yet"); }
int zett_retok_destroy(zett_retok*) { return 0; }
int zett_retokenize(zett_retok*, const uint8_t*, const int32_t*, int64_t, int32_t, int32_t, int32_t*, int64_t*, int64_t*, void*) {
    return zett::fail(ZETT_E_NOT_IMPLEMENTED, "retokenizer not built yet");
}
}
