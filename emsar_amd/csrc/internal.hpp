// internal.hpp -- what the other translation units of libemsar_hip.so may know about a context (not part of the ABI)
#ifndef EMSAR_INTERNAL_HPP
#define EMSAR_INTERNAL_HPP
#include <hip/hip_runtime.h>

#include "../../include/emsar_hip.h"

hipStream_t emsar_internal_stream(emsar_hip_ctx *ctx);
int emsar_internal_device(const emsar_hip_ctx *ctx);
void emsar_internal_set_error(emsar_hip_ctx *ctx, const char *call, const char *what);
#endif
