// ipk_internal.hpp -- what the translation units of libimagepipe_amd.so share besides the public header
#pragma once
#include <hip/hip_runtime_api.h>

namespace ipk {
// records the thread-local ipk_last_error() text and returns `code` (ipk_api.cpp)
int internal_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// IPK_OK when ipk_init succeeded (binds the context's device on this thread), else IPK_ERR_NOT_INIT with the error text set
int internal_require_init();
int internal_device();
}  // namespace ipk
