// Shared between the translation units of libgadfly_hip.so; NOT part of the C-ABI (hidden visibility).
#pragma once
#include <hip/hip_runtime.h>

// records the message gf_last_error() returns (thread-local) and returns `code`
__attribute__((visibility("hidden"), format(printf, 2, 3)))
int gf_internal_error(int code, const char *fmt, ...);

// hipGetLastError() -> 0, or -2 with the message recorded
__attribute__((visibility("hidden")))
int gf_internal_check_launch(const char *what);

// opt a set of kernels in to `bytes` (> 64 KB) of dynamic LDS on the device the stream belongs to; remembered per
// (which, device).  false if the attribute could not be set.
__attribute__((visibility("hidden")))
bool gf_internal_lds_opt_in(int which, hipStream_t st, const void *const *funcs, int nfuncs, size_t bytes);
