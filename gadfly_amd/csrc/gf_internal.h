// Shared between the translation units of libgadfly_hip.so; NOT part of the C-ABI (hidden visibility).
#pragma once
#include <hip/hip_runtime.h>

// records the message gf_last_error() returns (thread-local) and returns `code`
__attribute__((visibility("hidden"), format(printf, 2, 3)))
int gf_internal_error(int code, const char *fmt, ...);

// hipGetLastError() -> 0, or -2 with the message recorded
__attribute__((visibility("hidden")))
int gf_internal_check_launch(const char *what);
