// sift_internal.h -- host-side helpers shared by the translation units of libcusift_amd.so (not installed).
#pragma once

#include "../../include/cusift_amd_all.h"

// Sets the thread's error text (cusift_last_error) and returns `code`: `return cusift_fail(CUSIFT_ERR_HIP, "...", ...)`.
int cusift_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
