// Internal to libt5vq.so: shared between its translation units, not part of the C ABI (include/t5vq.h).
#ifndef T5VQ_INTERNAL_H_
#define T5VQ_INTERNAL_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "t5vq.h"

int t5vq_internal_fail(int code, const char* fmt, const char* a, const char* b);

#define T5VQ_I_CHECK_HIP(expr)                                                                         \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return t5vq_internal_fail(T5VQ_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define T5VQ_I_REQUIRE(cond, msg)                                                          \
    do {                                                                                   \
        if (!(cond)) return t5vq_internal_fail(T5VQ_ERR_INVALID, "%s (%s)", msg, #cond);   \
    } while (0)
#endif
