// Internal to libt5vq.so: shared between its translation units, not part of the C ABI (include/t5vq.h).
#ifndef T5VQ_INTERNAL_H_
#define T5VQ_INTERNAL_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "t5vq.h"

int t5vq_internal_fail(int code, const char* fmt, const char* a, const char* b);
extern "C" int t5vq_internal_scan_listed(const float* x, const float* w, const float* xx, const float* ee,
                                         const int64_t* range, int64_t N, int64_t K, int64_t D, int64_t code_offset,
                                         const int* tok_list, const int* n_listed, int64_t* keys, hipStream_t stream);

// Buffers the filtered search's first kernel zeroes on the way (the composite forward's statistics and sort
// counters: three memset launches less).  Lengths in 4-byte words; pointers 16-byte aligned.
struct T5vqZeroArgs {
    void* p[3];
    long long n[3];
};
extern "C" int t5vq_internal_dist_argmin_filtered(const float* x, const float* w, float* xx, float* ee, int compute_norms,
                                                  const int64_t* range, const int32_t* order, int64_t N, int64_t K,
                                                  int64_t D, int64_t code_offset, int exact_keys, void* workspace,
                                                  int64_t workspace_bytes, int64_t* keys, int32_t* n_overflow,
                                                  T5vqZeroArgs zero, hipStream_t stream);

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
