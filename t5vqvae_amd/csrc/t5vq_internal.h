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
// code_image (nullable) with code_image_valid != 0: the codebook's image is read from the block; == 0: the call
// writes it there (for the next call).  token_image (nullable): a valid image of x in token order (ignored with order).
// range_sparse: the caller expects most tokens' ranges to miss this codebook slice (more, smaller code splits).
extern "C" int t5vq_internal_dist_argmin_filtered(const float* x, const float* w, float* xx, float* ee, int compute_norms,
                                                  const int64_t* range, const int32_t* order, int64_t N, int64_t K,
                                                  int64_t D, int64_t code_offset, int exact_keys, void* workspace,
                                                  int64_t workspace_bytes, int64_t* keys, int32_t* n_overflow,
                                                  T5vqZeroArgs zero, void* code_image, int code_image_valid,
                                                  const void* token_image, int range_sparse, hipStream_t stream);

// ---- fp16 images of rows for the filtered search (t5vq_filter.hip); the kernels that produce the rows (codebook
// update, fused encoder tail) can emit them, so the search's preparation pass need not read those rows again.
// A row is scaled by a power of two so that its largest magnitude lies in [2^14, 2^15) (exact; rows of any magnitude
// fit fp16's range); inv = the inverse factor.  Blocks (t5vq_code_image_bytes / t5vq_token_image_bytes):
//   code image:  [K * D fp16 | pad to 256 B | kpad f32 inverse row scales | pad | kpad f32 sqrt(|e|^2)], kpad = padded K;
//                entries K .. kpad-1: inverse scale 1, sqrt(|e|^2) = +inf (padded codes are never candidates)
//   token image: [N * D fp16 | pad to 256 B | N f32 inverse row scales]
struct T5vqCodeImage {
    _Float16* rows;
    float* inv;
    float* se;
};
struct T5vqTokenImage {
    _Float16* rows;
    float* inv;
};
static inline int64_t t5vq_align256(int64_t v) { return (v + 255) / 256 * 256; }
static inline T5vqCodeImage t5vq_code_image_parts(void* base, int64_t K, int64_t D) {
    T5vqCodeImage im = {nullptr, nullptr, nullptr};
    if (!base) return im;
    char* b = static_cast<char*>(base);
    const int64_t kpad = t5vq_padded_codes(K);
    im.rows = reinterpret_cast<_Float16*>(b);
    im.inv = reinterpret_cast<float*>(b + t5vq_align256(2 * K * D));
    im.se = reinterpret_cast<float*>(b + t5vq_align256(2 * K * D) + t5vq_align256(4 * kpad));
    return im;
}
static inline T5vqTokenImage t5vq_token_image_parts(void* base, int64_t N, int64_t D) {
    T5vqTokenImage im = {nullptr, nullptr};
    if (!base) return im;
    im.rows = reinterpret_cast<_Float16*>(base);
    im.inv = reinterpret_cast<float*>(static_cast<char*>(base) + t5vq_align256(2 * N * D));
    return im;
}
// scale factor of a row whose largest magnitude is m (the same in every lane); *inv = its inverse
__device__ __forceinline__ float t5vq_image_scale(float m, float* inv) {
    const int e = ((__float_as_int(m) >> 23) & 0xFF) - 127;   // floor(log2 m) for normal m
    int sexp = 14 - e;
    if (m == 0.f || e < -100) sexp = 0;                         // all-zero (or denormal) row: no scaling
    sexp = sexp > 100 ? 100 : (sexp < -100 ? -100 : sexp);
    *inv = __int_as_float((127 - sexp) << 23);
    return __int_as_float((sexp + 127) << 23);
}
__device__ __forceinline__ float t5vq_absmax4(float m, float x, float y, float z, float w) {
    return fmaxf(m, fmaxf(fmaxf(fabsf(x), fabsf(y)), fmaxf(fabsf(z), fabsf(w))));
}
// four consecutive elements of a row (float4 #c4) into its image
__device__ __forceinline__ void t5vq_image_store4(_Float16* row, int c4, float x, float y, float z, float w, float scale) {
    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
    f16x4_t o;
    o.x = static_cast<_Float16>(x * scale);
    o.y = static_cast<_Float16>(y * scale);
    o.z = static_cast<_Float16>(z * scale);
    o.w = static_cast<_Float16>(w * scale);
    reinterpret_cast<f16x4_t*>(row)[c4] = o;
}

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
