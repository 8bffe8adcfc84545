// c_abi_demo.cpp -- one VectorQuantizerEMA training step through the C ABI alone (include/t5vq.h): no Python, no torch.
// What a C / C++ serving or training loop would write.  Build: make -C t5vqvae_amd/csrc demo
//
//   hipMalloc inputs / state / outputs / one workspace  ->  t5vq_vq_forward_f32  ->  t5vq_vq_ema_finalize_f32
//   and, beside it, the filtered search (t5vq_dist_argmin_filtered_f32), whose indices must equal the scan's.
// The program checks the indices on the host with a plain double-precision nearest-code loop (a self-check of the
// demo, not a code path of the library) and exits non-zero on any disagreement outside near-ties.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "t5vq.h"

#define HIP_OK(e)                                                                   \
    do {                                                                            \
        hipError_t err_ = (e);                                                      \
        if (err_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_));               \
            return 2;                                                               \
        }                                                                           \
    } while (0)
#define VQ_OK(e)                                                                    \
    do {                                                                            \
        if ((e) != T5VQ_OK) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #e, t5vq_last_error());              \
            return 3;                                                               \
        }                                                                           \
    } while (0)

static float gauss(uint64_t& s) {   // sum of 8 uniforms, centred: good enough for a demo
    float a = 0.f;
    for (int i = 0; i < 8; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        a += static_cast<float>((s >> 40) & 0xFFFFFF) / 16777216.0f;
    }
    return (a - 4.0f) * 1.2247f;
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 4096, K = argc > 2 ? atoll(argv[2]) : 1024, D = 768;
    const float decay = 0.99f, eps = 1e-5f, cc = 0.25f;
    std::vector<float> hx(N * D), hw(K * D), hema(K * D), hcs(K, 0.f);
    uint64_t seed = 7;
    for (auto& v : hw) v = gauss(seed);
    for (auto& v : hema) v = gauss(seed);
    for (int64_t i = 0; i < N; ++i) {   // tokens near codes, so that the nearest code is well separated
        const int64_t c = (i * 2654435761ULL) % K;
        for (int64_t j = 0; j < D; ++j) hx[i * D + j] = hw[c * D + j] + 0.5f * gauss(seed);
    }

    float *x, *w, *ema_w, *cs, *ste, *stats, *scalars, *xx, *ee;
    int64_t *idx, *keys_f;
    int32_t* unsettled;
    void *ws, *ws_f;
    const int64_t ws_bytes = t5vq_forward_workspace_bytes(N, K), wsf_bytes = t5vq_filter_workspace_bytes(N, K, D);
    HIP_OK(hipMalloc(&x, N * D * 4));
    HIP_OK(hipMalloc(&w, K * D * 4));
    HIP_OK(hipMalloc(&ema_w, K * D * 4));
    HIP_OK(hipMalloc(&cs, K * 4));
    HIP_OK(hipMalloc(&ste, N * D * 4));
    HIP_OK(hipMalloc(&idx, N * 8));
    HIP_OK(hipMalloc(&keys_f, N * 8));
    HIP_OK(hipMalloc(&stats, (K * D + K + 4) * 4));   // [embed_sum K*D | counts K | loss_sum 1]: one all-reduce unit
    HIP_OK(hipMalloc(&scalars, 16));
    HIP_OK(hipMalloc(&xx, N * 4));
    HIP_OK(hipMalloc(&ee, t5vq_padded_codes(K) * 4));
    HIP_OK(hipMalloc(&unsettled, 4));
    HIP_OK(hipMalloc(&ws, ws_bytes));
    HIP_OK(hipMalloc(&ws_f, wsf_bytes));
    HIP_OK(hipMemcpy(x, hx.data(), N * D * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(w, hw.data(), K * D * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ema_w, hema.data(), K * D * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(cs, hcs.data(), K * 4, hipMemcpyHostToDevice));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    float* embed_sum = stats;
    float* counts = stats + K * D;
    float* loss_sum = counts + K;

    // the filtered search first (it reads the pre-update codebook, like the forward below)
    VQ_OK(t5vq_dist_argmin_filtered_f32(x, w, xx, ee, /*compute_norms=*/1, /*range=*/nullptr, /*order=*/nullptr, N, K, D,
                                        /*code_offset=*/0,
                                        /*exact_keys=*/1, ws_f, wsf_bytes, keys_f, unsettled, stream));
    // forward: indices, straight-through output, loss terms, EMA statistics (vqvae/vqvae.py:103-160)
    VQ_OK(t5vq_vq_forward_f32(x, w, nullptr, /*xx_in=*/nullptr, /*ee_in=*/nullptr, N, K, D, T5VQ_LOSS_MSE, ws, ws_bytes,
                              ste, nullptr, nullptr, idx, counts, embed_sum, /*loss_terms=*/nullptr, loss_sum, stream));
    // (a token-sharded job all-reduces `stats` here)
    // finalize: cluster sizes, Laplace smoothing, codebook, loss and perplexity (vqvae/vqvae.py:131-162)
    // (ee_out = ee: the norms of the codebook just written, ready to be the next step's ee_in)
    VQ_OK(t5vq_vq_ema_finalize_f32(cs, ema_w, w, counts, embed_sum, loss_sum, /*n_loss_terms=*/1, K, D, N,
                                   static_cast<float>(N * D), decay, static_cast<float>(1.0 - 0.99), eps,
                                   static_cast<float>(K * 1e-5), cc, 0, 1, scalars, /*ee_out=*/ee, stream));
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<int64_t> hidx(N), hkeys(N);
    float hs[4];
    int32_t hu = 0;
    HIP_OK(hipMemcpy(hidx.data(), idx, N * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hkeys.data(), keys_f, N * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hs, scalars, 16, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&hu, unsettled, 4, hipMemcpyDeviceToHost));

    int64_t bad = 0, near_tie = 0, filt_diff = 0;
    for (int64_t i = 0; i < N; ++i) {
        double best = 1e300, second = 1e300;
        int64_t bi = -1;
        for (int64_t k = 0; k < K; ++k) {
            double d = 0;
            for (int64_t j = 0; j < D; ++j) {
                const double t = static_cast<double>(hx[i * D + j]) - hw[k * D + j];
                d += t * t;
            }
            if (d < best) { second = best; best = d; bi = k; }
            else if (d < second) second = d;
        }
        if (hidx[i] != bi) {
            if (second - best <= 1e-6 * (2.0 * D)) ++near_tie; else ++bad;
        }
        if ((hkeys[i] & 0xFFFFFFFFLL) != hidx[i]) ++filt_diff;
    }
    printf("N=%lld K=%lld D=%lld: loss=%.6f perplexity=%.3f n=%.1f | index mismatches %lld (near-ties %lld) | "
           "filtered search differs on %lld tokens, %d unsettled\n",
           (long long)N, (long long)K, (long long)D, hs[0], hs[1], hs[2], (long long)bad, (long long)near_tie,
           (long long)filt_diff, hu);
    return (bad == 0 && filt_diff == 0) ? 0 : 1;
}
