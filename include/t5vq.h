/* t5vq.h -- C ABI of the MI355X-native (gfx950) vector-quantization bottleneck.
 *
 * Drop-in boundary for ONE hot path of SnowYJ/T5VQVAE: the VQ bottleneck of
 * vqvae/vqvae.py (VectorQuantizer / VectorQuantizerEMA forward + EMA codebook update).
 * The reference has no FFI for this path -- it is a chain of stock ATen ops inside two
 * nn.Modules -- so each entry point below cites the lines of vqvae/vqvae.py (relative to
 * the reference root) whose ATen chain it replaces.  The Python host (t5vqvae_amd/vqvae.py)
 * mirrors the reference's nn.Module interface on top of these calls via ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'ed / torch CUDA tensor storage) unless noted;
 *    tensors are dense row-major fp32; indices are int64 at the boundary;
 *  - every call enqueues work on `stream` (a hipStream_t; NULL = null stream) and returns
 *    without synchronising; no call allocates or frees device memory -- callers bring
 *    outputs and workspaces;
 *  - return value: T5VQ_OK, or an error code with a thread-local message in t5vq_last_error();
 *  - N = tokens (B*S), K = codes, D = embedding dim; D % 4 == 0 and 16-byte aligned rows.
 *  - packed key (int64): high word = order-preserving signed image of the fp32 distance,
 *    low word = code index; min() of keys == (min distance, then lowest index) == torch.argmin's
 *    first-minimum rule (vqvae/vqvae.py:122).
 */
#ifndef T5VQ_H_
#define T5VQ_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T5VQ_OK 0
#define T5VQ_ERR_INVALID 1 /* bad argument: null pointer, shape, alignment, unsupported D */
#define T5VQ_ERR_HIP 2     /* a HIP runtime call or kernel launch failed */

#define T5VQ_CODE_PAD 128  /* |e|^2 buffers are padded to a multiple of this, pad value +inf */
#define T5VQ_LOSS_MSE 0
#define T5VQ_LOSS_COSINE 1

typedef void* t5vq_stream_t; /* hipStream_t */

/* Library identity / diagnostics (no reference counterpart). */
int t5vq_version(void);
const char* t5vq_last_error(void);
/* Diagnostic: select a tile configuration of the distance kernel (0 = shipped default; other values only
 * exist in builds made with -DT5VQ_TUNING and otherwise fall back to the default).  The setting belongs to the
 * calling thread: launches made from other threads keep the default.  No reference counterpart. */
int t5vq_set_tuning(int variant);
/* Length of the padded |e|^2 buffer for K codes: K rounded up to T5VQ_CODE_PAD. */
int64_t t5vq_padded_codes(int64_t K);
/* Number of loss terms t5vq_gather_stats_f32 writes for N tokens when no token order (perm) is given: one per TOKEN
 * when N < 32768 (the 4- and 16-token workgroups of small batches; the term then does not depend on how tokens are
 * grouped), one per 64-token workgroup above. */
int64_t t5vq_gather_blocks(int64_t N);

/* Row squared norms of two matrices in one launch.
 * Replaces torch.sum(flat_input**2, dim=1, keepdim=True) and torch.sum(weight**2, dim=1)
 * (vqvae/vqvae.py:106-107, :27-28, :174-175).
 *   xx[i] = sum_d x[i,d]^2, i < N;  ee[k] = sum_d w[k,d]^2, k < K;  ee[k] = +inf, K <= k < padded(K).
 * Either matrix may be skipped by passing rows = 0. */
int t5vq_row_sqnorms_f32(const float* x, int64_t N, float* xx, const float* w, int64_t K, float* ee,
                         int64_t D, t5vq_stream_t stream);

/* Fused distance + argmin on fp32 MFMA; the N x K distance matrix never reaches HBM.
 * Replaces distances = (xx + ee) - 2 * matmul(x, W^T) and torch.argmin(distances, 1)
 * (vqvae/vqvae.py:106-108,122; plain variant :27-29,43) and, when `range` is non-null, the per-token
 * Python loop argmin(distances[i, start:end]) of the disentangled mode (:110-119, :31-40).
 *   d_ik = fl(fl(xx_i + ee_k) - 2 * dot_ik)   (same rounding order as the reference expression)
 *   keys[i] = min over codes of pack(d_ik, index)
 * index = code_offset + k, or, with range != NULL ([N,2] int64 [start,end) in global code ids), the
 * slice-relative index code_offset + k - start_i restricted to start_i <= code_offset + k < end_i --
 * the reference uses the slice-relative argmin as an absolute row (:116-119), and so does this.
 * code_offset is the global id of w's first row (0 unless the codebook is sharded by code).
 * merge == 0: keys are overwritten.  merge != 0: keys already hold valid packed keys (e.g. from another
 * codebook shard) and are min-merged.  Rows whose range misses this shard keep INT64_MAX / their old key.
 * ee must be the padded buffer written by t5vq_row_sqnorms_f32.
 * order (nullable; only with range): a permutation of the token ids, the sequence in which the kernel's 128-token
 * tiles take the tokens (t5vq_order_by_range puts tokens of one role next to each other, so a tile carries one or two
 * ranges and skips every code tile outside them: 2 D sum_r N_r K_r flops instead of 2 D N K).  Speed only: x, xx,
 * range and keys stay indexed by token, nothing is moved, the keys are the same with and without it. */
int t5vq_dist_argmin_f32(const float* x, const float* w, const float* xx, const float* ee,
                         const int64_t* range, const int32_t* order, int64_t N, int64_t K, int64_t D,
                         int64_t code_offset, int merge, int64_t* keys, t5vq_stream_t stream);

/* order[0..N) = the token ids grouped by their code range (disentangled mode: the (start, end) pairs of
 * batch['srl'], t5/load_data.py:68-99, as the loop at vqvae/vqvae.py:110-119 reads them): a counting sort over
 * (start, end) buckets of 128 codes or more; the order inside a bucket is not specified.  K, code_offset: the
 * codebook (slice) the ranges will be searched in.  work: K + 8 int32 of scratch.  The composite forwards do this
 * themselves (from 1024 tokens on); the call is for users of the fine-grained searches.  No reference counterpart
 * (the reference visits token after token). */
int t5vq_order_by_range(const int64_t* range, int64_t N, int64_t K, int64_t code_offset, int32_t* order,
                        int32_t* work, t5vq_stream_t stream);

/* Unpack keys -> indices (int64, what torch.argmin returns, vqvae/vqvae.py:122) and/or distances.
 * Either output may be NULL. */
int t5vq_unpack_keys(const int64_t* keys, int64_t N, int64_t* idx, float* dmin, t5vq_stream_t stream);

/* Full distance matrix dist[N,K] (get_latent, vqvae/vqvae.py:167-178), same MFMA mainloop. */
int t5vq_distances_f32(const float* x, const float* w, const float* xx, const float* ee, int64_t N, int64_t K,
                       int64_t D, float* dist, t5vq_stream_t stream);

/* Codeword gather + straight-through output + latent-loss terms + EMA statistics in one pass over x.
 * Replaces encodings = zeros().scatter_(); quantized = matmul(encodings, W) (vqvae/vqvae.py:124-128),
 * the loss terms (:147-152), quantized = inputs + (quantized - inputs).detach() (:160) and, for the EMA
 * statistics, torch.sum(encodings, 0) (:133) and matmul(encodings.t(), flat_input) (:141).
 *   idx_i   = low word of keys[i] (clamped to [0,K));   q_i = w[idx_i]
 *   out_ste[i] = fl(x_i + fl(q_i - x_i))   (nullable)      out_q[i] = q_i (nullable, get_hidden :236)
 *   out_diff[i] = fl(q_i - x_i)            (nullable; saved for backward)
 *   idx_out[i] = idx_i as int64            (nullable)
 *   counts[k - stats_begin]  = #{i : idx_i = k} as fp32  (nullable; zeroed by this call)
 *   embed_sum[k - stats_begin,:] = sum_{i : idx_i = k} x_i (nullable; zeroed by this call; fp32 atomics,
 *                    one flush per (workgroup, code) after an in-LDS segment sum)
 *                    both only for stats_begin <= k < stats_begin + stats_count (the whole codebook:
 *                    0, K; a code shard keeps statistics for its own slice only)
 *   loss_partial[b] = sums of (q-x)^2 (MSE) or of 1 - cos(q_i, x_i) (COSINE, ATen cosine_embedding_loss form, eps
 *                    1e-12), one per token (N < 32768) or one per 64-token workgroup; length t5vq_gather_blocks(N)
 *                    (nullable)
 *   perm (nullable): the tokens in an order that puts equal codes next to each other (t5vq_order_by_code); a
 *                    workgroup then flushes one partial row per run of equal codes instead of one per token
 *                    Outputs stay in token order.  With perm, loss_partial holds one term per TOKEN (length N,
 *                    indexed by token): which tokens share a workgroup then varies from run to run, and the
 *                    loss must not. */
int t5vq_gather_stats_f32(const float* x, const float* w, const int64_t* keys, int64_t N, int64_t K, int64_t D,
                          int loss_kind, float* out_ste, float* out_q, float* out_diff, int64_t* idx_out,
                          int64_t stats_begin, int64_t stats_count, float* counts, float* embed_sum,
                          float* loss_partial, const int32_t* perm, t5vq_stream_t stream);

/* perm[0..N) = the token ids grouped by the code in their key (counting sort; the order inside a code is not
 * specified) -- or the identity, when the histogram shows that grouping would not save the statistics kernel
 * enough flushes to cover the sort (batches whose neighbouring tokens already share codes); decided on the
 * device, no host read-back.  work: K + 2 int32 of scratch.  No reference counterpart (an ordering for
 * t5vq_gather_stats_f32). */
int t5vq_order_by_code(const int64_t* keys, int64_t N, int64_t K, int32_t* perm, int32_t* work, t5vq_stream_t stream);

/* loss_sum[0] = sum of the n partials (fixed order, fp64 accumulation). */
int t5vq_reduce_partials_f32(const float* partials, int64_t n, float* loss_sum, t5vq_stream_t stream);

/* Cluster-size EMA + Laplace smoothing + loss/perplexity scalars.
 * Replaces vqvae/vqvae.py:132-139 (mode != 0), :148/:152,:157 (loss), :161-162 (perplexity):
 *   cs <- cs*decay + one_minus_decay*counts;  n = sum(cs);  cs <- (cs + epsilon) / (n + k_epsilon) * n
 *   m = loss_sum / loss_denom;   loss = plain_variant ? m + commitment_cost*m : commitment_cost*m
 *   perplexity = exp(-sum_k p_k log(p_k + 1e-10)), p_k = counts[k] / n_tokens
 * scalars[0..3] = loss, perplexity, n (0 if mode == 0), m.   one_minus_decay and k_epsilon are passed
 * in because the reference evaluates (1 - decay) and K*epsilon in Python double before they meet fp32.
 * mode 0: scalars only (eval).  mode 1: update, all in one call (K = the whole codebook).
 * Code-sharded codebooks (K = this shard's codes, k_epsilon still global K * epsilon) split the update
 * around one scalar all-reduce:  mode 2 writes cs <- cs*decay + one_minus_decay*counts and the LOCAL
 * partial sums scalars[1] = sum p log(p + 1e-10), scalars[2] = sum cs;  the caller all-reduces
 * scalars[1..2];  mode 3 reads them back, applies the Laplace smoothing and writes the perplexity.
 * In modes 2 and 3 cluster_size may be NULL (evaluation: only the perplexity is assembled).
 * loss_sum[0 .. n_loss_terms) are added up by this kernel (fixed order, fp64 accumulation -- the order of
 * t5vq_reduce_partials_f32, bit-identical to it): n_loss_terms = 1 for a ready sum, or the forward's unreduced loss
 * terms (t5vq_vq_forward_f32's loss_terms), which saves the reduce launch. */
int t5vq_ema_cluster_f32(float* cluster_size, const float* counts, const float* loss_sum, int64_t n_loss_terms,
                         int64_t K, int64_t n_tokens, float loss_denom, float decay, float one_minus_decay,
                         float epsilon, float k_epsilon, float commitment_cost, int plain_variant, int mode,
                         float* scalars, t5vq_stream_t stream);

/* Codebook EMA: ema_w <- ema_w*decay + one_minus_decay*embed_sum;  w <- ema_w / cluster_size[:,None]
 * (vqvae/vqvae.py:141-144; in place instead of re-binding new Parameters).
 * ee_out (nullable, [>= K]): |e_k|^2 of the codebook just written, bit-identical to t5vq_row_sqnorms_f32 run on it
 * (same lane layout and order, taken from the values in registers): the next forward can take it as ee_in instead of
 * re-reading the codebook (vqvae/vqvae.py:107 recomputes the norms every call).  Only rows < K are written: the
 * +inf padding of a t5vq_padded_codes(K) buffer is the caller's to set once (t5vq_row_sqnorms_f32 does). */
int t5vq_ema_embed_f32(float* ema_w, float* w, const float* embed_sum, const float* cluster_size, int64_t K,
                       int64_t D, float decay, float one_minus_decay, float* ee_out, t5vq_stream_t stream);

/* ---- Order-independent EMA statistics (opt-in; no reference counterpart: the reference's matmul(encodings.t(),
 * flat_input), vqvae/vqvae.py:141, is deterministic on its CPU path, float atomics are not) ----------------------
 * embed_sum accumulated in 64-bit fixed point: x * 2^s with 2^s * max|x| <= 2^38.  Integer addition is associative,
 * so the sums are bit-reproducible from run to run and for any sharding of the tokens (all-reduce them as int64),
 * and they are the exact sums of the quantised inputs, rounded once in t5vq_ema_embed_fixed_f32.
 *   t5vq_max_norm_f32:        max_xx_bits[0] = bit image of max_i xx[i]  (a token-sharded caller all-reduces it with
 *                             MAX, as int32, so that every rank uses the same scale)
 *   t5vq_embed_sum_fixed_f32: embed_sum[k - stats_begin, :] = sum_{i : idx_i = k} round(x_i * 2^s)   (int64 [.,D],
 *                             zeroed by the call; N < 2^24, D <= 1024)
 *   t5vq_ema_embed_fixed_f32: t5vq_ema_embed_f32 with dw = float(embed_sum * 2^-s) (ee_out as there). */
int t5vq_max_norm_f32(const float* xx, int64_t N, int32_t* max_xx_bits, t5vq_stream_t stream);
int t5vq_embed_sum_fixed_f32(const float* x, const int64_t* keys, const int32_t* max_xx_bits, int64_t N, int64_t K,
                             int64_t D, int64_t stats_begin, int64_t stats_count, int64_t* embed_sum,
                             t5vq_stream_t stream);
int t5vq_ema_embed_fixed_f32(float* ema_w, float* w, const int64_t* embed_sum, const int32_t* max_xx_bits,
                             const float* cluster_size, int64_t K, int64_t D, float decay, float one_minus_decay,
                             float* ee_out, t5vq_stream_t stream);

/* Gradient of the straight-through output and the MSE commitment term with respect to the inputs:
 *   grad_x = g_out - diff * (g_loss[0] * coef),  diff = q - x,  coef = commitment_cost * 2 / (N*D)
 * (autograd of vqvae/vqvae.py:148,157,160 and :55,69,71).  g_out or g_loss may be NULL (= 0). */
int t5vq_ste_backward_f32(const float* g_out, const float* diff, const float* g_loss, float coef,
                          int64_t n_elems, float* grad_x, t5vq_stream_t stream);

/* out[k,:] = scale * sum_{i : idx_i = k} rows[i,:]  with scale = scale_host * (scale_dev ? scale_dev[0] : 1).
 * Plain variant's codebook gradient: rows = q - x, scale = g_loss * 2/(N*D)
 * (autograd of matmul(encodings, W) under mse_loss, vqvae/vqvae.py:51,56).  out is zeroed by the call. */
int t5vq_scatter_rows_f32(const float* rows, const int64_t* idx, const float* scale_dev, float scale_host,
                          int64_t N, int64_t K, int64_t D, float* out, t5vq_stream_t stream);

/* Cosine latent loss gradients (autograd of F.cosine_embedding_loss(..., reduction="sum") / B at
 * vqvae/vqvae.py:58-64,150-152):  with c_i = 1 - cos(q_i, x_i),
 *   grad_x[i]   = g_out[i] + s_x * dc_i/dx_i        s_x = g_loss[0] * coef_x   (coef_x = commitment_cost / B)
 *   grad_q[i]   =            s_q * dc_i/dq_i        s_q = g_loss[0] * coef_q   (coef_q = 1 / B; plain variant)
 * grad_q may be NULL (EMA variant); g_out may be NULL. */
int t5vq_cosine_backward_f32(const float* x, const float* q, const float* g_out, const float* g_loss,
                             float coef_x, float coef_q, int64_t N, int64_t D, float* grad_x, float* grad_q,
                             t5vq_stream_t stream);

/* enc[N,K] = one-hot(idx) in fp32 (vqvae/vqvae.py:124-125).  Off the hot path: every caller of the
 * reference discards it. */
int t5vq_one_hot_f32(const int64_t* idx, int64_t N, int64_t K, float* enc, t5vq_stream_t stream);

/* ---- The same search, filtered in fp16 and decided in fp32 (no reference counterpart: an implementation choice) --
 * keys[i] are bit-identical to t5vq_dist_argmin_f32(same range and code_offset, merge = 0): an fp16 MFMA pass
 * bounds every distance from both sides with a proven error bound, a second pass lists per token the codes whose
 * lower bound does not exceed the smallest upper bound (the fp32 argmin and everything tied with it are among them),
 * and the fp32 distance expression of t5vq_dist_argmin_f32 -- same summation order -- is evaluated for those codes
 * only.  Tokens the filter cannot settle (more than 16 such codes) are decided by the fp32 scan kernel run on those
 * tokens alone.  Needs D % 64 == 0.
 * exact_keys != 0: keys are bit-identical to t5vq_dist_argmin_f32's.  exact_keys == 0: the INDEX word is identical;
 * for a token with a single candidate the distance word holds an upper bound instead of the distance (its fp32
 * evaluation is skipped) -- enough for every consumer that only gathers by index.
 * compute_norms == 0: xx [N] and ee [padded(K)] are inputs (from t5vq_row_sqnorms_f32 / t5vq_rmsnorm_sqnorms_f32 /
 * t5vq_ema_embed_f32's ee_out).
 * compute_norms == 1: they are OUTPUTS: the pass that writes the fp16 images also writes the row norms (bit-identical
 * to t5vq_row_sqnorms_f32), which saves that launch and its read of x.
 * compute_norms == 2: xx is an input, ee an output.  compute_norms == 3: xx is an output, ee an input.
 * workspace: t5vq_filter_workspace_bytes(N, K, D) bytes, 256-byte aligned (fp16 images of x and w, bounds,
 * candidate lists).  n_overflow (nullable, device int32): number of tokens that took the fp32 scan.
 * order (nullable; only with range): as for t5vq_dist_argmin_f32 -- the fp16 image of x is laid out in that order,
 * x itself is not moved; xx and keys stay indexed by token. */
int64_t t5vq_filter_workspace_bytes(int64_t N, int64_t K, int64_t D);
int t5vq_dist_argmin_filtered_f32(const float* x, const float* w, float* xx, float* ee, int compute_norms,
                                  const int64_t* range, const int32_t* order, int64_t N, int64_t K, int64_t D,
                                  int64_t code_offset, int exact_keys, void* workspace, int64_t workspace_bytes,
                                  int64_t* keys, int32_t* n_overflow, t5vq_stream_t stream);

/* ---- Evaluation-side searches (callers of the path in evaluate_vqvae.py) ----------------
 *
 * The topk nearest codes per token in ascending (distance, index) order -- what
 *   a, idx1 = torch.sort(distances[dim]); idx = idx1[:top_k]          (evaluate_vqvae.py:129-135)
 * takes from the full get_latent matrix, for every row at once and without the N x K matrix: topk passes of the
 * fused distance + argmin kernel, pass j admitting only (distance, index) pairs that order after pass j-1's
 * winner.  keys is [topk, N] (row j = the j-th nearest code of every token); unpack with t5vq_unpack_keys.
 * Equal distances come out lowest index first (torch.sort leaves their order unspecified). */
int t5vq_dist_topk_f32(const float* x, const float* w, const float* xx, const float* ee, int64_t N, int64_t K,
                       int64_t D, int64_t topk, int64_t* keys, t5vq_stream_t stream);

/* Argmin of a blend of two distance matrices [N,K] (latent-space interpolation, evaluate_vqvae.py:189-200):
 *   keys[i] = min_k pack(fl(fl(weight_left * dist_left[i,k]) + fl(weight_right * dist_right[i,k])), k)
 * = torch.argmin(source_t * left_distances + target_t * right_distances, dim=1), first minimum. */
int t5vq_blend_argmin_f32(const float* dist_left, const float* dist_right, float weight_left, float weight_right,
                          int64_t N, int64_t K, int64_t* keys, t5vq_stream_t stream);

/* ---- Encoder tail fused into the VQ prologue ------------------------------------------
 *
 * T5LayerNorm (python_transformers/modeling_t5.py:191-207: no mean subtraction, learned scale) followed by dropout
 * (:864-865, and :853-854 at the mid-encoder call site) produce the tensor the VQ receives (t5/load_t5.py:518).
 * One pass over `hidden` [N,D] writes that tensor AND its row norms, so t5vq_row_sqnorms_f32 need not re-read it:
 *   r_i = sqrt(mean_d hidden[i,d]^2 + eps);  y = gamma * (hidden / r_i)          (the reference's op order)
 *   x_out = keep ? (keep[i,d] ? y * scale : 0) : y       keep: uint8 [N,D] mask, nullable (eval / p = 0);
 *                                                        scale = 1 / (1 - p), as torch's dropout applies it
 *   xx[i] = sum_d x_out[i,d]^2 (nullable);   rms[i] = r_i (nullable; saved for backward)
 * The mask is an input (drawn by the caller's RNG): bit parity with torch's Philox stream is not a goal. */
int t5vq_rmsnorm_sqnorms_f32(const float* hidden, const float* gamma, const uint8_t* keep, float scale, float eps,
                             int64_t N, int64_t D, float* x_out, float* xx, float* rms, t5vq_stream_t stream);

/* Autograd of the above:  gy = grad_x * keep * scale;  a = gy * gamma;
 *   grad_hidden = a / r - hidden * (sum_d a_d hidden_d) / (D r^3);   grad_gamma[d] = sum_i gy[i,d] hidden[i,d] / r_i
 * grad_gamma is nullable; when given, `workspace` holds t5vq_rmsnorm_backward_blocks(N) * D floats of per-workgroup
 * partial sums that a second launch adds in fixed order (deterministic, no atomics). */
int64_t t5vq_rmsnorm_backward_blocks(int64_t N);
int t5vq_rmsnorm_backward_f32(const float* grad_x, const float* hidden, const float* gamma, const uint8_t* keep,
                              float scale, const float* rms, int64_t N, int64_t D, float* grad_hidden,
                              float* grad_gamma, float* workspace, t5vq_stream_t stream);

/* ---- Composite calls: what a replacement of the module's forward must export ------------------
 * They chain the entry points above on `stream`; results are bit-identical to calling them one by one.
 *
 * t5vq_vq_forward_f32 = vqvae/vqvae.py:103-128,147-152,160 (plain variant :24-64,71) plus the EMA statistics
 * of :133,:141 in one host call: row norms -> distance + argmin -> gather / straight-through / loss terms /
 * counts / embed_sum (-> loss_sum[0] = sum of the loss terms, if loss_sum is given).  `workspace` holds the per-call
 * temporaries (keys, |x|^2, |e|^2, loss terms): t5vq_forward_workspace_bytes(N, K) bytes, 16-byte aligned, caller-owned.
 * range, out_ste, out_q, out_diff, idx_out, counts, embed_sum, loss_sum are nullable as in
 * t5vq_dist_argmin_f32 / t5vq_gather_stats_f32.
 * xx_in (nullable): |x_i|^2 [N] already computed by the producer of x (t5vq_rmsnorm_sqnorms_f32); the pass over x for
 * the row norms is then skipped.  ee_in (nullable): |e_k|^2 [t5vq_padded_codes(K)], +inf padded, 16-byte aligned,
 * e.g. the ee_out of the t5vq_vq_ema_finalize_f32 call that wrote this codebook; the pass over w is then skipped.
 * loss_terms (nullable): caller buffer of t5vq_forward_loss_terms(N, embed_sum != NULL) floats that receives the
 * unreduced loss terms; t5vq_vq_ema_finalize_f32 / t5vq_ema_cluster_f32 add them up themselves, so a single-GPU
 * caller passes loss_terms and no loss_sum and saves the reduce launch.  Whole-codebook form (code_offset 0,
 * statistics for [0,K)); a code-sharded caller uses the fine-grained calls. */
int64_t t5vq_forward_workspace_bytes(int64_t N, int64_t K);
int64_t t5vq_forward_loss_terms(int64_t N, int with_embed_sum);
int t5vq_vq_forward_f32(const float* x, const float* w, const int64_t* range, const float* xx_in, const float* ee_in,
                        int64_t N, int64_t K, int64_t D, int loss_kind, void* workspace, int64_t workspace_bytes,
                        float* out_ste, float* out_q, float* out_diff, int64_t* idx_out, float* counts,
                        float* embed_sum, float* loss_terms, float* loss_sum, t5vq_stream_t stream);

/* t5vq_vq_forward_f32 for callers that cannot allocate.  NO counterpart in the reference: it exists for callers
 * that may not call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid it -- and
 * so cannot bring the workspace the entry point above asks for.  Same arguments minus the workspace, same kernels,
 * same results; the per-call temporaries (keys, |x|^2, |e|^2, loss terms, token order, sort cursors) live in a
 * library-owned static device array of T5VQ_NOALLOC_SCRATCH_BYTES (one copy per GPU the library runs on, held until
 * the process ends; its device address is looked up once per device on the first call there, with that device
 * current).  The call allocates and frees nothing.  Limit: t5vq_forward_workspace_bytes(N, K) (about 20 N + 8 K bytes,
 * plus 512 KB of key slots for problems of fewer than 256 tiles of 128 x 128)
 * must not exceed T5VQ_NOALLOC_SCRATCH_BYTES -- e.g. N = 131072 with K = 16384, or N = 65536 with K = 65536;
 * larger shapes return T5VQ_ERR_INVALID with a message.  Every call on a GPU shares that GPU's array: calls that
 * overlap on one GPU (two streams, two host threads) race on it -- serialise them; calls on one stream are ordered
 * by the stream.  Nothing is carried from one call to the next (every temporary is rewritten before it is read).
 * Under HIP graph capture: make ONE call per device outside capture first (any shape of the kind to be captured) --
 * the first call on a device looks up the scratch symbol (hipGetSymbolAddress) and sets the kernels' dynamic-LDS
 * attribute (hipFuncSetAttribute), which a capture may refuse; every later call only enqueues kernels and is
 * capturable (tests/test_zz_no_alloc.py captures, replays and compares). */
#define T5VQ_NOALLOC_SCRATCH_BYTES (4 << 20)
int t5vq_vq_forward_noalloc_f32(const float* x, const float* w, const int64_t* range, const float* xx_in,
                                const float* ee_in, int64_t N, int64_t K, int64_t D, int loss_kind, float* out_ste,
                                float* out_q, float* out_diff, int64_t* idx_out, float* counts, float* embed_sum,
                                float* loss_terms, float* loss_sum, t5vq_stream_t stream);

/* The forward for SMALL problems in ONE launch (vqvae/vqvae.py:103-128,147-152,160, the statistics of :133,:141 and,
 * with `scalars`, the loss / perplexity of :148,:157,:161-162 of a call WITHOUT an EMA update -- evaluation, or the plain
 * variant): distance + argmin on fp32 MFMA, the code splits of a token tile merged inside the launch, gather /
 * straight-through / loss terms / counts / embed_sum by the workgroup that merged them, the scalars by the workgroup that
 * finishes last.  Results are those of t5vq_vq_forward_f32 (+ t5vq_vq_ema_finalize_f32 with update = 0) bit for bit
 * (embed_sum up to the order of its atomic adds).  Shapes: t5vq_vq_forward_small_supported(N, K, D) -- fewer than 256
 * tiles of 128 x 128, N < T5VQ_SMALL_MAX_TOKENS, D % 256 == 0, D <= 1024; plain search, mse loss.
 * workspace: t5vq_forward_workspace_bytes(N, K) bytes, as for t5vq_vq_forward_f32.
 * state: t5vq_small_state_bytes(K) bytes of device memory that the caller ZEROES ONCE (hipMemset) and then hands,
 *   untouched, to every call (the tickets of the hand-offs between workgroups; each launch leaves it zeroed again, so
 *   no call pays a memset).  Calls that share a state block must not overlap in time (one stream, or otherwise ordered).
 * xx_in, ee_in (nullable) as in t5vq_vq_forward_f32 (a missing ee_in costs a row-norm launch over w).
 * counts, embed_sum (nullable): zeroed by the call (a memset each), then accumulated.  loss_terms (nullable): N floats,
 *   one term per token.  scalars (nullable, needs K % 4 == 0 and no embed_sum): [loss, perplexity, 0, raw latent loss]
 *   as t5vq_ema_cluster_f32 mode 0 writes them from loss_denom / commitment_cost / plain_variant (K <= 16384); counts
 *   (if given) is then written once by the last workgroup from its histogram of the codes and costs no memset.
 * No reference counterpart as a call of its own: it replaces the same lines as the chain it is identical to. */
#define T5VQ_SMALL_MAX_TOKENS 32768
int64_t t5vq_small_state_bytes(int64_t K);
int t5vq_vq_forward_small_supported(int64_t N, int64_t K, int64_t D);
int t5vq_vq_forward_small_f32(const float* x, const float* w, const float* xx_in, const float* ee_in, int64_t N,
                              int64_t K, int64_t D, void* workspace, int64_t workspace_bytes, void* state,
                              float* out_ste, float* out_q, float* out_diff, int64_t* idx_out, float* counts,
                              float* embed_sum, float* loss_terms, float* scalars, float loss_denom,
                              float commitment_cost, int plain_variant, t5vq_stream_t stream);

/* The same forward (vqvae/vqvae.py:103-128,147-152,160 + :133,:141) with the filtered search of
 * t5vq_dist_argmin_filtered_f32 in place of the scan: identical indices and outputs, one host call.  Needs
 * D % 64 == 0, 64 <= D <= 4096.  workspace: t5vq_forward_filtered_workspace_bytes(N, K, D) bytes, 256-byte aligned.
 * xx_in, ee_in, loss_terms as in t5vq_vq_forward_f32 (norms that are not given are written by the search's own
 * first pass).  n_overflow (nullable, device int32[1]): tokens the filter could not settle (decided by the scan kernel). */
int64_t t5vq_forward_filtered_workspace_bytes(int64_t N, int64_t K, int64_t D);
int t5vq_vq_forward_filtered_f32(const float* x, const float* w, const int64_t* range, const float* xx_in,
                                 const float* ee_in, int64_t N, int64_t K, int64_t D, int loss_kind, void* workspace,
                                 int64_t workspace_bytes, float* out_ste, float* out_q, float* out_diff,
                                 int64_t* idx_out, float* counts, float* embed_sum, float* loss_terms, float* loss_sum,
                                 int32_t* n_overflow, t5vq_stream_t stream);

/* t5vq_vq_ema_finalize_f32 = vqvae/vqvae.py:131-144 (if update != 0) and the loss / perplexity scalars
 * (:148/:152,:157,:161-162): t5vq_ema_cluster_f32 (mode 1 or 0) followed by t5vq_ema_embed_f32.  A
 * token-sharded caller all-reduces counts / embed_sum / loss_sum between the two composite calls.
 * loss_sum / n_loss_terms as in t5vq_ema_cluster_f32, ee_out (nullable) as in t5vq_ema_embed_f32. */
int t5vq_vq_ema_finalize_f32(float* cluster_size, float* ema_w, float* w, const float* counts,
                             const float* embed_sum, const float* loss_sum, int64_t n_loss_terms, int64_t K, int64_t D,
                             int64_t n_tokens, float loss_denom, float decay, float one_minus_decay, float epsilon,
                             float k_epsilon, float commitment_cost, int plain_variant, int update, float* scalars,
                             float* ee_out, t5vq_stream_t stream);

/* ---- fp16 images made by the producers of the rows (no reference counterpart: the filtered search is an
 * implementation choice) ------------------------------------------------------------------------------------------
 * The filtered search works on fp16 images of x and w (every row scaled by a power of two, with its inverse scale and,
 * for codes, sqrt(|e|^2)).  Its preparation pass makes them from the fp32 rows in every call; these variants let the
 * kernels that WRITE the rows emit the images, so that pass reads no row a producer has covered:
 *   code image  (t5vq_code_image_bytes(K, D) bytes, 256-byte aligned, caller-owned, kept between calls): written by
 *     the codebook update (the _img forms of t5vq_ema_embed_f32 / _fixed_f32 / t5vq_vq_ema_finalize_f32: the image of
 *     the codebook they have just written, vqvae/vqvae.py:144; D <= 2048), or by a search called with code_image_valid = 0 (it
 *     makes the image as before, but in the caller's block: an evaluation loop pays for it once).  A search called
 *     with code_image_valid != 0 reads it and needs ee as an input.  The image must be that of w, bit for bit: the
 *     caller keeps it only while w is unchanged (the module keys it by the codebook's pointer and version).
 *   token image (t5vq_token_image_bytes(N, D) bytes, 256-byte aligned): written by t5vq_rmsnorm_sqnorms_img_f32 for the
 *     x it produces (python_transformers/modeling_t5.py:864-865); a search given one needs xx as an input and ignores
 *     it when the tokens are visited in role order (order != NULL, or ranges in the composite forward from 1024 tokens).
 * Images are bit-identical to the ones the search makes itself, so are all results (tests/test_gpu_filter.py).
 * range_sparse (t5vq_dist_argmin_filtered_img_f32, ranged search only): != 0 when the caller expects most tokens' ranges
 *   to miss this codebook slice (one code shard of a disentangled codebook): the codes are then cut into more, smaller
 *   pieces so that the few token tiles with work fill the chip.  A performance hint: the keys do not depend on it.
 * Every other argument as in the form without _img. */
int64_t t5vq_code_image_bytes(int64_t K, int64_t D);
int64_t t5vq_token_image_bytes(int64_t N, int64_t D);
int t5vq_dist_argmin_filtered_img_f32(const float* x, const float* w, float* xx, float* ee, int compute_norms,
                                      const int64_t* range, const int32_t* order, int64_t N, int64_t K, int64_t D,
                                      int64_t code_offset, int exact_keys, void* workspace, int64_t workspace_bytes,
                                      int64_t* keys, int32_t* n_overflow, void* code_image, int code_image_valid,
                                      const void* token_image, int range_sparse, t5vq_stream_t stream);
int t5vq_vq_forward_filtered_img_f32(const float* x, const float* w, const int64_t* range, const float* xx_in,
                                     const float* ee_in, int64_t N, int64_t K, int64_t D, int loss_kind,
                                     void* workspace, int64_t workspace_bytes, float* out_ste, float* out_q,
                                     float* out_diff, int64_t* idx_out, float* counts, float* embed_sum,
                                     float* loss_terms, float* loss_sum, int32_t* n_overflow, void* code_image,
                                     int code_image_valid, const void* token_image, t5vq_stream_t stream);
int t5vq_ema_embed_img_f32(float* ema_w, float* w, const float* embed_sum, const float* cluster_size, int64_t K,
                           int64_t D, float decay, float one_minus_decay, float* ee_out, void* code_image,
                           t5vq_stream_t stream);
int t5vq_ema_embed_fixed_img_f32(float* ema_w, float* w, const int64_t* embed_sum, const int32_t* max_xx_bits,
                                 const float* cluster_size, int64_t K, int64_t D, float decay, float one_minus_decay,
                                 float* ee_out, void* code_image, t5vq_stream_t stream);
int t5vq_vq_ema_finalize_img_f32(float* cluster_size, float* ema_w, float* w, const float* counts,
                                 const float* embed_sum, const float* loss_sum, int64_t n_loss_terms, int64_t K,
                                 int64_t D, int64_t n_tokens, float loss_denom, float decay, float one_minus_decay,
                                 float epsilon, float k_epsilon, float commitment_cost, int plain_variant, int update,
                                 float* scalars, float* ee_out, void* code_image, t5vq_stream_t stream);
int t5vq_rmsnorm_sqnorms_img_f32(const float* hidden, const float* gamma, const uint8_t* keep, float scale, float eps,
                                 int64_t N, int64_t D, float* x_out, float* xx, float* rms, void* token_image,
                                 t5vq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* T5VQ_H_ */
