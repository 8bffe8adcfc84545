// t5vq_torch.cpp -- the thin PyTorch-ROCm extension over the C ABI (include/t5vq.h): TORCH_LIBRARY ops that take
// tensors, allocate outputs / workspaces with torch's caching allocator, and enqueue the library's kernels on
// torch's current HIP stream.  No arithmetic happens here.  The nn.Modules of t5vqvae_amd/vqvae.py (the drop-in
// for vqvae/vqvae.py's VectorQuantizer / VectorQuantizerEMA) call these ops from a torch.autograd.Function; one
// module call = ONE op call (t5vq::vq_step) on a single GPU, two (t5vq::vq_forward, all-reduce, t5vq::vq_finalize)
// when the tokens are sharded.
//
// Built by t5vqvae_amd/csrc/Makefile into libt5vq_torch.so (links libt5vq.so, libtorch, libc10, libc10_hip) and
// loaded with torch.ops.load_library (t5vqvae_amd/_lib.py).
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <optional>
#include <tuple>

#include "t5vq.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

inline void ok(int rc, const char* what) { TORCH_CHECK(rc == T5VQ_OK, "t5vq::", what, " failed (code ", rc, "): ", t5vq_last_error()); }

inline void* stream_of(const Tensor& t) { return at::hip::getCurrentHIPStream(t.device().index()).stream(); }

inline const Tensor& need(const Tensor& t, const char* name, at::ScalarType dtype = at::kFloat, int64_t ndim = -1) {
    TORCH_CHECK(t.is_cuda(), name, " must live on a ROCm device (got ", t.device(), "); t5vqvae_amd has no CPU path");
    TORCH_CHECK(t.scalar_type() == dtype, name, " must be ", dtype, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    TORCH_CHECK(ndim < 0 || t.dim() == ndim, name, " must have ", ndim, " dims, got shape ", t.sizes());
    return t;
}
// every tensor of a call lives on the device of its first one: a pointer into another GPU's memory would fault there
inline void same_device(const Tensor& ref, const OptTensor& t, const char* name) {
    if (t.has_value() && t->defined())
        TORCH_CHECK(t->device() == ref.device(), name, " is on ", t->device(), ", expected ", ref.device());
}
template <class T = float>
inline T* ptr(const OptTensor& t) { return t.has_value() && t->defined() ? static_cast<T*>(t->data_ptr()) : nullptr; }

// One contiguous statistics buffer per call: [embed_sum K*D | counts K | loss_sum 1 | pad | loss terms] -- its head
// [0, K*D + K + 1) is the unit a token-sharded run all-reduces in ONE collective (t5vqvae_amd/sharded.py).
struct StatsLayout {
    int64_t kd, counts, loss_sum, terms, n_terms, total;
    StatsLayout(int64_t N, int64_t K, int64_t D, bool embed_sum) {
        kd = embed_sum ? K * D : 0;
        counts = kd;
        loss_sum = kd + K;
        terms = (loss_sum + 1 + 3) / 4 * 4;
        n_terms = t5vq_forward_loss_terms(N, embed_sum ? 1 : 0);
        total = terms + n_terms;
    }
};

struct Fwd {
    Tensor ste, idx, stats;
    OptTensor diff, q, unsettled;
};

// code_image / token_image (filtered search only): the fp16 image blocks of include/t5vq.h's last section.  A code image
// is read when code_image_valid, else written by this call's preparation pass; a token image is always a valid one.
Fwd forward_impl(const Tensor& x, const Tensor& w, const OptTensor& ranges, const OptTensor& xx, const OptTensor& ee,
                 int64_t loss_kind, bool filtered, bool want_diff, bool want_q, bool want_embed_sum, bool reduce_loss,
                 const OptTensor& code_image = OptTensor(), bool code_image_valid = false,
                 const OptTensor& token_image = OptTensor()) {
    need(x, "x", at::kFloat, 2), need(w, "w", at::kFloat, 2);
    const int64_t N = x.size(0), D = x.size(1), K = w.size(0);
    TORCH_CHECK(w.size(1) == D, "x and w disagree on D");
    TORCH_CHECK(N > 0 && K > 0, "empty input");
    if (ranges.has_value()) {
        need(*ranges, "ranges", at::kLong, 2);
        TORCH_CHECK(ranges->size(0) == N && ranges->size(1) == 2, "ranges must be [", N, ", 2]");
    }
    if (xx.has_value()) TORCH_CHECK(need(*xx, "xx").numel() == N, "xx must hold ", N, " row norms");
    if (ee.has_value()) TORCH_CHECK(need(*ee, "ee").numel() == t5vq_padded_codes(K), "ee must be the padded |e|^2 buffer");
    if (filtered) TORCH_CHECK(D % 64 == 0 && D <= 4096, "the filtered search needs D % 64 == 0, D <= 4096");
    if (code_image.has_value()) {
        TORCH_CHECK(need(*code_image, "code_image", at::kByte).numel() >= t5vq_code_image_bytes(K, D),
                    "code_image smaller than t5vq_code_image_bytes(K, D)");
        TORCH_CHECK(!code_image_valid || ee.has_value(), "a valid code image comes with the codebook's norms (ee)");
    }
    if (token_image.has_value()) {
        TORCH_CHECK(need(*token_image, "token_image", at::kByte).numel() >= t5vq_token_image_bytes(N, D),
                    "token_image smaller than t5vq_token_image_bytes(N, D)");
        TORCH_CHECK(xx.has_value(), "a token image comes with the row norms of x (xx)");
    }
    same_device(x, code_image, "code_image"), same_device(x, token_image, "token_image");
    same_device(x, w, "w"), same_device(x, ranges, "ranges"), same_device(x, xx, "xx"), same_device(x, ee, "ee");
    const auto f32 = x.options();
    Fwd r;
    r.ste = at::empty_like(x);
    r.idx = at::empty({N}, f32.dtype(at::kLong));
    if (want_diff) r.diff = at::empty_like(x);
    if (want_q) r.q = at::empty_like(x);
    const StatsLayout st(N, K, D, want_embed_sum);
    r.stats = at::empty({st.total}, f32);
    float* sp = r.stats.data_ptr<float>();
    const int64_t ws_bytes = filtered ? t5vq_forward_filtered_workspace_bytes(N, K, D) : t5vq_forward_workspace_bytes(N, K);
    Tensor ws = at::empty({ws_bytes}, f32.dtype(at::kByte));
    void* stream = stream_of(x);
    if (filtered) {
        r.unsettled = at::empty({1}, f32.dtype(at::kInt));
        ok(t5vq_vq_forward_filtered_img_f32(
               x.data_ptr<float>(), w.data_ptr<float>(), ptr<int64_t>(ranges), ptr(xx), ptr(ee), N, K, D,
               static_cast<int>(loss_kind), ws.data_ptr(), ws_bytes, r.ste.data_ptr<float>(), ptr(r.q), ptr(r.diff),
               r.idx.data_ptr<int64_t>(), sp + st.counts, want_embed_sum ? sp : nullptr, sp + st.terms,
               reduce_loss ? sp + st.loss_sum : nullptr, r.unsettled->data_ptr<int32_t>(), ptr<void>(code_image),
               code_image_valid ? 1 : 0, ptr<void>(token_image), stream),
           "vq_forward (filtered)");
    } else {
        ok(t5vq_vq_forward_f32(x.data_ptr<float>(), w.data_ptr<float>(), ptr<int64_t>(ranges), ptr(xx), ptr(ee), N, K, D,
                               static_cast<int>(loss_kind), ws.data_ptr(), ws_bytes, r.ste.data_ptr<float>(), ptr(r.q),
                               ptr(r.diff), r.idx.data_ptr<int64_t>(), sp + st.counts, want_embed_sum ? sp : nullptr,
                               sp + st.terms, reduce_loss ? sp + st.loss_sum : nullptr, stream),
           "vq_forward");
    }
    return r;   // ws goes back to the caching allocator here: stream-ordered, the kernels above were enqueued first
}

Tensor finalize_impl(const OptTensor& cs, const OptTensor& ema_w, const OptTensor& w, const Tensor& stats, int64_t N_local,
                     int64_t K, int64_t D, bool with_embed_sum, bool reduced_loss, int64_t n_tokens, double loss_denom,
                     double decay, double one_minus_decay, double epsilon, double k_epsilon, double commitment_cost,
                     bool plain_variant, bool update, const OptTensor& ee_out,
                     const OptTensor& code_image = OptTensor()) {
    need(stats, "stats");
    if (code_image.has_value())
        TORCH_CHECK(need(*code_image, "code_image", at::kByte).numel() >= t5vq_code_image_bytes(K, D),
                    "code_image smaller than t5vq_code_image_bytes(K, D)");
    same_device(stats, code_image, "code_image");
    const StatsLayout st(N_local, K, D, with_embed_sum);
    TORCH_CHECK(stats.numel() == st.total, "stats buffer does not match (N, K, D, with_embed_sum)");
    if (update) {
        TORCH_CHECK(cs.has_value() && ema_w.has_value() && w.has_value() && with_embed_sum,
                    "update needs cluster_size, ema_w, w and the embed sums");
        need(*cs, "cluster_size"), need(*ema_w, "ema_w", at::kFloat, 2), need(*w, "w", at::kFloat, 2);
        TORCH_CHECK(cs->numel() == K && ema_w->size(0) == K && ema_w->size(1) == D && w->size(0) == K && w->size(1) == D,
                    "state tensors do not match (K, D)");
    }
    if (ee_out.has_value()) TORCH_CHECK(need(*ee_out, "ee_out").numel() >= K, "ee_out must hold at least K norms");
    same_device(stats, cs, "cluster_size"), same_device(stats, ema_w, "ema_w"), same_device(stats, w, "w");
    same_device(stats, ee_out, "ee_out");
    Tensor scalars = at::empty({4}, stats.options());
    const float* sp = stats.data_ptr<float>();
    ok(t5vq_vq_ema_finalize_img_f32(update ? ptr(cs) : nullptr, update ? ptr(ema_w) : nullptr, update ? ptr(w) : nullptr,
                                sp + st.counts, update ? sp : nullptr, reduced_loss ? sp + st.loss_sum : sp + st.terms,
                                reduced_loss ? 1 : st.n_terms, K, D, n_tokens, static_cast<float>(loss_denom),
                                static_cast<float>(decay), static_cast<float>(one_minus_decay),
                                static_cast<float>(epsilon), static_cast<float>(k_epsilon),
                                static_cast<float>(commitment_cost), plain_variant ? 1 : 0, update ? 1 : 0,
                                scalars.data_ptr<float>(), update ? ptr(ee_out) : nullptr,
                                update ? ptr<void>(code_image) : nullptr, stream_of(stats)),
       "vq_finalize");
    return scalars;
}

// ---- ops ----------------------------------------------------------------------------------------------------
// vq_forward = vqvae/vqvae.py:103-128,147-152,160 + the statistics of :133,:141 (t5vq_vq_forward_f32 /
// t5vq_vq_forward_filtered_f32).  Returns (ste, idx, diff?, q?, stats, unsettled?).
std::tuple<Tensor, Tensor, OptTensor, OptTensor, Tensor, OptTensor> vq_forward(
    const Tensor& x, const Tensor& w, const OptTensor& ranges, const OptTensor& xx, const OptTensor& ee, int64_t loss_kind,
    bool filtered, bool want_diff, bool want_q, bool want_embed_sum, bool reduce_loss, const OptTensor& code_image,
    bool code_image_valid, const OptTensor& token_image) {
    const c10::DeviceGuard guard(x.device());
    Fwd r = forward_impl(x, w, ranges, xx, ee, loss_kind, filtered, want_diff, want_q, want_embed_sum, reduce_loss,
                         code_image, code_image_valid, token_image);
    return {r.ste, r.idx, r.diff, r.q, r.stats, r.unsettled};
}

// vq_finalize = vqvae/vqvae.py:131-144 + the loss / perplexity scalars (:148/:152,:157,:161-162)
// (t5vq_vq_ema_finalize_f32).  Returns scalars [loss, perplexity, n, raw latent loss].
Tensor vq_finalize(const OptTensor& cs, const OptTensor& ema_w, const OptTensor& w, const Tensor& stats, int64_t n_local,
                   int64_t K, int64_t D, bool with_embed_sum, bool reduced_loss, int64_t n_tokens, double loss_denom,
                   double decay, double one_minus_decay, double epsilon, double k_epsilon, double commitment_cost,
                   bool plain_variant, bool update, const OptTensor& ee_out, const OptTensor& code_image) {
    const c10::DeviceGuard guard(stats.device());
    return finalize_impl(cs, ema_w, w, stats, n_local, K, D, with_embed_sum, reduced_loss, n_tokens, loss_denom, decay,
                         one_minus_decay, epsilon, k_epsilon, commitment_cost, plain_variant, update, ee_out, code_image);
}

// vq_step = the whole module call on one GPU: vq_forward then vq_finalize, one dispatcher crossing.
// Returns (ste, idx, diff?, q?, scalars, unsettled?); loss / perplexity are scalars[0..1] -- the 0-dim tensors over them
// are made by the Python caller (vqvae._scalar_of): outputs that alias each other would break the schema's promise.
//
// small_state (optional): a zeroed block of t5vq_small_state_bytes(K) bytes that only calls ordered after one another
// share (the module keeps one).  With it, a call WITHOUT an EMA update (evaluation, or the plain variant) on a small
// problem (t5vq_vq_forward_small_supported) is ONE launch: t5vq_vq_forward_small_f32 writes the scalars itself.
std::tuple<Tensor, Tensor, OptTensor, OptTensor, Tensor, OptTensor> vq_step(
    const Tensor& x, const Tensor& w, const OptTensor& ranges, const OptTensor& xx, const OptTensor& ee,
    const OptTensor& cs, const OptTensor& ema_w, int64_t loss_kind, bool filtered, bool want_diff, bool want_q,
    double loss_denom, double decay, double one_minus_decay, double epsilon, double k_epsilon, double commitment_cost,
    bool plain_variant, bool update, const OptTensor& ee_out, const OptTensor& small_state, const OptTensor& code_image,
    bool code_image_valid, const OptTensor& token_image) {
    const c10::DeviceGuard guard(x.device());
    if (!update && small_state.has_value() && small_state->defined() && !ranges.has_value() && !filtered &&
        loss_kind == T5VQ_LOSS_MSE && x.dim() == 2 && w.dim() == 2 && w.size(0) % 4 == 0 && w.size(0) <= 16384 &&
        t5vq_vq_forward_small_supported(x.size(0), w.size(0), x.size(1))) {
        need(x, "x", at::kFloat, 2), need(w, "w", at::kFloat, 2), need(*small_state, "small_state", at::kByte);
        const int64_t N = x.size(0), D = x.size(1), K = w.size(0);
        TORCH_CHECK(w.size(1) == D, "x and w disagree on D");
        TORCH_CHECK(small_state->numel() >= t5vq_small_state_bytes(K), "small_state smaller than t5vq_small_state_bytes(K)");
        if (xx.has_value()) TORCH_CHECK(need(*xx, "xx").numel() == N, "xx must hold ", N, " row norms");
        if (ee.has_value()) TORCH_CHECK(need(*ee, "ee").numel() == t5vq_padded_codes(K), "ee must be the padded |e|^2 buffer");
        same_device(x, w, "w"), same_device(x, xx, "xx"), same_device(x, ee, "ee"), same_device(x, small_state, "small_state");
        const auto f32 = x.options();
        Tensor ste = at::empty_like(x), idx = at::empty({N}, f32.dtype(at::kLong)), scalars = at::empty({4}, f32);
        OptTensor diff, q;
        if (want_diff) diff = at::empty_like(x);
        if (want_q) q = at::empty_like(x);
        const int64_t ws_bytes = t5vq_forward_workspace_bytes(N, K);
        Tensor ws = at::empty({ws_bytes}, f32.dtype(at::kByte));
        ok(t5vq_vq_forward_small_f32(x.data_ptr<float>(), w.data_ptr<float>(), ptr(xx), ptr(ee), N, K, D, ws.data_ptr(),
                                     ws_bytes, small_state->data_ptr(), ste.data_ptr<float>(), ptr(q), ptr(diff),
                                     idx.data_ptr<int64_t>(), nullptr, nullptr, nullptr, scalars.data_ptr<float>(),
                                     static_cast<float>(loss_denom), static_cast<float>(commitment_cost),
                                     plain_variant ? 1 : 0, stream_of(x)),
           "vq_step (single launch)");
        return {ste, idx, diff, q, scalars, OptTensor()};
    }
    Fwd r = forward_impl(x, w, ranges, xx, ee, loss_kind, filtered, want_diff, want_q, update, false, code_image,
                         code_image_valid, token_image);
    const OptTensor wopt = update ? OptTensor(w) : OptTensor();
    Tensor scalars = finalize_impl(cs, ema_w, wopt, r.stats, x.size(0), w.size(0), x.size(1), update, false, x.size(0),
                                   loss_denom, decay, one_minus_decay, epsilon, k_epsilon, commitment_cost, plain_variant,
                                   update, ee_out, code_image);   // the update leaves the NEW codebook's image there
    return {r.ste, r.idx, r.diff, r.q, scalars, r.unsettled};
}

// vq_backward (MSE): grad_x = g_out - diff * (g_loss * coef)   (t5vq_ste_backward_f32)
Tensor ste_backward(const OptTensor& g_out, const Tensor& diff, const OptTensor& g_loss, double coef) {
    const c10::DeviceGuard guard(diff.device());
    need(diff, "diff");
    OptTensor go = g_out, gl = g_loss;
    if (go.has_value()) go = need(go->contiguous(), "g_out");
    if (gl.has_value()) gl = need(gl->contiguous(), "g_loss");
    TORCH_CHECK(!go.has_value() || go->numel() == diff.numel(), "g_out must have diff's element count");
    TORCH_CHECK(!gl.has_value() || gl->numel() == 1, "g_loss must be a scalar");
    same_device(diff, go, "g_out"), same_device(diff, gl, "g_loss");
    Tensor out = at::empty_like(diff);
    ok(t5vq_ste_backward_f32(ptr(go), diff.data_ptr<float>(), ptr(gl), static_cast<float>(coef), diff.numel(),
                             out.data_ptr<float>(), stream_of(diff)),
       "ste_backward");
    return out;
}

// codebook gradient of the plain variant: out[k] = scale * sum_{i: idx_i = k} rows[i]   (t5vq_scatter_rows_f32)
Tensor scatter_rows(const Tensor& rows, const Tensor& idx, const OptTensor& scale_dev, double scale_host, int64_t K) {
    const c10::DeviceGuard guard(rows.device());
    need(rows, "rows", at::kFloat, 2), need(idx, "idx", at::kLong);
    OptTensor sd = scale_dev;
    if (sd.has_value()) sd = need(sd->contiguous(), "scale_dev");
    TORCH_CHECK(K > 0 && idx.numel() == rows.size(0), "idx must hold one code per row, K must be positive");
    TORCH_CHECK(!sd.has_value() || sd->numel() == 1, "scale_dev must be a scalar");
    same_device(rows, idx, "idx"), same_device(rows, sd, "scale_dev");
    Tensor out = at::empty({K, rows.size(1)}, rows.options());
    ok(t5vq_scatter_rows_f32(rows.data_ptr<float>(), idx.data_ptr<int64_t>(), ptr(sd), static_cast<float>(scale_host),
                             rows.size(0), K, rows.size(1), out.data_ptr<float>(), stream_of(rows)),
       "scatter_rows");
    return out;
}

std::tuple<Tensor, OptTensor> cosine_backward(const Tensor& x, const Tensor& q, const OptTensor& g_out,
                                              const OptTensor& g_loss, double coef_x, double coef_q, bool want_grad_q) {
    const c10::DeviceGuard guard(x.device());
    need(x, "x", at::kFloat, 2), need(q, "q", at::kFloat, 2);
    OptTensor go = g_out, gl = g_loss;
    if (go.has_value()) go = need(go->contiguous(), "g_out");
    if (gl.has_value()) gl = need(gl->contiguous(), "g_loss");
    TORCH_CHECK(q.sizes() == x.sizes(), "x and q must have the same shape");
    TORCH_CHECK(!go.has_value() || go->numel() == x.numel(), "g_out must have x's element count");
    TORCH_CHECK(!gl.has_value() || gl->numel() == 1, "g_loss must be a scalar");
    same_device(x, q, "q"), same_device(x, go, "g_out"), same_device(x, gl, "g_loss");
    Tensor gx = at::empty_like(x);
    OptTensor gq;
    if (want_grad_q) gq = at::empty_like(x);
    ok(t5vq_cosine_backward_f32(x.data_ptr<float>(), q.data_ptr<float>(), ptr(go), ptr(gl), static_cast<float>(coef_x),
                                static_cast<float>(coef_q), x.size(0), x.size(1), gx.data_ptr<float>(), ptr(gq),
                                stream_of(x)),
       "cosine_backward");
    return {gx, gq};
}

// get_latent (vqvae/vqvae.py:167-178): the full [N, K] distance matrix
Tensor distances(const Tensor& x, const Tensor& w) {
    const c10::DeviceGuard guard(x.device());
    need(x, "x", at::kFloat, 2), need(w, "w", at::kFloat, 2);
    const int64_t N = x.size(0), D = x.size(1), K = w.size(0);
    TORCH_CHECK(w.size(1) == D, "x and w disagree on D");
    same_device(x, w, "w");
    Tensor xx = at::empty({N}, x.options()), ee = at::empty({t5vq_padded_codes(K)}, x.options());
    Tensor out = at::empty({N, K}, x.options());
    void* stream = stream_of(x);
    ok(t5vq_row_sqnorms_f32(x.data_ptr<float>(), N, xx.data_ptr<float>(), w.data_ptr<float>(), K, ee.data_ptr<float>(), D,
                            stream),
       "row_sqnorms");
    ok(t5vq_distances_f32(x.data_ptr<float>(), w.data_ptr<float>(), xx.data_ptr<float>(), ee.data_ptr<float>(), N, K, D,
                          out.data_ptr<float>(), stream),
       "distances");
    return out;
}

Tensor one_hot(const Tensor& idx, int64_t K) {
    const c10::DeviceGuard guard(idx.device());
    need(idx, "idx", at::kLong);
    TORCH_CHECK(K > 0, "K must be positive");
    Tensor enc = at::empty({idx.numel(), K}, idx.options().dtype(at::kFloat));
    ok(t5vq_one_hot_f32(idx.data_ptr<int64_t>(), idx.numel(), K, enc.data_ptr<float>(), stream_of(idx)), "one_hot");
    return enc;
}

// |e|^2 of a codebook in the padded layout every search takes (vqvae/vqvae.py:107)
Tensor code_sqnorms(const Tensor& w) {
    const c10::DeviceGuard guard(w.device());
    need(w, "w", at::kFloat, 2);
    Tensor ee = at::empty({t5vq_padded_codes(w.size(0))}, w.options());
    ok(t5vq_row_sqnorms_f32(nullptr, 0, nullptr, w.data_ptr<float>(), w.size(0), ee.data_ptr<float>(), w.size(1),
                            stream_of(w)),
       "row_sqnorms");
    return ee;
}

// Tensors that are not on a ROCm device: there is no CPU path, say so instead of the dispatcher's listing
void no_cpu_path(const c10::OperatorHandle& op, c10::Stack*) {
    TORCH_CHECK(false, op.schema().name(), ": tensors must live on a ROCm device; t5vqvae_amd has no CPU path");
}

}  // namespace

TORCH_LIBRARY(t5vq, m) {
    m.def("vq_forward(Tensor x, Tensor w, Tensor? ranges, Tensor? xx, Tensor? ee, int loss_kind, bool filtered, "
          "bool want_diff, bool want_q, bool want_embed_sum, bool reduce_loss, Tensor(f!)? code_image=None, "
          "bool code_image_valid=False, Tensor? token_image=None) -> "
          "(Tensor, Tensor, Tensor?, Tensor?, Tensor, Tensor?)");
    m.def("vq_finalize(Tensor(a!)? cluster_size, Tensor(b!)? ema_w, Tensor(c!)? w, Tensor stats, int n_local, int K, int D, "
          "bool with_embed_sum, bool reduced_loss, int n_tokens, float loss_denom, float decay, float one_minus_decay, "
          "float epsilon, float k_epsilon, float commitment_cost, bool plain_variant, bool update, Tensor(d!)? ee_out, "
          "Tensor(f!)? code_image=None) "
          "-> Tensor");
    m.def("vq_step(Tensor x, Tensor(c!) w, Tensor? ranges, Tensor? xx, Tensor? ee, Tensor(a!)? cluster_size, "
          "Tensor(b!)? ema_w, int loss_kind, bool filtered, bool want_diff, bool want_q, float loss_denom, float decay, "
          "float one_minus_decay, float epsilon, float k_epsilon, float commitment_cost, bool plain_variant, bool update, "
          "Tensor(d!)? ee_out, Tensor(e!)? small_state, Tensor(f!)? code_image=None, bool code_image_valid=False, "
          "Tensor? token_image=None) -> (Tensor, Tensor, Tensor?, Tensor?, Tensor, Tensor?)");
    m.def("ste_backward(Tensor? g_out, Tensor diff, Tensor? g_loss, float coef) -> Tensor");
    m.def("scatter_rows(Tensor rows, Tensor idx, Tensor? scale_dev, float scale_host, int K) -> Tensor");
    m.def("cosine_backward(Tensor x, Tensor q, Tensor? g_out, Tensor? g_loss, float coef_x, float coef_q, "
          "bool want_grad_q) -> (Tensor, Tensor?)");
    m.def("distances(Tensor x, Tensor w) -> Tensor");
    m.def("one_hot(Tensor idx, int K) -> Tensor");
    m.def("code_sqnorms(Tensor w) -> Tensor");
}

TORCH_LIBRARY_IMPL(t5vq, CUDA, m) {   // ROCm devices carry the CUDA dispatch key in torch
    m.impl("vq_forward", &vq_forward);
    m.impl("vq_finalize", &vq_finalize);
    m.impl("vq_step", &vq_step);
    m.impl("ste_backward", &ste_backward);
    m.impl("scatter_rows", &scatter_rows);
    m.impl("cosine_backward", &cosine_backward);
    m.impl("distances", &distances);
    m.impl("one_hot", &one_hot);
    m.impl("code_sqnorms", &code_sqnorms);
}

TORCH_LIBRARY_IMPL(t5vq, CPU, m) {
    for (const char* name : {"vq_forward", "vq_finalize", "vq_step", "ste_backward", "scatter_rows", "cosine_backward",
                             "distances", "one_hot", "code_sqnorms"})
        m.impl(name, torch::CppFunction::makeFromBoxedFunction<&no_cpu_path>());
}
