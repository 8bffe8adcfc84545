"""Encoder tail fused into the VQ prologue.

In the reference the tensor the VQ receives is ``dropout(final_layer_norm(hidden_states))``
(python_transformers/modeling_t5.py:864-865, consumed at t5/load_t5.py:518; the mid-encoder call site applies the
same two ops first, :853-855).  ``T5LayerNorm`` is an RMS norm: ``weight * (x / sqrt(mean(x^2) + eps))``
(:191-207).  In eager PyTorch that tail is half a dozen elementwise passes over [N, D], and the VQ then reads the
result once more just to form |x|^2.  ``EncoderTailVQ`` does norm + dropout + |x|^2 in ONE pass
(``t5vq_rmsnorm_sqnorms_f32``) and hands both to the quantizer; its backward (``t5vq_rmsnorm_backward_f32``)
returns the gradients of the hidden states and of the norm weight.

A maintainer who wants the fusion replaces ``final_layer_norm`` / ``dropout`` of the encoder stack by identities
and calls this module where ``self.vqvae(encoding, srl_ids)`` is called.  The plain
drop-in (``VectorQuantizerEMA`` alone) does not need it.

The dropout mask is drawn with torch's generator (``bernoulli_``) and passed to the kernel as a uint8 tensor, or
given by the caller (``keep_mask=``); torch's own fused dropout draws a different Philox stream, so masks -- not
results for a given mask -- differ from eager ``nn.Dropout``.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class _TailFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, gamma, keep, scale, eps, want_image):
        ximg = ops.token_image(hidden.shape[0], hidden.shape[1], hidden.device) if want_image else None
        x, xx, rms = ops.rmsnorm_sqnorms(hidden, gamma.detach(), keep, scale, eps, want_rms=True, token_image=ximg)
        ctx.save_for_backward(hidden, gamma.detach(), keep, rms)
        ctx.scale = scale
        if ximg is None:                      # (an autograd Function returns tensors: an empty block stands for "none")
            ximg = torch.empty(0, dtype=torch.uint8, device=hidden.device)
        ctx.mark_non_differentiable(xx, ximg)
        return x, xx, ximg

    @staticmethod
    def backward(ctx, g_x, g_xx, g_ximg):
        hidden, gamma, keep, rms = ctx.saved_tensors
        gh, gg = ops.rmsnorm_backward(g_x, hidden, gamma, keep, ctx.scale, rms,
                                      want_grad_gamma=ctx.needs_input_grad[1])
        return (gh if ctx.needs_input_grad[0] else None), gg, None, None, None, None


class EncoderTailVQ(nn.Module):
    """``vq(dropout(T5LayerNorm(hidden_states)), srl)`` with the norm, the dropout and the row norms in one kernel.

    ``vq``: a ``t5vqvae_amd`` quantizer module.  ``weight``: the ``final_layer_norm.weight`` Parameter to share (so
    checkpoints keep their key), or None for a fresh ``ones(hidden_size)`` (modeling_t5.py:197)."""

    # the norm kernel also writes the fp16 image of its output from this many elements on: at C3 (5e7) tail + filtered
    # step 1226 -> 1214 us; at C1 (2.2e6) the extra buffer and output cost more than the preparation pass they save
    token_image_min_elements = 1 << 24

    def __init__(self, vq, hidden_size: int = 768, eps: float = 1e-6, dropout_rate: float = 0.1, weight=None):
        super().__init__()
        self.vq = vq
        self.weight = weight if weight is not None else nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps
        self.dropout_rate = float(dropout_rate)

    def forward(self, hidden_states, srl=None, keep_mask=None):
        d = hidden_states.shape[-1]
        flat = hidden_states.reshape(-1, d)
        if flat.dtype != torch.float32:
            raise RuntimeError(f"hidden_states must be float32 (got {flat.dtype})")
        flat = flat.contiguous()
        keep, scale = None, 1.0
        if keep_mask is not None:
            keep = keep_mask.reshape(-1, d).to(torch.uint8).contiguous()
            scale = 1.0 / (1.0 - self.dropout_rate)
        elif self.training and self.dropout_rate > 0.0:
            keep = torch.empty(flat.shape, dtype=torch.uint8, device=flat.device).bernoulli_(1.0 - self.dropout_rate)
            scale = 1.0 / (1.0 - self.dropout_rate)
        x, xx, ximg = _TailFunction.apply(flat, self.weight, keep, scale, float(self.variance_epsilon),
                                          self._wants_image(flat.shape[0], d, srl))
        return self.vq(x.view(hidden_states.shape), srl, _xx=xx, _ximg=ximg if ximg.numel() else None)

    def _wants_image(self, n: int, d: int, srl) -> bool:
        """Whether the quantizer's search of this call will be the filtered one in token order -- then the norm kernel
        also writes the fp16 image of its output and the search's preparation pass does not read x again.  (A wrong
        guess costs one unused write of N * D * 2 bytes, or the usual preparation pass: never a result.)"""
        vq = self.vq
        search = getattr(vq, "search", "scan")
        if search == "scan" or d % 64 or d > 2048 or n * d < self.token_image_min_elements:
            return False
        if srl is not None and getattr(vq, "disentanglement", False) and n >= 1024:
            return False                       # role order: the search lays its image out in that order itself
        return search == "filter" or n * vq._num_embeddings >= vq.filter_min_work
