"""nn.Module-shaped adapter over the CPU oracle, so the shared golden runner can drive it (tests only)."""
import os
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import vq_oracle as vo  # noqa: E402


class OracleQuantizer(nn.Module):
    def __init__(self, cls, num_embeddings, embedding_dim, commitment_cost, decay=0.99, epsilon=1e-5,
                 disentanglement=False, loss="mse", path="faithful", **kwargs):
        super().__init__()
        self.ema = cls == "VectorQuantizerEMA"
        self._num_embeddings, self._embedding_dim = num_embeddings, embedding_dim
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._commitment_cost, self._decay, self._epsilon = commitment_cost, decay, epsilon
        self.disentanglement, self.loss, self.path = disentanglement, loss, path
        if self.ema:
            self.register_buffer("_ema_cluster_size", torch.zeros(num_embeddings))
            self._ema_w = nn.Parameter(torch.zeros(num_embeddings, embedding_dim))

    def _state(self):
        if self.ema:
            return vo.VQState(self._embedding.weight, self._ema_w.data, self._ema_cluster_size)
        return vo.VQState(self._embedding.weight)

    def forward(self, inputs, srl=None):
        st = self._state()
        fn = vo.faithful_forward if self.path == "faithful" else vo.algorithmic_forward
        o = fn(st, inputs, srl, ema=self.ema, training=self.training, commitment_cost=self._commitment_cost,
               decay=self._decay, epsilon=self._epsilon, disentanglement=self.disentanglement, loss=self.loss)
        if self.ema and self.training:
            self._ema_cluster_size = st.cluster_size
            self._ema_w = nn.Parameter(st.ema_w)
            self._embedding.weight = nn.Parameter(st.weight)
        enc = o.encodings
        if enc is None:
            enc = torch.zeros(o.indices.shape[0], self._num_embeddings).scatter_(1, o.indices.unsqueeze(1), 1)
        return o.loss, o.quantized, o.perplexity, enc

    def get_latent(self, inputs):
        return vo.get_latent(self._state(), inputs)

    def get_indices(self, inputs, srl=None):
        return vo.get_indices(self._state(), inputs, srl, self.disentanglement)

    def get_hidden(self, inputs, srl=None):
        return vo.get_hidden(self._state(), inputs, srl, self.disentanglement)


    def topk_codes(self, inputs, top_k):
        return vo.topk_codes(self._state(), inputs, top_k)

    def blended_indices(self, source_code, target_code, source_t, target_t):
        return vo.blended_indices(self._state(), source_code, target_code, source_t, target_t)


def factory(path):
    def make(cls, **kw):
        return OracleQuantizer(cls, path=path, **kw)
    return make
