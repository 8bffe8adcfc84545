"""Multi-GPU forms of the VQ bottleneck: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests of the host logic).

Token sharding (BASELINE config 4): every rank holds N/G tokens and a full replica of the codebook state.
Distances, argmin, gather and the loss terms are per token, so the only exchange is ONE all-reduce(sum) of
the fused statistics buffer [embed_sum K*D | counts K | loss_sum 1] that ``vqvae._VQFunction`` lays out
contiguously; every rank then runs the identical finalize kernels on identical reduced values, so replicas
stay bit-identical without a broadcast.  Loss and perplexity are those of the concatenated batch.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_tokens(module, group=None, world_tokens: int = 0):
    """Turn a VectorQuantizerEMA / VectorQuantizer replica into a token-sharded one (in place).

    ``group``: process group to reduce over (default: the world group).  ``world_tokens``: the global
    token count if ranks hold unequal shares (default: local count x world size)."""
    if not dist.is_initialized():
        raise RuntimeError("shard_tokens needs an initialised torch.distributed process group")
    module.process_group = group if group is not None else dist.group.WORLD
    module.world_tokens = int(world_tokens)
    return module


def broadcast_state(module, src: int = 0, group=None):
    """Make replicas identical at start-up (the reference seeds one process; here rank `src` wins)."""
    for t in module.state_dict().values():
        dist.broadcast(t, src=src, group=group)
    return module
