"""Host side of the disentangled (role-grouped) codebook: per-token code ranges as the reference's loader builds
them, and the index convention the quantizer applies to them.

The reference stores one ``(start, end)`` pair per token in ``batch['srl']`` (t5/load_data.py:68-99): every word's
semantic-role label is looked up in ``dimension_dict`` (e.g. ``{"V": (0, 1000), "residual": (1000, K)}``,
train_t5.py:309), repeated once per word piece, labels without an entry and the padding fall back to
``"residual"``.  The quantizer then searches only codes ``[start, end)`` of each token (vqvae/vqvae.py:110-119).

Index convention.  The reference returns the argmin's position INSIDE the slice and uses it unchanged as a codebook
row (no ``+ start``), so a token of a role that starts at code 1000 is quantized to one of rows ``0 .. end-start``.
``module.index_mode`` makes the choice explicit:
  "reference" (default)  slice-relative indices used as absolute rows: bit-for-bit the reference's behaviour;
  "absolute"             ``start + argmin``: each role really owns its code range.  NOT what the published
                         checkpoint was trained with; offered for new training runs.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import torch

INDEX_MODES = ("reference", "absolute")


def role_vocab(label_rows: Iterable[Sequence[str]]) -> Dict[str, int]:
    """Role label -> id, most frequent first, then "FIX" and "PAD" (t5/utils.py:22-44)."""
    freq: Dict[str, int] = {}
    for row in label_rows:
        for lab in row:
            freq[lab] = freq.get(lab, 0) + 1
    vocab = {lab: i for i, (lab, _) in enumerate(sorted(freq.items(), key=lambda kv: kv[1], reverse=True))}
    n = len(vocab)
    vocab["FIX"] = n
    vocab["PAD"] = n + 1
    return vocab


def expand_role_ranges(labels: Sequence[str], pieces_per_word: Sequence[int],
                       dimension_dict: Dict[str, Tuple[int, int]], seq_len: int) -> torch.Tensor:
    """[seq_len, 2] int64 code ranges of one example (t5/load_data.py:68-86,97): word i's range repeated
    ``pieces_per_word[i]`` times, unknown labels and padding get ``dimension_dict["residual"]``."""
    if len(labels) != len(pieces_per_word):
        raise ValueError("one label per word is required")            # the reference asserts the same (:71)
    residual = tuple(dimension_dict["residual"])
    rows: List[Tuple[int, int]] = []
    for lab, t in zip(labels, pieces_per_word):
        rows.extend([tuple(dimension_dict[lab]) if lab in dimension_dict else residual] * int(t))
    if len(rows) > seq_len:
        raise ValueError(f"{len(rows)} word pieces do not fit seq_len={seq_len}")   # reference skips such lines (:64)
    rows.extend([residual] * (seq_len - len(rows)))
    return torch.tensor(rows, dtype=torch.int64).view(seq_len, 2)


def bucket_by_role(ranges: torch.Tensor):
    """(order, sorted_ranges): the stable permutation that makes tokens of one role contiguous, so that every
    128-token tile of the distance kernel carries one or two ranges and skips the code tiles outside them."""
    order = torch.argsort(ranges[:, 0] * (1 << 32) + ranges[:, 1], stable=True)
    return order, ranges.index_select(0, order)
