"""Counterpart of the reference's ``src/utils/model_utils.py``.

The HIP attention kernel derives the causal mask from (query, key) indices in registers and never reads a
(T,T) tensor; this helper exists only so callers of the reference API keep working.
"""
import torch


def generate_square_subsequent_mask(sz: int) -> torch.Tensor:
    """bool (sz,sz), True strictly above the diagonal (= key j > query i is masked); model_utils.py:4-7."""
    idx = torch.arange(sz)
    return idx[None, :] > idx[:, None]
