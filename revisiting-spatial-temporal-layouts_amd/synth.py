"""Seeded synthetic layouts and deterministic closed-form weights.

Everything the GPU box needs to rebuild the exact inputs and weights the golden
fixtures under ``tests/golden/`` were captured with: no ``torch.manual_seed``
module init (that depends on torch version / constructor order), only
splitmix64 over (name-hash, element index), computed in numpy uint64.

Input contract follows the reference collater
(``src/modelling/datasets.py:52-125,239-288``):

* slot 0 of every frame (padded frames too) is the CLS object: category =
  ``cls`` id, box ``[0,0,1,1]``, score 1.0;
* real frames are a prefix, the last real frame is the "extract" frame with
  only the CLS object, padded frames have frame type 0;
* padded object slots: category 0, box zeros, score 0;
* ``src_key_padding_mask_boxes = categories == 0``,
  ``src_key_padding_mask_frames = frame_types == 0``, ``lengths`` = real
  frames (incl. extract).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

_MASK64 = (1 << 64) - 1

# dataset vocabularies (src/modelling/configs.py:40-89)
DATASETS = {
    # name: (unique_categories, cls_id, first_object_id, regular, empty, extract)
    "something": dict(unique_categories=4, cls=3, obj_ids=(1, 2), regular=2, empty=3, extract=4),
    "action_genome": dict(unique_categories=38, cls=1, obj_ids=tuple(range(2, 38)), regular=1, empty=3, extract=2),
}


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK64
    return h


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser over a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(key: int, n: int) -> np.ndarray:
    """n float64 in [0,1): element i = splitmix64(splitmix64(key) + i) >> 11 * 2^-53."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.array([key & _MASK64], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64) + base
    z = splitmix64(idx)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _sym(key: int, shape, bound: float) -> torch.Tensor:
    n = int(np.prod(shape))
    u = uniform01(key, n)
    return torch.from_numpy(((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape))


def make_state_dict(shapes: Dict[str, tuple], seed: int = 1234, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic fp32 weights for every key of a Stlt / StltBackbone state dict.

    ``shapes`` maps state-dict key -> shape (take it from ``module.state_dict()``).
    Bounds are torch-like (1/sqrt(fan_in) for matrices) so activations have
    realistic magnitudes and the softmaxes are not uniform; ``gain`` scales the
    attention in-projections to sharpen (gain>1) the attention distributions.
    """
    out: Dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        key = fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15 & _MASK64)
        if name.endswith("position_ids"):
            out[name] = torch.arange(shape[-1], dtype=torch.int64).expand(shape).clone()
            continue
        leaf = name.rsplit(".", 1)[-1]
        parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
        if leaf in ("cls_token", "pos_embed"):
            out[name] = _sym(key, shape, 0.5)
            continue
        if parent in ("norm1", "norm2", "layer_norm", "ln"):
            if leaf == "weight":
                out[name] = 1.0 + _sym(key, shape, 0.1)
            else:
                out[name] = _sym(key, shape, 0.05)
        elif parent in ("category_embeddings", "position_embeddings", "frame_type_embedding"):
            out[name] = _sym(key, shape, 1.0)
        elif leaf == "in_proj_weight":
            out[name] = _sym(key, shape, gain * (6.0 / (shape[0] + shape[1])) ** 0.5 * 1.5)
        elif leaf in ("in_proj_bias",) or leaf == "bias":
            out[name] = _sym(key, shape, 0.05)
        elif leaf == "weight" and len(shape) in (2, 5):  # Linear, or the 1x1x1 Conv3d projector
            out[name] = _sym(key, shape, 1.0 / (shape[1] ** 0.5))
        else:
            raise KeyError(f"no init rule for {name} {shape}")
    return out


def make_batch(
    B: int,
    T: int,
    N: int,
    dataset: str = "something",
    seed: int = 0,
    dense: bool = False,
    with_scores: Optional[bool] = None,
    min_len: Optional[int] = None,
) -> Dict[str, torch.Tensor]:
    """Seeded collater-shaped batch (SURVEY.md §8d 'Synthetic inputs').

    ``T`` includes the extract frame, ``N`` includes the CLS slot.  Clip 0 always
    has ``length == T``.  ``dense=True`` makes every clip full length and every
    object slot real.
    """
    ds = DATASETS[dataset]
    if with_scores is None:
        with_scores = dataset == "action_genome"
    rng = np.random.Generator(np.random.PCG64(seed))
    cats = np.zeros((B, T, N), dtype=np.int64)
    boxes = np.zeros((B, T, N, 4), dtype=np.float32)
    scores = np.zeros((B, T, N), dtype=np.float32)
    ftypes = np.zeros((B, T), dtype=np.int64)
    lengths = np.zeros((B,), dtype=np.int64)
    lo = max(2, (T + 1) // 2) if min_len is None else max(2, min_len)
    lo = min(lo, T)
    obj_ids = np.asarray(ds["obj_ids"], dtype=np.int64)
    for b in range(B):
        ln = T if (dense or b == 0) else int(rng.integers(lo, T + 1))
        lengths[b] = ln
        # CLS slot in every frame, padded frames included (datasets.py:247-264)
        cats[b, :, 0] = ds["cls"]
        boxes[b, :, 0] = (0.0, 0.0, 1.0, 1.0)
        scores[b, :, 0] = 1.0
        for t in range(ln - 1):
            k = (N - 1) if dense else int(rng.integers(0, N))
            if k > 0:
                cats[b, t, 1 : k + 1] = obj_ids[rng.integers(0, len(obj_ids), size=k)]
                xy = np.sort(rng.random((k, 2, 2), dtype=np.float64), axis=1)  # [k,(lo,hi),(x,y)]
                boxes[b, t, 1 : k + 1, 0] = xy[:, 0, 0]
                boxes[b, t, 1 : k + 1, 1] = xy[:, 0, 1]
                boxes[b, t, 1 : k + 1, 2] = xy[:, 1, 0]
                boxes[b, t, 1 : k + 1, 3] = xy[:, 1, 1]
                scores[b, t, 1 : k + 1] = 0.5 + 0.5 * rng.random(k)
                ftypes[b, t] = ds["regular"]
            else:
                ftypes[b, t] = ds["empty"]
        ftypes[b, ln - 1] = ds["extract"]
    batch = {
        "categories": torch.from_numpy(cats),
        "boxes": torch.from_numpy(boxes),
        "frame_types": torch.from_numpy(ftypes),
        "lengths": torch.from_numpy(lengths),
        "src_key_padding_mask_boxes": torch.from_numpy(cats == 0),
        "src_key_padding_mask_frames": torch.from_numpy(ftypes == 0),
    }
    if with_scores:
        batch["scores"] = torch.from_numpy(scores)
    return batch


def make_video_samples(dataset: str, n_videos: int, N: int, seed: int):
    """Seeded per-video dicts shaped like ``StltDataset.__getitem__`` output (reference datasets.py:52-125): variable
    frame counts, fixed N object slots, last frame = extract.  Input of the collater (device or reference)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for i in range(n_videos):
        n_frames = int(rng.integers(2, 9))
        b = make_batch(1, n_frames, N, dataset=dataset, seed=seed * 100 + i, with_scores=True, min_len=n_frames)
        label = torch.tensor(int(rng.integers(0, 174))) if dataset == "something" else torch.from_numpy(rng.random(157).astype(np.float32))
        out.append({"video_id": f"vid{i}", "categories": b["categories"][0], "boxes": b["boxes"][0], "scores": b["scores"][0],
                    "frame_types": b["frame_types"][0], "lengths": torch.tensor(n_frames), "labels": label})
    return out


# Named shape configurations (BASELINE.json "configs")
CONFIGS = {
    "micro": dict(T=5, N=3, hidden_size=32, num_attention_heads=4, num_spatial_layers=1, num_temporal_layers=1,
                  num_classes=7, dataset="something"),
    # the smallest model the checkpoint round-trip fixture trains on the GPU box (tools/make_ckpt_fixture.py: 58 KB of weights; head dim 8)
    "nano": dict(T=9, N=3, hidden_size=16, num_attention_heads=2, num_spatial_layers=1, num_temporal_layers=1,
                 num_classes=7, dataset="something"),
    "cfg1": dict(T=16, N=4, hidden_size=256, num_attention_heads=4, num_spatial_layers=4, num_temporal_layers=8,
                 num_classes=174, dataset="something"),
    "cfg2": dict(T=32, N=7, hidden_size=768, num_attention_heads=12, num_spatial_layers=4, num_temporal_layers=8,
                 num_classes=174, dataset="something"),
    "cfg2p": dict(T=33, N=8, hidden_size=768, num_attention_heads=12, num_spatial_layers=4, num_temporal_layers=8,
                  num_classes=174, dataset="something"),
    # the reference's defaults: layout_num_frames 16 (+ 1 extract frame), 4 object slots (+ CLS) — utils/parser.py:62-66, datasets.py:97-113
    "refdef": dict(T=17, N=5, hidden_size=768, num_attention_heads=12, num_spatial_layers=4, num_temporal_layers=8,
                   num_classes=174, dataset="something"),
    # a head dim other than 64 (96): the reference takes any hidden_size % num_attention_heads == 0 (configs.py:92-111)
    "heads": dict(T=9, N=6, hidden_size=384, num_attention_heads=4, num_spatial_layers=2, num_temporal_layers=3,
                  num_classes=174, dataset="something"),
    # a hidden size that is not a multiple of 32 (head dim 25)
    "odd": dict(T=7, N=4, hidden_size=100, num_attention_heads=4, num_spatial_layers=2, num_temporal_layers=2,
                num_classes=174, dataset="something"),
    "cfg4": dict(T=64, N=36, hidden_size=768, num_attention_heads=12, num_spatial_layers=4, num_temporal_layers=8,
                 num_classes=157, dataset="action_genome"),
}


def model_kwargs(name: str) -> dict:
    c = CONFIGS[name]
    return dict(
        num_classes=c["num_classes"],
        unique_categories=DATASETS[c["dataset"]]["unique_categories"],
        hidden_size=c["hidden_size"],
        num_attention_heads=c["num_attention_heads"],
        num_spatial_layers=c["num_spatial_layers"],
        num_temporal_layers=c["num_temporal_layers"],
        hidden_dropout_prob=0.0,
    )


# The small learnable task of the epoch-shell fixture (tools/gen_golden_fit.py -> tests/golden/fit_micro.npz): the label of a clip is a
# function of its number of real frames, which the temporal tower sees through the frame types and positions.
FIT_TASK = dict(config="micro", T=9, N=3, clips_per_batch=16, train_batches=2, val_batches=2, epochs=6, warmup_epochs=1, lr=6e-3,
                weight_decay=1e-3, clip_val=5.0, weight_seed=77)


def fit_batch(split: str, epoch: int, index: int) -> Dict[str, torch.Tensor]:
    """Batch `index` of `epoch` of the fit task's train split (a fresh shuffle per epoch, as a DataLoader(shuffle=True) gives) or of
    its fixed validation split, with labels."""
    t = FIT_TASK
    seed = 40000 + 100 * epoch + index if split == "train" else 90000 + index
    b = make_batch(t["clips_per_batch"], t["T"], t["N"], seed=seed)
    b["labels"] = (b["lengths"] - 2) % CONFIGS[t["config"]]["num_classes"]
    return b


def make_appearance_features(B: int, seed: int = 0, channels: int = 2048) -> torch.Tensor:
    """Stand-in for the R3D-50 feature map of Resnet3D.forward_features (reference models.py:221-222):
    (B, 2048, 2, 4, 4), non-negative like a post-ReLU map."""
    u = uniform01(fnv1a64("appearance_features") ^ (seed * 0x9E3779B97F4A7C15 & _MASK64), B * channels * 32)
    return torch.from_numpy((u * 1.5).astype(np.float32).reshape(B, channels, 2, 4, 4))


def flops_per_clip(T: int, N: int, d: int, n_sp: int, n_tp: int, classes: int) -> float:
    """Algorithmic dense forward FLOPs per clip."""
    return (
        n_sp * (24.0 * T * N * d * d + 4.0 * T * N * N * d)
        + n_tp * (24.0 * T * d * d + 4.0 * T * T * d)
        + 2.0 * d * d
        + 2.0 * d * classes
    )
