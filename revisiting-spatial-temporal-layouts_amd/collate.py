"""Device-side counterpart of the reference's ``StltCollater`` (src/modelling/datasets.py:239-288).

``DeviceCollater(dataset_name, device)(samples)`` takes the list of per-video dicts that ``StltDataset.__getitem__``
produces (datasets.py:52-125: ``categories (T_i,N)``, ``boxes (T_i,N,4)``, ``scores (T_i,N)``, ``frame_types (T_i)``,
``lengths``, ``labels``, ``video_id``) and returns the collated batch ON THE GPU with the reference's keys.  The ragged
per-video tensors are concatenated once, copied once per field, and one HIP kernel writes the padded tensors and both
key-padding masks (instead of five ``pad_sequence`` loops plus seven host-to-device copies of padded data).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import _lib as L
from . import ops
from .synth import DATASETS


class DeviceCollater:
    def __init__(self, dataset_name: str = "something", device="cuda"):
        self.dataset_name = dataset_name
        self.cls_id = DATASETS[dataset_name]["cls"]  # category2id["cls"] (src/modelling/configs.py:40-78)
        self.device = torch.device(device)

    def __call__(self, samples: List[Dict[str, torch.Tensor]]) -> Dict[str, object]:
        lib = L.load()
        dev = self.device
        B = len(samples)
        lens = [int(s["categories"].shape[0]) for s in samples]
        T, N = max(lens), int(samples[0]["categories"].shape[1])
        off = torch.zeros(B + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64), 0)
        cat_r = torch.cat([s["categories"].to(torch.int64) for s in samples]).contiguous().to(dev, non_blocking=True)
        box_r = torch.cat([s["boxes"].to(torch.float32) for s in samples]).contiguous().to(dev, non_blocking=True)
        ft_r = torch.cat([s["frame_types"].to(torch.int64) for s in samples]).contiguous().to(dev, non_blocking=True)
        keep_scores = self.dataset_name == "action_genome"  # datasets.py:253-260
        sc_r = (torch.cat([s["scores"].to(torch.float32) for s in samples]).contiguous().to(dev, non_blocking=True)
                if keep_scores else None)
        off_d = off.to(dev, non_blocking=True)
        out = {
            "categories": torch.empty(B, T, N, dtype=torch.int64, device=dev),
            "boxes": torch.empty(B, T, N, 4, dtype=torch.float32, device=dev),
            "frame_types": torch.empty(B, T, dtype=torch.int64, device=dev),
            "src_key_padding_mask_boxes": torch.empty(B, T, N, dtype=torch.bool, device=dev),
            "src_key_padding_mask_frames": torch.empty(B, T, dtype=torch.bool, device=dev),
        }
        if keep_scores:
            out["scores"] = torch.empty(B, T, N, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.stlt_collate_fwd(cat_r.data_ptr(), box_r.data_ptr(), ops._p(sc_r), ft_r.data_ptr(), off_d.data_ptr(), B, T, N,
                                         self.cls_id, out["categories"].data_ptr(), out["boxes"].data_ptr(),
                                         ops._p(out.get("scores")), out["frame_types"].data_ptr(),
                                         out["src_key_padding_mask_boxes"].data_ptr(),
                                         out["src_key_padding_mask_frames"].data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "stlt_collate_fwd")
        out["lengths"] = torch.stack([torch.as_tensor(s["lengths"]) for s in samples]).to(torch.int64).to(dev)
        out["labels"] = torch.stack([torch.as_tensor(s["labels"]) for s in samples]).to(dev)
        out["video_id"] = [s.get("video_id") for s in samples]
        return out


def real_counts(batch, round_up_to: int = 1) -> dict:
    """{"num_real_tokens", "num_real_frames"} of a collated batch, counted from its masks (on whatever device they live: call it on the host
    side of the loader, where the masks are made — reference datasets.py:274-286): frames = zeros of src_key_padding_mask_frames, tokens =
    zeros of src_key_padding_mask_boxes inside those frames.  Added to the batch as host integers they let a skip-padding forward / training
    step run without reading its row counts back (include/stlt_hip.h: stlt_inputs.n_real_tokens / n_real_frames).  round_up_to > 1 rounds both
    up to a multiple (capped at the padded sizes): upper bounds, which the inference calls accept."""
    real = ~batch["src_key_padding_mask_frames"].bool()
    tokens = int(((~batch["src_key_padding_mask_boxes"].bool()) & real[:, :, None]).sum())
    frames = int(real.sum())
    if round_up_to > 1:  # inference only: upper bounds are enough, so one captured hipGraph (or one launch plan) serves a bucket of batches
        B, T, N = batch["src_key_padding_mask_boxes"].shape
        up = lambda v, cap: min(cap, (v + round_up_to - 1) // round_up_to * round_up_to)  # noqa: E731
        tokens, frames = up(tokens, B * T * N), up(frames, B * T)
    return {"num_real_tokens": tokens, "num_real_frames": frames}
