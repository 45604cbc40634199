"""CPU oracle of the collater.  TEST INFRASTRUCTURE ONLY (see oracle/stlt_oracle.py header).

Restates ``StltCollater.__call__`` (reference src/modelling/datasets.py:243-288) and ``pad_sequence``
(src/utils/data_utils.py:93-102) with plain tensor ops.  Pinned by tests/golden/collate_*.npz, captured from the
reference's own collater by tools/gen_golden_collate.py.
"""
from typing import Dict, List

import torch

CLS = {"something": 3, "action_genome": 1}  # category2id["cls"], src/modelling/configs.py:40-78


def collate(samples: List[Dict[str, torch.Tensor]], dataset_name: str) -> Dict[str, torch.Tensor]:
    B = len(samples)
    T = max(int(s["categories"].shape[0]) for s in samples)
    N = int(samples[0]["categories"].shape[1])
    cat = torch.zeros(B, T, N, dtype=torch.int64)
    cat[:, :, 0] = CLS[dataset_name]                      # pad_categories_tensor, datasets.py:247-251
    box = torch.zeros(B, T, N, 4)
    box[:, :, 0] = torch.tensor([0.0, 0.0, 1.0, 1.0])     # pad_boxes_tensor, :262-264
    sc = torch.zeros(B, T, N)
    sc[:, :, 0] = 1.0                                     # pad_scores_tensor, :254-258
    ft = torch.zeros(B, T, dtype=torch.int64)             # frame2type["pad"] = 0, :266-269
    for i, s in enumerate(samples):
        n = s["categories"].shape[0]
        cat[i, :n] = s["categories"]; box[i, :n] = s["boxes"]; sc[i, :n] = s["scores"]; ft[i, :n] = s["frame_types"]
    out = {"categories": cat, "boxes": box, "frame_types": ft,
           "lengths": torch.stack([torch.as_tensor(s["lengths"]) for s in samples]),
           "labels": torch.stack([torch.as_tensor(s["labels"]) for s in samples]),
           "src_key_padding_mask_boxes": cat == 0,        # :274-278
           "src_key_padding_mask_frames": ft == 0}        # :280-286
    if dataset_name == "action_genome":                   # :253-260
        out["scores"] = sc
    return out
