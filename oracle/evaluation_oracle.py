"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's metrics (src/utils/evaluation.py).

Pinned by tests/golden/evaluation.npz, captured from the reference's own `EvaluatorSomething` / `charades_map`
(tools/gen_golden_evaluation.py).  Plain Python / numpy loops, written for clarity at small sizes.
"""
import numpy as np


def topk_correct(logits: np.ndarray, labels: np.ndarray, k: int) -> int:
    """evaluation.py:24-34: count rows whose label is among the k largest logits."""
    hits = 0
    for row, y in zip(logits, labels):
        order = sorted(range(len(row)), key=lambda j: (-row[j], j))[:k]
        hits += int(y) in order
    return hits


def charades_map(scores: np.ndarray, truths: np.ndarray):
    """evaluation.py:100-132, float64."""
    scores = np.array(scores, dtype=np.float64)
    truths = np.asarray(truths, dtype=np.float64)
    scores[truths.sum(axis=1) == 0, :] = -np.inf
    aps = []
    for c in range(scores.shape[1]):
        order = sorted(range(scores.shape[0]), key=lambda i: (-scores[i, c], i))
        n_pos, seen_pos, total = int((truths[:, c] == 1).sum()), 0, 0.0
        if n_pos == 0:
            aps.append(float("nan"))
            continue
        for rank, i in enumerate(order, start=1):
            if truths[i, c] == 1:
                seen_pos += 1
                total += seen_pos / rank
        aps.append(total / n_pos)
    aps = np.array(aps)
    return float(np.mean(aps)), aps * truths.sum(axis=0) / truths.sum(), aps
