"""The evaluation metrics step next to the path.

* `language_metrics`: word accuracy and perplexity of the teacher-forced pass
  (misc/Crit/crit_lang.py:75-103), from the fused device scoring (engine.score_teacher_forced).
* `concept_metrics`: F1@k and mAP of the concept probabilities against multi-hot labels
  (misc/Crit/crit_attribute.py:58-89), evaluated on the (all-gathered) `preds_attr`.
COCO text metrics (BLEU/METEOR/ROUGE/CIDEr) need the reference's Java tools and stay external.
"""
import math
from typing import Dict

import torch

from .constants import PAD

TOPK_LIST = (5, 10, 20, 30, 40, 50)  # crit_attribute.py:20


def language_metrics(logp: torch.Tensor, pred: torch.Tensor, labels: torch.Tensor) -> Dict[str, float]:
    labels = labels.to(pred.device)
    mask = labels.ne(PAD)
    n = float(mask.sum())
    acc = float(((pred.long() == labels) & mask).sum()) / n
    ce = float(-(logp * mask).sum()) / n
    return {"Word Acc0": acc, "Perplexity": math.exp(ce), "n_words": n}


def concept_metrics(preds_attr: torch.Tensor, labels_attr: torch.Tensor, calculate_mAP: bool = True) -> Dict[str, float]:
    preds = torch.clamp(preds_attr.float(), 0.01, 0.99)
    labels = labels_attr[:, : preds.shape[1]].to(preds.device).float()
    out = {}
    _, cand = preds.topk(max(TOPK_LIST), dim=1, sorted=True, largest=True)
    n_pos = labels.sum(1)
    for k in TOPK_LIST:
        hit = labels.gather(1, cand[:, :k]).sum(1)
        hit[hit.eq(0)] = 1e-3
        precision, recall = hit / k, hit / n_pos
        out["F1-%02d" % k] = float((2 * precision * recall / (precision + recall)).mean())
    if calculate_mAP:
        _, idx = preds.sort(dim=1, descending=True)
        _, rank = idx.sort(dim=1)
        aps = []
        for i in range(labels.shape[0]):
            pos = labels[i].nonzero().squeeze(1)
            hit_rank, _ = rank[i][pos].sort()
            ids = torch.arange(len(pos), device=pos.device)
            aps.append(float(((ids + 1).float() / (hit_rank + 1)).mean()))
        out["mAP"] = sum(aps) / len(aps)
    return out
