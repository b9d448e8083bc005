"""Return-type contracts shared by the models and the experiment harness.

Field names and order follow IGN/utils/shapelet_util.py:17-41 so that code written against the
reference's ``ModelInfo`` / ``ClassificationResult`` keeps working.  The plotting half of the
reference module (t-SNE / shapelet figures, :44-195) is out of scope.
"""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class ModelInfo:
    d: Optional[torch.Tensor] = None                # (B, G*K*C) min_t distance per shapelet
    p: Optional[torch.Tensor] = None                # (B, G*K*C) gate output per shapelet
    eta: Optional[torch.Tensor] = None              # (B, 1) gini gate (InterpGN only)
    shapelet_preds: Optional[torch.Tensor] = None   # (B, N) SBM expert logits
    dnn_preds: Optional[torch.Tensor] = None        # (B, N) deep expert logits
    preds: Optional[torch.Tensor] = None            # (B, N) mixture logits
    loss: Optional[torch.Tensor] = None             # (1,) model regulariser
    # extension (appended, default None: code written against the reference's seven fields is unaffected):
    t: Optional[torch.Tensor] = None                # (B, G*K*C) int32 window index of each shapelet's best match


@dataclass
class ClassificationResult:
    x_data: Optional[torch.Tensor] = None
    shapelets: Optional[list] = None
    trues: Optional[torch.Tensor] = None
    preds: Optional[torch.Tensor] = None
    shapelet_preds: Optional[torch.Tensor] = None
    dnn_preds: Optional[torch.Tensor] = None
    p: Optional[torch.Tensor] = None
    d: Optional[torch.Tensor] = None
    w: Optional[torch.Tensor] = None
    eta: Optional[torch.Tensor] = None
    loss: Optional[float] = None
    accuracy: Optional[float] = None
    # extension: where each shapelet matched each test series -- window index per (sample, shapelet) in p's column order,
    # and the (start, length) in samples it corresponds to.  The reference's visualize_shapelets re-derives this on the host
    # (IGN/utils/shapelet_util.py:153); the forward kernel already has it.
    t: Optional[torch.Tensor] = None
    match_start: Optional[torch.Tensor] = None      # (n, G*K*C) first sample of the best-matching window
    match_len: Optional[torch.Tensor] = None        # (G*K*C,)   shapelet length per column
