"""Per-class anchor/query matcher, batched on the device.

Semantics of transoar/models/matcher.py:9-65: for every (sample, organ class)
that has a ground-truth box, cost = cost_bbox*L1 + cost_class*(-sigmoid(logit))
+ cost_giou*(-GIoU) over that class's queries (their ANCHORS when
anchor_matching, else their predicted boxes); the cheapest query is the match;
soft labels are the min-max normalised GIoU of the class's queries against the
target; classes without a target get soft label -1 and no match.

The reference walks samples x classes in Python on the CPU (two .cpu() syncs per
call, three calls per step).  This version is one batched pass on the device
with no host synchronisation.  ``targets`` may be the reference's list of dicts
({'boxes','labels'}) or a DenseTargets built once per step.
"""
import torch
from torch import nn

from .bboxes import box_cxcyczwhd_to_xyzxyz, elementwise_giou_3d


class DenseTargets:
    """boxes (N, num_organs, 6) and present (N, num_organs) bool, class c at
    index c-1; num_boxes = total number of target boxes (python int, or a
    device scalar when summed over data-parallel ranks); n_present = number of
    (sample, class) slots with a target, None = count locally."""

    def __init__(self, boxes, present, num_boxes, n_present=None):
        self.boxes, self.present, self.num_boxes, self.n_present = boxes, present, num_boxes, n_present

    @staticmethod
    def from_list(targets, num_organs, device=None):
        device = device or targets[0]["boxes"].device
        n = len(targets)
        boxes = torch.zeros(n, num_organs, 6, device=device)
        present = torch.zeros(n, num_organs, dtype=torch.bool, device=device)
        num = 0
        for i, t in enumerate(targets):
            lab = t["labels"].to(device=device, dtype=torch.long) - 1
            boxes[i, lab] = t["boxes"].to(device=device, dtype=torch.float32)
            present[i, lab] = True
            num += int(t["labels"].shape[0])
        return DenseTargets(boxes, present, num)


class Matcher(nn.Module):
    def __init__(self, cost_class=1, cost_bbox=1, cost_giou=1, anchor_matching=True, num_organs=None):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs can't be 0"
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.anchor_matching = anchor_matching
        self.num_organs = num_organs

    @torch.no_grad()
    def forward(self, outputs, targets, anchors, num_top_queries=1):
        """-> matches (N, organs, qpo) long 0/1, soft_labels (N, organs, qpo) float."""
        logits = outputs["pred_logits"]
        if not isinstance(targets, DenseTargets):
            targets = DenseTargets.from_list(targets, self.num_organs, logits.device)
        geo = self.geometry(outputs, targets, anchors)
        return self.assign(logits, geo, num_top_queries), geo[1]

    def geometry(self, outputs, targets, anchors):
        """The box side of the matching: (cost_giou, soft_labels, boxes, tgt, present).  With
        anchor matching it depends on the anchors and the targets only -- the criterion computes it
        once and reuses it for the auxiliary outputs."""
        logits = outputs["pred_logits"]
        n, n_q, _ = logits.shape
        qpo = n_q // self.num_organs
        if self.anchor_matching:
            boxes = anchors[None].expand(n, -1, -1)
        else:
            boxes = outputs["pred_boxes"]
        boxes = boxes.reshape(n, self.num_organs, qpo, -1).float()
        tgt = targets.boxes[:, :, None, :]                                       # (N, organs, 1, 6)
        present = targets.present[:, :, None]
        cost_giou = -elementwise_giou_3d(box_cxcyczwhd_to_xyzxyz(boxes.clamp(min=0)),
                                         box_cxcyczwhd_to_xyzxyz(tgt))
        if qpo == 1:     # the reference's TypeError branch (matcher.py:59-61)
            soft = torch.ones_like(cost_giou)
        else:
            hi = cost_giou.max(-1, keepdim=True).values
            lo = cost_giou.min(-1, keepdim=True).values
            soft = ((cost_giou - hi) / (lo - hi)).clamp(min=0)
        soft = torch.where(present, soft, torch.full_like(soft, -1.0))
        return cost_giou, soft, boxes, tgt, present

    def assign(self, logits, geo, num_top_queries=1):
        """Top-k queries per organ under cost = class + giou (+ bbox) terms -> matches (N, organs, qpo)."""
        cost_giou, _, boxes, tgt, present = geo
        n, n_q, _ = logits.shape
        qpo = n_q // self.num_organs
        probs = logits.reshape(n, self.num_organs, qpo).float().sigmoid()
        cost = self.cost_class * (-probs) + self.cost_giou * cost_giou
        if self.cost_bbox != 0:
            cost = cost + self.cost_bbox * (boxes - tgt).abs().sum(-1)
        best = cost.topk(num_top_queries, dim=-1, largest=False).indices
        matches = torch.zeros(n, self.num_organs, qpo, dtype=torch.long, device=logits.device)
        matches.scatter_(-1, best, 1)
        return matches * present
