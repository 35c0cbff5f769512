"""Set criterion: per-class matching, BCE on soft labels, L1 + GIoU on the
matched boxes, optional segmentation proxy losses.

Semantics of transoar/models/criterion.py:9-125 including:
  * the aux-loss quirk: for every intermediate decoder layer the
    MATCHING uses that layer's outputs, but the box / class losses are taken on
    the FINAL outputs (criterion.py:114-123);
  * num_boxes = number of target boxes in the (local) batch (:96).
Everything stays on the device; no .cpu()/.item() inside.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .bboxes import box_cxcyczwhd_to_xyzxyz, elementwise_giou_3d
from .matcher import DenseTargets


class SoftDiceLoss(nn.Module):
    """Soft Dice over softmax probabilities (criterion.py:127-166)."""

    def __init__(self, nonlin=None, batch_dice=False, do_bg=False, smooth_nom=1e-5, smooth_denom=1e-5):
        super().__init__()
        self.nonlin, self.batch_dice, self.do_bg = nonlin, batch_dice, do_bg
        self.smooth_nom, self.smooth_denom = smooth_nom, smooth_denom

    def forward(self, inp, target, loss_mask=None):
        axes = ([0] if self.batch_dice else []) + list(range(2, inp.dim()))
        if self.nonlin is not None:
            inp = self.nonlin(inp)
        with torch.no_grad():
            if target.dim() != inp.dim():
                target = target.unsqueeze(1)
            onehot = target if target.shape == inp.shape else torch.zeros_like(inp).scatter_(1, target.long(), 1)
        tp, fp, fn = inp * onehot, inp * (1 - onehot), (1 - inp) * onehot
        if loss_mask is not None:
            tp, fp, fn = (t * loss_mask[:, :1] for t in (tp, fp, fn))
        tp, fp, fn = tp.sum(axes), fp.sum(axes), fn.sum(axes)
        dc = (2 * tp + self.smooth_nom) / (2 * tp + fp + fn + self.smooth_denom)
        if not self.do_bg:
            dc = dc[1:] if self.batch_dice else dc[:, 1:]
        return 1 - dc.mean()


class TransoarCriterion(nn.Module):
    def __init__(self, num_classes, matcher, seg_proxy, seg_fg_bg):
        super().__init__()
        self.num_classes, self.matcher = num_classes, matcher
        self._seg_proxy, self._seg_fg_bg = seg_proxy, seg_fg_bg
        if seg_proxy:
            self._dice_loss = SoftDiceLoss(nonlin=nn.Softmax(dim=1), batch_dice=True, smooth_nom=1e-05,
                                           smooth_denom=1e-05, do_bg=False)

    def loss_class(self, outputs, soft_labels, n_valid=None):
        logits = outputs["pred_logits"].flatten().float()
        labels = soft_labels.flatten().to(logits.device)
        valid = labels != -1
        # mean over the valid entries only (criterion.py:46-49); n_valid overrides the
        # local count when the batch is sharded over data-parallel ranks
        per = F.binary_cross_entropy_with_logits(logits, labels.clamp(min=0), reduction="none")
        return (per * valid).sum() / (valid.sum() if n_valid is None else n_valid)

    def bbox_terms(self, outputs, targets):
        """Per (sample, class, query) L1 and 1-GIoU of the predicted boxes against the class target."""
        preds = outputs["pred_boxes"]
        n, n_q, _ = preds.shape
        qpo = n_q // self.num_classes
        preds = preds.reshape(n, self.num_classes, qpo, -1).float()
        tgt = targets.boxes[:, :, None, :]
        l1 = (preds - tgt).abs().sum(-1)
        giou = elementwise_giou_3d(box_cxcyczwhd_to_xyzxyz(preds.clamp(min=0)), box_cxcyczwhd_to_xyzxyz(tgt))
        return l1, 1 - giou

    def loss_bboxes(self, outputs, targets, matches, num_boxes, matches_per_class=1, terms=None):
        l1, one_minus_giou = self.bbox_terms(outputs, targets) if terms is None else terms
        hit = matches.bool()
        zero = torch.zeros_like(l1)
        # unmatched entries may hold 0-volume targets -> NaN giou; mask by select, not multiply
        denom = num_boxes * matches_per_class
        return torch.where(hit, l1, zero).sum() / denom, torch.where(hit, one_minus_giou, zero).sum() / denom

    def loss_segmentation(self, outputs, targets):
        if self._seg_fg_bg:
            targets = (targets > 0).to(targets.dtype)
        targets = targets.squeeze(1).long()
        return F.cross_entropy(outputs["pred_seg"], targets), self._dice_loss(outputs["pred_seg"], targets)

    def forward(self, outputs, targets, seg_targets, anchors):
        if not isinstance(targets, DenseTargets):
            targets = DenseTargets.from_list(targets, self.num_classes, outputs["pred_logits"].device)
        from . import fused_criterion
        if fused_criterion.usable(self, outputs, targets, seg_targets):
            # one forward and one backward launch (csrc/criterion.hip) instead of ~270 tiny ones; the code below is what it computes
            return fused_criterion.run(self, outputs, targets, anchors)
        num_boxes = targets.num_boxes
        n_valid = None
        if targets.n_present is not None:
            qpo = outputs["pred_logits"].shape[1] // self.num_classes
            n_valid = targets.n_present * qpo
        geo = self.matcher.geometry(outputs, targets, anchors)
        soft = geo[1]
        matches = self.matcher.assign(outputs["pred_logits"], geo)
        terms = self.bbox_terms(outputs, targets)      # of the FINAL outputs: the aux terms reuse them (sic, below)
        loss_bbox, loss_giou = self.loss_bboxes(outputs, targets, matches, num_boxes, terms=terms)
        zero = torch.zeros((), device=outputs["pred_logits"].device)
        loss_cls = self.loss_class(outputs, soft, n_valid)
        losses = {"bbox": loss_bbox, "giou": loss_giou, "cls": loss_cls, "segce": zero, "segdice": zero}
        if self._seg_proxy:
            losses["segce"], losses["segdice"] = self.loss_segmentation(outputs, seg_targets)
        shared = self.matcher.anchor_matching      # then geometry and soft labels do not depend on the output
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            geo_i = geo if shared else self.matcher.geometry(aux, targets, anchors)
            matches = self.matcher.assign(aux["pred_logits"], geo_i)
            lb, lg = self.loss_bboxes(outputs, targets, matches, num_boxes, terms=terms)      # sic: final outputs
            losses["bbox_%d" % i], losses["giou_%d" % i] = lb, lg
            # sic: the class loss of the FINAL logits against this output's soft labels (criterion.py:117-124)
            losses["cls_%d" % i] = loss_cls if shared else self.loss_class(outputs, geo_i[1], n_valid)
        return losses
