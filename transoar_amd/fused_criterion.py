"""The set criterion of the training step as ONE forward and ONE backward launch (include/transoar_criterion.h,
csrc/criterion.hip): matcher geometry, one assignment per decoder output, L1 / GIoU on the matched boxes and the BCE on
the soft labels -- transoar/models/criterion.py:9-125 with matcher.py:9-65 in its anchor-matching form.

transoar_amd/criterion.py (the torch mirror, ~270 launches of ~5 us per step with its autograd graph) stays the path for
everything this kernel does not cover -- CPU tensors, predicted-box matching, one query per class, the segmentation
proxy losses -- and is what the GPU tests compare this kernel with (tests/test_criterion_gpu.py), next to the goldens
generated from the reference (g7, g10, g11 run through here).  TRANSOAR_FUSED_CRITERION=0 switches it off.
"""
import ctypes
import os

import torch

from . import _native  # noqa: F401  (torch's HIP runtime first)

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtransoar_criterion.so")
if not os.path.exists(_LIB_PATH):
    raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
lib = ctypes.CDLL(_LIB_PATH)
_p, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
lib.transoar_set_criterion_forward.restype = _i
lib.transoar_set_criterion_forward.argtypes = [_p, _i, _i, _p, _i, _p, _p, _p, _p, _f, _p, _f, _f, _f, _i, _i, _i, _p, _p, _p, _p, _p, _p]
lib.transoar_set_criterion_backward.restype = _i
lib.transoar_set_criterion_backward.argtypes = [_p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p]
lib.transoar_criterion_abi_version.restype = _i
if lib.transoar_criterion_abi_version() != 1:
    raise _native.NativeLibraryError("%s: ABI version mismatch, rebuild" % _LIB_PATH)

ENABLED = os.environ.get("TRANSOAR_FUSED_CRITERION", "1") != "0"
MAX_LAYERS, MAX_R = 8, 64
_DT = {torch.float32: 0, torch.bfloat16: 2}


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class LossDict(dict):
    """The criterion's dict of losses, plus `vector`: the same values as ONE differentiable tensor in the dict's order, so
    that the weighted total is one dot product (TrainStep._weighted_total) instead of a stack of eleven selected scalars."""
    vector = None


def usable(criterion, outputs, targets, seg_targets):
    """Can the fused kernel take this call?  (Everything else runs the torch mirror in criterion.py.)"""
    logits, boxes = outputs.get("pred_logits"), outputs.get("pred_boxes")
    if not (ENABLED and torch.is_tensor(logits) and logits.is_cuda and torch.is_tensor(boxes) and boxes.is_cuda):
        return False
    m = criterion.matcher
    if criterion._seg_proxy or not m.anchor_matching or m.num_organs != criterion.num_classes:
        return False
    aux = outputs.get("aux_outputs", [])
    n, q = logits.shape[0], logits.shape[1]
    if logits.dim() != 3 or logits.shape[2] != 1 or boxes.shape != (n, q, 6) or q % criterion.num_classes:
        return False
    r = q // criterion.num_classes
    if not (2 <= r <= MAX_R and 1 + len(aux) <= MAX_LAYERS):
        return False
    if logits.dtype not in _DT or boxes.dtype not in _DT or not logits.is_contiguous() or not boxes.is_contiguous():
        return False
    for a in aux:
        al = a["pred_logits"]
        if al.shape != logits.shape or al.dtype != logits.dtype or not al.is_cuda or not al.is_contiguous():
            return False
    tb = targets.boxes
    return tb.is_cuda and tb.dtype == torch.float32 and tb.shape == (n, criterion.num_classes, 6)


class _SetCriterion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, boxes, aux_logits, anchors, tgt_boxes, present, num_boxes, n_present, costs, num_classes):
        n, q = logits.shape[0], logits.shape[1]
        r = q // num_classes
        layers = 1 + len(aux_logits)
        dev = logits.device
        ptrs = (ctypes.c_void_p * layers)(logits.data_ptr(), *[a.data_ptr() for a in aux_logits])
        anchors = anchors.contiguous().float()
        tgt_boxes = tgt_boxes.contiguous()
        present8 = present.contiguous().view(torch.uint8) if present.dtype == torch.bool else present.contiguous().to(torch.uint8)
        nb_dev = np_dev = None
        nb_host = 0.0
        if torch.is_tensor(num_boxes):
            nb_dev = num_boxes.to(device=dev, dtype=torch.float32)
        else:
            nb_host = float(num_boxes)
        if n_present is not None:
            np_dev = n_present.to(device=dev, dtype=torch.float32) if torch.is_tensor(n_present) else torch.tensor(float(n_present), device=dev)
        losses = torch.empty(5 + 3 * (layers - 1), dtype=torch.float32, device=dev)
        d_l1 = torch.empty(n, q, 6, dtype=torch.float32, device=dev)
        d_giou = torch.empty(n, q, 6, dtype=torch.float32, device=dev)
        d_cls = torch.empty(n, q, dtype=torch.float32, device=dev)
        hit = torch.empty(layers, n, q, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _check(lib.transoar_set_criterion_forward(
                ctypes.cast(ptrs, ctypes.c_void_p), layers, _DT[logits.dtype], boxes.data_ptr(), _DT[boxes.dtype], anchors.data_ptr(),
                tgt_boxes.data_ptr(), present8.data_ptr(), None if nb_dev is None else nb_dev.data_ptr(), nb_host,
                None if np_dev is None else np_dev.data_ptr(), float(costs[0]), float(costs[1]), float(costs[2]), n, num_classes, r,
                losses.data_ptr(), d_l1.data_ptr(), d_giou.data_ptr(), d_cls.data_ptr(), hit.data_ptr(), _stream()),
                "transoar_set_criterion_forward")
        ctx.save_for_backward(d_l1, d_giou, d_cls, hit)
        ctx.dims = (layers, n, num_classes, r, logits.dtype, boxes.dtype, tuple(logits.shape))
        return losses

    @staticmethod
    def backward(ctx, g):
        d_l1, d_giou, d_cls, hit = ctx.saved_tensors
        layers, n, o, r, ldt, bdt, lshape = ctx.dims
        g = g.contiguous().float()
        grad_boxes = torch.empty(n, o * r, 6, dtype=bdt, device=g.device)
        grad_logits = torch.empty(lshape, dtype=ldt, device=g.device)
        with torch.cuda.device(g.device):
            _check(lib.transoar_set_criterion_backward(g.data_ptr(), layers, d_l1.data_ptr(), d_giou.data_ptr(), d_cls.data_ptr(),
                                                       hit.data_ptr(), n, o, r, grad_boxes.data_ptr(), _DT[bdt], grad_logits.data_ptr(),
                                                       _DT[ldt], _stream()), "transoar_set_criterion_backward")
        return grad_logits, grad_boxes, None, None, None, None, None, None, None, None


def run(criterion, outputs, targets, anchors):
    """-> LossDict with the keys and the order of TransoarCriterion.forward (criterion.py); .vector holds them as one tensor."""
    aux = outputs.get("aux_outputs", [])
    m = criterion.matcher
    losses = _SetCriterion.apply(outputs["pred_logits"], outputs["pred_boxes"], tuple(a["pred_logits"] for a in aux), anchors,
                                    targets.boxes, targets.present, targets.num_boxes, targets.n_present,
                                    (m.cost_class, m.cost_bbox, m.cost_giou), criterion.num_classes)
    # the kernel writes the vector in the dict's order: bbox, giou, cls, segce, segdice, then (bbox_i, giou_i, cls_i) per auxiliary output
    keys = ["bbox", "giou", "cls", "segce", "segdice"]
    for i in range(len(aux)):
        keys += ["bbox_%d" % i, "giou_%d" % i, "cls_%d" % i]
    out = LossDict(zip(keys, losses.unbind(0)))
    out.vector = losses
    return out
