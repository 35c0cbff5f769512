"""3-D box helpers used by the matcher and the losses (batched, device-side).
Semantics of transoar/utils/bboxes.py:6-43 (generalized_bbox_iou_3d,
box_cxcyczwhd_to_xyzxyz) and :98-148 (iou_3d, bboxes_volume)."""
import torch


# (split / unbind instead of b[..., :3] / t[..., 0]: the same values, but autograd's backward of a split is ONE cat where every
# slice and select costs a zero-fill, a copy and an accumulating add -- 25 tiny launches per step in the box losses, round 6)
def box_cxcyczwhd_to_xyzxyz(b):
    c, s = b.split(3, dim=-1)
    return torch.cat((c - 0.5 * s, c + 0.5 * s), dim=-1)


def _prod3(t):
    # explicit product of the three extents: Tensor.prod's backward checks for zeros on the
    # host (a device->host sync, and illegal inside a captured HIP graph)
    x, y, z = t.unbind(-1)
    return x * y * z


def _volume(b):
    lo, hi = b.split(3, dim=-1)
    return _prod3(hi - lo)


def elementwise_giou_3d(a, b):
    """GIoU of box pairs, broadcasting over leading dims; boxes x1y1z1x2y2z2.
    Equals the reference's pairwise matrix entry for (a_i, b_j)."""
    a_lo, a_hi = a.split(3, dim=-1)
    b_lo, b_hi = b.split(3, dim=-1)
    inter = _prod3((torch.min(a_hi, b_hi) - torch.max(a_lo, b_lo)).clamp(min=0))
    union = _prod3(a_hi - a_lo) + _prod3(b_hi - b_lo) - inter
    iou = inter / union
    hull = _prod3((torch.max(a_hi, b_hi) - torch.min(a_lo, b_lo)).clamp(min=0))
    return iou - (hull - union) / hull


def generalized_bbox_iou_3d(bboxes1, bboxes2):
    """[N, M] pairwise GIoU (reference signature)."""
    assert (bboxes1[:, 3:] >= bboxes1[:, :3]).all()
    assert (bboxes2[:, 3:] >= bboxes2[:, :3]).all()
    return elementwise_giou_3d(bboxes1[:, None, :], bboxes2[None, :, :])
