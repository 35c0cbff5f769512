"""3-D box helpers used by the matcher and the losses (batched, device-side).
Semantics of transoar/utils/bboxes.py:6-43 (generalized_bbox_iou_3d,
box_cxcyczwhd_to_xyzxyz) and :98-148 (iou_3d, bboxes_volume)."""
import torch


def box_cxcyczwhd_to_xyzxyz(b):
    c, s = b[..., :3], b[..., 3:]
    return torch.cat((c - 0.5 * s, c + 0.5 * s), dim=-1)


def _prod3(t):
    # explicit product of the three extents: Tensor.prod's backward checks for zeros on the
    # host (a device->host sync, and illegal inside a captured HIP graph)
    return t[..., 0] * t[..., 1] * t[..., 2]


def _volume(b):
    return _prod3(b[..., 3:] - b[..., :3])


def elementwise_giou_3d(a, b):
    """GIoU of box pairs, broadcasting over leading dims; boxes x1y1z1x2y2z2.
    Equals the reference's pairwise matrix entry for (a_i, b_j)."""
    inter = _prod3((torch.min(a[..., 3:], b[..., 3:]) - torch.max(a[..., :3], b[..., :3])).clamp(min=0))
    union = _volume(a) + _volume(b) - inter
    iou = inter / union
    hull = _prod3((torch.max(a[..., 3:], b[..., 3:]) - torch.min(a[..., :3], b[..., :3])).clamp(min=0))
    return iou - (hull - union) / hull


def generalized_bbox_iou_3d(bboxes1, bboxes2):
    """[N, M] pairwise GIoU (reference signature)."""
    assert (bboxes1[:, 3:] >= bboxes1[:, :3]).all()
    assert (bboxes2[:, 3:] >= bboxes2[:, :3]).all()
    return elementwise_giou_3d(bboxes1[:, None, :], bboxes2[None, :, :])
