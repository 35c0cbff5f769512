"""TEST INFRASTRUCTURE ONLY -- torch-CPU restatement of the reference's
``use_cuda=False`` path.

``msda3d_core_torch`` restates ``ms_deform_attn_core_pytorch``
(transoar/models/ops/functions/ms_deform_attn_func.py:41-65): every level is
resampled with 5-D ``F.grid_sample`` (trilinear, zero padding,
``align_corners=False``) at ``2*loc-1`` and the L*P samples are mixed with the
attention weights.  It is differentiable (autograd), which makes it the
gradient oracle too, and it is what ``bench.py`` times as ``cpu_baseline``
(kind "port") on the GPU box's host cores.

grid_sample itself is PyTorch (reference pins torch==1.10.0,
requirements.txt:1; the 5-D trilinear/zeros/align_corners=False semantics are
unchanged in the torch 2.10 used here).
"""
import torch
import torch.nn.functional as F


def msda3d_core_torch(value, spatial_shapes, sampling_locations, attention_weights):
    """value (N,S,M,C); spatial_shapes (L,3) [D,H,W]; sampling_locations
    (N,Lq,M,L,P,3) xyz in [0,1]; attention_weights (N,Lq,M,L,P) -> (N,Lq,M*C)."""
    n, _, m, c = value.shape
    lq, n_lvl, n_pts = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    sizes = [tuple(int(v) for v in s) for s in spatial_shapes]
    # (N*M, Lq, L, P, 3) grid in [-1,1]; grid_sample reads it as (x,y,z)=(W,H,D)
    grid = (sampling_locations * 2.0 - 1.0).permute(0, 2, 1, 3, 4, 5).reshape(n * m, lq, n_lvl, n_pts, 3)
    per_head = value.permute(0, 2, 3, 1).reshape(n * m, c, -1)      # (N*M, C, S)
    sampled, start = [], 0
    for lvl, (d, h, w) in enumerate(sizes):
        vol = per_head[:, :, start:start + d * h * w].reshape(n * m, c, d, h, w)
        start += d * h * w
        # output (N*M, C, 1, Lq, P)
        s = F.grid_sample(vol, grid[:, None, :, lvl], mode="bilinear",
                          padding_mode="zeros", align_corners=False)
        sampled.append(s[:, :, 0])
    sampled = torch.stack(sampled, dim=3)                            # (N*M, C, Lq, L, P)
    mix = attention_weights.permute(0, 2, 1, 3, 4).reshape(n * m, 1, lq, n_lvl, n_pts)
    out = (sampled * mix).sum(dim=(3, 4))                            # (N*M, C, Lq)
    return out.reshape(n, m * c, lq).permute(0, 2, 1).contiguous()
