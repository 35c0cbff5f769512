"""Experiment configuration for the hot path, as plain dicts with the
reference's keys (config/attn_fpn_foc_dec_{visceral,amos}.yaml merged with
config/{visceral,amos}.yaml and dataset/<name>/data_info.json by
transoar/utils/io.py:20-38).  A user's own YAML loaded with ``yaml.safe_load``
works as well: the model reads the same keys.

``data_info.json`` is produced by the reference's dataset preprocessing from
real CT data (data/preprocessor_*.py:114-157) and is not shipped, so
``synthetic_bbox_properties`` fabricates per-class box statistics with
non-empty attention volumes (an empty RoI would make a softmax row all -inf,
focused_decoder.py:243-247).
"""
import copy

import torch

_BACKBONE = dict(
    name="attn_fpn", use_encoder_attn=False,
    conv_kernels=[[3, 3, 3]] * 6, strides=[[1, 1, 1]] + [[2, 2, 2]] * 5,
    in_channels=1, start_channels=24,
    depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=[5, 5, 5], mlp_ratio=4, qkv_bias=True,
    qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2, conv_merging=False,
    use_decoder_attn=False, fpn_channels=384, out_fmaps=["P2"],
    pos_encoding="sine", feature_levels=["P2", "P3", "P4", "P5"], hidden_dim=384, dim_feedforward=1024,
    dropout=0.1, nheads=6, layers=2, n_points=4, use_cuda=False,
    use_seg_proxy_loss=False, fg_bg=True,
)
_NECK = dict(
    name="foc_attn", pos_encoding="sine", input_levels="P2", hidden_dim=384, dropout=0.1, nheads=8,
    dim_feedforward=1024, dec_layers=3, restrict_attn=True, obj_self_attn=False,
    anchor_gen_dynamic_offset=True, anchor_gen_offset=0.1, anchor_offset_pred=True,
    max_anchor_pred_offset=0.1, num_queries=540, num_organs=20, aux_loss=True,
)
_TOP = dict(
    lr=2e-4, lr_backbone=2e-5, weight_decay=1e-4, clip_max_norm=-1, lr_drop=2500, batch_size=2,
    anchor_matching=True, set_cost_class=1, set_cost_bbox=0, set_cost_giou=0,
    loss_coefs=dict(cls=2, bbox=5, giou=2, segce=2, segdice=2),
)


def visceral_config(refine=False, use_cuda=True, swin=False):
    """config/attn_fpn_foc_dec_visceral.yaml: 160x160x256 volumes, 20 organs,
    540 queries, neck on P2.  refine=True switches the deformable-attention
    refinement on (use_decoder_attn; shipped default is off, yaml:69); swin=True the Swin encoder stages
    (use_encoder_attn, yaml:48: BASELINE.json config #4)."""
    cfg = copy.deepcopy(_TOP)
    cfg.update(experiment_name="foc_dec_visceral", dataset="visceral_160_160_256_CT", num_classes=20,
               volume_shape=(160, 160, 256))
    cfg["backbone"] = copy.deepcopy(_BACKBONE)
    cfg["backbone"].update(use_decoder_attn=refine, use_cuda=use_cuda, use_encoder_attn=swin)
    cfg["neck"] = copy.deepcopy(_NECK)
    return cfg


def amos_config(refine=False, use_cuda=True):
    """config/attn_fpn_foc_dec_amos.yaml: 256x256x128 volumes, 15 organs, 405
    queries, neck on P3, refine levels P3..P5."""
    cfg = visceral_config(refine, use_cuda)
    cfg.update(experiment_name="foc_dec_amos", dataset="amos_256_256_128_CT", num_classes=15,
               volume_shape=(256, 256, 128))
    cfg["backbone"].update(out_fmaps=["P3"], feature_levels=["P3", "P4", "P5"])
    cfg["neck"].update(input_levels="P3", num_queries=405, num_organs=15)
    return cfg


def synthetic_bbox_properties(num_classes, seed=0):
    """Per class: median / min / max box (cx,cy,cz,w,h,d) and attention volume
    (x1,y1,z1,x2,y2,z2), all normalised to [0,1]."""
    g = torch.Generator().manual_seed(seed)
    props = {}
    for c in range(1, num_classes + 1):
        centre = 0.3 + 0.4 * torch.rand(3, generator=g)
        size = 0.10 + 0.10 * torch.rand(3, generator=g)
        roi_half = size / 2 + 0.05 + 0.05 * torch.rand(3, generator=g)
        lo, hi = (centre - roi_half).clamp(0, 1), (centre + roi_half).clamp(0, 1)
        props[str(c)] = {
            "median": torch.cat((centre, size)).tolist(),
            "min": torch.cat((centre - 0.03, size * 0.7)).tolist(),
            "max": torch.cat((centre + 0.03, size * 1.4)).tolist(),
            "attn_area": torch.cat((lo, hi)).tolist(),
        }
    return props


def synthetic_targets(batch, num_classes, seed=1, device="cpu"):
    """one box per class per sample, cxcycz~U(.3,.7),
    whd~U(.1,.2), labels 1..num_classes (sorted, as the reference's
    segmentation2bbox yields them)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(batch):
        boxes = torch.cat((0.3 + 0.4 * torch.rand(num_classes, 3, generator=g),
                           0.1 + 0.1 * torch.rand(num_classes, 3, generator=g)), -1)
        out.append({"boxes": boxes.to(device), "labels": torch.arange(1, num_classes + 1, device=device)})
    return out
