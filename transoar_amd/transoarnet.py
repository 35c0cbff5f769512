"""TransoarNet: AttnFPN backbone -> sine pos-enc -> Focused Decoder -> heads.

Mirrors transoar/models/transoarnet.py (TransoarNet :11-149, MLP :157-171) and
the builders of transoar/models/build.py: same config keys, same parameter
names (``_backbone``, ``_neck``, ``_cls_head``, ``_reg_head.layers``,
``_query_embed``, ``_seg_head``).  Anchors / offset restrictions are
non-persistent buffers (they follow ``.to(device)``; the reference pins them
with ``.cuda()`` in the constructor, transoarnet.py:27-28) so the state_dict
key set is unchanged.
"""
import itertools

import torch
import torch.nn.functional as F
from torch import nn

from . import shadow
from .backbone import AttnFPN
from .criterion import TransoarCriterion
from .focused_decoder import FocusedDecoder
from .matcher import Matcher
from .position_encoding import PositionEmbeddingLearned3D, PositionEmbeddingSine3D


def build_backbone(config):
    return AttnFPN(config)


def build_neck(config, bbox_props):
    return FocusedDecoder(d_model=config["hidden_dim"], nhead=config["nheads"],
                          num_decoder_layers=config["dec_layers"], dim_feedforward=config["dim_feedforward"],
                          dropout=config["dropout"], activation="relu", return_intermediate_dec=True,
                          bbox_props=bbox_props, config=config)


def build_pos_enc(config):
    if config["pos_encoding"] == "sine":
        return PositionEmbeddingSine3D(channels=config["hidden_dim"])
    if config["pos_encoding"] == "learned":
        return PositionEmbeddingLearned3D(channels=config["hidden_dim"])
    raise ValueError("Please select a implemented pos. encoding.")


def build_criterion(config):
    matcher = Matcher(cost_class=config["set_cost_class"], cost_bbox=config["set_cost_bbox"],
                      cost_giou=config["set_cost_giou"], anchor_matching=config["anchor_matching"],
                      num_organs=config["neck"]["num_organs"])
    return TransoarCriterion(num_classes=config["num_classes"], matcher=matcher,
                             seg_proxy=config["backbone"]["use_seg_proxy_loss"],
                             seg_fg_bg=config["backbone"]["fg_bg"])


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for layer in self.layers[:-1]:
            x = F.relu(shadow.linear(x, layer.weight, layer.bias))
        return shadow.linear(x, self.layers[-1].weight, self.layers[-1].bias)


def generate_anchors(neck_config, bbox_props):
    """One anchor box per query: the organ's median box placed at the centre of
    its attention volume, shifted by every combination of {+o, -o, 0} per axis
    with o = (attention volume size - median size)/3 (dynamic) or a fixed
    offset.  Also the per-query restriction of the predicted offset.
    (transoarnet.py:60-116)  -> anchors (Q,6) in [0,1], restrictions (Q,6)"""
    n_q, n_org = neck_config["num_queries"], neck_config["num_organs"]
    qpo = int(n_q / n_org)
    dynamic = neck_config["anchor_gen_dynamic_offset"]
    anchors, pos_limits = [], []
    for props in bbox_props.values():
        median = torch.tensor(props["median"], dtype=torch.float32)
        vol = torch.tensor(props["attn_area"], dtype=torch.float32)
        centre, extent = (vol[:3] + vol[3:]) / 2, vol[3:] - vol[:3]
        if dynamic:
            o = (extent - median[3:]) / 3
            per_axis = [torch.stack((o[a], -o[a], torch.zeros(()))) for a in range(3)]
        else:
            fixed = float(neck_config["anchor_gen_offset"])
            per_axis = [torch.tensor([0.0, fixed, -fixed])] * 3
        if qpo == 1:
            shifts = torch.zeros(1, 3)
        else:
            shifts = torch.tensor(list(itertools.product(*[ax.tolist() for ax in per_axis])), dtype=torch.float32)
            if qpo == 7:
                shifts = shifts[(shifts != 0).sum(-1) <= 1]
        anchors.append(torch.cat((shifts + centre, median[3:].expand(shifts.shape[0], 3)), dim=-1))
        pos_limits.append(shifts.max(dim=0).values)
    med = torch.tensor([p["median"] for p in bbox_props.values()], dtype=torch.float32)[:, 3:]
    lo = torch.tensor([p["min"] for p in bbox_props.values()], dtype=torch.float32)[:, 3:]
    hi = torch.tensor([p["max"] for p in bbox_props.values()], dtype=torch.float32)[:, 3:]
    size_limits = torch.max(med - lo, hi - med)
    restriction = torch.cat((torch.stack(pos_limits), size_limits), dim=-1).repeat_interleave(qpo, dim=0)
    return torch.cat(anchors).clamp(min=0, max=1), restriction


class TransoarNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        neck = config["neck"]
        hidden = neck["hidden_dim"]
        self._input_levels = neck["input_levels"]
        self._anchor_offset = neck["anchor_offset_pred"]
        self._aux_loss = neck["aux_loss"]

        self._backbone = build_backbone(config["backbone"])

        anchors, restrictions = generate_anchors(neck, config["bbox_properties"])
        if not neck["anchor_gen_dynamic_offset"]:
            restrictions = torch.full_like(restrictions, float(neck["max_anchor_pred_offset"]))
        restrictions[:, :3] /= 2
        self.register_buffer("_anchors", anchors, persistent=False)
        self.register_buffer("_restrictions", restrictions, persistent=False)

        self._neck = build_neck(neck, config["bbox_properties"])
        self._cls_head = nn.Linear(hidden, 1)
        self._reg_head = MLP(hidden, hidden, 6, 3)

        self._seg_proxy = config["backbone"]["use_seg_proxy_loss"]
        if self._seg_proxy:
            n_out = 2 if config["backbone"]["fg_bg"] else neck["num_organs"] + 1
            self._seg_head = nn.Conv3d(config["backbone"]["start_channels"], n_out, kernel_size=1, stride=1)

        self._query_embed = nn.Embedding(neck["num_queries"], hidden * 2)     # [query_pos | tgt]
        self._pos_enc = build_pos_enc(neck)
        if self._anchor_offset:
            # start exactly on the anchors with neutral scores (transoarnet.py:53-58)
            for t in (self._cls_head.weight, self._cls_head.bias,
                      self._reg_head.layers[-1].weight, self._reg_head.layers[-1].bias):
                nn.init.zeros_(t)

    def forward(self, x):
        feats = self._backbone(x)
        det_src = feats[self._input_levels]
        hs = self._neck(det_src, self._query_embed.weight, self._pos_enc(det_src))   # (layers, N, Q, C)
        logits = shadow.linear(hs, self._cls_head.weight, self._cls_head.bias)
        boxes = self._reg_head(hs)
        if self._anchor_offset:
            boxes = (boxes.tanh() * self._restrictions + self._anchors).clamp(min=0, max=1)
        else:
            boxes = boxes.sigmoid()
        out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1],
               "pred_seg": self._seg_head(feats["P0"]) if self._seg_proxy else 0}
        if self._aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b} for a, b in zip(logits[:-1], boxes[:-1])]
        return out
