"""3-D conv FPN backbone (AttnFPN): six conv stages down, lateral / transposed
conv / output convs up, optional deformable-attention refinement.

Mirrors transoar/models/backbones/attn_fpn.py (AttnFPN :18-32, Decoder :34-145,
Encoder :148-213) and EncoderCnnBlock (encoder_blocks.py:14-54): same config
keys, same parameter names (``_encoder._stages.N._block.K``, ``_decoder._lateral
/_up/_out/_refine``) so reference checkpoints load with strict=True.
The Swin encoder (``use_encoder_attn``) is outside this build's scope
 and raises.
"""
import os

import torch
from torch import nn

from . import instnorm
from .conv3d import Conv3dK3, to_ncdhw
from .position_encoding import PositionEmbeddingLearned3D, PositionEmbeddingSine3D
from .swin_encoder import ConvPatchMerging, EncoderSwinBlock, PatchMerging
from .refine_block import DecoderDefAttnBlock
from .token_linear import token_linear


class EncoderCnnBlock(nn.Module):
    """[Conv3d(k, stride, pad, no bias) -> InstanceNorm3d(affine) -> ReLU] x 2;
    only the first conv strides."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=1, bias=False,
                 affine=True, eps=1e-05):
        super().__init__()
        kernel_size, stride = tuple(kernel_size), tuple(stride)
        # same Sequential layout as the reference (checkpoint keys _block.{0,1,3,4}.*); the
        # convolutions are Conv3dK3 (nn.Conv3d subclasses with a hand-written bf16 GPU path)
        self._block = nn.Sequential(
            Conv3dK3(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias),
            nn.InstanceNorm3d(out_channels, affine=affine, eps=eps),
            nn.ReLU(inplace=True),
            Conv3dK3(out_channels, out_channels, kernel_size, stride=1, padding=padding, bias=bias),
            nn.InstanceNorm3d(out_channels, affine=affine, eps=eps),
            nn.ReLU(inplace=True),
        )
        self._affine = affine

    fused_norm = True      # class switch (A/B against MIOpen's batch-norm kernels)

    def _norm_relu(self, x, norm, part=None):
        if (EncoderCnnBlock.fused_norm and self._affine and x.dtype == torch.bfloat16
                and instnorm.supported(x, norm.num_features)):
            return instnorm.instance_norm_relu(x, norm.weight, norm.bias, norm.eps, relu=True, part=part)
        return torch.relu_(norm(x))

    def _conv_norm_relu(self, conv, norm, x):
        # the statistics of the InstanceNorm come out of the convolution's epilogue where its kernel offers them (the
        # full-resolution layers of stage 0: one pass over the 629-MB map less per layer)
        if (EncoderCnnBlock.fused_norm and self._affine and isinstance(conv, Conv3dK3) and norm.num_features <= 32):
            y, part = conv.forward_with_stats(x)
            if part is not None and not (y.dtype == torch.bfloat16 and instnorm.supported(y, norm.num_features)):
                part = None
            return self._norm_relu(y, norm, part)
        return self._norm_relu(conv(x), norm)

    def forward(self, x):
        if not x.is_cuda:
            return self._block(x)
        blk = self._block
        x = self._conv_norm_relu(blk[0], blk[1], x)
        return self._conv_norm_relu(blk[3], blk[4], x)


class Encoder(nn.Module):
    def __init__(self, config, debug=False):
        super().__init__()
        self._debug = debug
        self._stages = nn.ModuleList()
        cin, cout = config["in_channels"], config["start_channels"]
        swin = bool(config["use_encoder_attn"])
        depths = list(config["depths"])
        # stochastic-depth rate grows linearly over all Swin blocks (attn_fpn.py:160-162)
        dpr = [float(v) for v in torch.linspace(0, config["drop_path_rate"], sum(depths))]
        merge = ConvPatchMerging if config["conv_merging"] else PatchMerging
        for stage_id, (kernel, stride) in enumerate(zip(config["conv_kernels"], config["strides"])):
            if swin and stage_id > 1:         # stages 0-1 are the convolutional patch embedding (attn_fpn.py:172)
                k = stage_id - 2
                stage = EncoderSwinBlock(
                    dim=cin, depth=depths[k], num_heads=config["num_heads"][k], window_size=config["window_size"],
                    mlp_ratio=config["mlp_ratio"], qkv_bias=config["qkv_bias"], qk_scale=config["qk_scale"],
                    drop=config["drop_rate"], attn_drop=config["attn_drop_rate"],
                    drop_path=dpr[sum(depths[:k]):sum(depths[:k + 1])], downsample=merge)
            else:
                stage = EncoderCnnBlock(cin, cout, kernel, stride)
            self._stages.append(stage)
            cin, cout = cout, cout * 2

    def forward(self, x):
        outputs = {}
        for i, stage in enumerate(self._stages):
            x = stage(x)
            outputs["C%d" % i] = x
        if self._debug:
            print("AttnFPN encoder shapes:", {k: list(v.shape) for k, v in outputs.items()})
            self._debug = False
        return outputs


def _gemm_conv_ok(f):
    """channels-last bf16 map on the GPU under bf16 autocast: its voxels are the rows of a token matrix (a view)"""
    return (f.is_cuda and f.dtype == torch.bfloat16 and f.dim() == 5 and f.shape[1] % 8 == 0
            and f.is_contiguous(memory_format=torch.channels_last_3d)
            and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16)


def _conv1_as_gemm(conv, f):
    """Conv3d(kernel 1) of attn_fpn.py:55-63 as a token GEMM on csrc/gemm.hip: rows = voxels, (Cout, Cin) weights."""
    n, c, d, h, w = f.shape
    tok = f.permute(0, 2, 3, 4, 1).reshape(n, d * h * w, c)
    y = token_linear(tok, conv.weight.view(conv.out_channels, c), conv.bias, force_hip=True, min_tokens=1024)
    return y.view(n, d, h, w, conv.out_channels).permute(0, 4, 1, 2, 3)


def _is_up2(up):
    return (tuple(up.kernel_size) == (2, 2, 2) and tuple(up.stride) == (2, 2, 2) and tuple(up.padding) == (0, 0, 0)
            and tuple(up.output_padding) == (0, 0, 0) and tuple(up.dilation) == (1, 1, 1) and up.groups == 1)


def _up2_as_gemm(up, f):
    """ConvTranspose3d(kernel = stride = 2) of attn_fpn.py:75-83: every input voxel writes its own 2x2x2 output block,
    so it is one GEMM voxels x (8 Cout, Cin) followed by a pixel shuffle."""
    n, c, d, h, w = f.shape
    co = up.out_channels
    tok = f.permute(0, 2, 3, 4, 1).reshape(n, d * h * w, c)
    w8 = up.weight.permute(2, 3, 4, 1, 0).reshape(8 * co, c)                   # rows ((kd, kh, kw), cout)
    b8 = None if up.bias is None else up.bias.repeat(8)
    y = token_linear(tok, w8, b8, force_hip=True, min_tokens=1024)              # (n, dhw, 8 co)
    y = y.view(n, d, h, w, 2, 2, 2, co).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(n, 2 * d, 2 * h, 2 * w, co)
    return y.permute(0, 4, 1, 2, 3)


class Decoder(nn.Module):
    # the FPN's 1x1x1 lateral and k = s = 2 transposed convolutions as token GEMMs on the hand-written kernel
    gemm_pointwise = os.environ.get("TRANSOAR_FPN_STOCK_POINTWISE") is None

    def __init__(self, config, debug=False):
        super().__init__()
        self._debug = debug
        n_stages = len(config["conv_kernels"])
        self._num_stages = n_stages
        self._refine_fmaps = config["use_decoder_attn"]
        self._refine_feature_levels = list(config["feature_levels"])
        self._seg_proxy = config["use_seg_proxy_loss"]
        fpn = int(config["fpn_channels"])
        enc_channels = [config["start_channels"] * 2 ** s for s in range(n_stages)]

        wanted = list(config["out_fmaps"]) + (list(config["feature_levels"]) if self._refine_fmaps else [])
        required = {int(name[-1]) for name in wanted}
        if self._seg_proxy:
            required.add(0)
        self._required_stages = sorted(required)
        first = 0 if self._seg_proxy else self._required_stages[0]
        self._first_stage = first

        lateral_in = enc_channels[first:]
        lateral_out = [min(c, fpn) for c in lateral_in]
        self._lateral = nn.ModuleList(nn.Conv3d(i, o, kernel_size=1) for i, o in zip(lateral_in, lateral_out))
        self._lateral_levels = len(self._lateral)

        self._out = nn.ModuleList()
        for n, stage in enumerate(self._required_stages):
            cout = enc_channels[0] if (self._seg_proxy and n == 0) else fpn
            # nn.Conv3d's parameters and state-dict keys; bf16 GPU path on the implicit-GEMM kernels (conv_gemm.hip)
            self._out.append(Conv3dK3(lateral_out[stage - first], cout, kernel_size=3, padding=1))

        # top-down path, coarsest first
        self._up = nn.ModuleList()
        coarse_to_fine = lateral_out[::-1]
        strides = [tuple(s) for s in config["strides"]][::-1]
        for lvl in range(len(coarse_to_fine) - 1):
            self._up.append(nn.ConvTranspose3d(coarse_to_fine[lvl], coarse_to_fine[lvl + 1],
                                               kernel_size=strides[lvl], stride=strides[lvl]))

        if self._refine_fmaps:
            if config["pos_encoding"] == "sine":
                self._pos_enc = PositionEmbeddingSine3D(channels=config["hidden_dim"])
            elif config["pos_encoding"] == "learned":
                self._pos_enc = PositionEmbeddingLearned3D(channels=config["hidden_dim"])
            else:
                raise ValueError("Please select a implemented pos. encoding.")
            self._refine = DecoderDefAttnBlock(
                d_model=config["hidden_dim"], nhead=config["nheads"], num_layers=config["layers"],
                dim_feedforward=config["dim_feedforward"], dropout=config["dropout"],
                feature_levels=config["feature_levels"], n_points=config["n_points"],
                use_cuda=config["use_cuda"])

    def forward(self, x):
        # the encoder hands over channels-last (NDHWC) bf16 maps on the GPU; everything from here on is a
        # torch/MIOpen convolution.  Conv3dK3.ndhwc_everywhere (default): keep channels-last all the way -- MIOpen's
        # CK solvers are NDHWC natively and miopen_db/ has entries for these keys; otherwise convert to NCDHW first
        feats = list(x.values())[-self._lateral_levels:]
        if not Conv3dK3.ndhwc_everywhere:
            feats = [to_ncdhw(f) for f in feats]
        as_gemm = Decoder.gemm_pointwise and all(_gemm_conv_ok(f) for f in feats)
        laterals = [(_conv1_as_gemm(conv, f) if as_gemm else conv(f)) for conv, f in zip(self._lateral, feats)]
        # merged[s - first] = lateral_s + up(merged_{s+1})
        merged = [None] * self._lateral_levels
        carry = None
        for k, lat in enumerate(reversed(laterals)):
            cur = lat if carry is None else lat + carry
            merged[self._lateral_levels - 1 - k] = cur
            if k < self._lateral_levels - 1:
                up = self._up[k]
                carry = _up2_as_gemm(up, cur) if (as_gemm and _is_up2(up)) else up(cur)
        outputs = {"P%d" % s: self._out[n](merged[s - self._first_stage])
                   for n, s in enumerate(self._required_stages)}

        if self._refine_fmaps:
            fmaps = [outputs[name] for name in self._refine_feature_levels]
            refined = self._refine(fmaps, [self._pos_enc(f) for f in fmaps])
            outputs.update(zip(self._refine_feature_levels, refined))
        if self._debug:
            print("AttnFPN decoder shapes:", {k: list(v.shape) for k, v in outputs.items()})
            self._debug = False
        return outputs


class AttnFPN(nn.Module):
    def __init__(self, fpn_config, debug=False):
        super().__init__()
        self._encoder = Encoder(fpn_config, debug)
        self._decoder = Decoder(fpn_config, debug)

    def forward(self, src):
        return self._decoder(self._encoder(src))

    def init_weights(self):
        pass
