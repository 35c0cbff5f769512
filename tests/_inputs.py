"""Seeded / closed-form input generators shared by the CPU and GPU tests.
Nothing here touches /root/reference (it does not exist on the GPU box)."""
import numpy as np
import torch


def level_starts(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def rand_inputs(seed, N, M, C, Lq, L, P, shapes, dtype, loc_lo=0.0, loc_hi=1.0):
    """The distributions of the reference's op test (ops/test.py:70-73) with a
    fixed seed: value = U*0.01, loc = U[lo,hi), attn normalised over (L,P)."""
    g = torch.Generator().manual_seed(seed)
    S = int(shapes.prod(1).sum())
    value = (torch.rand(N, S, M, C, generator=g, dtype=torch.float64) * 0.01).to(dtype)
    loc = (torch.rand(N, Lq, M, L, P, 3, generator=g, dtype=torch.float64) * (loc_hi - loc_lo) + loc_lo).to(dtype)
    attn = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    return value, loc, attn


def medium_inputs(dtype=torch.float64):
    """RNG-free inputs of the g3 fixture (ops/test.py "Medium" shape, :28-31);
    must stay in sync with tests/golden/make_golden.py:medium_inputs."""
    N, M, C, Lq, L, P = 1, 16, 16, 4860, 3, 4
    shapes = torch.as_tensor([(8, 15, 39), (4, 4, 10), (2, 2, 5)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    i = torch.arange(N * S * M * C, dtype=torch.float64).reshape(N, S, M, C)
    value = 0.01 * (0.5 + 0.5 * torch.sin(0.37 * i + 0.11 * torch.sqrt(i + 1.0)))
    j = torch.arange(N * Lq * M * L * P * 3, dtype=torch.float64).reshape(N, Lq, M, L, P, 3)
    loc = -0.1 + 1.2 * (0.5 + 0.5 * torch.cos(1.93 * j + 0.007 * j ** 1.5 / (1.0 + 1e-4 * j)))
    k = torch.arange(N * Lq * M * L * P, dtype=torch.float64).reshape(N, Lq, M, L, P)
    attn = 1.0 + torch.sin(0.77 * k) ** 2
    attn = attn / attn.sum((-1, -2), keepdim=True)
    return value.to(dtype), shapes, loc.to(dtype), attn.to(dtype)


VISCERAL_LEVELS = [(40, 40, 64), (20, 20, 32), (10, 10, 16), (5, 5, 8)]   # P2..P5 of 160x160x256
AMOS_LEVELS = [(32, 32, 16), (16, 16, 8), (8, 8, 4)]                     # P3..P5 of 256x256x128


def reference_points(shapes):
    """Voxel centres of every level, (1, S, 3) xyz in [0,1]
    (decoder_blocks.py:107-131 semantics)."""
    pts = []
    for D, H, W in shapes.tolist():
        z, y, x = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
        pts.append(torch.stack(((x.reshape(-1) + 0.5) / W, (y.reshape(-1) + 0.5) / H,
                                (z.reshape(-1) + 0.5) / D), -1))
    return torch.cat(pts, 0)[None].float()


def model_like_inputs(seed, N, levels, M=6, C=64, P=4, dtype=torch.float32, jitter=0.3,
                      device="cpu"):
    """Sampling pattern the refine block produces (self-attention over the
    pyramid: Lq = S, reference points = voxel centres, offsets = k voxels along
    the head's axis, ms_deform_attn.py:67-82) plus a jitter so trilinear weights
    are generic.  value ~ N(0,1), attn = softmax of random logits."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(levels, dtype=torch.long)
    S = int(shapes.prod(1).sum())
    L = len(levels)
    ref = reference_points(shapes)                                   # (1,S,3)
    dirs = torch.tensor([(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0)],
                        dtype=torch.float32)[:M]
    step = torch.arange(1, P + 1, dtype=torch.float32)
    off = dirs[:, None, None, :] * step[None, None, :, None]         # (M,1,P,3)
    off = off.expand(M, L, P, 3) + jitter * (torch.rand(N, S, M, L, P, 3, generator=g) - 0.5) * 2
    whd = shapes.flip(-1).float()
    loc = ref[:, :, None, None, None, :] + off / whd[None, None, None, :, None, :]
    value = torch.randn(N, S, M, C, generator=g)
    attn = torch.softmax(torch.randn(N, S, M, L * P, generator=g), -1).view(N, S, M, L, P)
    return (value.to(dtype).to(device), shapes.to(device), level_starts(shapes).to(device),
            loc.to(dtype).to(device).contiguous(), attn.to(dtype).to(device))


def load_case(npz, prefix):
    keys = ["value", "shapes", "lsi", "loc", "attn", "out", "grad_out", "grad_value", "grad_loc", "grad_attn"]
    return {k: torch.from_numpy(np.asarray(npz[prefix + "." + k])) for k in keys}


def case_prefixes(npz):
    return sorted({k.split(".")[0] for k in npz.files})


# ---------------------------------------------------------------------------
# model-level fixtures: deterministic weights + analytic volume, so the golden
# files only need to hold outputs
# ---------------------------------------------------------------------------
def fill_deterministic(module, gain=1.0):
    """Overwrite every parameter of `module` with a closed-form pattern that
    depends only on the parameter's position in state_dict order and its shape.
    Used identically on the reference model (make_golden.py) and on ours."""
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            n = p.numel()
            idx = torch.arange(n, dtype=torch.float64)
            fan_in = p[0].numel() if p.dim() > 1 else 1
            scale = gain * (1.0 / max(fan_in, 1)) ** 0.5 if p.dim() > 1 else 0.1
            vals = scale * torch.sin(idx * (0.37 + 0.01 * (k % 7)) + 0.5 * k)
            if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm3.weight") \
                    or (p.dim() == 1 and ("_block.1.weight" in name or "_block.4.weight" in name)):
                vals = 1.0 + 0.1 * vals          # norm gains around 1
            p.copy_(vals.reshape(p.shape).to(p.dtype))


def analytic_volume(shape, batch=1):
    """x = 0.5 + 0.5 sin(.05 i) cos(.07 j) sin(.03 k + 1) (+0.1*b) on (b,1,i,j,k)."""
    i = torch.arange(shape[0], dtype=torch.float32)[:, None, None]
    j = torch.arange(shape[1], dtype=torch.float32)[None, :, None]
    k = torch.arange(shape[2], dtype=torch.float32)[None, None, :]
    vol = 0.5 + 0.5 * torch.sin(0.05 * i) * torch.cos(0.07 * j) * torch.sin(0.03 * k + 1.0)
    return torch.stack([vol + 0.1 * b for b in range(batch)])[:, None]


def small_backbone_config(refine, use_cuda=False, levels=("P2", "P3", "P4", "P5")):
    return dict(
        name="attn_fpn", use_encoder_attn=False, conv_kernels=[[3, 3, 3]] * 6,
        strides=[[1, 1, 1]] + [[2, 2, 2]] * 5, in_channels=1, start_channels=4,
        depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=[5, 5, 5], mlp_ratio=4, qkv_bias=True,
        qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2, conv_merging=False,
        use_decoder_attn=refine, fpn_channels=48, out_fmaps=["P2"], pos_encoding="sine",
        feature_levels=list(levels), hidden_dim=48, dim_feedforward=64, dropout=0.1, nheads=6, layers=2,
        n_points=4, use_cuda=use_cuda, use_seg_proxy_loss=False, fg_bg=True)


def small_swin_config(conv_merging=False):
    """Reduced-width backbone with the Swin encoder (use_encoder_attn=True): start_channels 6 so that the Swin
    stage widths 12/24/48/96 divide by the heads 3/6/12/24."""
    cfg = small_backbone_config(False)
    cfg.update(use_encoder_attn=True, start_channels=6, conv_merging=conv_merging, drop_path_rate=0.2)
    return cfg


def small_model_config(refine, use_cuda=False):
    """Reduced-width VISCERAL-geometry model (the Focused Decoder's mask table
    only admits 160x160x256, focused_decoder.py:99-117)."""
    from transoar_amd.config import synthetic_bbox_properties, visceral_config
    cfg = visceral_config(refine=refine, use_cuda=use_cuda)
    cfg["backbone"] = small_backbone_config(refine, use_cuda)
    cfg["neck"].update(hidden_dim=48, dim_feedforward=64)
    cfg["bbox_properties"] = synthetic_bbox_properties(20, seed=0)
    return cfg
