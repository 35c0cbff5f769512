"""The ``MSDA`` object: the two functions the reference's pybind module
``MultiScaleDeformableAttention`` exports (ops/src/vision.cpp:13-16), bound to
the gfx950 C ABI instead of the CUDA extension.

    import transoar_amd.msda as MSDA
    out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, im2col_step)
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_out, im2col_step)

Argument order, layouts, the contiguity / device checks and the returned shapes
follow ops/src/cuda/ms_deform_attn_cuda.cu:20-80 (forward) and :83-154
(backward).  Differences: bf16/f16 storage is accepted (fp32 accumulate; the
reference dispatches float/double only, .cu:64), kernel launch errors raise
(the reference printf()s them, ms_deform_im2col_cuda.cuh:1119-1123), and
``im2col_step`` only takes part in the reference's divisibility check -- the
batch is processed in one launch.
"""
import ctypes
import os

import torch

from . import _native

_DT = {torch.float32: _native.F32, torch.float64: _native.F64,
       torch.bfloat16: _native.BF16, torch.float16: _native.F16}

# bit set forwarded to the C ABI (see include/transoar_msda3d.h); module-level so
# tests can force the generic kernels.
flags = int(os.environ.get("TRANSOAR_MSDA_FLAGS", "0"))
DETERMINISTIC = 64      # TRANSOAR_MSDA3D_DETERMINISTIC
ERR_MODE = -8           # TRANSOAR_ERR_MODE
# Bit-stable backward ("deterministic-mode backward"): grad_value is accumulated in an order that does not depend
# on atomics (stable sort of the points by (cell, point index), brick-owner walk on every level) -- about 4x the default
# backward's time.  On when this switch is set (strict: a form the mode does not cover raises) or when
# torch.use_deterministic_algorithms(True) is in force (a form that is not covered raises like torch's own ops do, or warns
# and runs the default order under warn_only=True).  grad_sampling_loc / grad_attn_weight and the forward have no atomics.
deterministic = os.environ.get("TRANSOAR_MSDA_DETERMINISTIC", "0") == "1"


# host copy of a spatial_shapes tensor, kept ON the tensor object (it dies with it; an address-keyed
# cache would hand a recycled allocation the shapes of a dead tensor): the reference passes the level
# shapes as a device tensor only; one .tolist() (a sync) per distinct tensor buys the brick schedule
# of the kernels.  locality_hint = False skips it; the per-brick kernels then fall back to the
# per-item ones (same results).
locality_hint = True


def _shapes_on_host(spatial_shapes):
    if not locality_hint:
        return None, None
    hit = getattr(spatial_shapes, "_transoar_host_shapes", None)
    if hit is None or hit[0] != spatial_shapes._version:
        vals = [int(v) for v in spatial_shapes.flatten().tolist()]
        hit = (spatial_shapes._version, (ctypes.c_int64 * len(vals))(*vals))
        spatial_shapes._transoar_host_shapes = hit
    return hit[1], ctypes.addressof(hit[1])


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_inputs(named):
    for name, t in named:
        _require(t.is_contiguous(), "%s tensor has to be contiguous" % name)
    if not named[0][1].is_cuda:
        # ops/src/ms_deform_attn.h:38,60
        raise RuntimeError("Not implemented on the CPU")
    dev = named[0][1].device
    for name, t in named:
        _require(t.is_cuda and t.device == dev, "%s must be a CUDA tensor on %s" % (name, dev))


def _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    _require(value.dim() == 4, "value must be (N, S, M, C)")
    _require(sampling_loc.dim() == 6 and sampling_loc.size(-1) == 3,
             "sampling_loc must be (N, Lq, M, L, P, 3)")
    N, S, M, C = value.shape
    L = spatial_shapes.size(0)
    Lq, P = sampling_loc.size(1), sampling_loc.size(4)
    _require(spatial_shapes.dtype == torch.int64 and tuple(spatial_shapes.shape) == (L, 3),
             "spatial_shapes must be int64 (L, 3)")
    _require(level_start_index.dtype == torch.int64 and tuple(level_start_index.shape) == (L,),
             "level_start_index must be int64 (L,)")
    _require(tuple(sampling_loc.shape) == (N, Lq, M, L, P, 3), "sampling_loc shape mismatch")
    _require(tuple(attn_weight.shape) == (N, Lq, M, L, P), "attn_weight shape mismatch")
    step = min(N, int(im2col_step))
    _require(step > 0 and N % step == 0,
             "batch(%d) must divide im2col_step(%d)" % (N, step))      # .cu:52
    _require(value.dtype in _DT, "unsupported value dtype %s" % value.dtype)
    return N, S, M, C, L, Lq, P


def _coerce_loc(value, sampling_loc, attn_weight):
    # bf16/f16 value may come with fp32 locations (what autocast produces) or
    # with same-dtype locations; anything else is converted to fp32 / value dtype.
    half = value.dtype in (torch.bfloat16, torch.float16)
    if sampling_loc.dtype != attn_weight.dtype or not (
            sampling_loc.dtype == value.dtype or (half and sampling_loc.dtype == torch.float32)):
        want = torch.float32 if half else value.dtype
        sampling_loc = sampling_loc.to(want).contiguous()
        attn_weight = attn_weight.to(want).contiguous()
    return sampling_loc, attn_weight


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step):
    """-> Tensor (N, Lq, M*C); replaces ops/src/ms_deform_attn.h:20-39."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)])
    N, S, M, C, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc,
                                 attn_weight, im2col_step)
    sampling_loc, attn_weight = _coerce_loc(value, sampling_loc, attn_weight)
    out = torch.empty((N, Lq, M * C), dtype=value.dtype, device=value.device)
    _keep, host_ptr = _shapes_on_host(spatial_shapes)
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = _native.lib.transoar_msda3d_forward(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(),
            N, S, M, C, L, Lq, P, _DT[value.dtype], _DT[sampling_loc.dtype], host_ptr, flags, stream)
    _native.check(rc, "transoar_msda3d_forward")
    return out


def ms_deform_attn_forward_fused(value, spatial_shapes, proj, reference_points, strict=True):
    """Gather with the module's sampling head in its prologue (no reference counterpart as one call: it is
    ops/modules/ms_deform_attn.py:114-136 -- softmax, ``ref + offsets / (W, H, D)``, MSDeformAttnFunction -- without
    materialising sampling_locations / attention_weights).

    value (N, S, M, C) bf16/f16; proj (N, S, 4*M*L*P) bf16 = [offsets (M, L, P, 3) | logits (M, L*P)] per query;
    reference_points (1 or N, S, L, 3) fp32.  -> (N, S, M*C).  Queries are the pyramid's voxels (Lq == S)."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes), ("proj", proj),
                   ("reference_points", reference_points)])
    _require(value.dim() == 4 and value.dtype in (torch.bfloat16, torch.float16), "value must be 16-bit (N, S, M, C)")
    N, S, M, C = value.shape
    L = spatial_shapes.size(0)
    P = proj.size(-1) // (4 * M * L)
    _require(proj.dtype == torch.bfloat16 and tuple(proj.shape) == (N, S, 4 * M * L * P), "proj must be bf16 (N, S, 4*M*L*P)")
    _require(reference_points.dtype == torch.float32 and reference_points.dim() == 4 and
             tuple(reference_points.shape[1:]) == (S, L, 3) and reference_points.size(0) in (1, N),
             "reference_points must be fp32 (1 or N, S, L, 3)")
    out = torch.empty((N, S, M * C), dtype=value.dtype, device=value.device)
    _keep, host_ptr = _shapes_on_host(spatial_shapes)
    _require(host_ptr is not None, "the fused gather needs the level shapes on the host (locality_hint)")
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = _native.lib.transoar_msda3d_forward_fused(
            value.data_ptr(), proj.data_ptr(), reference_points.data_ptr(), reference_points.size(0) * S,
            out.data_ptr(), N, S, M, C, L, P, _DT[value.dtype], host_ptr, stream)
    if rc == -2 and not strict:           # TRANSOAR_ERR_DIM: not a form the fused kernel covers
        return None
    _native.check(rc, "transoar_msda3d_forward_fused")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]; replaces
    ops/src/ms_deform_attn.h:41-61."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight), ("grad_output", grad_output)])
    N, S, M, C, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc,
                                 attn_weight, im2col_step)
    _require(tuple(grad_output.shape) == (N, Lq, M * C) and grad_output.dtype == value.dtype,
             "grad_output must be (N, Lq, M*C) with value's dtype")
    loc_in_dtype, attn_in_dtype = sampling_loc.dtype, attn_weight.dtype
    sampling_loc, attn_weight = _coerce_loc(value, sampling_loc, attn_weight)
    grad_value = torch.empty_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    dims = (N, S, M, C, L, Lq, P, _DT[value.dtype], _DT[sampling_loc.dtype])
    _keep, host_ptr = _shapes_on_host(spatial_shapes)
    torch_det = torch.are_deterministic_algorithms_enabled()

    def run(call_flags):
        ws_bytes = _native.lib.transoar_msda3d_backward_workspace_bytes(*dims, call_flags)
        workspace = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=value.device)
        with torch.cuda.device(value.device):
            stream = torch.cuda.current_stream().cuda_stream
            return _native.lib.transoar_msda3d_backward(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                workspace.data_ptr(), ws_bytes, *dims, host_ptr, call_flags, stream)

    if deterministic or torch_det or (flags & DETERMINISTIC):
        rc = run(flags | DETERMINISTIC)
        if rc == ERR_MODE and not (deterministic or (flags & DETERMINISTIC)):
            if not torch.is_deterministic_algorithms_warn_only_enabled():
                raise RuntimeError("ms_deform_attn_backward does not have a deterministic implementation for this form (16-bit "
                                   "storage, 64 channels, 4 points, <= 4 levels with queries = the pyramid's voxels are covered), "
                                   "but torch.use_deterministic_algorithms(True) is set; use warn_only=True to run the default order")
            import warnings
            warnings.warn("ms_deform_attn_backward: no deterministic implementation for this form; running the default "
                          "(atomic-order dependent) accumulation")
            rc = run(flags)
    else:
        rc = run(flags)
    _native.check(rc, "transoar_msda3d_backward")
    return [grad_value, grad_loc.to(loc_in_dtype), grad_attn.to(attn_in_dtype)]


def ms_deform_attn_backward_proj(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_proj] or None: the backward with the sampling head's backward folded into the query kernel
    (transoar_msda3d_backward_proj): grad_proj (N, Lq, 4*M*L*P) bf16 is the gradient of the stacked [sampling_offsets |
    attention_weights] projection.  None when the form is not covered (TRANSOAR_ERR_MODE): the caller then runs
    ms_deform_attn_backward and the head's own backward kernel."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)])
    N, S, M, C, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    _require(tuple(grad_output.shape) == (N, Lq, M * C) and grad_output.dtype == value.dtype,
             "grad_output must be (N, Lq, M*C) with value's dtype")
    if sampling_loc.dtype != torch.float32 or attn_weight.dtype != torch.float32 or value.dtype not in (torch.bfloat16, torch.float16):
        return None
    call_flags = flags | (DETERMINISTIC if (deterministic or torch.are_deterministic_algorithms_enabled()) else 0)
    grad_value = torch.empty_like(value)
    grad_proj = torch.empty((N, Lq, 4 * M * L * P), dtype=torch.bfloat16, device=value.device)
    dims = (N, S, M, C, L, Lq, P, _DT[value.dtype], _DT[sampling_loc.dtype])
    _keep, host_ptr = _shapes_on_host(spatial_shapes)
    ws_bytes = _native.lib.transoar_msda3d_backward_workspace_bytes(*dims, call_flags)
    workspace = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=value.device)
    with torch.cuda.device(value.device):
        rc = _native.lib.transoar_msda3d_backward_proj(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), grad_output.data_ptr(), grad_value.data_ptr(), grad_proj.data_ptr(),
            workspace.data_ptr(), ws_bytes, *dims, host_ptr, call_flags, torch.cuda.current_stream().cuda_stream)
    if rc == ERR_MODE:
        return None
    _native.check(rc, "transoar_msda3d_backward_proj")
    return [grad_value, grad_proj]
