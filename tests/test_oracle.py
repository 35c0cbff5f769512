"""CPU: the two oracle restatements against the golden vectors generated from
the reference's own Python path (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import msda3d_oracle as c_oracle
from oracle.torch_ref import msda3d_core_torch
from tests._inputs import case_prefixes, load_case, medium_inputs, level_starts


def _tol(dtype):
    # fp64: the restatements differ from the reference only by summation order
    return (1e-12, 1e-14) if dtype == torch.float64 else (2e-5, 1e-7)


def _close(a, b, dtype, what):
    rtol, atol = _tol(dtype)
    scale = float(b.abs().max()) or 1.0
    assert torch.allclose(a, b, rtol=rtol, atol=atol * max(scale, 1.0) + rtol * scale * 1e-3), (
        what, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("fixture", ["g1_op_small.npz", "g2_op_edge.npz"])
def test_c_oracle_matches_golden(golden_dir, fixture):
    z = np.load(os.path.join(golden_dir, fixture))
    for prefix in case_prefixes(z):
        c = load_case(z, prefix)
        dt = c["value"].dtype
        out = c_oracle.forward(c["value"].numpy(), c["shapes"].numpy(), c["lsi"].numpy(),
                               c["loc"].numpy(), c["attn"].numpy())
        _close(torch.from_numpy(out), c["out"], dt, prefix + ".out")
        gv, gl, ga = c_oracle.backward(c["value"].numpy(), c["shapes"].numpy(), c["lsi"].numpy(),
                                       c["loc"].numpy(), c["attn"].numpy(), c["grad_out"].numpy())
        _close(torch.from_numpy(gv), c["grad_value"], dt, prefix + ".grad_value")
        if prefix != "centres":
            # On exact voxel centres the pixel coordinate is an integer and the
            # trilinear sample is not differentiable there: the kernel formula
            # (loc*size-0.5, .cuh:424-426) and grid_sample's un-normalisation
            # (((2*loc-1)+1)*size-1)/2 round differently, floor() picks
            # different cells and grad_loc is a different one-sided derivative.
            # out / grad_value / grad_attn are continuous and must still agree.
            _close(torch.from_numpy(gl), c["grad_loc"], dt, prefix + ".grad_loc")
        _close(torch.from_numpy(ga), c["grad_attn"], dt, prefix + ".grad_attn")


@pytest.mark.parametrize("fixture", ["g1_op_small.npz", "g2_op_edge.npz"])
def test_torch_restatement_matches_golden(golden_dir, fixture):
    z = np.load(os.path.join(golden_dir, fixture))
    for prefix in case_prefixes(z):
        c = load_case(z, prefix)
        dt = c["value"].dtype
        v = c["value"].clone().requires_grad_()
        loc = c["loc"].clone().requires_grad_()
        a = c["attn"].clone().requires_grad_()
        out = msda3d_core_torch(v, c["shapes"], loc, a)
        _close(out.detach(), c["out"], dt, prefix + ".out")
        gv, gl, ga = torch.autograd.grad(out, (v, loc, a), c["grad_out"])
        _close(gv, c["grad_value"], dt, prefix + ".grad_value")
        _close(gl, c["grad_loc"], dt, prefix + ".grad_loc")
        _close(ga, c["grad_attn"], dt, prefix + ".grad_attn")


def test_medium_shape_against_golden(golden_dir):
    """ops/test.py "Medium" shape, closed-form inputs: every 8th output row,
    checksums of all gradients and their first 4096 entries."""
    z = np.load(os.path.join(golden_dir, "g3_op_medium.npz"))
    value, shapes, loc, attn = medium_inputs()
    lsi = level_starts(shapes)
    out = c_oracle.forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), attn.numpy())
    np.testing.assert_allclose(out[:, ::8], z["out_rows"], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(out.sum(), z["out_sum"], rtol=1e-10)
    g = torch.Generator().manual_seed(99)
    grad_out = torch.randn(out.shape, generator=g, dtype=torch.float64)
    np.testing.assert_allclose(grad_out.sum().item(), z["grad_out_sum"], rtol=1e-12)
    gv, gl, ga = c_oracle.backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(),
                                   attn.numpy(), grad_out.numpy())
    for name, arr in (("grad_value", gv), ("grad_loc", gl), ("grad_attn", ga)):
        np.testing.assert_allclose(arr.sum(), z[name + "_sum"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.abs(arr).sum(), z[name + "_abs_sum"], rtol=1e-10)
        np.testing.assert_allclose(arr.reshape(-1)[:4096], z[name + "_head"], rtol=1e-10, atol=1e-14)
    # and the torch restatement agrees with the C one on the full tensors
    out_t = msda3d_core_torch(value, shapes, loc, attn)
    np.testing.assert_allclose(out_t.numpy(), out, rtol=1e-11, atol=1e-15)


def test_float32_oracle_close_to_float64():
    value, shapes, loc, attn = medium_inputs()
    lsi = level_starts(shapes)
    o64 = c_oracle.forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), attn.numpy())
    o32 = c_oracle.forward(value.float().numpy(), shapes.numpy(), lsi.numpy(),
                           loc.double().float().numpy(), attn.float().numpy())
    # loc rounding to fp32 moves samples by <=1e-7*size voxels: small relative change
    assert np.abs(o32 - o64).max() <= 2e-5 * np.abs(o64).max()
