"""AdamW with the update of every parameter in ONE launch (include/transoar_optim.h, csrc/optim.hip).

A subclass of torch.optim.AdamW: parameter groups, state (step / exp_avg / exp_avg_sq per parameter, the step counts
and learning rates on the device as with capturable=True), state_dict / load_state_dict and the LR schedulers are
torch's own -- only step() is replaced.  torch's fused implementation runs the flagship's 335 tensors as eight
multi_tensor_apply launches, 0.60 ms per step; this one moves the same 28 bytes per parameter in 0.3 ms.
Reference: scripts/train.py:52-63 (AdamW, two learning rates, weight decay 1e-4)."""
import ctypes
import os

import torch

from . import _native  # noqa: F401

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtransoar_optim.so")
if not os.path.exists(_LIB_PATH):
    raise _native.NativeLibraryError("%s is not built (python transoar_amd/_build.py)" % _LIB_PATH)
lib = ctypes.CDLL(_LIB_PATH)
lib.transoar_adamw_step.restype = ctypes.c_int
lib.transoar_adamw_step.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_double] * 2 + [ctypes.c_float] * 2 + [ctypes.c_void_p]
lib.transoar_optim_abi_version.restype = ctypes.c_int
if lib.transoar_optim_abi_version() != 1:
    raise _native.NativeLibraryError("%s: ABI version mismatch, rebuild" % _LIB_PATH)
CHUNK = 16384
_MAX_TABLES = 32


class FlatAdamW(torch.optim.AdamW):
    """AdamW(params, lr, weight_decay, betas, eps) on CUDA fp32 parameters; lr may differ per group and is kept as a
    device tensor (a scheduler changes it in place), so the step can be part of a captured HIP graph."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True, capturable=True)
        for g in self.param_groups:
            if g["amsgrad"] or g["maximize"]:
                raise ValueError("FlatAdamW: amsgrad / maximize are not implemented")
            if not torch.is_tensor(g["lr"]):
                dev = g["params"][0].device
                g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)
        self._tables = {}          # pointer signature -> (device table, chunk tensor ids, chunk offsets, n_chunks, keep-alive)

    def prepare_capture(self):
        """Call (eagerly) right before a stream capture that contains step(): the addresses of the gradients a captured
        backward produces are only known while capturing, and their table has to reach the device from PINNED host memory
        that exists already (no host allocation inside a capture) and is never rewritten afterwards (every replay copies
        it again)."""
        params = [p for g in self.param_groups for p in g["params"]]
        n_chunks = sum((p.numel() + CHUNK - 1) // CHUNK for p in params)
        dev = params[0].device
        self._capture_slot = (torch.zeros((len(params), 8), dtype=torch.int64).pin_memory(),
                              torch.zeros(n_chunks, dtype=torch.int32).pin_memory(),
                              torch.zeros(n_chunks, dtype=torch.int64).pin_memory(),
                              torch.zeros((len(params), 8), dtype=torch.int64, device=dev),
                              torch.zeros(n_chunks, dtype=torch.int32, device=dev),
                              torch.zeros(n_chunks, dtype=torch.int64, device=dev))

    def _table(self, rows, device):
        """rows: [(p, g, m, v, lr, step)] -> device work list; cached per set of addresses.  A captured graph keeps using
        the buffers of ITS signature, so entries are never rewritten in place, only added."""
        sig = tuple(t.data_ptr() for r in rows for t in r)
        hit = self._tables.get(sig)
        if hit is not None:
            if torch.cuda.is_current_stream_capturing() and not hit[5]:
                # an eager entry found again while capturing (data-parallel: the bucket views have the same addresses in the
                # warm-up steps and in the capture): the graph bakes this table's address in, so it must never be evicted
                hit = self._tables[sig] = hit[:5] + (True,)
            return hit
        import numpy as np
        tab = np.zeros((len(rows), 8), dtype=np.int64)
        ids, offs = [], []
        for i, (p, g, m, v, lr, st) in enumerate(rows):
            if p.numel() % 4 == 0 and any(t.data_ptr() % 16 for t in (p, g, m, v)):
                # the kernel's float4 path (n % 4 == 0) needs 16-byte aligned pointers: a view at an odd offset of a storage
                # would fault.  torch allocations and the reducer's bucket views (128-byte aligned) are.  (Checked when a work
                # table is made, i.e. once per set of addresses.)
                raise ValueError("FlatAdamW: parameter, gradient and moments must be 16-byte aligned")
            tab[i, :6] = [p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lr.data_ptr(), st.data_ptr()]
            tab[i, 6] = p.numel()
            for off in range(0, p.numel(), CHUNK):
                ids.append(i)
                offs.append(off)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            slot = getattr(self, "_capture_slot", None)
            if slot is None:
                raise RuntimeError("FlatAdamW.step() inside a stream capture: call prepare_capture() before the capture starts")
            self._capture_slot = None                       # consumed: this capture's replays read it for good
            h_tab, h_ids, h_offs, d_tab, d_ids, d_offs = slot
            h_tab[: len(rows)].copy_(torch.from_numpy(tab))
            h_ids[: len(ids)].copy_(torch.tensor(ids, dtype=torch.int32))
            h_offs[: len(offs)].copy_(torch.tensor(offs, dtype=torch.int64))
            d_tab.copy_(h_tab, non_blocking=True)
            d_ids.copy_(h_ids, non_blocking=True)
            d_offs.copy_(h_offs, non_blocking=True)
            entry = (d_tab, d_ids, d_offs, len(ids), slot, True)
        else:
            if len(self._tables) >= _MAX_TABLES:
                # eager steps get their gradient tensors from the allocator: a handful of recurring address sets; drop the
                # oldest table that no captured graph can be holding
                for k, v in list(self._tables.items()):
                    if not v[5]:
                        del self._tables[k]
                        break
            # from PINNED host memory, without waiting: a pageable copy is synchronous, i.e. the host sat out the whole queue
            # once per step whenever the gradients had new addresses (fresh tensors every eager step) -- 33 ms of "host
            # time" per eager step where the enqueue work is 18 (bench.py host_enqueue_drained_ms)
            h = (torch.from_numpy(tab).pin_memory(), torch.tensor(ids, dtype=torch.int32).pin_memory(),
                 torch.tensor(offs, dtype=torch.int64).pin_memory())
            entry = (h[0].to(device, non_blocking=True), h[1].to(device, non_blocking=True), h[2].to(device, non_blocking=True),
                     len(ids), h, False)
        self._tables[sig] = entry
        return entry

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        rows, steps, device = [], [], None
        betas = eps = wd = None
        for group in self.param_groups:
            cfg = (tuple(group["betas"]), float(group["eps"]), float(group["weight_decay"]))
            if betas is None:
                (betas, eps, wd) = cfg
            elif cfg != (betas, eps, wd):
                raise ValueError("FlatAdamW: betas / eps / weight_decay must be the same in every group (only lr differs)")
            if group.get("amsgrad") or group.get("maximize"):          # (a loaded state dict can switch them on after __init__)
                raise ValueError("FlatAdamW: amsgrad / maximize are not implemented")
            if not torch.is_tensor(group["lr"]):            # (a state dict written next to a plain AdamW was loaded)
                group["lr"] = torch.tensor(float(group["lr"]), dtype=torch.float32, device=group["params"][0].device)
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise ValueError("FlatAdamW: dense contiguous fp32 CUDA parameters only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = p.grad if p.grad.is_contiguous() and p.grad.dtype == torch.float32 else p.grad.float().contiguous()
                lr = group["lr"]
                if lr.dtype != torch.float32 or lr.device != p.device:        # e.g. a state dict loaded with map_location="cpu"
                    lr = group["lr"] = lr.to(device=p.device, dtype=torch.float32)
                if device is not None and p.device != device:
                    raise ValueError("FlatAdamW: every parameter must live on one device (one launch, one work table)")
                rows.append((p, g, st["exp_avg"], st["exp_avg_sq"], lr, st["step"]))
                steps.append(st["step"])
                device = p.device
        if not rows:
            return loss
        torch._foreach_add_(steps, 1.0)
        tab, ids, offs, n_chunks = self._table(rows, device)[:4]
        with torch.cuda.device(device):
            rc = lib.transoar_adamw_step(tab.data_ptr(), ids.data_ptr(), offs.data_ptr(), n_chunks, betas[0], betas[1], eps, wd,
                                         torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError("transoar_adamw_step failed with code %d" % rc)
        # the kernel wrote through raw pointers: tell autograd (and the bf16 mirrors' staleness check) that the weights changed
        torch.autograd.graph.increment_version([r[0] for r in rows])
        return loss
