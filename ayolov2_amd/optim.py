"""``SGD``: torch.optim.SGD's interface and arithmetic with the whole step in ONE HIP launch (``ayolo_sgd_step``).

The reference builds ``optim.SGD(pg_bn, lr, momentum, nesterov=True)`` plus a weight-decay group and a bias group
(scripts/train/yolo_trainer.py:149-168) and steps it through ``torch.cuda.amp.GradScaler``
(yolo_trainer.py:332-338).  This class keeps that surface -- param groups, ``state_dict`` with ``momentum_buffer``
entries, LR schedulers, ``GradScaler.step`` (``_step_supports_amp_scaling``: the scaler hands over ``grad_scale`` and
``found_inf`` tensors and the kernel divides / skips on the device, no host sync) -- but the ~180 parameter tensors are
updated by a single kernel instead of several torch launches per parameter group.  CPU parameters fall back to
torch's own arithmetic (the class is then a plain torch.optim.SGD)."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib

_MAX_GROUPS = 8
_CHUNK = 32768


class _Group(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_float), ("momentum", ctypes.c_float), ("weight_decay", ctypes.c_float),
                ("dampening", ctypes.c_float), ("nesterov", ctypes.c_int), ("reserved", ctypes.c_int)]


class _Groups(ctypes.Structure):
    _fields_ = [("g", _Group * _MAX_GROUPS)]


_JOB = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i8"), ("group", "<i4"), ("first", "<i4")])


class SGD(torch.optim.SGD):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                         nesterov=nesterov)
        self._table = None          # {gradient pointers: built job table}
        self._plist = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._table = None                  # momentum buffers were replaced

    @staticmethod
    def _dense_like(p: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        """The gradient in the parameter's own memory order (a no-op for the plan's gradients)."""
        if g.stride() == p.stride():
            return g
        out = torch.empty_like(p)          # preserve_format: same strides as p for dense tensors
        out.copy_(g)
        return out

    def _build_table(self, plist, grads):
        """Validate + lay out the job table (only when a parameter / gradient / momentum pointer changed)."""
        rows, keep, dev = [], [], None
        for (gi, p), g in zip(plist, grads):
            if g is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse):
                return None
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            if not dense:
                raise _lib.AyoloError("ayolov2_amd.optim.SGD: parameter is not a dense block of memory")
            dev = p.device
            gr = self._dense_like(p, g)
            if gr is not g:
                keep.append(gr)            # a re-laid-out copy lives as long as the table; the caller's own gradient
                                           # tensors are NOT retained (the key validates their addresses)
            buf = None
            if float(self.param_groups[gi]["momentum"]) != 0.0:
                st = self.state[p]
                buf = st.get("momentum_buffer")
                if buf is None:
                    # NaN = "not initialised yet": the kernel then takes buf = g, and a first step skipped by the
                    # GradScaler (found_inf is only known on the device) leaves the buffer uninitialised, like torch
                    buf = st["momentum_buffer"] = torch.full_like(p, float("nan"))
                elif buf.stride() != p.stride() or not buf.is_cuda:
                    nb = torch.empty_like(p)
                    nb.copy_(buf)
                    buf = st["momentum_buffer"] = nb
            # chunks of <= _CHUNK elements: one grid row each, so the few multi-million-element conv weights are spread
            # over hundreds of workgroups instead of being walked by one row of the grid
            n, pb, gb, bb = p.numel(), p.data_ptr(), gr.data_ptr(), buf.data_ptr() if buf is not None else 0
            for o in range(0, n, _CHUNK):
                rows.append((pb + 4 * o, gb + 4 * o, bb + 4 * o if bb else 0, min(_CHUNK, n - o), gi, 0))
        if not rows:
            return ()
        jobs = np.array(rows, dtype=_JOB)
        # pinned staging + asynchronous copy: a pageable .to(device) would block the host until the stream has drained
        host = torch.from_numpy(jobs.view(np.uint8).copy()).pin_memory()
        tab = host.to(dev, non_blocking=True)
        copied = len(keep) > 0
        keep.append(host)
        return tab, len(rows), keep, copied

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ngroups = len(self.param_groups)
        sig = tuple(len(g["params"]) for g in self.param_groups)
        if self._plist is None or self._plist[0] != sig:
            self._plist = (sig, [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"]])
            self._table = None
        plist = self._plist[1]
        grads = [p.grad for _, p in plist]
        # host cost matters (the train step keeps the host ~1 step ahead of the GPU at best): the table is rebuilt only
        # when a pointer changed -- the plan hands out views of one flat gradient buffer, normally at the same address
        key = tuple(0 if g is None else g.data_ptr() for g in grads) + tuple(p.data_ptr() for _, p in plist)
        if self._table is None:
            self._table = {}
        built = self._table.get(key, False)
        if built is False:
            # (the plan's per-step gradient buffer alternates between a few allocator blocks: keep a table per address set)
            built = None
            if ngroups <= _MAX_GROUPS and not any(g.get("maximize", False) for g in self.param_groups):
                built = self._build_table(plist, grads)
            if len(self._table) >= 4:
                self._table.pop(next(iter(self._table)))
            self._table[key] = built
        if built is None:                                      # CPU / exotic parameters: torch's own arithmetic
            if getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
                raise _lib.AyoloError("ayolov2_amd.optim.SGD: GradScaler hand-over needs fp32 parameters on the GPU")
            return super().step()
        if built == ():
            return loss
        tab, n, keep, copied = built
        if copied:                                             # a gradient had to be re-laid-out: its content is per step
            self._table.pop(key, None)
        groups = _Groups()
        for gi, g in enumerate(self.param_groups):
            gg = groups.g[gi]
            gg.lr, gg.momentum, gg.weight_decay = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"])
            gg.dampening, gg.nesterov = float(g["dampening"]), int(bool(g["nesterov"]))
        scale = getattr(self, "grad_scale", None)
        found = getattr(self, "found_inf", None)
        _lib.call("ayolo_sgd_step", tab.data_ptr(), n, ctypes.byref(groups),
                  scale.data_ptr() if scale is not None else None, found.data_ptr() if found is not None else None,
                  torch.cuda.current_stream().cuda_stream)
        _lib.bump_versions(p for (_, p), g in zip(plist, grads) if g is not None)     # written through raw pointers
        return loss
