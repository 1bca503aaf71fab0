"""Static execution plans: the model is compiled ONCE per (input shape, dtype) into two straight-line lists of
kernel launches over static HBM buffers -- forward and backward -- which ``ayolo_run_ops`` enqueues with a single
host call each (csrc/plan.hip).

Compared with the per-module autograd path (functional.py) the plan
  * removes ~1600 Python-level launches / allocations per training step (the step was host-bound);
  * eliminates the glue kernels torch ran between the HIP kernels: every Concat input is written by its producer
    directly into a channel slice of the concat buffer, the Bottleneck shortcut is added inside the BN+SiLU pass,
    gradients of multi-consumer tensors are accumulated by the dgrad epilogue (``accumulate``), BN accumulators and
    weight gradients are zeroed by one memset each;
  * keeps the reference's surface: the plan is wrapped in ONE ``torch.autograd.Function`` whose outputs are the
    YOLOHead raw tensors and whose backward returns ordinary per-parameter gradients (DDP / GradScaler / any torch
    optimiser work unchanged).

The arithmetic is the same kernels in the same order as the module path (parity-tested against the CPU oracle).
"""
from __future__ import annotations

import ctypes
from ctypes import c_double, c_float, c_int, c_int64, c_void_p
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from . import functional as F_
from ._lib import EPI_HEAD, EPI_NONE, ConvDesc
from .modules import C3, SPPF, Bottleneck, Concat, Conv, UpSample, YOLOHead, _act_code, _pair

(OP_CONV_FWD, OP_CONV_DGRAD, OP_CONV_WGRAD, OP_CAST_WEIGHT, OP_BN_FINALIZE, OP_AFFINE_ACT, OP_BN_BWD_REDUCE,
 OP_BN_BWD_APPLY, OP_MAXPOOL_FWD, OP_MAXPOOL_BWD, OP_UPSAMPLE_FWD, OP_UPSAMPLE_BWD, OP_PACK_INPUT, OP_HEAD_GRAD_PACK,
 OP_COPY2D, OP_MEMSET, OP_BN_EVAL_AFFINE, OP_BN_TRAIN_ACT, OP_CAST_WEIGHTS) = range(1, 20)
OP_JOIN_SIDE = 21
OP_STEM_BN_WGRAD = 22
OP_WGRAD_GROUP = 23
OP_BN_BWD_APPLY2 = 24
OP_SPPF_FWD = 25
OP_SPPF_BWD = 26


class Op(ctypes.Structure):
    _fields_ = [("kind", c_int), ("i", c_int * 12), ("f", c_float * 2), ("d", c_double * 1), ("l", c_int64 * 1),
                ("p", c_void_p * 16), ("conv", ConvDesc)]


def _op(kind, i=(), f=(), d=(), l=(), p=(), conv: Optional[ConvDesc] = None) -> Op:
    o = Op()
    o.kind = kind
    for k, v in enumerate(i):
        o.i[k] = int(v)
    for k, v in enumerate(f):
        o.f[k] = float(v)
    for k, v in enumerate(d):
        o.d[k] = float(v)
    for k, v in enumerate(l):
        o.l[k] = int(v)
    for k, v in enumerate(p):
        o.p[k] = None if v is None else (v if isinstance(v, int) else v.data_ptr())
    if conv is not None:
        o.conv = conv
    return o


import os as _os

OP_SIDE = 0x100
WGRAD_SIDE_STREAM = _os.environ.get("AYOLO_WGRAD_STREAM", "1") == "1"
FUSE_STEM_BACKWARD = True     # stem block: BatchNorm-backward apply inside its weight-gradient kernel (ayolo_stem_bn_wgrad)
FOLD_SHORTCUT_GRAD = True     # Bottleneck shortcut gradients written by the BatchNorm-backward apply pass (ayolo_bn_act_bwd_apply_res)
MAX_PLANS = int(_os.environ.get("AYOLO_MAX_PLANS", "4"))                 # cached plans per model (multi-scale training)
MERGE_SIBLINGS = _os.environ.get("AYOLO_MERGE_SIBLINGS", "1") == "1"     # C3: cv1 | cv2 as one conv
# the BatchNorm-backward apply passes of a merged cv1 | cv2 pair as ONE launch over whole rows of the shared z / dz buffers
# (ayolo_bn_act_bwd_apply2) instead of two over half rows
BN_APPLY_PAIR = True          # (tests flip the attribute to pin the pair against two launches)
# SPPF's three chained max-pools as one launch per direction on the LDS-resident map (ayolo_sppf_pool_fwd / _bwd; fp16 plans)
SPPF_FUSED = True
# BatchNorm-backward sums (the first of the two backward passes of a Conv-BN-act block) computed in the epilogue of the
# dgrad that produces the block's output gradient, instead of a pass of its own over da and z (ayolo_conv_dgrad_bn)
BN_REDUCE_IN_DGRAD = _os.environ.get("AYOLO_BNR", "1") == "1"
# Weight gradients run as a few GROUPED launches (ayolo_wgrad_group_run: one launch per tile class over the item list of all
# layers of a group + one fixed-order reduction of the split-K partials) instead of one launch per layer.  The backward list is
# cut into this many groups of similar work (the last one is halved WGRAD_TAIL more times: what the final group still has
# to do when the main stream's backward ends is exposed).
# Transform on load (ayolo_conv_fwd_xf): the BatchNorm + SiLU pass of a Conv block whose activation has exactly ONE reader, a
# 1x1 / stride-1 conv, is not launched -- that conv (and its weight gradient) read the block's pre-activation z and form the
# activation on the way to the MFMAs; the block keeps a tiny finalize launch (batch statistics -> scale / shift, running stats)
XF_ON_LOAD = _os.environ.get("AYOLO_XF", "1") == "1"
XF_FINALIZE_IN_READER = _os.environ.get("AYOLO_XF_FIN", "1") == "1"      # the folded blocks' finalize inside the reader's launch
# What the reader's WEIGHT GRADIENT reads: 0 (default) = the materialised activation, which the reader's forward launch writes back
# from its first channel tile ("store-back": the pass and its z read are still gone, the write stays); 1 = the pre-activation z,
# transformed on load in k_wgrad too (no write at all, but the transform then runs on the weight-gradient stream, where it cost
# the step more than the write does: same-box A/B, profiles/r04_ab_xf_*)
# (no environment switch since round 5: the on-load route lost its A/B; tests flip this attribute to pin it against the default)
XF_WGRAD_ON_LOAD = False
# The group count follows the weight gradients' total work (32-pixel steps x dw tiles, `cost` below): one group per ~0.36 M
# tile-steps, between 4 and 8 -- YOLOv5s at batch 64 (1.45 M) is fastest with 4 (profiles/r04_ab_wgrad_sweep.txt), YOLOv5l at
# batch 32 (3.56 M) with 8 (profiles/r04_cfg3_ab.txt); AYOLO_WGRAD_GROUPS=n pins it (the one-launch-per-layer
# route of round 4's A/B -- +0.67 ms, profiles/r04_ab_wgrad_groups.txt -- is gone).
WGRAD_GROUPS = int(_os.environ.get("AYOLO_WGRAD_GROUPS", "-1"))
WGRAD_GROUP_WORK = 0.36e6
WGRAD_TAIL = 2            # re-swept on round 5's final code (profiles/r05_ab_wgrad_retune.txt); environment switch retired in round 6
# Fork placement, measured in round 6 and NOT kept: launching group g later than the slot where its last layer's dz
# is complete -- so that the small-map groups run under the large-map part of the main chain -- cost +0.07 ... +0.17 ms in every
# placement tried (26 / 26 + 8 / 19 + 7 jobs later), and the groups on the main stream +0.06 ms: the main-stream kernels behind a
# fork wait for workgroup slots exactly as long as the group's work takes, wherever it is placed (profiles/r06_ab_fork_placement_1.txt).
# (A cap on a group's resident workgroups per CU -- so that the kernels of backward's dependent chain forked behind it find free
# slots at once -- was measured and lost: 13.58 ms uncapped, 14.45 with two workgroups per CU, 16.3 with one; the weight
# gradients are latency-bound per workgroup and need every slot they can get, profiles/r04_ab_wgrad_cap.txt; code removed.)


class PlanUnsupported(Exception):
    pass


class Act:
    """An activation in NHWC memory; possibly a channel slice of a wider root buffer (concat elimination)."""

    def __init__(self, t: torch.Tensor, root: Optional["Act"] = None, c0: int = 0):
        self.t = t
        self.root = root if root is not None else self
        self.c0 = c0
        if root is None:
            self.g: Optional[torch.Tensor] = None
            self.ginit = np.zeros(t.shape[1], dtype=bool)

    @property
    def C(self) -> int:
        return self.t.shape[1]

    def slice(self, c0: int, c1: int) -> "Act":
        return Act(self.t[:, c0:c1], self.root, self.c0 + c0)

    def grad(self) -> torch.Tensor:
        r = self.root
        if r.g is None:
            r.g = torch.empty_like(r.t)
        return r.g[:, self.c0:self.c0 + self.C]

    def is_init(self) -> bool:
        return bool(self.root.ginit[self.c0:self.c0 + self.C].all())

    def mark_init(self) -> None:
        self.root.ginit[self.c0:self.c0 + self.C] = True


class _FloatArena:
    """Bump allocator over ONE flat tensor (fp32 by default; the BatchNorm accumulators are fp64, see ayolo.h)."""

    def __init__(self, dtype: torch.dtype = torch.float32):
        self.reqs: List[Tuple[int, list]] = []
        self.total = 0
        self.dtype = dtype
        self.buf: Optional[torch.Tensor] = None

    def request(self, n: int) -> int:
        off = self.total
        self.total += (n + 63) // 64 * 64
        return off

    def allocate(self, device):
        self.buf = torch.zeros(max(self.total, 64), dtype=self.dtype, device=device)

    def view(self, off: int, n: int) -> torch.Tensor:
        return self.buf[off:off + n]


class TrainPlan:
    def __init__(self, model, x_shape: Sequence[int], dt: torch.dtype, device):
        self.model = model
        self.dt = dt
        self.device = device
        self.B, self.Cimg, self.H, self.W = x_shape
        self.keep: List[torch.Tensor] = []          # everything the op lists point to
        self.fwd: List[Op] = []
        self.casts: List[Op] = []
        self.bwd_emitters: List[Callable[[], None]] = []
        self.bwd: List[Op] = []
        self.stats = _FloatArena(torch.float64)     # BN forward accumulators (zeroed at the start of forward)
        self.sums = _FloatArena(torch.float64)      # BN backward accumulators (zeroed at the start of forward)
        self.small = _FloatArena()                  # mean / invstd / scale / shift
        self.gradarena = _FloatArena()              # every parameter gradient (zeroed at the start of backward)
        self.param_grad_view: Dict[int, Callable[[], torch.Tensor]] = {}
        self.params: List[nn.Parameter] = []
        self.dz_list: List[torch.Tensor] = []
        self.late: List[Callable[[], None]] = []    # closures run after the arenas exist (pointer binding)
        self.bn_counters: List[torch.Tensor] = []
        self.bn_buffers: List[torch.Tensor] = []            # running statistics the forward kernels update in place
        self.bn_sync_fix: list = []                         # sync_bn: (bn, small-arena offset, C, local count) per layer
        self.bn_in_dgrad = 0                                # BN layers whose backward sums ride in a dgrad epilogue
        self._gwrites: List[tuple] = []                     # (backward op index, root Act id, c_lo, c_hi, is_dgrad)
        self._bn_layers: List[dict] = []                    # two-pass BN backward layers (candidates for the dgrad epilogue)
        self.collect_times = False                          # bench.py: per-op in-situ timing (ayolo_run_ops_timed)
        self.op_times: Dict[str, list] = {}
        self.raw_specs = []
        self.grad_done: List[Tuple[int, int, int]] = []     # (index of the backward op that completes it, arena offset, n)
        self.fwd_sync: list = []                            # sync_bn: (conv op, stats offset, n)
        self.bwd_sync: list = []                            # sync_bn: (index of the reduce op, sums view)
        self.draw_ops: List[Op] = []
        self._head_dz: Dict[int, Tuple[int, int]] = {}
        self._wjobs: List[dict] = []                        # weight-gradient jobs in backward order (grouped after emission)
        self._reads: List[tuple] = []                       # (root Act id, c_lo, c_hi, kind, consumer record): who reads which activation
        self._reads_at: set = set()                         # forward-list lengths at which a read was registered (see _read)
        self._producers: List[dict] = []                    # Conv-BN-act blocks: candidates for transform on load
        self.xf_layers = 0                                  # blocks whose BatchNorm + activation pass was folded into the consumer
        self.wgroup_costs: Dict[int, Tuple[float, float, int]] = {}   # backward op index -> (bytes, flop, layers) of a group launch
        self.wgroup_slots: List[Tuple[int, int]] = []       # (backward op index, head level) pairs: dy override pointers to patch
        self.pack_op: Optional[Op] = None
        self._compile()

    # ------------------------------------------------------------------ helpers
    def _new_act(self, C, H, W) -> Act:
        t = ops.new_act(self.B, C, H, W, self.dt, self.device)
        self.keep.append(t)
        return Act(t)

    def _register_param(self, p: nn.Parameter, numel_pad: int, view_fn) -> int:
        off = self.gradarena.request(numel_pad)
        self.params.append(p)
        self.param_grad_view[id(p)] = (off, numel_pad, view_fn)
        return off

    def _wrote(self, off: Optional[int], n: int, at: Optional[int] = None) -> None:
        """The backward op just appended (or op index `at`) completes the gradient arena range [off, off+n)."""
        if off is not None:
            self.grad_done.append((len(self.bwd) - 1 if at is None else at, off, n))

    def _read(self, act: "Act", kind: str, consumer: Optional[dict] = None) -> None:
        """EVERY forward emitter registers the activations its op reads here (the transform-on-load pass decides from
        this list whether an activation has exactly one 1x1 reader and may stay virtual -- a reader that is not registered
        would see an unmaterialised activation).  _emit_fwd checks that no forward op was appended without a registration."""
        self._reads.append((id(act.root), act.c0, act.c0 + act.C, kind, consumer))
        self._reads_at.add(len(self.fwd))

    def _gw(self, act: "Act", is_dgrad: bool) -> None:
        """The backward op just appended writes (or accumulates into) the gradient of `act`."""
        self._gwrites.append((len(self.bwd) - 1, id(act.root), act.c0, act.c0 + act.C, is_dgrad))

    def _dz(self, n: int) -> torch.Tensor:
        """Backward operand dz of one layer: a buffer of its own -- the grouped weight-gradient launches on the side stream read it
        long after the layer's turn on the main stream."""
        t = torch.empty(n, dtype=self.dt, device=self.device)
        self.keep.append(t)
        self.dz_list.append(t)                               # tools/grad_dump.py: layer-by-layer comparison of two routes
        return t

    def _wgrad_jobs_of(self, cx: dict, desc: ConvDesc, dy: torch.Tensor, off: int, n: int, dy_slot: int = -1) -> None:
        """The weight-gradient job(s) of a conv whose x the transform-on-load pass may have redirected: one job over the plain
        input, one over the producer's z, or -- two input segments -- one job per segment, each writing its column block of dw."""
        if not cx["segs"]:
            return self._wgrad_job(desc, cx["x"], dy, off, n, dy_slot=dy_slot)
        K = desc.kh * desc.kw * desc.Cin
        for t, ld, c0, C, xf in cx["segs"]:
            d = ConvDesc.from_buffer_copy(desc)
            d.Cin, d.ldx = C, ld
            self._wgrad_job(d, t, dy, off, n, dy_slot=dy_slot, xf=xf, col0=c0, dw_ld=(K if len(cx["segs"]) > 1 else 0))

    def _wgrad_job(self, desc: ConvDesc, x: torch.Tensor, dy: torch.Tensor, off: int, n: int, dy_slot: int = -1, xf=None, col0: int = 0,
                   dw_ld: int = 0) -> None:
        """One layer's weight gradient: a slot in the backward list now (dz is complete here), the launch later
        (_group_wgrads: the slot of the LAST layer of a group becomes the group's launch, the others stay empty)."""
        self.bwd.append(_op(0))
        self._wjobs.append(dict(idx=len(self.bwd) - 1, desc=desc, x=x, dy=dy, off=off, n=n, slot=dy_slot, xf=xf, col0=col0, dw_ld=dw_ld))

    def _group_wgrads(self) -> None:
        from ._lib import WgradJob
        jobs = self._wjobs
        if not jobs:
            return
        ga = self.gradarena
        side = OP_SIDE if WGRAD_SIDE_STREAM else 0
        lib = _lib.lib()
        es = 2 if self.dt == torch.float16 else 4

        def cost(j):                                       # 32-pixel steps x dw tiles: what a launch's duration scales with
            d = j["desc"]
            tm = 32 if d.Cout <= 32 else (64 if d.Cout <= 64 else 128)
            tiles = -(-(d.kh * d.kw * d.Cin) // 128) * -(-d.Cout // tm)
            return tiles * -(-(d.B * d.Ho * d.Wo) // 32)

        def account(js):
            by = sum(es * (j["desc"].B * j["desc"].H * j["desc"].W * j["desc"].Cin + j["desc"].B * j["desc"].Ho * j["desc"].Wo * j["desc"].Cout)
                     + 4 * j["desc"].Cout * j["desc"].kh * j["desc"].kw * j["desc"].Cin for j in js)
            fl = sum(2.0 * j["desc"].B * j["desc"].Ho * j["desc"].Wo * j["desc"].Cout * j["desc"].kh * j["desc"].kw * j["desc"].Cin for j in js)
            return float(by), fl, len({j["off"] for j in js})       # layers (a two-segment conv is two jobs over one weight)

        # ---- cut the jobs (backward order) into groups of similar work; the tail is cut finer
        costs = [cost(j) for j in jobs]
        total = float(sum(costs))
        ngroups = WGRAD_GROUPS if WGRAD_GROUPS > 0 else min(8, max(4, int(round(total / WGRAD_GROUP_WORK))))
        targets = [total / ngroups] * max(ngroups - 1, 0)
        rest = total / ngroups
        for _ in range(max(WGRAD_TAIL, 0)):
            rest /= 2
            targets.append(rest)
        targets.append(rest)
        groups, cur, acc, t = [], [], 0.0, 0
        for j, c in zip(jobs, costs):
            cur.append(j)
            acc += c
            if acc >= targets[min(t, len(targets) - 1)] * 0.999 and j is not jobs[-1]:
                groups.append(cur)
                cur, acc, t = [], 0.0, t + 1
        if cur:
            groups.append(cur)
        # ---- one table per group (host copy + device copy), one workspace for all (they run back to back on one stream)
        built = []
        ws_need = 16
        for js in groups:
            arr = (WgradJob * len(js))()
            for k, j in enumerate(js):
                arr[k].conv = j["desc"]
                arr[k].x, arr[k].dy = j["x"].data_ptr(), j["dy"].data_ptr()
                arr[k].dw = ga.view(j["off"], j["n"]).data_ptr() + 4 * j["col0"]
                arr[k].dw_ld = j["dw_ld"]
                arr[k].alpha, arr[k].dy_slot, arr[k].overwrite = 1.0, j["slot"], 1       # nothing else writes these arena ranges
                if j["xf"] is not None:
                    arr[k].xscale, arr[k].xshift, arr[k].xact = j["xf"][0].data_ptr(), j["xf"][1].data_ptr(), j["xf"][2]
            tb, wb = ctypes.c_size_t(0), ctypes.c_size_t(0)
            _lib.check(lib.ayolo_wgrad_group_size(arr, len(js), ctypes.byref(tb), ctypes.byref(wb)), "ayolo_wgrad_group_size")
            host = ctypes.create_string_buffer(tb.value)
            _lib.check(lib.ayolo_wgrad_group_build(arr, len(js), host, tb.value), "ayolo_wgrad_group_build")
            dev = torch.frombuffer(host, dtype=torch.uint8).clone().to(self.device)
            ws_need = max(ws_need, wb.value)
            built.append((js, host, dev))
            self.keep += [host, dev]
        ws = torch.empty(ws_need, dtype=torch.uint8, device=self.device)
        self.keep.append(ws)
        self.wgroup_ws_bytes = ws_need
        job_pos = {id(j): k for k, j in enumerate(jobs)}
        taken = set()
        for g, (js, host, dev) in enumerate(built):
            # the group launches where its LAST layer's dz is complete
            k = job_pos[id(js[-1])]
            while jobs[k]["idx"] in taken and k + 1 < len(jobs):
                k += 1
            while jobs[k]["idx"] in taken:
                k -= 1
            assert k >= job_pos[id(js[-1])], "no free slot behind the group's last layer"
            idx = jobs[k]["idx"]
            taken.add(idx)
            slots = sorted({j["slot"] for j in js if j["slot"] >= 0})
            nov = (max(slots) + 1) if slots else 0
            o = _op(OP_WGRAD_GROUP | side, i=(nov,), l=(ws.numel(),), p=(ctypes.addressof(host), dev, ws))
            for j in js:
                if j["slot"] >= 0:
                    o.p[3 + j["slot"]] = j["dy"].data_ptr()
                    self.wgroup_slots.append((idx, j["slot"]))
                self.grad_done.append((idx, j["off"], j["n"]))
            self.bwd[idx] = o
            self.wgroup_costs[idx] = account(js)

    # ------------------------------------------------------------------ conv + BN + act
    def _conv_block(self, mod: Conv, x: Optional[Act], dst: Optional[Act], residual: Optional[Act] = None,
                    image: bool = False) -> Act:
        return self._conv_group([mod], x, [dst], residual, image)[0]

    @staticmethod
    def _mergeable(a: Conv, b: Conv) -> bool:
        """Two Conv blocks over the SAME input can run as one conv with concatenated output channels (C3's cv1 | cv2)."""
        ca, cb = a.conv, b.conv
        if not (isinstance(ca, nn.Conv2d) and isinstance(cb, nn.Conv2d)):
            return False
        same = (ca.in_channels == cb.in_channels and ca.kernel_size == cb.kernel_size and ca.stride == cb.stride
                and ca.padding == cb.padding and ca.weight.dtype == cb.weight.dtype)
        k = ca.in_channels * ca.kernel_size[0] * ca.kernel_size[1]
        # the two weight gradients must be adjacent in the gradient arena (64-element slots), outputs 8-channel aligned
        return bool(same and MERGE_SIBLINGS and (ca.out_channels * k) % 64 == 0 and ca.out_channels % 8 == 0)

    def _conv_group(self, mods: Sequence[Conv], x: Optional[Act], dsts: Sequence[Optional[Act]],
                    residual: Optional[Act] = None, image: bool = False) -> List[Act]:
        """One or several Conv-BN-act blocks reading the same input.  Several blocks (C3's cv1 and cv2) become ONE conv
        whose output channels are the blocks' channels side by side: the input is read once by the forward conv and once
        by the weight-gradient kernel, and ONE dgrad over the concatenated dz replaces a dgrad plus an accumulating
        (read-modify-write) dgrad.  BatchNorm / activation stay per block (channel slices of the shared z / dz)."""
        dt, dev = self.dt, self.device
        convs = [m.conv for m in mods]
        bns = [getattr(m, "batch_norm", None) for m in mods]
        for conv, bn in zip(convs, bns):
            if not isinstance(conv, nn.Conv2d) or bn is None or conv.bias is not None or bn.momentum is None:
                raise PlanUnsupported("non-standard Conv block")
            if bn.running_mean is None or bn.running_mean.dtype != torch.float32 or conv.weight.dtype != torch.float32:
                raise PlanUnsupported("needs fp32 master weights / BN buffers")
            if not conv.weight.detach().permute(0, 2, 3, 1).is_contiguous():
                conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
        assert len(mods) == 1 or (residual is None and not image)
        couts = [c.weight.shape[0] for c in convs]
        Ct = sum(couts)
        _, Cin, kh, kw = convs[0].weight.shape
        if image:
            xshape = (self.B, self.Cimg, self.H, self.W)
        else:
            xshape = tuple(x.t.shape)
        geo = F_._Geometry(xshape, (Ct, Cin, kh, kw), _pair(convs[0].stride), _pair(convs[0].padding), dt)
        if image:
            packed = ops.new_act(self.B, geo.cin_pad, self.H, self.W, dt, dev)
            self.keep.append(packed)
            self.pack_op = _op(OP_PACK_INPUT, i=(self.B, self.Cimg, self.H, self.W, ops.dtype_code(dt), geo.cin_pad), p=(None, packed))
            self.fwd.append(self.pack_op)
            xk = packed.as_strided((self.B, 8, self.H, self.W // 2), (self.H * self.W * 4, 1, self.W * 4, 8)) if geo.packed_stem else packed
        else:
            if geo.needs_pack:
                raise PlanUnsupported("channel count not a multiple of the vector width")
            xk = x.t
        _, _, _, _, ldx = ops.nhwc_info(xk)
        code = ops.dtype_code(dt)
        # what the weight gradient will read as x (the transform-on-load pass may point it at the producer's z)
        cx = dict(x=xk, ldx=ldx, xf=None, segs=None, xs_off=None)
        # weights: compute-dtype copy and its transpose, refreshed by a cast op at the start of every forward
        wc = torch.empty((Ct, kh, kw, geo.cin_pad), dtype=dt, device=dev)
        wt = torch.empty((geo.cin_pad, kh, kw, Ct), dtype=dt, device=dev)
        self.keep += [wc, wt]
        c0s = [sum(couts[:j]) for j in range(len(couts))]
        for conv, c0, co in zip(convs, c0s, couts):
            self.casts.append(_op(OP_CAST_WEIGHT, i=(co, kh, kw, Cin, co, geo.cin_pad, code, Ct),
                                  p=(conv.weight, wc[c0:c0 + co], wt[..., c0:])))
        z = self._new_act(Ct, geo.Ho, geo.Wo)
        R = ops.STAT_REPS
        st_off = self.stats.request(R * 2 * Ct)
        npix = self.B * geo.Ho * geo.Wo
        op_conv = _op(OP_CONV_FWD, i=(EPI_NONE, R, 0), p=(xk, wc, z.t, None, None, None), conv=geo.desc(dt, ldx, Ct))
        self.fwd.append(op_conv)
        if image:
            self._reads_at.add(len(self.fwd))                 # the stem reads the packed image, not an activation
        else:
            pointwise = (kh, kw) == (1, 1) and _pair(convs[0].stride) == (1, 1) and _pair(convs[0].padding) == (0, 0) and geo.Cin_k == Cin
            if pointwise:
                cx["xs_off"] = self.small.request(2 * Cin)           # scale | shift over the conv's input channels (transform on load)
            self._read(x, "conv", dict(op=op_conv, cx=cx, pointwise=pointwise, cin=Cin, cout=Ct, x=x))
            if residual is not None:
                self._read(residual, "residual")
        self.late.append(lambda: op_conv.p.__setitem__(5, self.stats.view(st_off, R * 2 * Ct).data_ptr()))
        self.fwd_sync.append((op_conv, st_off, R * 2 * Ct))            # sync_bn: all-reduce of the batch statistics
        K = geo.kdims[0] * geo.kdims[1] * geo.Cin_k
        outs: List[Act] = []
        per = []          # per block: (bn, act, a, zj, c0, co, sm_off, su_off, gg_off, gb_off)
        gw_off0 = None
        for j, (mod, conv, bn, c0, co) in enumerate(zip(mods, convs, bns, c0s, couts)):
            act = _act_code(mod.activation)
            dst = dsts[j]
            a = dst if dst is not None else self._new_act(co, geo.Ho, geo.Wo)
            assert a.C == co and tuple(a.t.shape[2:]) == (geo.Ho, geo.Wo)
            _, _, _, _, lda = ops.nhwc_info(a.t)
            zj = z.t[:, c0:c0 + co] if len(mods) > 1 else z.t
            sm_off = self.small.request(4 * co)
            res_t = residual.t if residual is not None else None
            ldr = ops.nhwc_info(res_t)[4] if res_t is not None else 0
            # batch statistics -> scale/shift, running-stat update and a = act(bn(z)) (+ residual) in one kernel
            op_act = _op(OP_BN_TRAIN_ACT, i=(code, Ct, lda, co, R, act, ldr, Ct), l=(npix,), d=(float(npix),),
                         f=(bn.eps, bn.momentum),
                         p=(zj, a.t, None, bn.weight, bn.bias, bn.running_mean, bn.running_var, None, None, res_t))
            self.fwd.append(op_act)

            def bind_fwd(op_act=op_act, c0=c0, co=co, sm_off=sm_off):
                st = self.stats.view(st_off, R * 2 * Ct)
                sm = self.small.view(sm_off, 4 * co)
                op_act.p[2] = st.data_ptr() + 8 * c0
                op_act.p[7] = sm[0:co].data_ptr()              # save_mean
                op_act.p[8] = sm[co:2 * co].data_ptr()         # save_invstd

            self.late.append(bind_fwd)
            if bn.num_batches_tracked is not None:
                self.bn_counters.append(bn.num_batches_tracked)
            self.bn_buffers += [bn.running_mean, bn.running_var]
            self.bn_sync_fix.append((bn, sm_off, co, npix))

            # ---- parameters / gradient slots (the blocks' weight gradients are adjacent: one wgrad writes them all)
            def wview(buf, co=co, kh=kh, kw=kw, cp=geo.cin_pad, Cin=Cin):
                g = buf.view(co, kh, kw, cp)[..., :Cin].permute(0, 3, 1, 2)
                return g if cp == Cin else g.contiguous(memory_format=torch.channels_last)

            gw_off = self._register_param(conv.weight, co * K, wview)
            if j == 0:
                gw_off0 = gw_off
            else:
                assert gw_off == gw_off0 + c0 * K, "merged conv: weight gradients must be adjacent in the arena"
            per.append([bn, act, a, zj, c0, co, sm_off])
            outs.append(a)
            self._producers.append(dict(a=a, op=op_act, z=zj, ldz=Ct, co=co, sm_off=sm_off, act=act, bn=bn, residual=residual is not None,
                                        R=R, npix=npix))
        for e in per:
            bn, co = e[0], e[5]
            e.append(self.sums.request(R * 2 * co))
            e.append(self._register_param(bn.weight, co, lambda b: b) if bn.weight is not None else None)
            e.append(self._register_param(bn.bias, co, lambda b: b) if bn.bias is not None else None)
        x_act = x

        # the fused stem backward (ayolo_stem_bn_wgrad) exists for exactly the kernel's geometry -- pixel-pair taps 6 x 3, stride
        # (2, 1), padding (2, 1), <= 64 output channels -- with x, z and da each inside one 2 GiB buffer descriptor; anything
        # else (other packed stems, a batch / resolution whose stem tensors reach 2 GiB) keeps the separate apply + weight
        # gradient ops, which split large batches on the C side
        lim = (1 << 31) - 4096
        stem_fused = (FUSE_STEM_BACKWARD and image and geo.packed_stem and dt == torch.float16 and len(mods) == 1 and residual is None
                      and Ct <= 64 and Ct % 8 == 0 and tuple(geo.kdims) == (6, 3) and tuple(geo.sdims) == (2, 1)
                      and tuple(geo.pdims) == (2, 1) and self.B * self.H * self.W * 8 < lim and 2 * npix * Ct < lim
                      and (dsts[0] is None or 2 * npix * dsts[0].root.C < lim))

        def emit_bwd():
            ga = self.gradarena
            dz = self._dz(npix * Ct) if not stem_fused else None
            dzv = dz.view(self.B, geo.Ho, geo.Wo, Ct).permute(0, 3, 1, 2) if dz is not None else None
            pair = (BN_APPLY_PAIR and len(per) == 2 and residual is None and not stem_fused and per[0][1] == per[1][1]
                    and Ct <= 2048 and all(e[5] % (8 if dt == torch.float16 else 4) == 0 for e in per))
            pair_p, pair_i, pair_done = [], [], []
            for bn, act, a, zj, c0, co, sm_off, su_off, gg_off, gb_off in per:
                da = a.grad()
                if not a.is_init():
                    raise RuntimeError("plan: gradient of a conv output was never produced")
                _, _, _, _, ldda = ops.nhwc_info(da)
                sm = self.small.view(sm_off, 4 * co)
                su = self.sums.view(su_off, R * 2 * co)
                dgam = ga.view(gg_off, co) if gg_off is not None else None
                dbet = ga.view(gb_off, co) if gb_off is not None else None
                self.bwd.append(_op(OP_BN_BWD_REDUCE, i=(code, Ct, ldda, co, act, R), l=(npix,),
                                    p=(zj, da, sm[0:co], sm[co:2 * co], bn.weight, bn.bias, su)))
                # shortcut (Bottleneck: out = x + a): d(x) (+)= d(a) leaves with the same pass (it reads da anyway) instead of a
                # strided copy of its own
                dr = residual.grad() if residual is not None else None
                fold_res = dr is not None and FOLD_SHORTCUT_GRAD
                if stem_fused:
                    # the stem has no input gradient: its dz has one reader, the weight-gradient kernel, which forms it from
                    # (da, z) on the way to LDS -- no apply pass, no dz buffer traffic (ayolo_stem_bn_wgrad)
                    self.bwd.append(_op(OP_STEM_BN_WGRAD, i=(Ct, act, R), f=(1.0, 1.0),
                                        p=(xk, zj, da, sm[0:co], sm[co:2 * co], bn.weight, bn.bias, su, ga.view(gw_off0, Ct * K), dgam, dbet),
                                        conv=geo.desc(dt, ldx, ldda)))
                    self._wrote(gw_off0, Ct * K)
                elif pair:
                    pair_p += [da, sm[0:2 * co], bn.weight, bn.bias, su, dgam, dbet]
                    pair_i += [co, ldda]
                else:
                    self.bwd.append(_op(OP_BN_BWD_APPLY, i=(code, Ct, ldda, Ct, co, act, R) + ((ops.nhwc_info(dr)[4], int(residual.is_init())) if fold_res else (0, 0)),
                                        l=(npix,), f=(1.0,),
                                        p=(zj, da, dzv[:, c0:c0 + co], sm[0:co], sm[co:2 * co], bn.weight, bn.bias, su, dgam, dbet,
                                           dr if fold_res else None)))
                if pair:
                    pair_done += [(gg_off, co), (gb_off, co)]   # dgamma / dbeta leave with the pair's apply pass, appended below
                else:
                    self._wrote(gg_off, co)
                    self._wrote(gb_off, co)
                ridx = len(self.bwd) - (1 if pair else 2)       # this block's reduce op
                self.bwd_sync.append((ridx, su))               # sync_bn: all-reduce of the sums between reduce and apply
                self._bn_layers.append(dict(reduce=ridx, a=a, z=zj, ldz=Ct, sm=sm[0:2 * co], gamma=bn.weight,
                                            beta=bn.bias, sums=su, C=co, act=act, R=R))
                if residual is not None:      # shortcut: d(residual) += d(a)
                    if not fold_res:
                        self.bwd.append(_op(OP_COPY2D, i=(code, ldda, ops.nhwc_info(dr)[4], co, int(residual.is_init())), l=(npix,),
                                            p=(da, dr)))
                    self._gw(residual, False)
                    residual.mark_init()
            if pair:
                # both blocks' sums are complete (two reduce ops / dgrad epilogues above): one apply pass over whole rows
                self.bwd.append(_op(OP_BN_BWD_APPLY2, i=(code, Ct, Ct, per[0][1], R, pair_i[0], pair_i[2], pair_i[1], pair_i[3]),
                                    l=(npix,), f=(1.0,), p=[z.t, dzv] + pair_p))
                for off, n in pair_done:                        # gradient-arena ranges this op completes (bucket readiness, DDP)
                    self._wrote(off, n)
            if stem_fused:
                return
            # the weight gradient only needs dz: its slot comes BEFORE the layer's dgrad, so that a (grouped) launch forked
            # here does not wait for that dgrad
            self._wgrad_jobs_of(cx, geo.desc(dt, ldx, Ct), dz, gw_off0, Ct * K)
            if not image:
                dx = x_act.grad()
                self.bwd.append(_op(OP_CONV_DGRAD, i=(int(x_act.is_init()),), p=(dz, wt, dx),
                                    conv=geo.desc(dt, ops.nhwc_info(dx)[4], Ct)))
                self._gw(x_act, True)
                x_act.mark_init()

        self.bwd_emitters.append(emit_bwd)
        return outs

    def _fold_bn_act_into_consumers(self) -> None:
        """Transform on load (stage A).  A Conv-BN-act block whose activation `a` has exactly one reader -- a
        1x1 / stride-1 conv (a Conv block, C3's merged cv1 | cv2, C3's cv3 over its two-part concat, a YOLOHead level) -- does
        not materialise `a`: its BatchNorm + activation pass becomes the finalize launch (batch statistics -> scale / shift,
        saved and running statistics), the reader's forward conv and weight gradient take the block's pre-activation z plus the
        two per-channel vectors and form act(z * scale + shift) on the way to the MFMAs -- bit for bit the activation the pass
        would have written.  The reader's input channels are one or two SEGMENTS (ayolo_xf_seg): each a virtual block output or
        a plain activation as it lies in the concat buffer (a Bottleneck output with its shortcut added, an up-sampled map).
        Blocks with a shortcut added in their pass, activations that are also read by a pool, an up-sampling, a 3x3 conv or a
        shortcut keep the materialised pass.  Measured per layer: profiles/r04_xf_forward_sweep.txt (the pair of launches costs
        1.2-2.5x the fused one; it loses only where the channel table and the per-channel-tile repetition of the transform
        weigh in: Cin x channel tiles > 1024)."""
        if not XF_ON_LOAD or self.dt != torch.float16:
            return
        # the invariant this pass stands on: every forward op that reads an activation registered the read (_read) right behind
        # its append -- a new reader kind added without it must fail here, not compute on an unmaterialised activation
        for i, o in enumerate(self.fwd):
            if (o.kind & 0xff) in (OP_CONV_FWD, OP_MAXPOOL_FWD, OP_UPSAMPLE_FWD, OP_SPPF_FWD) and (i + 1) not in self._reads_at:
                raise AssertionError(f"forward op {i} (kind {o.kind & 0xff}) reads an activation without a _read registration")
        for rd in self._reads:
            root, rlo, rhi, kind, c = rd
            if kind != "conv" or not c["pointwise"] or c["cx"]["segs"] is not None:
                continue
            if c["cin"] * -(-c["cout"] // 128) > 1024 or c["cin"] % 8 or c["cx"]["xs_off"] is None:
                continue
            prods = sorted((P for P in self._producers if id(P["a"].root) == root and rlo <= P["a"].c0 and P["a"].c0 + P["a"].C <= rhi),
                           key=lambda P: P["a"].c0)
            # segments of [rlo, rhi): (lo, hi, producer or None = plain)
            segs, cur = [], rlo
            for P in prods:
                lo, hi = P["a"].c0, P["a"].c0 + P["a"].C
                if lo < cur:
                    segs = None
                    break
                only = [r for r in self._reads if r[0] == root and r[1] < hi and r[2] > lo]
                virt = (not P["residual"]) and len(only) == 1 and only[0] is rd
                if lo > cur:
                    segs.append([cur, lo, None])
                segs.append([lo, hi, P if virt else None])
                cur = hi
            if segs is None:
                continue
            if cur < rhi:
                segs.append([cur, rhi, None])
            merged = []
            for sg in segs:                                   # adjacent plain pieces are one plain segment
                if merged and merged[-1][2] is None and sg[2] is None:
                    merged[-1][1] = sg[1]
                else:
                    merged.append(sg)
            if len(merged) > 2 or not any(sg[2] is not None for sg in merged):
                continue
            # two segments: the weight gradient runs as one job per segment over 128-column dw tiles -- a segment that is not a
            # whole number of tiles would spend up to half of its MFMA / DMA work on padding (measured: folding the 32- / 64-channel
            # halves of the early C3 blocks cost the step what the saved passes bought, profiles/r04_ab_xf_two_segments.txt)
            if len(merged) == 2 and ((merged[0][1] - rlo) % 32 or (XF_WGRAD_ON_LOAD and any((hi - lo) % 128 for lo, hi, _ in merged))):
                continue
            if any((hi - lo) % 8 for lo, hi, _ in merged):
                continue
            cin = c["cin"]
            xs = self.small.view(c["cx"]["xs_off"], 2 * cin)
            scale_all, shift_all = xs[:cin], xs[cin:]
            xin = c["x"]                                      # the consumer's materialised input (for plain segments)
            ld_plain = ops.nhwc_info(xin.t)[4]
            wsegs, bits, ptrs, fins = [], 0, [], []
            for k, (lo, hi, P) in enumerate(merged):
                c0, C = lo - rlo, hi - lo
                if P is None:
                    t = xin.t[:, c0:c0 + C]
                    wsegs.append((t, ld_plain, c0, C, None))
                    ptrs.append((t, ld_plain, C))
                    continue
                co = P["co"]
                assert co == C
                sm = self.small.view(P["sm_off"], 4 * co)
                scale, shift = scale_all[c0:c0 + C], shift_all[c0:c0 + C]
                o, bn = P["op"], P["bn"]
                stats_ptr, smean, sinv = o.p[2], o.p[7], o.p[8]            # bound by the late closures of _conv_group
                assert (o.kind & 0xff) == OP_BN_TRAIN_ACT and stats_ptr and smean and sinv
                fins.append(dict(stats=stats_ptr, reps=P["R"], sld=P["ldz"], C=co, c0=c0, count=float(P["npix"]), bn=bn, smean=smean, sinv=sinv,
                                 op=o, scale=scale, shift=shift))
                bits |= ((1 if P["act"] else 0) | 2) << (2 * k)
                wsegs.append((P["z"], P["ldz"], c0, C, (scale, shift, P["act"])))
                ptrs.append((P["z"], P["ldz"], C))
            op = c["op"]
            # the finalize of the folded blocks: inside the reader's launch (every workgroup derives scale / shift in its prologue,
            # workgroup 0 writes the vectors and the saved / running statistics) unless the reader may have to be cut into batch
            # halves (2 GiB tensors): then each block keeps a finalize launch of its own
            B_, _, H_, W_ = xin.t.shape
            small_enough = B_ * H_ * W_ * max(max(pt[1] for pt in ptrs), c["cout"]) * 4 < (1 << 31) - 4096
            if not XF_WGRAD_ON_LOAD and not small_enough:
                continue                                      # (store-back cannot follow a batch split; nothing was modified yet)
            self.xf_layers += len(fins)
            if XF_FINALIZE_IN_READER and small_enough:
                from ._lib import XfFin
                farr = (XfFin * len(fins))()
                for k, f in enumerate(fins):
                    bn = f["bn"]
                    farr[k].stats, farr[k].reps, farr[k].sld, farr[k].C, farr[k].c0, farr[k].count = f["stats"], f["reps"], f["sld"], f["C"], f["c0"], f["count"]
                    farr[k].gamma = bn.weight.data_ptr() if bn.weight is not None else None
                    farr[k].beta = bn.bias.data_ptr() if bn.bias is not None else None
                    farr[k].eps, farr[k].momentum = bn.eps, bn.momentum
                    farr[k].running_mean = bn.running_mean.data_ptr() if bn.running_mean is not None else None
                    farr[k].running_var = bn.running_var.data_ptr() if bn.running_var is not None else None
                    farr[k].save_mean, farr[k].save_invstd = f["smean"], f["sinv"]
                    f["op"].kind = 0                             # no launch of its own
                self.keep.append(farr)
                op.p[9] = ctypes.addressof(farr)
                op.i[6] = len(fins)
            else:
                for f in fins:
                    bn = f["bn"]
                    new = _op(OP_BN_FINALIZE, i=(f["reps"], f["C"], f["sld"]), d=(f["count"],), f=(bn.eps, bn.momentum),
                              p=(f["stats"], bn.weight, bn.bias, bn.running_mean, bn.running_var, f["smean"], f["sinv"], f["scale"], f["shift"]))
                    ctypes.memmove(ctypes.addressof(f["op"]), ctypes.addressof(new), ctypes.sizeof(Op))    # in place: self.fwd holds this object
            op.p[0] = ptrs[0][0].data_ptr()
            op.conv.ldx = ptrs[0][1]
            op.p[6], op.p[7] = scale_all.data_ptr(), shift_all.data_ptr()
            op.i[3] = bits
            if len(ptrs) > 1:
                op.p[8] = ptrs[1][0].data_ptr()
                op.i[4], op.i[5] = ptrs[1][1], ptrs[0][2]
            if XF_WGRAD_ON_LOAD:
                c["cx"]["segs"] = wsegs
            else:
                op.p[10] = xin.t.data_ptr()                   # store-back: the first channel tile writes the activation it forms
                op.i[7] = ld_plain
                c["cx"]["segs"] = []                          # (marks the consumer as folded; the weight gradient reads xin as before)

    def _batched_casts(self) -> List[Op]:
        """All per-layer fp32 -> compute-dtype weight casts (and transposes) as ONE launch: the job table lives in
        device memory and is written once here (parameter storages are stable for the lifetime of the plan)."""
        if not self.casts:
            return []
        import numpy as np
        job_t = np.dtype([("w32", "<u8"), ("w", "<u8"), ("wt", "<u8"), ("Cout", "<i4"), ("taps", "<i4"), ("Cin", "<i4"),
                          ("Cout_pad", "<i4"), ("Cin_pad", "<i4"), ("wt_ld", "<i4")])
        jobs = np.zeros(len(self.casts), dtype=job_t)
        for k, o in enumerate(self.casts):
            jobs[k] = (o.p[0] or 0, o.p[1] or 0, o.p[2] or 0, o.i[0], o.i[1] * o.i[2], o.i[3], o.i[4], o.i[5], o.i[7])
        tab = torch.from_numpy(jobs.view(np.uint8).copy()).to(self.device)
        self.keep.append(tab)
        return [_op(OP_CAST_WEIGHTS, i=(len(self.casts), self.casts[0].i[6]), p=(tab,))]

    # ------------------------------------------------------------------ composite blocks
    def _bottleneck(self, b: Bottleneck, x: Act, dst: Optional[Act]) -> Act:
        y1 = self._conv_block(b.cv1, x, None)
        return self._conv_block(b.cv2, y1, dst, residual=x if b.add else None)

    def _c3(self, m: C3, x: Act, dst: Optional[Act]) -> Act:
        h = m.cv1.conv.out_channels
        _, _, H, W = x.t.shape
        cat = self._new_act(2 * h, H, W)
        n = len(m.m)
        d1 = cat.slice(0, h) if n == 0 else None
        if self._mergeable(m.cv1, m.cv2):
            t = self._conv_group([m.cv1, m.cv2], x, [d1, cat.slice(h, 2 * h)])[0]
        else:
            t = self._conv_block(m.cv1, x, d1)
        for bi, b in enumerate(m.m):
            t = self._bottleneck(b, t, cat.slice(0, h) if bi == n - 1 else None)
        if not self._mergeable(m.cv1, m.cv2):
            self._conv_block(m.cv2, x, cat.slice(h, 2 * h))
        return self._conv_block(m.cv3, cat, dst)

    def _pool(self, k: int, src: Act, dst: Act) -> None:
        B, C, H, W = src.t.shape
        arg = torch.empty((B, H, W, C), dtype=torch.uint8, device=self.device)
        self.keep.append(arg)
        code = ops.dtype_code(self.dt)
        lds, ldd = ops.nhwc_info(src.t)[4], ops.nhwc_info(dst.t)[4]
        self.fwd.append(_op(OP_MAXPOOL_FWD, i=(code, lds, ldd, B, H, W, C, k), p=(src.t, dst.t, arg)))
        self._read(src, "pool")

        def emit():
            dy, dx = dst.grad(), src.grad()
            self.bwd.append(_op(OP_MAXPOOL_BWD, i=(code, ops.nhwc_info(dy)[4], ops.nhwc_info(dx)[4], B, H, W, C, k, int(src.is_init())),
                                p=(arg, dy, dx)))
            self._gw(src, False)
            src.mark_init()

        self.bwd_emitters.append(emit)

    def _sppf(self, m: SPPF, x: Act, dst: Optional[Act]) -> Act:
        h = m.cv1.conv.out_channels
        _, _, H, W = x.t.shape
        cat = self._new_act(4 * h, H, W)
        self._conv_block(m.cv1, x, cat.slice(0, h))
        k = m.pool.kernel_size
        code = ops.dtype_code(self.dt)
        if SPPF_FUSED and k == 5 and self.dt == torch.float16 and _lib.lib().ayolo_sppf_pool_supported(code, H, W, h):
            B = self.B
            arg = torch.empty((3, B, H, W, h), dtype=torch.uint8, device=self.device)
            self.keep.append(arg)
            ld = ops.nhwc_info(cat.t)[4]
            self.fwd.append(_op(OP_SPPF_FWD, i=(code, ld, B, H, W, h), p=(cat.t, arg)))
            self._read(cat.slice(0, h), "pool")

            def emit():
                if not cat.is_init():
                    raise RuntimeError("plan: SPPF concat gradient incomplete before the pool cascade's backward")
                dcat = cat.grad()
                self.bwd.append(_op(OP_SPPF_BWD, i=(code, ops.nhwc_info(dcat)[4], B, H, W, h), p=(arg, dcat)))
                self._gw(cat.slice(0, h), False)

            self.bwd_emitters.append(emit)
        else:
            for j in range(3):
                self._pool(k, cat.slice(j * h, (j + 1) * h), cat.slice((j + 1) * h, (j + 2) * h))
        return self._conv_block(m.cv2, cat, dst)

    def _upsample(self, x: Act, dst: Optional[Act]) -> Act:
        B, C, H, W = x.t.shape
        out = dst if dst is not None else self._new_act(C, 2 * H, 2 * W)
        code = ops.dtype_code(self.dt)
        self.fwd.append(_op(OP_UPSAMPLE_FWD, i=(code, ops.nhwc_info(x.t)[4], ops.nhwc_info(out.t)[4], B, H, W, C), p=(x.t, out.t)))
        self._read(x, "upsample")

        def emit():
            dy, dx = out.grad(), x.grad()
            self.bwd.append(_op(OP_UPSAMPLE_BWD, i=(code, ops.nhwc_info(dy)[4], ops.nhwc_info(dx)[4], B, H, W, C, int(x.is_init())),
                                p=(dy, dx)))
            self._gw(x, False)
            x.mark_init()

        self.bwd_emitters.append(emit)
        return out

    def _head(self, head: YOLOHead, xs: List[Act]) -> None:
        dt, dev = self.dt, self.device
        if len(xs) > 4:
            raise PlanUnsupported("more than four detection levels")      # dy override slots of a grouped weight gradient
        for lvl, x in enumerate(xs):
            conv = head.conv[lvl]
            if not isinstance(conv, nn.Conv2d) or conv.weight.dtype != torch.float32:
                raise PlanUnsupported("non-standard head conv")
            Cout, Cin = conv.weight.shape[:2]
            cp = F_._round_up(Cout, 8)
            B, _, H, W = x.t.shape
            geo = F_._Geometry(tuple(x.t.shape), conv.weight.shape, (1, 1), (0, 0), dt)
            # a grouped weight-gradient job whose dy arrives through an override slot cannot be cut into batch halves
            # (csrc/conv.hip wgrad_halves): a level over the 2 GiB descriptor range goes to the per-module path
            es_ = 2 if dt == torch.float16 else 4
            if B * H * W * max(ops.nhwc_info(x.t)[4], cp) * es_ >= (1 << 31) - 4096:
                raise PlanUnsupported("head level over the 2 GiB buffer-descriptor range")
            wc = torch.empty((cp, 1, 1, Cin), dtype=dt, device=dev)
            wt = torch.empty((Cin, 1, 1, cp), dtype=dt, device=dev)
            buf = torch.empty((B, H, W, cp), dtype=torch.float32, device=dev)
            self.keep += [wc, wt, buf]
            self.casts.append(_op(OP_CAST_WEIGHT, i=(Cout, 1, 1, Cin, cp, Cin, ops.dtype_code(dt)), p=(conv.weight, wc, wt)))
            ldx = ops.nhwc_info(x.t)[4]
            op_head = _op(OP_CONV_FWD, i=(EPI_HEAD, 1, head.no), p=(x.t, wc, buf, None, conv.bias, None), conv=geo.desc(dt, ldx, cp))
            self.fwd.append(op_head)
            cx = dict(x=x.t, ldx=ldx, xf=None, segs=None, xs_off=self.small.request(2 * Cin))
            self._read(x, "conv", dict(op=op_head, cx=cx, pointwise=True, cin=Cin, cout=cp, x=x))
            gw_off = self._register_param(conv.weight, cp * Cin, lambda b, Cout=Cout, Cin=Cin: b.view(-1, Cin)[:Cout].view(Cout, Cin, 1, 1))
            gb_off = self._register_param(conv.bias, Cout, lambda b: b) if conv.bias is not None else None
            self.raw_specs.append((buf, (B, head.na, H, W, head.no), (H * W * cp, head.no, W * cp, cp, 1), gb_off, Cout))
            npix = B * H * W
            code = ops.dtype_code(dt)

            def emit(x=x, geo=geo, wt=wt, cp=cp, Cin=Cin, Cout=Cout, B=B, H=H, W=W, gw_off=gw_off, gb_off=gb_off, ldx=ldx, npix=npix, lvl=lvl, cx=cx):
                ga = self.gradarena
                dz = self._dz(npix * cp)
                op = _op(OP_HEAD_GRAD_PACK, i=(B, head.na, H, W, head.no, code, cp),
                         p=(None, dz, ga.view(gb_off, Cout) if gb_off is not None else None))
                self.draw_ops.append(op)
                self.bwd.append(op)
                dx = x.grad()
                self.bwd.append(_op(OP_CONV_DGRAD, i=(int(x.is_init()),), p=(dz, wt, dx),
                                    conv=geo.desc(dt, ops.nhwc_info(dx)[4], cp, cout=cp)))
                self._gw(x, True)
                x.mark_init()
                # dy of a head level changes from step to step (the fused loss hands its own buffer over): override slot = level
                self._wgrad_jobs_of(cx, geo.desc(dt, ldx, cp, cout=cp), dz, gw_off, cp * Cin, dy_slot=lvl)
                self._wrote(gb_off, Cout, at=0)          # bias gradient: pack op, or the fused loss before the list runs

            self.bwd_emitters.append(emit)

    # ------------------------------------------------------------------ whole model
    def _compile(self) -> None:
        model = self.model
        layers = list(model.model)
        routes = model.routes
        n = len(layers)
        # -- shape inference (channels, spatial) per layer output
        ch: List[int] = []
        hw: List[Tuple[int, int]] = []
        for i, m in enumerate(layers):
            frm = routes[i]
            fl = frm if isinstance(frm, list) else [frm]
            srcs = [(i + f) if f < 0 else f for f in fl]
            cin = [self.Cimg if s < 0 else ch[s] for s in srcs]
            sin = [(self.H, self.W) if s < 0 else hw[s] for s in srcs]
            if isinstance(m, Conv):
                if not isinstance(m.conv, nn.Conv2d):
                    raise PlanUnsupported("decomposed Conv block (per-module path)")
                s = _pair(m.conv.stride)
                k = _pair(m.conv.kernel_size)
                p = _pair(m.conv.padding)
                ch.append(m.conv.out_channels)
                hw.append(((sin[0][0] + 2 * p[0] - k[0]) // s[0] + 1, (sin[0][1] + 2 * p[1] - k[1]) // s[1] + 1))
            elif isinstance(m, C3):
                ch.append(m.cv3.conv.out_channels); hw.append(sin[0])
            elif isinstance(m, SPPF):
                ch.append(m.cv2.conv.out_channels); hw.append(sin[0])
            elif isinstance(m, UpSample):
                ch.append(cin[0]); hw.append((sin[0][0] * 2, sin[0][1] * 2))
            elif isinstance(m, Concat):
                if m.dimension != 1:
                    raise PlanUnsupported("concat on a non-channel dim")
                ch.append(sum(cin)); hw.append(sin[0])
            elif isinstance(m, YOLOHead):
                ch.append(0); hw.append((0, 0))
            else:
                raise PlanUnsupported(type(m).__name__)
        # -- concat destinations: producer layer -> (concat layer, channel offset)
        dest: Dict[int, Tuple[int, int]] = {}
        for j, m in enumerate(layers):
            if isinstance(m, Concat):
                off = 0
                fl = routes[j]
                for f in fl:
                    s = (j + f) if f < 0 else f
                    if s in dest or s < 0:
                        raise PlanUnsupported("layer feeds two concats")
                    dest[s] = (j, off)
                    off += ch[s]
        cat_bufs: Dict[int, Act] = {}
        outs: List[Optional[Act]] = []
        for i, m in enumerate(layers):
            frm = routes[i]
            fl = frm if isinstance(frm, list) else [frm]
            srcs = [(i + f) if f < 0 else f for f in fl]
            xin = [None if s < 0 else outs[s] for s in srcs]
            dst = None
            if i in dest:
                j, off = dest[i]
                if j not in cat_bufs:
                    cat_bufs[j] = self._new_act(ch[j], hw[j][0], hw[j][1])
                dst = cat_bufs[j].slice(off, off + ch[i])
            if isinstance(m, Conv):
                out = self._conv_block(m, xin[0], dst, image=(srcs[0] < 0))
            elif isinstance(m, C3):
                out = self._c3(m, xin[0], dst)
            elif isinstance(m, SPPF):
                out = self._sppf(m, xin[0], dst)
            elif isinstance(m, UpSample):
                out = self._upsample(xin[0], dst)
            elif isinstance(m, Concat):
                out = cat_bufs[i]
            elif isinstance(m, YOLOHead):
                self._head(m, xin)
                out = None
            outs.append(out)
        # -- arenas, late binding, backward emission (reverse order)
        dev = self.device
        for ar in (self.stats, self.sums, self.small, self.gradarena):
            ar.allocate(dev)
        for fn in self.late:
            fn()
        self._fold_bn_act_into_consumers()
        # all accumulators are zeroed at the START OF THE FORWARD (one fill each): the gradient arena too, because the
        # fused loss adds the head bias gradients into it before the backward list runs
        head_ops = [_op(OP_MEMSET, l=(self.stats.buf.numel() * 8,), p=(self.stats.buf,)),
                    _op(OP_MEMSET, l=(self.sums.buf.numel() * 8,), p=(self.sums.buf,)),
                    _op(OP_MEMSET, l=(self.gradarena.buf.numel() * 4,), p=(self.gradarena.buf,))]
        nhead = len(head_ops) + (1 if self.casts else 0)
        self.fwd = head_ops + self._batched_casts() + self.fwd
        self.bwd = []
        for emit in reversed(self.bwd_emitters):
            emit()
        self._group_wgrads()
        self._fold_bn_reduce()
        self.fwd_arr = (Op * len(self.fwd))(*self.fwd)
        self.bwd_arr = (Op * len(self.bwd))(*self.bwd)
        # pointer-patch slots inside the arrays (ctypes copies structs into the array)
        self.pack_idx = next(k for k, o in enumerate(self.fwd) if o is self.pack_op)
        self.draw_idx = [next(k for k, o in enumerate(self.bwd) if o is d) for d in self.draw_ops]
        # draw ops were emitted in reverse level order
        self.draw_levels = list(reversed(range(len(self.raw_specs))))
        self.param_ptrs = tuple(p.data_ptr() for p in self.params)
        self.fwd_sync_idx = [(next(k for k, o in enumerate(self.fwd) if o is c), off, n) for c, off, n in self.fwd_sync]
        self.buckets = self._make_buckets()

    def _fold_bn_reduce(self) -> None:
        """Move the first pass of the BatchNorm + activation backward (sum du, sum du * xhat over all pixels) into the
        epilogue of the dgrad that PRODUCES the layer's output gradient: that kernel has every value of da in registers, so
        instead of a pass that re-reads da and z from HBM it reads z once and the sums cost no launch of their own.  Valid
        when the LAST writer of the layer's da (backward order; fan-out tensors are accumulated by several ops) is a conv
        dgrad whose dx covers the layer's channels -- it then holds the TOTAL gradient, and the sums are what the separate
        pass would compute from the same rounded values.  A dgrad serves up to two layers side by side in its dx (C3's
        concat buffer: the last Bottleneck's output and the cv2 half).  Layers fed by pool / upsample / shortcut-copy
        backward ops keep the separate pass; so do fp32 plans (the exact-parity mode)."""
        if not BN_REDUCE_IN_DGRAD or self.dt != torch.float16:
            return
        from ._lib import BnSeg
        by_op: Dict[int, list] = {}
        for L in self._bn_layers:
            a = L["a"]
            root, lo, hi = id(a.root), a.c0, a.c0 + L["C"]
            ws = [w for w in self._gwrites if w[1] == root and w[2] < hi and w[3] > lo and w[0] < L["reduce"]]
            if not ws:
                continue
            last = max(ws, key=lambda w: w[0])
            if not last[4] or not (last[2] <= lo and hi <= last[3]):
                continue
            by_op.setdefault(last[0], []).append((lo - last[2], L))
        for idx, items in by_op.items():
            items.sort(key=lambda t: t[0])
            op = self.bwd[idx]
            segs = [(c0, L) for c0, L in items]
            ok = (len(segs) <= 2 and all(c0 % 8 == 0 and L["C"] % 8 == 0 for c0, L in segs)
                  and len({L["act"] for _, L in segs}) == 1 and len({L["R"] for _, L in segs}) == 1
                  and (len(segs) == 1 or (segs[1][0] % 32 == 0 and segs[0][0] + segs[0][1]["C"] <= segs[1][0])))
            if not ok:
                continue
            arr = (BnSeg * len(segs))()
            for k, (c0, L) in enumerate(segs):
                arr[k].z, arr[k].mean_invstd = L["z"].data_ptr(), L["sm"].data_ptr()
                arr[k].gamma = L["gamma"].data_ptr() if L["gamma"] is not None else None
                arr[k].beta = L["beta"].data_ptr() if L["beta"] is not None else None
                arr[k].sums = L["sums"].data_ptr()
                arr[k].ldz, arr[k].c0, arr[k].C = L["ldz"], c0, L["C"]
                self.bwd[L["reduce"]].kind = 0                       # the separate reduce pass is not needed
                self.bn_in_dgrad += 1
            self.keep.append(arr)
            op.p[3] = ctypes.addressof(arr)
            op.i[1], op.i[2], op.i[3] = len(segs), segs[0][1]["act"], segs[0][1]["R"]

    def _make_buckets(self, target_bytes: Optional[int] = None) -> List[Tuple[int, int, int]]:
        """Gradient buckets for data-parallel training: [(ready, lo, hi)] -- the arena range [lo, hi) is complete once
        backward op `ready` has been ENQUEUED (weight gradients run on the side stream).  The arena is laid out in forward
        order and backward runs in reverse, so buckets are cut from the top of the arena downwards, ~6 per model (>= 4 MB:
        an xGMI ring all-reduce is per-link bound and latency-dominated below that)."""
        total = self.gradarena.total
        if target_bytes is None:
            target_bytes = int(_os.environ.get("AYOLO_BUCKET_MB", "0")) << 20 or max(4 << 20, total * 4 // 6)
        done = sorted(self.grad_done, key=lambda t: -t[1])            # highest arena offset (last layers) first
        out: List[Tuple[int, int, int]] = []
        hi, ready = total, -1
        for k, (idx, off, n) in enumerate(done):
            ready = max(ready, idx)
            last = k == len(done) - 1
            if (hi - off) * 4 >= target_bytes or last:
                lo = 0 if last else off
                out.append((ready, lo, hi))
                hi = off
        # The LAST bucket ends with the first layer, whose gradients are the last thing backward produces (the stem's weight
        # gradient runs at the very end of the list on the largest activations of the net): as one bucket its all-reduce can
        # only start after the backward window has closed (round 2: bucket 5 at +10.02 ms of a 9.98 ms window).  Split it: the
        # part that is complete BEFORE the first layer's own ops leaves early, what remains exposed is the first layer's few
        # KB (a latency-bound collective either way).
        if out:
            ready_l, lo_l, hi_l = out[-1]
            inside = [(idx, off, n) for idx, off, n in done if lo_l <= off < hi_l]
            first_ops = max(idx for idx, _, _ in inside)                        # the op that completes the bucket
            early = [(idx, off, n) for idx, off, n in inside if idx < first_ops - 2]   # not the first layer's wgrad / BN apply
            if early and len(early) < len(inside):
                cut = min(off for _, off, _ in early)
                late_hi = max(off + n for idx, off, n in inside if idx >= first_ops - 2)
                if late_hi <= cut and (hi_l - cut) * 4 >= 64 << 10:
                    out[-1] = (max(idx for idx, _, _ in early), cut, hi_l)
                    out.append((ready_l, lo_l, cut))
        fixed, run = [], -1
        for ready, lo, hi in out:                                      # ready indices must not decrease bucket to bucket
            run = max(run, ready)
            fixed.append((run, lo, hi))
        return fixed

    # ------------------------------------------------------------------ execution
    def valid_for(self, params_ptrs) -> bool:
        return params_ptrs == self.param_ptrs

    def run_forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        self._x_keep = x
        # Bound the host's run-ahead: the ROCm runtime degrades badly (long stalls with the GPU idle) once several
        # thousand launches are queued, and with one host call per pass the host is ~4 steps ahead within
        # milliseconds.  Waiting for the PREVIOUS step's forward (long finished when the GPU is the bottleneck)
        # keeps at most ~1 step in flight and never idles the GPU.
        prev = getattr(self, "_fwd_done", None)
        if prev is not None:
            prev.synchronize()
        self.generation = getattr(self, "generation", 0) + 1
        self.fwd_arr[self.pack_idx].p[0] = x.data_ptr()
        sync = getattr(self.model, "_ayolo_grad_sync", None)
        st = torch.cuda.current_stream().cuda_stream
        if sync is not None and getattr(sync, "sync_bn", False) and sync.active():
            # SyncBatchNorm (train_model_builder.py:135-136): every layer's batch statistics are averaged over the ranks
            # between its conv and its normalise pass -- the list runs in one segment per BN layer
            a = 0
            for k, off, n in self.fwd_sync_idx:
                self._run(self.fwd_arr, a, k + 1, st, "forward")
                sync.average_now(self.stats.view(off, n))
                a = k + 1
            self._run(self.fwd_arr, a, len(self.fwd), st, "forward")
            # The kernels scale the running variance by n / (n - 1) with the LOCAL sample count n; torch's SyncBatchNorm uses
            # the global count N = world * n.  Add the difference momentum * var * (N/(N-1) - n/(n-1)) (var from the saved
            # inverse standard deviation) -- a handful of tiny ops per layer, on a path that already runs ~100 blocking
            # collectives per step.
            world = sync.world
            if world > 1:
                with torch.no_grad():
                    for bn, sm_off, co, n in self.bn_sync_fix:
                        if bn.running_var is None or n < 2 or bn.momentum is None:     # (cumulative-average BN never plans)
                            continue
                        inv = self.small.view(sm_off, 4 * co)[co:2 * co]
                        N = world * n
                        var = (inv.pow(-2) - bn.eps).clamp_min_(0.0)                   # rounding must not make it negative
                        bn.running_var.add_(var * (bn.momentum * (N / (N - 1) - n / (n - 1))))
        else:
            self._run(self.fwd_arr, 0, len(self.fwd), st, "forward")
        if self.bn_counters:
            torch._foreach_add_(self.bn_counters, 1)
        _lib.bump_versions(self.bn_buffers)                     # written through raw pointers by k_bn_train_act
        self._fwd_done = torch.cuda.Event()
        self._fwd_done.record()
        raws = [spec[0].as_strided(spec[1], spec[2]) for spec in self.raw_specs]
        for r, (buf, _, _, gb_off, cout) in zip(raws, self.raw_specs):
            # the fused loss may hand its gradient over in dz layout, adding the bias gradient straight into the arena
            slot = self.gradarena.view(gb_off, cout) if gb_off is not None else None
            r._ayolo_head = (buf.shape[-1], self.dt, slot)
        return raws

    def _run(self, arr, a: int, b: int, stream: int, what: str, no_join: bool = False) -> None:
        if b <= a:
            return
        if getattr(self, "collect_times", False) and a == 0 and b == len(arr):
            # measurement mode (bench.py): per-op durations in situ, events on each op's own stream; synchronous
            ms = (c_float * (b - a))()
            _lib.check(_lib.lib().ayolo_run_ops_timed(arr, b - a, stream, ms), f"ayolo_run_ops_timed({what})")
            self.op_times.setdefault(what, []).append(np.frombuffer(ms, dtype=np.float32).copy())
            return
        _lib.check(_lib.lib().ayolo_run_ops_ex(ctypes.byref(arr, a * ctypes.sizeof(Op)), b - a, stream, 1 if no_join else 0),
                   f"ayolo_run_ops({what})")

    def op_costs(self, what: str):
        """Algorithmic cost of every op of the forward / backward list, for roofline accounting:
        [(family, bytes, flop)] with bytes = the tensors an op must read and write once (activations in the compute dtype,
        fp32 weight gradients), flop = 2 * MAC of the convolutions."""
        es = 2 if self.dt == torch.float16 else 4
        out = []
        for k_op, o in enumerate(self.fwd if what == "forward" else self.bwd):
            kind = o.kind & 0xff
            if kind == OP_WGRAD_GROUP and what != "forward":
                by, fl, _ = self.wgroup_costs[k_op]
                out.append(("conv_wgrad", by, fl))
                continue
            d = o.conv
            macs = d.B * d.Ho * d.Wo * d.Cout * d.kh * d.kw * d.Cin
            xin, yout, wts = d.B * d.H * d.W * d.Cin, d.B * d.Ho * d.Wo * d.Cout, d.Cout * d.kh * d.kw * d.Cin
            if kind == OP_CONV_FWD:
                yes = 4 if o.i[0] == EPI_HEAD else es
                back = 0
                if o.p[6] and o.p[10]:               # transform on load with store-back: the virtual segments' activation is written once
                    c_first = o.i[5] if o.p[8] else d.Cin
                    cv = (c_first if (o.i[3] >> 1) & 1 else 0) + ((d.Cin - c_first) if (o.i[3] >> 3) & 1 else 0)
                    back = es * d.B * d.H * d.W * cv
                out.append(("conv_fwd", es * (xin + wts) + yes * yout + back, 2.0 * macs))
            elif kind == OP_CONV_DGRAD:
                zb = 0
                if o.i[1] > 0:                      # BatchNorm-backward sums in the epilogue: z of the served layers read once
                    from ._lib import BnSeg
                    segs = ctypes.cast(o.p[3], ctypes.POINTER(BnSeg))
                    zb = es * d.B * d.H * d.W * sum(segs[k].C for k in range(o.i[1]))
                out.append(("conv_dgrad", es * (yout + wts + xin * (2 if o.i[0] else 1)) + zb, 2.0 * macs))
            elif kind == OP_CONV_WGRAD:
                out.append(("conv_wgrad", es * (xin + yout) + 4 * wts, 2.0 * macs))
            elif kind == OP_STEM_BN_WGRAD:          # x, da and z read once, dw written
                out.append(("conv_wgrad", es * (xin + 2 * yout) + 4 * wts, 2.0 * macs))
            elif kind == OP_BN_TRAIN_ACT:
                out.append(("bn_act_fwd", es * o.l[0] * o.i[3] * (3 if o.p[9] else 2), 0.0))
            elif kind == OP_BN_FINALIZE:
                out.append(("bn_finalize", 8.0 * o.i[0] * 2 * o.i[1] + 4.0 * 6 * o.i[1], 0.0))
            elif kind == OP_BN_BWD_REDUCE:
                out.append(("bn_bwd_reduce", es * o.l[0] * o.i[3] * 2, 0.0))
            elif kind == OP_BN_BWD_APPLY:
                # + the shortcut gradient written (and, accumulating, read) by the same pass
                out.append(("bn_bwd_apply", es * o.l[0] * o.i[4] * (3 + ((2 if o.i[8] else 1) if o.p[10] else 0)), 0.0))
            elif kind == OP_BN_BWD_APPLY2:
                out.append(("bn_bwd_apply", es * o.l[0] * (o.i[5] + o.i[6]) * 3, 0.0))
            elif kind in (OP_MAXPOOL_FWD, OP_UPSAMPLE_FWD):
                n = o.i[3] * o.i[4] * o.i[5] * o.i[6] * (4 if kind == OP_UPSAMPLE_FWD else 1)
                out.append(("pool_upsample", es * n * (1.25 if kind == OP_UPSAMPLE_FWD else 2) + (n if kind == OP_MAXPOOL_FWD else 0), 0.0))
            elif kind in (OP_MAXPOOL_BWD, OP_UPSAMPLE_BWD):
                n = o.i[3] * o.i[4] * o.i[5] * o.i[6]
                out.append(("pool_upsample", es * n * (5 if kind == OP_UPSAMPLE_BWD else 2) + (n if kind == OP_MAXPOOL_BWD else 0), 0.0))
            elif kind == OP_SPPF_FWD:                 # x read, three outputs + three position planes written
                n = o.i[2] * o.i[3] * o.i[4] * o.i[5]
                out.append(("pool_upsample", es * n * 4 + 3 * n, 0.0))
            elif kind == OP_SPPF_BWD:                 # four gradient slices + three position planes read, d(x) written
                n = o.i[2] * o.i[3] * o.i[4] * o.i[5]
                out.append(("pool_upsample", es * n * 5 + 3 * n, 0.0))
            elif kind == OP_COPY2D:
                out.append(("copy", es * o.l[0] * o.i[3] * (3 if o.i[4] else 2), 0.0))
            elif kind == OP_PACK_INPUT:
                n = o.i[0] * o.i[2] * o.i[3]
                out.append(("pack_input", n * (4 * o.i[1] + es * o.i[5]), 0.0))
            elif kind == OP_MEMSET:
                out.append(("fill", float(o.l[0]), 0.0))
            elif kind == OP_CAST_WEIGHTS:
                n = sum(c.i[4] * c.i[1] * c.i[2] * c.i[5] for c in self.casts)
                out.append(("cast_weights", n * (4 + 2 * es), 0.0))
            else:
                out.append(("other", 0.0, 0.0))
        return out

    @staticmethod
    def _storage_refs(t: torch.Tensor) -> int:
        """Tensors (and Python storage wrappers) that share `t`'s storage right now, `t` included; -1 if this torch cannot
        tell (then pooled buffers are never reused)."""
        fn = getattr(torch._C, "_storage_Use_Count", None)
        return int(fn(t.untyped_storage()._cdata)) if fn is not None else -1

    def _grad_out_buffer(self):
        """The flat buffer this step's gradients are handed out in.  A pool of two persistent buffers: the gradient addresses
        then repeat from step to step, which is what lets optim.SGD keep its job table instead of rebuilding and re-uploading
        it every step.  A pooled buffer is reused only when it is PROVABLY unreferenced -- its storage (and the storage of
        every side buffer handed out with it) is back at the reference count it had when only the pool held it.  That is the
        case after `zero_grad(set_to_none=True)` (yolo_trainer.py:336) once nobody else holds a gradient of that backward: a
        `.grad` that is still set (gradient accumulation), a `torch.autograd.grad` result, a list of gradients kept across
        steps or a hook that stashed one all keep the count up, and a fresh buffer is allocated instead -- gradients returned
        by one backward are never overwritten by a later one (autograd semantics)."""
        pool = self.__dict__.setdefault("_flat_pool", [])
        for entry in pool:
            flat, extra, base = entry
            if base[0] > 0 and self._storage_refs(flat) == base[0] and all(self._storage_refs(e) == base[1] for e in extra.values()):
                return entry
        flat = torch.empty_like(self.gradarena.buf)
        probe = torch.empty(1, device=flat.device)
        entry = (flat, {}, [self._storage_refs(flat), self._storage_refs(probe)])     # idle counts: the flat buffer / a side buffer
        if len(pool) < 2:
            pool.append(entry)
        return entry

    def _head_wgrad_dy(self, lvl: int, ptr: int) -> None:
        """this step's dz of head level `lvl` -> the weight-gradient launch that carries that level"""
        for k, slot in self.wgroup_slots:
            if slot == lvl:
                self.bwd_arr[k].p[3 + lvl] = ptr

    def run_backward(self, draws: Sequence[Optional[torch.Tensor]]) -> List[torch.Tensor]:
        from .losses import take_packed_head_grad
        keep = []
        late_bias = []
        for idx, lvl in zip(self.draw_idx, self.draw_levels):
            d = draws[lvl]
            buf, shape = self.raw_specs[lvl][:2]
            op, dg = self.bwd_arr[idx], self.bwd_arr[idx + 1]                              # pack, head dgrad
            if idx not in self._head_dz:
                self._head_dz[idx] = (op.p[1], op.p[2])
            dz0, dbias0 = self._head_dz[idx]
            pk = take_packed_head_grad(d, buf.shape[-1], self.dt, buf.data_ptr())
            if pk is not None:
                # gradient already in the head conv's operand layout: skip the pack op, point dgrad / wgrad at it
                op.kind = 0
                dg.p[0] = pk[0].data_ptr()
                self._head_wgrad_dy(lvl, pk[0].data_ptr())
                keep += [pk[0], pk[1]]
                if dbias0 and pk[1].data_ptr() != dbias0:
                    late_bias.append((dbias0, pk[1]))
                continue
            op.kind = OP_HEAD_GRAD_PACK
            dg.p[0] = dz0
            self._head_wgrad_dy(lvl, dz0)
            if d is None:
                d = torch.zeros(shape, dtype=torch.float32, device=self.device)
            if d.dtype != torch.float32 or not d.is_contiguous():
                d = d.float().contiguous()
            keep.append(d)
            op.p[0] = d.data_ptr()
        for ptr, dbias in late_bias:         # bias gradient of a packed head level that is not already in its arena slot
            off = (ptr - self.gradarena.buf.data_ptr()) // 4
            self.gradarena.buf[off:off + dbias.numel()].add_(dbias)
        st = torch.cuda.current_stream().cuda_stream
        sync = getattr(self.model, "_ayolo_grad_sync", None)
        active = sync is not None and sync.active()
        # segment boundaries: after op k ... (kind, payload)
        cuts = []
        if active and getattr(sync, "sync_bn", False):
            cuts += [(k, 0, su) for k, su in self.bwd_sync]
        if active and getattr(sync, "overlap", True):
            cuts += [(ready, 1, (lo, hi)) for ready, lo, hi in self.buckets]
        cuts.sort(key=lambda c: (c[0], c[1]))
        if active and getattr(sync, "measure", False):
            sync.mark_backward_start()
        a = 0
        for k, kind, payload in cuts:
            # AYOLO_RUN_NO_JOIN: the compute stream does not wait for the side-stream weight gradients here; the
            # communication stream does (sync.launch_bucket) -- the all-reduce of a bucket overlaps the rest of backward
            self._run(self.bwd_arr, a, k + 1, st, "backward", no_join=True)
            a = max(a, k + 1)
            if kind == 0:
                sync.average_now(payload)
            else:
                sync.launch_bucket(self.gradarena.buf[payload[0]:payload[1]])
        self._run(self.bwd_arr, a, len(self.bwd), st, "backward")
        self._d_keep = keep
        if active:
            if getattr(sync, "overlap", True):
                sync.wait_all()                                   # compute stream waits for the bucket all-reduces
            else:
                sync.reduce_flat(self.gradarena.buf)              # one blocking all-reduce of the whole arena
        # The arena is scratch that the next forward zeroes: hand out gradients that OWN their memory (autograd steals
        # them as p.grad and may keep them across steps for gradient accumulation) -- one flat copy, views into it.
        flat, extra, _ = self._grad_out_buffer()
        flat.copy_(self.gradarena.buf)
        lo_, hi_ = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        grads = []
        for p in self.params:
            off, n, view_fn = self.param_grad_view[id(p)]
            g = view_fn(flat[off:off + n])
            if not (lo_ <= g.data_ptr() < hi_):
                # not a view (the stem: 3 of its 4 packed input channels are compacted): the copy lives in a persistent
                # buffer of the pool entry too, so that EVERY gradient address repeats from step to step
                keep = extra.get(id(p))
                if keep is None or keep.shape != g.shape or keep.stride() != g.stride():
                    keep = extra[id(p)] = torch.empty_like(g)
                keep.copy_(g)
                g = keep.as_strided(keep.shape, keep.stride())      # a fresh view object: autograd only adopts a gradient
                                                                    # tensor nobody else references (it would clone `keep`)
            if g.stride() != p.stride() and g.is_contiguous(memory_format=torch.channels_last) and p.dim() == 4 \
                    and p.shape[2] == 1 and p.shape[3] == 1:
                g = g.as_strided(p.shape, p.stride())                # 1x1 kernels: same memory, the parameter's strides
            grads.append(g)
        return grads


class _PlanTrainFn(torch.autograd.Function):
    """The plan's activations and its outputs (the raw head tensors are views of plan-owned buffers) live in STATIC
    storage: a later training forward of the same shape overwrites them.  Forward-only calls are fine; a backward
    through an overwritten forward (two micro-batch forwards before the first backward) is detected and raises."""

    @staticmethod
    def forward(ctx, plan: TrainPlan, x: torch.Tensor, *params):
        ctx.plan = plan
        outs = tuple(plan.run_forward(x))
        ctx.generation = plan.generation
        return outs

    @staticmethod
    def backward(ctx, *draws):
        if ctx.generation != ctx.plan.generation:
            raise RuntimeError("ayolov2_amd plan: backward of a forward whose saved activations were overwritten by a later "
                               "forward of the same shape (the plan executor keeps ONE set of static buffers per input "
                               "shape); run backward before the next forward, or set model.use_plan = False")
        if getattr(ctx.plan, "_bwd_generation", None) == ctx.generation:
            # the BatchNorm sums, the gradient arena and the head bias gradients the fused loss adds into it are zeroed once
            # per FORWARD: a second backward through the same forward would accumulate on top of the first
            raise RuntimeError("ayolov2_amd plan: second backward through the same forward (retain_graph / two losses "
                               "back-propagated separately): sum the losses and call backward once, or set "
                               "model.use_plan = False")
        ctx.plan._bwd_generation = ctx.generation
        grads = ctx.plan.run_backward(draws)
        return (None, None) + tuple(grads)


def plan_forward_train(model, x: torch.Tensor):
    """Training forward through the cached plan; returns the list of raw head tensors (or None if unsupported)."""
    dt = torch.float16 if torch.is_autocast_enabled() else torch.float32
    key = (tuple(x.shape), dt, x.device)
    cache = model.__dict__.setdefault("_plans", {})
    plan = cache.get(key)
    if plan is False:
        return None
    if plan is not None and not plan.valid_for(tuple(p.data_ptr() for p in plan.params)):
        plan = None
    if plan is None:
        try:
            plan = TrainPlan(model, tuple(x.shape), dt, x.device)
        except PlanUnsupported:
            cache[key] = False
            return None
        live = [k for k, v in cache.items() if v is not False]
        while len(live) >= MAX_PLANS:            # a plan owns every activation of its shape (GBs): keep the newest few
            cache.pop(live.pop(0))
        cache[key] = plan
    return list(_PlanTrainFn.apply(plan, x, *plan.params))
