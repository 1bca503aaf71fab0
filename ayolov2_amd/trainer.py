"""Data-parallel training glue with the reference's interface for this path
(scripts/train/train_model_builder.py:75-141 ``TrainModelBuilder``, scripts/train/yolo_trainer.py:289-358
``training_step``).  One process per GPU; the plan executor's flat gradient arena is all-reduced over torch.distributed's
"nccl" backend (= RCCL on ROCm, xGMI inside a node) in reverse-layer buckets launched from a communication stream while
the remaining backward kernels run (``FlatGradDDP`` / ``_FlatSync``); non-plannable models use torch DDP.

Conventions kept from the reference: the loss is multiplied by WORLD_SIZE under DDP
(yolo_trainer.py:325-326) on top of ``loss * batch_size`` (losses.py:297-300); BatchNorm statistics stay local
(``sync_bn`` defaults to false, train_config.yaml:17); the process group falls back to "gloo" when no GPU backend is
available (train_model_builder.py:112-114).
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn

LOCAL_RANK = int(os.getenv("LOCAL_RANK", -1))
RANK = int(os.getenv("RANK", -1))
WORLD_SIZE = int(os.getenv("WORLD_SIZE", 1))


class ModelEMA:
    """Exponential moving average of every floating tensor of the state dict, with the reference's interface and
    arithmetic (scripts/utils/torch_utils.py:377-426).  On the GPU the ~180 tensors are updated by ONE kernel launch
    (``ayolo_ema_update``) instead of two torch ops per tensor."""

    def __init__(self, model: nn.Module, decay: float = 0.9999, updates: int = 0) -> None:
        import math
        from copy import deepcopy
        plans = model.__dict__.pop("_plans", None)          # cached executor plans own GBs of activations: not part of the EMA
        try:
            self.ema = deepcopy(model).eval()                # FP32 EMA
        finally:
            if plans is not None:
                model.__dict__["_plans"] = plans
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._jobs = None

    @staticmethod
    def _slots(module: nn.Module):
        """{state-dict key: (owner module, '_parameters' | '_buffers', leaf name)} -- where the tensor of a key lives NOW.
        Looking the tensor up through its owner every step follows `.data` swaps and buffer re-assignment (`model.half()`,
        `.to(...)`) without building two ~300-entry state dicts per step (1.9 ms of host time)."""
        out = {}
        for name, mod in module.named_modules(remove_duplicate=False):
            pre = name + "." if name else ""
            for leaf, t in mod._parameters.items():
                if t is not None:
                    out[pre + leaf] = (mod, "_parameters", leaf)
            for leaf, t in mod._buffers.items():
                if t is not None and leaf not in mod._non_persistent_buffers_set:
                    out[pre + leaf] = (mod, "_buffers", leaf)
        return out

    def _job_table(self, model: nn.Module):
        import numpy as np
        refs = getattr(self, "_refs", None)
        if refs is None or refs[0] is not model or refs[1] != sum(1 for _ in model.modules()):
            es, ms = self._slots(self.ema), self._slots(model)
            slots = []
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    slots.append((es[k], ms[k] if k in ms else ms[f"module.{k}"]))
            self._refs = refs = (model, sum(1 for _ in model.modules()), slots)
        pairs = [(getattr(eo, ek)[el], getattr(mo, mk)[ml]) for (eo, ek, el), (mo, mk, ml) in refs[2]]
        key = tuple((v.data_ptr(), s.data_ptr()) for v, s in pairs)
        if self._jobs is None or self._jobs[0] != key:
            def dense(t):                                   # one dense block of memory (any of the two layouts in use)
                return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))

            ok = all(v.is_cuda and s.is_cuda and v.dtype == torch.float32 and s.dtype == torch.float32 and v.shape == s.shape
                     and v.stride() == s.stride() and dense(v) for v, s in pairs)
            tab = None
            if ok and pairs:
                # one grid row per job: the few multi-million-element conv weights are cut into 32 K-element chunks so that
                # they are spread over hundreds of workgroups instead of being walked by one row of the grid
                CH = 32768
                rows = [(v.data_ptr() + 4 * o, s_.data_ptr() + 4 * o, min(CH, v.numel() - o))
                        for v, s_ in pairs for o in range(0, v.numel(), CH)]
                job_t = np.dtype([("ema", "<u8"), ("src", "<u8"), ("n", "<i8")])
                jobs = np.array(rows, dtype=job_t)
                tab = torch.from_numpy(jobs.view(np.uint8).copy()).to(pairs[0][0].device)
                self._njobs = len(rows)
            self._jobs = (key, tab, pairs)
        return self._jobs

    def update(self, model: nn.Module) -> None:
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            _, tab, pairs = self._job_table(model)
            if tab is not None:
                from . import _lib
                _lib.call("ayolo_ema_update", tab.data_ptr(), self._njobs, float(d), torch.cuda.current_stream().cuda_stream)
                _lib.bump_versions(v for v, _ in pairs)          # raw-pointer writes: caches keyed on _version must see them
                return
            for v, src in pairs:                             # CPU / mixed-dtype state: the reference's two in-place ops
                v *= d
                v += (1.0 - d) * src.detach()

    def update_attr(self, model: nn.Module, include=(), exclude=("process_group", "reducer")) -> None:
        for k, v in model.__dict__.items():
            if (len(include) and k not in include) or k.startswith("_") or k in exclude:
                continue
            setattr(self.ema, k, v)


class _FlatSync:
    """What the plan calls during backward (a plain object: it must not become a submodule of the model).

    Gradient exchange of the data-parallel train step (the reference wraps the model in torch DDP,
    scripts/train/train_model_builder.py:75-78, whose reducer overlaps bucketed all-reduces with backward).  The plan
    executor runs backward as ONE op list, so the overlap is built here: the plan cuts the list where a reverse-layer
    bucket of its flat gradient arena is complete and calls ``launch_bucket``; the bucket's all-reduce (RCCL over xGMI
    through torch.distributed) is enqueued from a COMMUNICATION stream that waits for the executor's side stream (weight
    gradients) and for the compute stream (BatchNorm / bias gradients) -- the compute stream itself never waits, so the
    collective runs under the remaining backward kernels.  ``wait_all`` joins before the optimiser."""

    def __init__(self, group, world: int, sync_bn: bool = False, compress: Optional[str] = None) -> None:
        self.group, self.world, self.sync_bn = group, world, sync_bn
        self.overlap = os.getenv("AYOLO_DDP_OVERLAP", "1") == "1"
        # gradient compression of the bucket exchange (SURVEY.md 8e "fp16/bf16 gradient compression optional"): the bucket is
        # cast to a 16-bit buffer, averaged over the ranks in that type and written back into the fp32 arena -- half the
        # bytes on the xGMI links (YOLOv5l at 4 images per GPU is the exchange-bound case: 186 MB of fp32 gradients against
        # a short backward).  "bf16" keeps fp32's range (gradients still carry the GradScaler's factor here); "fp16" has
        # three more mantissa bits and overflows above 65504.  Off by default: the reference exchanges fp32.
        self.compress = compress if compress is not None else (os.getenv("AYOLO_DDP_COMPRESS") or None)
        assert self.compress in (None, "fp16", "bf16"), self.compress
        self._works = []
        self._comm = None
        # measurement mode (bench.py): timing events at every bucket's end on the communication stream and at the compute
        # stream's arrival at wait_all -> exposed_ms()
        self.measure = False
        self._marks = []

    def active(self) -> bool:
        return self.world > 1 or os.getenv("AYOLO_FORCE_DDP") == "1"      # the env switch exercises the RCCL calls on one GPU

    def _avg(self, t: torch.Tensor, async_op: bool):
        if dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        w = dist.all_reduce(t, group=self.group, async_op=async_op)       # gloo has no AVG: sum, then scale
        if w is not None:
            w.wait()
        if self.world > 1:
            t.mul_(1.0 / self.world)
        return None

    def reduce_flat(self, flat: torch.Tensor) -> None:
        """One blocking averaged all-reduce of the whole arena (AYOLO_DDP_OVERLAP=0, and the reference for the buckets)."""
        if self.active():
            self._avg(flat, False)

    def _exchange(self, view: torch.Tensor, async_op: bool):
        """Average `view` over the ranks in place, through a 16-bit buffer when compression is on.  Returns
        (work handle or None, finish() or None): finish copies the reduced 16-bit buffer back and must run after the work."""
        if self.compress is None:
            return self._avg(view, async_op), None
        half = view.to(torch.bfloat16 if self.compress == "bf16" else torch.float16)
        w = self._avg(half, async_op)
        return w, (lambda: view.copy_(half))

    def launch_bucket(self, view: torch.Tensor) -> None:
        """Averaged all-reduce of one finished arena range, asynchronous to the compute stream."""
        if not self.active():
            return
        if not view.is_cuda:
            w, fin = self._exchange(view, False)
            if fin is not None:
                fin()
            return
        from . import _lib
        cur = torch.cuda.current_stream()
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=view.device)
        comm = self._comm
        _lib.call("ayolo_side_stream_join", comm.cuda_stream)              # weight gradients of the bucket (side stream)
        comm.wait_stream(cur)                                                # BatchNorm / bias gradients (compute stream)
        with torch.cuda.stream(comm):
            w, fin = self._exchange(view, True)
            if fin is not None:
                if w is not None:
                    w.wait()                                                 # the communication stream waits for the collective
                fin()
                w = None
            # whatever the backend did (NCCL: the collective runs on its own stream behind `w`; gloo / compression: the work
            # was waited for and the scaling / write-back was enqueued on the communication stream), an event on the
            # communication stream marks the point the compute stream must wait for
            if self.measure and w is not None:
                w.wait()                                                     # communication stream behind the collective
            ev = torch.cuda.Event(enable_timing=self.measure)
            ev.record(comm)
        self._works.append((w, ev))
        if self.measure:
            self._marks.append(("bucket", ev, view.numel() * view.element_size()))

    def mark_backward_start(self) -> None:
        """Measurement mode: the compute stream's time at the start of the backward list (exposed_ms reports every bucket's
        completion and the join against it: the backward window the exchange has to hide in)."""
        if self.measure and torch.cuda.is_available():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self._marks.append(("start", ev, 0))

    def wait_all(self) -> None:
        cur = torch.cuda.current_stream() if self._works and torch.cuda.is_available() else None
        if self.measure and cur is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(cur)
            self._marks.append(("join", ev0, 0))
        for w, ev in self._works:
            if w is not None:
                w.wait()                                                     # current stream waits for the collective
            if cur is not None:
                cur.wait_event(ev)                                           # ... and for what the communication stream did after it
        self._works = []

    def exposed_ms(self) -> dict:
        """Measurement mode: per step, how long each bucket's exchange ran PAST the moment the compute stream reached
        wait_all (0 = fully hidden behind backward), from the timing events of the measured steps (call after a device
        synchronize)."""
        steps, cur = [], []
        for kind, ev, nbytes in self._marks:
            cur.append((kind, ev, nbytes))
            if kind == "join":
                steps.append(cur)
                cur = []
        self._marks = []
        per_bucket, sizes, done, window, nwin = None, None, None, 0.0, 0
        for st in steps:
            join = st[-1][1]
            start = next((ev for kind, ev, _ in st if kind == "start"), None)
            bk = [m for m in st[:-1] if m[0] == "bucket"]
            edge, row = 0.0, []
            for kind, ev, nbytes in bk:
                t = max(join.elapsed_time(ev), 0.0)          # ms the bucket finished after the join point (<= 0: hidden)
                row.append(max(t - edge, 0.0))
                edge = max(edge, t)
            if per_bucket is None:
                per_bucket, sizes, done = [0.0] * len(row), [b for _, _, b in bk], [0.0] * len(row)
            if len(row) == len(per_bucket):
                per_bucket = [a + b for a, b in zip(per_bucket, row)]
                if start is not None:
                    done = [a + start.elapsed_time(ev) for a, (_, ev, _) in zip(done, bk)]
                    window += start.elapsed_time(join)
                    nwin += 1
        n = max(len(steps), 1)
        per_bucket = [round(v / n, 4) for v in (per_bucket or [])]
        out = {"buckets": len(per_bucket), "bucket_mb": [round(b / 1e6, 2) for b in (sizes or [])],
               "exposed_ms_per_bucket": per_bucket, "exposed_ms_per_step": round(sum(per_bucket), 4),
               "overlap": self.overlap, "compress": self.compress, "measured_steps": len(steps)}
        if nwin:
            # the backward window (start of the backward list -> the compute stream reaches wait_all) and when, inside it, each
            # bucket's averaged all-reduce was complete
            out["backward_window_ms"] = round(window / nwin, 3)
            out["bucket_done_ms_after_backward_start"] = [round(v / nwin, 3) for v in done]
        return out

    def average_now(self, t: torch.Tensor) -> None:
        """sync_bn: in-stream average of one layer's BatchNorm accumulators (sum, sum of squares / backward sums) over
        the ranks.  Averaging instead of summing keeps the kernels' local pixel count valid:
        (sum over ranks / world) / n_local = global sum / global count for equal per-rank batches."""
        if self.active():
            self._avg(t, False)

    def __getstate__(self):                       # checkpoints pickle the whole model: a process group cannot travel
        return {"group": None, "world": 1, "sync_bn": False, "overlap": True, "compress": None, "_works": [], "_comm": None,
                "measure": False, "_marks": []}


class FlatGradDDP(nn.Module):
    """Data-parallel wrapper for models that train through the plan executor.

    The plan produces every parameter gradient of a step in ONE flat fp32 arena inside its single autograd node, so
    torch DDP's per-parameter hooks have nothing to overlap with.  This wrapper keeps DDP's contract -- parameters /
    buffers broadcast from rank 0 at construction, gradients averaged over the group -- and overlaps the exchange with
    backward itself: the arena is all-reduced in ~5 reverse-layer buckets as they complete (see ``_FlatSync``).
    ``sync_bn=True`` additionally averages every BatchNorm layer's batch statistics (forward) and gradient sums
    (backward) over the ranks, i.e. torch.nn.SyncBatchNorm semantics (train_config.yaml:17 ``sync_bn``, default
    false: statistics then stay local, as in the reference)."""

    def __init__(self, module: nn.Module, process_group=None, sync_bn: bool = False, compress: Optional[str] = None) -> None:
        super().__init__()
        assert dist.is_initialized(), "init_process_group first (TrainModelBuilder.ddp_init)"
        self.module = module
        self.sync = _FlatSync(process_group, dist.get_world_size(process_group), sync_bn, compress)
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t, 0, group=process_group)
        module._ayolo_grad_sync = self.sync

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class TrainModelBuilder:
    def __init__(self, model: nn.Module, cfg: Dict[str, Any], log_dir: str = "exp", full_cfg: Optional[dict] = None) -> None:
        self.model = model
        self.cfg = cfg
        self.log_dir = log_dir
        self.device = torch.device("cuda", max(LOCAL_RANK, 0)) if torch.cuda.is_available() else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.rank = int(os.getenv("RANK", -1))
        self.local_rank = int(os.getenv("LOCAL_RANK", -1))
        self.world_size = int(os.getenv("WORLD_SIZE", 1))

    def ddp_init(self) -> None:
        """One process per GPU: bind the device and join the process group (RCCL when GPUs are present)."""
        if self.local_rank == -1:
            return
        if self.cuda:
            assert torch.cuda.device_count() > self.local_rank, "insufficient GPUs for DDP"
            torch.cuda.set_device(self.local_rank)
            self.device = torch.device("cuda", self.local_rank)
        train_cfg = self.cfg.get("train", {})
        assert train_cfg.get("batch_size", self.world_size) % self.world_size == 0, "--batch-size must be multiple of GPU count"
        assert not train_cfg.get("image_weights", False), "--image-weights argument is not compatible with DDP training"
        if not dist.is_initialized():
            backend = "nccl" if (self.cuda and dist.is_nccl_available()) else "gloo"
            dist.init_process_group(backend=backend)

    def to_ddp(self) -> nn.Module:
        if self.cuda and type(self.model).__name__ == "YOLOModel" and getattr(self.model, "use_plan", True) \
                and os.getenv("AYOLO_TORCH_DDP") != "1":
            return FlatGradDDP(self.model, sync_bn=bool(self.cfg.get("train", {}).get("sync_bn", False)))
        if self.cuda and self.cfg.get("train", {}).get("sync_bn", False):
            raise NotImplementedError("sync_bn needs the plan executor (YOLOModel with use_plan): the per-module HIP path keeps "
                                      "BatchNorm statistics local")
        if self.cuda:
            return nn.parallel.DistributedDataParallel(self.model, device_ids=[self.local_rank], output_device=self.local_rank,
                                                       gradient_as_bucket_view=True)
        return nn.parallel.DistributedDataParallel(self.model)

    def prepare(self) -> Tuple[nn.Module, Optional[Any], torch.device]:
        torch.manual_seed(1 + max(self.rank, 0))
        self.model.to(self.device)
        ema = ModelEMA(self.model) if self.rank in (-1, 0) else None          # train_model_builder.py:130
        if self.rank != -1:
            self.model = self.to_ddp()
        return self.model, ema, self.device


def training_step(model: nn.Module, loss_fn, optimizer: torch.optim.Optimizer, scaler, imgs: torch.Tensor,
                  targets: torch.Tensor, world_size: int = 1, amp: bool = True, ema: Optional["ModelEMA"] = None):
    """autocast forward -> ComputeLoss -> (x world_size under DDP) -> scaled backward -> step (accumulate = 1) -> EMA
    (yolo_trainer.py:322-338)."""
    from ._lib import roctx_range
    with roctx_range("train.forward+loss"), torch.autocast(imgs.device.type, dtype=torch.float16, enabled=amp and imgs.is_cuda):
        pred = model(imgs)
        loss, items = loss_fn(pred, targets)
    if world_size > 1:
        loss = loss * world_size
    with roctx_range("train.backward"):
        if scaler is not None:
            scaler.scale(loss).backward()
        else:
            loss.backward()
    with roctx_range("train.optimizer"):
        if scaler is not None:
            scaler.step(optimizer)
            scaler.update()
        else:
            optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        if ema is not None:
            ema.update(model)
    return loss.detach(), items
