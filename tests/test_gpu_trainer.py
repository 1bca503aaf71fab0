"""Trainer glue on the GPU: `training_step` (scripts/train/yolo_trainer.py:289-358) through the HIP plan executor with
GradScaler, the one-launch SGD and ModelEMA, against the CPU oracle network stepped by torch's own SGD; and the
data-parallel wrapper's overlapped bucket exchange / sync_bn code path on a single-rank RCCL group."""
import copy
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "ayolov2_amd", "configs")
HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


def _models(seed=0):
    from ayolov2_amd import YOLOModel
    from oracle.model_ref import RefYOLO
    torch.manual_seed(seed)
    cfg = os.path.join(CFG, "yolov5n.yaml")
    m = YOLOModel(cfg)
    r = RefYOLO(cfg)
    r.load_state_dict(m.state_dict())
    for mod in (m, r):
        mod.hyp, mod.gr, mod.nc = dict(HYP), 1.0, 80
    return m.cuda().train(), r.train()


def _batch(seed, B=4, hw=(128, 160)):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, *hw, generator=g)
    nt = 5 * B
    t = torch.cat((torch.randint(0, B, (nt, 1), generator=g).float(), torch.randint(0, 80, (nt, 1), generator=g).float(),
                   torch.rand(nt, 2, generator=g) * 0.8 + 0.1, torch.rand(nt, 2, generator=g) * 0.3 + 0.05), 1)
    return x, t


def _groups(model):
    """yolo_trainer.py:149-168: BatchNorm weights / conv weights (decayed) / biases."""
    pg_w, pg_bn, pg_b = [], [], []
    for mod in model.modules():
        if hasattr(mod, "bias") and isinstance(mod.bias, torch.nn.Parameter):
            pg_b.append(mod.bias)
        if isinstance(mod, torch.nn.BatchNorm2d):
            pg_bn.append(mod.weight)
        elif hasattr(mod, "weight") and isinstance(mod.weight, torch.nn.Parameter):
            pg_w.append(mod.weight)
    return pg_bn, pg_w, pg_b


def _optim(cls, model, lr=0.01, **kw):
    pg_bn, pg_w, pg_b = _groups(model)
    opt = cls(pg_bn, lr=lr, momentum=0.937, nesterov=True, **kw)
    opt.add_param_group({"params": pg_w, "weight_decay": 5e-4})
    opt.add_param_group({"params": pg_b})
    return opt


def test_training_step_matches_cpu_reference():
    """Three `training_step`s (fp32 mode, so that the comparison is about the step logic and not fp16 rounding): forward
    -> fused ComputeLoss -> backward through the plan -> ayolo_sgd_step -> ModelEMA, against RefYOLO + the torch-op loss +
    torch.optim.SGD + the reference's EMA arithmetic on the CPU.  Parameters, BN running statistics and the EMA copy."""
    import math
    from ayolov2_amd.losses import ComputeLoss
    from ayolov2_amd.optim import SGD
    from ayolov2_amd.trainer import ModelEMA, training_step
    m, r = _models(1)
    # a small learning rate keeps the comparison about the step LOGIC: with lr 0.01 the fp32 summation-order noise of one
    # step, amplified by the next forward's batch statistics (few samples per channel at stride 32), occasionally moved the
    # third step's loss by > 1 % between runs
    opt_g, opt_c = _optim(SGD, m, lr=0.002), _optim(torch.optim.SGD, r, lr=0.002)
    loss_g, loss_c = ComputeLoss(m), ComputeLoss(r)
    ema = ModelEMA(m)
    ema_ref = {k: v.detach().clone() for k, v in copy.deepcopy(r).state_dict().items()}
    for step in range(3):
        x, t = _batch(10 + step, B=4, hw=(160, 192))
        lg, items = training_step(m, loss_g, opt_g, None, x.cuda(), t.cuda(), world_size=1, amp=False, ema=ema)
        lc, _ = loss_c(r(x), t)
        opt_c.zero_grad(set_to_none=True)
        lc.backward()
        opt_c.step()
        d = 0.9999 * (1 - math.exp(-(step + 1) / 2000))
        for k, v in r.state_dict().items():
            if v.dtype.is_floating_point:
                ema_ref[k].mul_(d).add_((1.0 - d) * v.detach())
            else:
                ema_ref[k] = v.clone()
        # step 0 starts from identical weights; later steps inherit the (BatchNorm-amplified: 80 samples per channel at
        # stride 32) fp32 summation-order noise of the previous updates
        np.testing.assert_allclose(float(lg), float(lc), rtol=2e-4 if step == 0 else 1e-2)
    sd_g = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    for k, v in r.state_dict().items():
        if v.dtype.is_floating_point:
            err = float((sd_g[k] - v).abs().max())
            assert err <= 2e-3 * float(v.abs().max()) + 1e-4, (k, err)
        else:
            assert int(sd_g[k]) == int(v), k
    for k, v in ema.ema.state_dict().items():
        if v.dtype.is_floating_point:
            assert float((v.cpu() - ema_ref[k]).abs().max()) <= 2e-3 * float(ema_ref[k].abs().max()) + 1e-4, k


def test_training_step_amp_gradscaler():
    """The reference's actual configuration: autocast fp16 + GradScaler (yolo_trainer.py:322-338) through training_step:
    the first step's loss equals the fp32 mode's within fp16 tolerance, every step is finite, the scaler never skips (no inf
    gradients at scale 1024), parameters move and the EMA counts every step."""
    from ayolov2_amd.losses import ComputeLoss
    from ayolov2_amd.optim import SGD
    from ayolov2_amd.trainer import ModelEMA, training_step
    m, _ = _models(2)
    m32 = copy.deepcopy(m)
    opt = _optim(SGD, m)
    loss_fn = ComputeLoss(m)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    ema = ModelEMA(m)
    x, t = _batch(20)
    x, t = x.cuda(), t.cuda()
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    losses = [float(training_step(m, loss_fn, opt, scaler, x, t, amp=True, ema=ema)[0]) for _ in range(4)]
    l32 = float(training_step(m32, ComputeLoss(m32), _optim(SGD, m32), None, x, t, amp=False)[0])
    assert all(np.isfinite(losses)) and abs(losses[0] - l32) <= 5e-3 * abs(l32), (losses, l32)
    assert float(scaler.get_scale()) == 1024.0 and ema.updates == 4
    moved = [float((p.detach() - before[k]).abs().max()) for k, p in m.named_parameters()]
    assert min(moved) > 0.0 and all(np.isfinite(moved))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("sync_bn", [False, True])
def test_flat_grad_ddp_single_rank_rccl(sync_bn, monkeypatch):
    """FlatGradDDP on a one-rank RCCL group with the exchange forced on: the backward list runs in bucket segments, every
    bucket's all-reduce is enqueued from the communication stream (after the executor's side stream), the compute stream
    joins before the gradients are handed out; with sync_bn every BN layer's statistics / gradient sums are averaged in
    stream order.  On one rank the averages are identities, so gradients and BN buffers must equal the plain model's."""
    import torch.distributed as dist
    from ayolov2_amd.losses import ComputeLoss
    from ayolov2_amd.trainer import FlatGradDDP
    monkeypatch.setenv("AYOLO_FORCE_DDP", "1")
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        m, _ = _models(3)
        m2 = copy.deepcopy(m)
        w = FlatGradDDP(m, sync_bn=sync_bn)
        assert m._ayolo_grad_sync.active()
        x, t = _batch(30)
        outs = []
        for mod, run in ((m, w), (m2, m2)):
            loss, _ = ComputeLoss(mod)(run(x.cuda()), t.cuda())
            loss.backward()
            outs.append(({k: p.grad.detach().float().cpu() for k, p in mod.named_parameters()},
                         {k: b.detach().float().cpu() for k, b in mod.named_buffers()}))
        plan = next(iter(m._plans.values()))
        assert len(plan.buckets) >= 2 and not m._ayolo_grad_sync._works       # all bucket works were waited for
        for k, g in outs[0][0].items():
            ref = outs[1][0][k]
            assert float((g - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-7, k
        for k, b in outs[0][1].items():
            torch.testing.assert_close(b, outs[1][1][k], rtol=1e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()


def test_eval_after_raw_pointer_updates_sees_new_weights():
    """the inference executor and the per-module weight cache key their folded BatchNorm vectors / fp16
    weight copies on tensor._version, while ayolo_sgd_step, ayolo_ema_update and the BatchNorm running-statistic update of a
    training forward write through raw pointers.  The reference validates the SAME ema.ema object after every epoch
    (yolo_trainer.py:120): validate once (plan built), train, validate again -- the second validation must see the new
    weights.  Checked against the CPU oracle loaded with the EMA state dict."""
    from oracle.model_ref import RefYOLO
    from ayolov2_amd.losses import ComputeLoss
    from ayolov2_amd.optim import SGD
    from ayolov2_amd.trainer import ModelEMA, training_step
    m, _ = _models(5)
    opt = _optim(SGD, m, lr=0.05)
    loss_fn = ComputeLoss(m)
    ema = ModelEMA(m)
    ema.decay = lambda x: 0.5                                   # make the EMA move visibly in a few steps
    xv = torch.rand(2, 3, 128, 160, generator=torch.Generator().manual_seed(7)).cuda()
    with torch.no_grad():
        z0 = ema.ema(xv)[0].clone()                             # first validation: builds (and caches) the inference plan
    for step in range(3):
        x, t = _batch(30 + step)
        training_step(m, loss_fn, opt, None, x.cuda(), t.cuda(), amp=False, ema=ema)
    with torch.no_grad():
        z1 = ema.ema(xv)[0].clone()                             # second validation through the SAME cached plan
    r = RefYOLO(os.path.join(CFG, "yolov5n.yaml")).eval()
    r.load_state_dict({k: v.cpu() for k, v in ema.ema.state_dict().items()})
    with torch.no_grad():
        zr = r(xv.cpu())[0]
    assert float((z1 - z0).abs().max()) > 1e-3, "the EMA model did not change: the test would prove nothing"
    np.testing.assert_allclose(z1.cpu().numpy(), zr.numpy(), rtol=1e-4, atol=2e-3)
    # the per-module path (use_plan off: fine-tuning of decomposed blocks) stepped by the one-launch SGD: its fp16 weight
    # cache must follow the update too
    m.use_plan = False
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a0 = m(xv)[0].float().clone()
    m.train()
    x, t = _batch(40)
    training_step(m, loss_fn, opt, None, x.cuda(), t.cuda(), amp=True, ema=None)
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a1 = m(xv)[0].float().clone()
    r.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        ar = r(xv.cpu())[0]
    assert float((a1 - a0).abs().max()) > 1e-3
    assert float((a1[..., 4:].cpu() - ar[..., 4:]).abs().max()) < 0.02


def test_model_copies_after_forwards_and_repeated_backward():
    """(medium) deepcopy / torch.save of a model that has cached executor plans (the reference deep-copies the EMA
    model into every checkpoint, yolo_trainer.py:379-386); (medium) the one-launch SGD reuses its job table from step to step
    (the plan hands out gradients at repeating addresses); (low) a second backward through one forward raises instead of
    double-accumulating the BatchNorm sums."""
    import io
    from ayolov2_amd.losses import ComputeLoss
    from ayolov2_amd.optim import SGD
    from ayolov2_amd.trainer import training_step
    m, _ = _models(6)
    x, t = _batch(50)
    x, t = x.cuda(), t.cuda()
    opt = _optim(SGD, m)
    loss_fn = ComputeLoss(m)
    builds = []
    orig = opt._build_table
    opt._build_table = lambda *a, **k: (builds.append(1), orig(*a, **k))[1]
    for _ in range(6):
        training_step(m, loss_fn, opt, None, x, t, amp=True)
    assert len(builds) <= 2, f"the SGD job table was rebuilt {len(builds)} times in 6 steps"
    m.eval()
    with torch.no_grad():
        m(x)
    assert m.__dict__.get("_plans"), "both executors should have cached plans by now"
    c = copy.deepcopy(m)
    assert not c.__dict__.get("_plans")
    buf = io.BytesIO()
    torch.save(m, buf)
    with torch.no_grad():
        np.testing.assert_allclose(c(x)[0].cpu().numpy(), m(x)[0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    m.train()
    raws = m(x)
    loss = sum(r.float().square().mean() for r in raws)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        loss.backward()
