"""Inference executor (infer_plan.py) and the BASELINE configurations round 1 left untested: cfg 4 (Tucker-decomposed
YOLOv5s, fp16, eval), cfg 5's forward leg (YOLOv5x 1280^2 fp16 eval), fp16 parity of YOLOv5s -- the model and dtype the
headline bench runs -- against the CPU oracle, and the full-size fp16 train step against its own exact-fp32 mode."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "ayolov2_amd", "configs")
HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


def _pair(name, seed=0):
    from ayolov2_amd import YOLOModel
    from oracle.model_ref import RefYOLO
    torch.manual_seed(seed)
    cfg = os.path.join(CFG, f"yolov5{name}.yaml")
    m = YOLOModel(cfg)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.6, 1.4)
                mod.bias.uniform_(-0.3, 0.3)
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.6, 1.6)
    r = RefYOLO(cfg)
    r.load_state_dict(m.state_dict())
    return m, r


def _has_plan(m, kind):
    return any(v is not False and k and k[0] == kind for k, v in m.__dict__.get("_plans", {}).items())


@pytest.mark.parametrize("fused", [False, True])
def test_infer_plan_fp32_equals_oracle_and_module_path(fused):
    """eval forward (decoded prediction + raw logits) through the one-call inference executor vs the CPU oracle (1e-4, the
    north-star bar) and vs the per-module path; unfused (BatchNorm folded into the epilogue) and after fuse()."""
    m, r = _pair("s", seed=21)
    m, r = m.cuda().eval(), r.eval()
    if fused:
        m.fuse()
    x = torch.rand(2, 3, 160, 192)
    with torch.no_grad():
        zr, raws_r = r(x)
        zg, raws_g = m(x.cuda())
        assert _has_plan(m, "eval"), "the inference executor must be the path that ran"
        zg, raws_g = zg.clone(), [t.clone() for t in raws_g]
        m.use_plan = False
        zm, raws_m = m(x.cuda())
    for a, b in zip(raws_g, raws_r):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(zg.cpu().numpy(), zr.numpy(), rtol=1e-4, atol=2e-3)      # pixels: 1e-4 of the 192-px range...
    for a, b in zip(raws_g, raws_m):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_infer_plan_tracks_weight_updates():
    """Folded BatchNorm vectors and compute-dtype weight copies are cached across forwards and must be refreshed when a
    parameter or a running statistic changes in place."""
    m, r = _pair("n", seed=22)
    m, r = m.cuda().eval(), r.eval()
    x = torch.rand(1, 3, 64, 64)
    with torch.no_grad():
        m(x.cuda())
        for mod, ref in ((m, r),):
            mod.model[0].conv.weight.mul_(1.5)
            mod.model[2].cv1.batch_norm.running_var.add_(0.5)
        sd = {k: v.cpu() for k, v in m.state_dict().items()}
        r.load_state_dict(sd)
        zg, _ = m(x.cuda())
        zr, _ = r(x)
    np.testing.assert_allclose(zg.cpu().numpy(), zr.numpy(), rtol=1e-4, atol=2e-3)


def test_eval_prediction_survives_the_next_forward():
    """A val.py-style loop may keep `out` of batch i while batch i + 1 runs: the decoded prediction belongs to the call
    that produced it; only with model.static_outputs = True is it the executor's static buffer."""
    m, _ = _pair("n", seed=36)
    m = m.cuda().eval()
    x0, x1 = torch.rand(2, 3, 64, 96).cuda(), torch.rand(2, 3, 64, 96).cuda()
    with torch.no_grad():
        out0, _ = m(x0)
        keep = out0.clone()
        out1, _ = m(x1)
        assert out1.data_ptr() != out0.data_ptr() and torch.equal(out0, keep) and not torch.equal(out1, keep)
        m.static_outputs = True
        s0, _ = m(x0)
        s1, _ = m(x1)
        assert s0.data_ptr() == s1.data_ptr()
        assert torch.equal(s1, out1)


def test_infer_plan_fp16_yolov5s():
    """The dtype validation runs in (`half` models / autocast): fp16 storage, fp32 accumulation.  Tolerance: 2 % of the
    logit range (same bar as the fp16 train-step tests)."""
    m, r = _pair("s", seed=23)
    m, r = m.cuda().eval().fuse(), r.eval()
    x = torch.rand(2, 3, 320, 320)
    with torch.no_grad():
        _, raws_r = r(x)
        with torch.autocast("cuda", dtype=torch.float16):
            _, raws_g = m(x.cuda())
    assert _has_plan(m, "eval")
    for a, b in zip(raws_g, raws_r):
        err = float((a.cpu() - b).abs().max())
        assert err <= 0.02 * float(b.max() - b.min()), err


_TUCKER_CACHE = {}


@pytest.mark.parametrize("form", ["factors", "first", "last", "dense", "auto"])
def test_tucker_launch_forms_fp32(form, monkeypatch):
    """The inference executor may multiply a Tucker block's linear convs out (infer_plan._tucker_form): every launch form --
    three factor launches, first factor merged into the core, core merged into the last factor, the dense conv again, and the
    cost model's own pick -- must give the decomposed module's result (per-module path, exact-fp32 mode) to 1e-4 of the logit
    range, and a weight update must reach the merged weights.  YOLOv5n, decomposed incl. the 6x6 stem."""
    from ayolov2_amd import decomposition as D, infer_plan as IP
    from ayolov2_amd.modules import Conv
    if "m" not in _TUCKER_CACHE:                             # decompose once (host-side SVDs), copy per launch form
        m0, _ = _pair("n", seed=31)
        with torch.no_grad():
            for mod in m0.modules():
                if isinstance(mod, Conv) and mod.conv.kernel_size != (1, 1):
                    w = mod.conv.weight.data
                    co, ci, kh, kw = w.shape
                    ro, ri = max(co // 4, 2), max(ci // 4, 2)
                    std = 1.0 / (ci * kh * kw) ** 0.5
                    w.copy_(torch.einsum("abhw,oa,ib->oihw", torch.randn(ro, ri, kh, kw), torch.randn(co, ro), torch.randn(ci, ri))
                            * (std / (ro * ri) ** 0.5) + 0.05 * std * torch.randn_like(w))
        D.decompose_model(m0, loss_thr=0.1, prune_step=0.0)
        assert len(D.decomposed_ranks(m0)) >= 10
        _TUCKER_CACHE["m"] = m0
    m = copy.deepcopy(_TUCKER_CACHE["m"])
    m = m.cuda().eval()
    x = torch.rand(2, 3, 128, 160).cuda()
    monkeypatch.setattr(IP, "TUCKER_FORMS", form)
    with torch.no_grad():
        m.use_plan = False
        _, raws_ref = m(x)
        raws_ref = [t.clone() for t in raws_ref]
        m.use_plan = True
        m.__dict__.pop("_plans", None)
        _, raws = m(x)
        assert _has_plan(m, "eval")
        plan = [p for p in m._plans.values() if p][0]
        assert len(plan.tucker_forms) >= 10 and (form == "auto" or set(plan.tucker_forms) == {form}), plan.tucker_forms
        for a, b in zip(raws, raws_ref):
            assert float((a - b).abs().max()) <= 1e-4 * float(b.max() - b.min()) + 1e-5
        # a weight update of a factor reaches the multiplied-out weights (version-tracked sources)
        seq = next(mod.conv for mod in m.modules() if isinstance(mod, Conv) and isinstance(mod.conv, torch.nn.Sequential))
        seq[0].weight.mul_(-2.0)
        m.use_plan = False
        _, raws_ref2 = m(x)
        raws_ref2 = [t.clone() for t in raws_ref2]
        m.use_plan = True
        _, raws2 = m(x)
        for a, b in zip(raws2, raws_ref2):
            assert float((a - b).abs().max()) <= 1e-4 * float(b.max() - b.min()) + 1e-5
        assert any(float((a - b).abs().max()) > 1e-6 * float(b.max() - b.min()) for a, b in zip(raws_ref2, raws_ref))   # the update is visible at all


def test_cfg4_tucker_decomposed_yolov5s_fp16_eval():
    """BASELINE cfg 4: decompose_model() (scripts/tensor_decomposition/decomposition.py:237-339, defaults of
    decompose_model.py:63-74 except prune_step 0 to bound the SVD count) on YOLOv5s -- every k > 1 conv incl. the 6x6
    stem -- on the HIP model and on the CPU oracle with the same weights; fp16 eval of the decomposed model through the
    inference executor (three launches per decomposed block over rank-padded buffers) vs the oracle's fp32 forward.
    Weights get a planted low-rank structure (random-init weights have no low-rank structure to find)."""
    from torch import nn
    from ayolov2_amd import decomposition as D
    from ayolov2_amd.modules import Conv
    m, r = _pair("s", seed=24)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, Conv) and mod.conv.kernel_size != (1, 1):
                w = mod.conv.weight.data
                co, ci, kh, kw = w.shape
                ro, ri = max(co // 4, 2), max(ci // 4, 2)
                core = torch.randn(ro, ri, kh, kw)
                std = 1.0 / (ci * kh * kw) ** 0.5                       # variance-preserving scale (fp16 range over ~20 layers)
                w.copy_(torch.einsum("abhw,oa,ib->oihw", core, torch.randn(co, ro), torch.randn(ci, ri)) * (std / (ro * ri) ** 0.5)
                        + 0.05 * std * torch.randn_like(w))
    r.load_state_dict(m.state_dict())
    D.decompose_model(m, loss_thr=0.1, prune_step=0.0)
    D.decompose_model(r, loss_thr=0.1, prune_step=0.0)
    ranks = D.decomposed_ranks(m)
    assert len(ranks) >= 12, ranks                       # of 18: stem, 6 strided Conv rows, 11 Bottleneck 3x3 convs
    assert D.count_param(m) < 7_235_389 * 0.8
    m, r = m.cuda().eval(), r.eval()
    x = torch.rand(2, 3, 256, 320)
    with torch.no_grad():
        _, raws_r = r(x)
        with torch.autocast("cuda", dtype=torch.float16):
            _, raws_g = m(x.cuda())
    assert _has_plan(m, "eval"), "decomposed blocks must run on the inference executor"
    for a, b in zip(raws_g, raws_r):
        err = float((a.cpu() - b).abs().max())
        assert err <= 0.02 * float(b.max() - b.min()), (err, float(b.max() - b.min()))
    # a .half() copy (what decompose_model.py:297 saves) runs the same way
    plans = m.__dict__.pop("_plans", None)                # executor plans own the activations: not part of a copy
    mh = copy.deepcopy(m).half()
    m.__dict__["_plans"] = plans
    with torch.no_grad():
        _, raws_h = mh(x.cuda().half())
    for a, b in zip(raws_h, raws_r):
        assert float((a.cpu().float() - b).abs().max()) <= 0.03 * float(b.max() - b.min())


def test_cfg5_forward_leg_yolov5x_1280_fp16():
    """BASELINE cfg 5's forward leg: YOLOv5x, 1280x1280, fuse().eval(), fp16, at the full-size launch geometry (batch 2;
    the CPU oracle would take minutes): the fp16 executor against the exact-fp32 mode of the same kernels."""
    from ayolov2_amd import YOLOModel
    torch.manual_seed(25)
    m = YOLOModel(os.path.join(CFG, "yolov5x.yaml")).cuda().eval().fuse()
    x = torch.rand(2, 3, 1280, 1280).cuda()
    with torch.no_grad():
        z32, raws32 = m(x)
        z32, raws32 = z32.clone(), [t.clone() for t in raws32]
        with torch.autocast("cuda", dtype=torch.float16):
            z16, raws16 = m(x)
    assert z16.shape == (2, 100800, 85)
    for a, b in zip(raws16, raws32):
        assert float((a - b).abs().max()) <= 0.02 * float(b.max() - b.min())
    assert float((z16[..., 4:] - z32[..., 4:]).abs().max()) < 0.02          # objectness / class probabilities


def test_cfg5_yolov5x_fp16_eval_vs_oracle():
    """BASELINE cfg 5's model and dtype against the ORACLE (not only against its own fp32 mode): YOLOv5x, fuse().eval(),
    fp16, one 640 x 640 image (the CPU oracle needs ~10 s for it; the 1280^2 launch geometry is the test above).
    Raw logits within 2 % of the logit range per level (the fp16 bar of every other model), decoded scores within 0.02."""
    m, r = _pair("x", seed=35)
    m, r = m.cuda().eval().fuse(), r.eval()
    x = torch.rand(1, 3, 640, 640)
    with torch.no_grad():
        zr, raws_r = r(x)
        with torch.autocast("cuda", dtype=torch.float16):
            zg, raws_g = m(x.cuda())
    assert _has_plan(m, "eval")
    for a, b in zip(raws_g, raws_r):
        err = float((a.cpu() - b).abs().max())
        assert err <= 0.02 * float(b.max() - b.min()), (err, float(b.max() - b.min()))
    assert float((zg[..., 4:].cpu() - zr[..., 4:]).abs().max()) < 0.02


def _train_step(m, x, t, amp):
    """One forward + loss + backward.  The fp16 mode scales the loss as the reference's GradScaler does
    (yolo_trainer.py:329; a fixed 2^12 here) -- unscaled fp16 gradients of the early layers sit in the subnormal range --
    and returns the unscaled gradients."""
    from ayolov2_amd.losses import ComputeLoss
    m.zero_grad(set_to_none=True)
    scale = 4096.0 if amp else 1.0
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        raws = m(x)
        loss, _ = ComputeLoss(m)(raws, t)
    (loss * scale).backward()
    return float(loss.detach()), [r.detach().float().clone() for r in raws], {k: p.grad.detach().float().clone() / scale for k, p in m.named_parameters()}


def _targets(B, seed):
    g = torch.Generator().manual_seed(seed)
    n = B * 6
    return torch.cat((torch.arange(B).repeat_interleave(6).float()[:, None], torch.randint(0, 80, (n, 1), generator=g).float(),
                      torch.rand(n, 2, generator=g) * 0.8 + 0.1, torch.rand(n, 2, generator=g) * 0.4 + 0.03), 1)


def test_yolov5s_train_step_fp32_and_fp16_vs_oracle():
    """YOLOv5s (the bench model) at 4 x 320^2 against the CPU oracle: the fp32 mode at the north-star bar (logits 1e-4),
    the fp16 autocast mode -- the bench's dtype -- at 2 % of the logit range, its gradients against the oracle's Jacobian applied
    to the oracle loss's gradient at the fp16 logits (_oracle_grads_at_logits: median error 15 % / 90th percentile 25 % of each
    parameter's largest element, whole-gradient cosine 0.985), with the direction of every weight gradient checked as well."""
    from ayolov2_amd.losses import ComputeLoss
    m, r = _pair("s", seed=26)
    for mod in (m, r):
        mod.hyp, mod.gr, mod.nc = dict(HYP), 1.0, 80
    m, r = m.cuda().train(), r.train()
    x, t = torch.rand(4, 3, 320, 320), _targets(4, 27)
    raws_r = r(x)
    loss_r, _ = ComputeLoss(r)(raws_r, t)
    loss_r.backward()
    gr = {k: p.grad for k, p in r.named_parameters()}
    sd = copy.deepcopy(m.state_dict())
    l32, raws32, g32 = _train_step(m, x.cuda(), t.cuda(), amp=False)
    m.load_state_dict(sd)                                     # same BN running statistics for the second run
    l16, raws16, g16 = _train_step(m, x.cuda(), t.cuda(), amp=True)
    for a, b in zip(raws32, raws_r):
        np.testing.assert_allclose(a.cpu().numpy(), b.detach().numpy(), rtol=1e-4, atol=1e-4)
    assert abs(l32 - float(loss_r.detach())) <= 1e-4 * abs(float(loss_r.detach()))
    for k, g in g32.items():
        assert float((g.cpu() - gr[k]).abs().max()) <= 2e-3 * float(gr[k].abs().max()) + 1e-7, k
    for a, b in zip(raws16, raws_r):
        assert float((a.cpu() - b.detach()).abs().max()) <= 0.02 * float(b.max() - b.min())
    assert abs(l16 - float(loss_r.detach())) <= 5e-3 * abs(float(loss_r.detach()))
    # fp16 gradients at random initialisation: every element is a sum of 10^4..10^6 signed terms that largely cancel while fp16
    # storage rounds every activation and activation gradient to 1e-3, so a parameter's worst element is a noisy quantity (the
    # first layers, at the END of the fp16 backward chain, and BatchNorm vectors lead) while the direction of the whole gradient
    # holds.  The step is bit-reproducible (fp64 statistics, no atomics in the weight gradients): these are fixed numbers of a
    # kernel route, and bit-equivalent routes (same MFMA order, another summation order of the statistics) move them.
    # Reference for the fp16 gradients: the oracle's Jacobian applied to the oracle loss's gradient AT THE FP16 LOGITS
    # (_oracle_grads_at_logits) -- the direct comparison (oracle loss gradient at the oracle's logits) is printed next to it.
    gr_direct = gr
    gr, cond = _oracle_grads_at_logits(r, x, t, raws16)
    direct = _cos(_flat({k: v.cpu() for k, v in g16.items()}), _flat(gr_direct))
    print("yolov5s fp16 vs oracle: loss-gradient cosine per level (fp16 logits vs oracle logits) %s; whole-gradient cosine against the "
          "oracle's own step %.5f" % ([round(c, 5) for c in cond], direct))
    errs, flat16, flat32 = {}, [], []
    for k, g in g16.items():
        errs[k] = float((g.cpu() - gr[k]).abs().max()) / (float(gr[k].abs().max()) + 1e-12)
        flat16.append(g.cpu().flatten().double())
        flat32.append(gr[k].flatten().double())
        if g.dim() == 4:
            a_, b_ = flat16[-1], flat32[-1]
            assert float((a_ @ b_) / (a_.norm() * b_.norm() + 1e-300)) >= 0.90, k
    e = np.sort(np.array(list(errs.values())))
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    a_, b_ = torch.cat(flat16), torch.cat(flat32)
    glob = float((a_ @ b_) / (a_.norm() * b_.norm()))
    print("yolov5s fp16 vs oracle: gradient error / max element: median %.4f, p90 %.4f, max %.4f %s; whole-gradient cosine %.5f"
          % (np.median(e), e[int(0.9 * len(e))], e[-1], [(k, round(v, 3)) for k, v in top], glob))
    # Measured on MI355X in round 6 over eight kernel routes that compute the same arithmetic (k_gconv / k_pw / eight-wavefront
    # tiles, profiles/r06_pw_fp16_numbers_v2.txt): median 0.066-0.085, 90th percentile 0.109-0.152, maximum 0.31-0.63, whole-gradient
    # cosine 0.9936-0.9959 -- against the oracle's OWN step the same routes span 0.9586-0.9913, all of it the loss's conditioning
    # (loss-gradient cosine of the 10 x 10 level 0.968-0.993).  A wrong tile / channel chunk moves the cosine below 0.9; the
    # comparison on a well-conditioned net is test_fp16_train_step_well_conditioned_vs_oracle.
    assert np.median(e) <= 0.15 and e[int(0.9 * len(e))] <= 0.25, (np.median(e), e[int(0.9 * len(e))], e[-1], top)
    assert glob >= 0.985, glob
    assert direct >= 0.9, direct


def _flat(g):
    return torch.cat([v.flatten().double() for v in g.values()])


def _oracle_grads_at_logits(r, x, t, raws):
    """Parameter gradients of the CPU oracle with the LOSS GRADIENT EVALUATED AT THE GIVEN LOGITS (chain rule: J^T dL/dlogits
    with J the oracle's own Jacobian and dL/dlogits the oracle loss's gradient at `raws`, the fp16 run's logits) -- the
    comparison of an fp16 step against this is decided by the conv / BatchNorm chain and not by the conditioning of ComputeLoss:
    at random initialisation one CIoU term can be near-singular, and logits that differ by 1e-3 then give loss gradients whose
    cosine is 0.98 on that level (round 6, profiles/r06_grad_cmp_loss.txt: measured by torch autograd on the logits of two
    bit-equivalent kernel routes).  Also returns that cosine per level (loss gradient at `raws` vs at the oracle's logits)."""
    from ayolov2_amd.losses import ComputeLoss
    leaf = [a.detach().float().cpu().clone().requires_grad_(True) for a in raws]
    loss, _ = ComputeLoss(r)(leaf, t)
    loss.backward()
    dl = [a.grad for a in leaf]
    r.zero_grad(set_to_none=True)
    raws_r = r(x)
    leaf_r = [a.detach().clone().requires_grad_(True) for a in raws_r]
    loss_r, _ = ComputeLoss(r)(leaf_r, t)
    loss_r.backward()
    cond = [_cos(a.grad.flatten().double(), b.flatten().double()) for a, b in zip(leaf_r, dl)]
    torch.autograd.backward(raws_r, dl)
    return {k: p.grad.detach().clone() for k, p in r.named_parameters()}, cond


def _cos(a, b):
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def test_fp16_train_step_is_reproducible_and_routes_agree(monkeypatch):
    """BatchNorm statistics are accumulated in fp64 from the workgroup level on, so the only order-dependent roundings of a
    step sit at 1e-16 -- below one fp32 ulp of every mean / variance -- and the fp16 rounding of every activation repeats:
    two identical fp16 steps give the same loss and the same gradients (conv weight gradients bit for bit since round 4).  That makes every ROUTE comparison discriminating in the bench's own dtype: the plan executor vs the
    per-module path, and the BatchNorm-backward sums in the dgrad epilogue vs the separate reduce pass (different fp32
    summation order of the same rounded values), all within cosine 1 - 1e-4 of each other -- three orders of magnitude below what fp16
    storage itself costs against fp32, and far below any wrong tile (cosine < 0.9)."""
    from ayolov2_amd import plan as P
    m, _ = _pair("s", seed=37)
    with torch.no_grad():                        # BatchNorm gains of 0.3: see test_fp16_train_step_well_conditioned_vs_oracle
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.fill_(0.3)
    m.hyp, m.gr, m.nc = dict(HYP), 1.0, 80
    m = m.cuda().train()
    x, t = torch.rand(4, 3, 320, 320).cuda(), _targets(4, 38).cuda()
    sd = copy.deepcopy(m.state_dict())

    def run(**kw):
        m.load_state_dict(sd)
        m.__dict__.pop("_plans", None)
        m.use_plan = kw.get("use_plan", True)
        old, old_xf, old_xw, old_pair, old_sppf = P.BN_REDUCE_IN_DGRAD, P.XF_ON_LOAD, P.XF_WGRAD_ON_LOAD, P.BN_APPLY_PAIR, P.SPPF_FUSED
        P.BN_REDUCE_IN_DGRAD = kw.get("bnr", True)
        P.XF_ON_LOAD = kw.get("xf", True)
        P.XF_WGRAD_ON_LOAD = kw.get("xf_wgrad", False)
        P.BN_APPLY_PAIR = kw.get("pair", True)
        P.SPPF_FUSED = kw.get("sppf", True)
        monkeypatch.setenv("AYOLO_WGRAD3", "1" if kw.get("wgrad3") else "0")      # read when the plan's group tables are built (1: every
                                                                                  # stride-1 3x3 layer on the retired k_wgrad3)
        try:
            loss, _, g = _train_step(m, x, t, amp=True)
            if kw.get("use_plan", True):
                pl = [v for v in m._plans.values() if v][0]
                assert (pl.xf_layers >= 10) == kw.get("xf", True), pl.xf_layers
        finally:
            P.BN_REDUCE_IN_DGRAD, P.XF_ON_LOAD, P.XF_WGRAD_ON_LOAD, P.BN_APPLY_PAIR, P.SPPF_FUSED = old, old_xf, old_xw, old_pair, old_sppf
            m.use_plan = True
        last.append(g)
        return loss, _flat(g)

    last = []
    l0, g0 = run()
    l1, g1 = run()
    assert l0 == l1, (l0, l1)
    assert _cos(g0, g1) >= 1.0 - 1e-9 and float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max())
    # round 4: weight gradients are split-K partials added in a fixed order (no atomics) -- every conv weight gradient of the
    # step repeats BIT FOR BIT (the stem's dedicated kernel still sends one set of atomics per workgroup; BatchNorm / bias
    # gradients are sums of atomically accumulated fp32 / fp64 partials)
    conv_w = [k for k, v in last[0].items() if v.dim() == 4 and not k.startswith("model.0.")]
    assert len(conv_w) >= 59
    not_equal = [k for k in conv_w if not torch.equal(last[0][k], last[1][k])]
    assert not not_equal, not_equal
    # transform on load (11 blocks whose BatchNorm + SiLU pass is folded into their 1x1 reader) against the materialised route:
    # the reader forms the SAME activation bits on the way to the MFMAs, in the forward conv and in its weight gradient, so the
    # loss and every conv weight gradient of the step are bit-identical with the switch off
    l4, g4 = run(xf=False)
    assert l4 == l0, (l4, l0)
    not_equal = [k for k in conv_w if not torch.equal(last[0][k], last[-1][k])]
    assert not not_equal, not_equal
    assert _cos(g0, g4) >= 1.0 - 1e-9
    # ... and with the weight gradients transforming on load as well (k_wgrad reads z; one job per input segment)
    l5, g5 = run(xf_wgrad=True)
    assert l5 == l0, (l5, l0)
    assert _cos(g0, g5) >= 1.0 - 1e-9
    bad = [k for k in conv_w if float((last[0][k] - last[-1][k]).abs().max()) > 1e-6 * float(last[0][k].abs().max())]
    assert not bad, bad
    # the stride-1 3x3 weight gradients on the patch-staged k_wgrad3 (all of them here): same fp16 operands, fp32 partial sums in
    # another order
    l6, g6 = run(wgrad3=True)
    assert l6 == l0, (l6, l0)
    assert _cos(g0, g6) >= 1.0 - 1e-9
    bad = [k for k in conv_w if float((last[0][k] - last[-1][k]).abs().max()) > 1e-5 * float(last[0][k].abs().max())]
    assert not bad, bad
    # round 6: the BatchNorm-backward apply of a merged cv1 | cv2 pair as one launch (ayolo_bn_act_bwd_apply2) against two launches
    # over half rows: the same arithmetic per element -- the loss and every conv weight gradient bit for bit, the rest as close as
    # two runs of one route are (BatchNorm gradients come from fp64 sums accumulated by atomics: their last fp32 place may move)
    l7, g7 = run(pair=False)
    assert l7 == l0, (l7, l0)
    not_equal = [k for k in conv_w if not torch.equal(last[0][k], last[-1][k])]
    assert not not_equal, not_equal
    assert _cos(g0, g7) >= 1.0 - 1e-9 and float((g0 - g7).abs().max()) <= 1e-5 * float(g0.abs().max())
    # ... and SPPF's pool cascade in one launch per direction (ayolo_sppf_pool_fwd / _bwd) against three pool launches: values and
    # window positions are the scan's, the backward sums are exact where the launches' sequential fp32 sums round -- bit-identical
    # loss, gradients equal up to a stray last place of the pooled map's gradient
    l8, g8 = run(sppf=False)
    assert l8 == l0, (l8, l0)
    assert _cos(g0, g8) >= 1.0 - 1e-9 and float((g0 - g8).abs().max()) <= 1e-4 * float(g0.abs().max())
    l2, g2 = run(bnr=False)
    l3, g3 = run(use_plan=False)
    print("fp16 step: same route twice cos %.9f; epilogue sums vs reduce pass cos %.9f; plan vs module path cos %.9f"
          % (_cos(g0, g1), _cos(g0, g2), _cos(g0, g3)))
    # measured: 1 - 1.9e-6 (epilogue sums vs reduce pass: the sums' fp32 partials add in a different order, which moves the
    # fp16 rounding of a few dz elements)
    assert abs(l2 - l0) <= 1e-6 * abs(l0) and _cos(g0, g2) >= 1.0 - 1e-5
    # the per-module path is different ARITHMETIC at the fp16 level, not only a different order: the Bottleneck shortcut is a
    # separate fp16 add there (the sum is rounded twice; the plan adds inside the BatchNorm + SiLU pass and rounds once), so it
    # is held to what one fp16 rounding per residual block costs on this well-conditioned net.  That cost is a BAND, not a number:
    # over six seeds 0.9985 ... 0.9992 with the round-4 summation order of the BatchNorm sums and 0.9973 ... 0.9993 with round 5's
    # (tools/route_noise.py, profiles/r05_route_noise.txt; this seed: 0.99910 / 0.99868) -- which fp16 roundings flip depends on the
    # last fp32 bit of every statistic.  The threshold sits at the band's lower edge, the same 0.997 the oracle comparison holds.
    assert abs(l3 - l0) <= 1e-4 * abs(l0) and _cos(g0, g3) >= 0.997, (l3, l0, _cos(g0, g3))


@pytest.mark.parametrize("name,size,thr", [("s", 320, (0.997, 0.99)), ("l", 256, (0.997, 0.99))])
def test_fp16_train_step_well_conditioned_vs_oracle(name, size, thr):
    """The bench's model and dtype against the CPU oracle where the comparison is decided by the kernels and not by the
    conditioning of a random-init BatchNorm network: BatchNorm gains of 0.3 (a trained net's are well below the 1.0 of
    the default initialisation; with them a 3e-4 perturbation of the input moves the exact-fp32 gradient by cosine 0.9995
    instead of 0.988) and reproducible statistics; the reference is the oracle's Jacobian applied to the oracle loss's gradient at
    the fp16 logits (_oracle_grads_at_logits; round 6: over eight bit-equivalent kernel routes 0.9979-0.9987 for YOLOv5s and
    0.9973-0.9988 for YOLOv5l, where the oracle's own step gives 0.9951-0.9988 with the loss gradient of the 8 x 8 level at cosine
    0.9938 on two of the routes).  YOLOv5s: whole-gradient cosine >= 0.997, every conv weight >= 0.99, loss within 2e-4.  YOLOv5l (BASELINE cfg 3's model, VERDICT r3 item 9 -- until
    round 4 it was held only to an in-test conditioning calibration): the same check and thresholds on the deeper net (measured
    0.9993 / 0.9990)."""
    from ayolov2_amd.losses import ComputeLoss
    m, r = _pair(name, seed=39)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.fill_(0.3)
    r.load_state_dict(m.state_dict())
    for mod in (m, r):
        mod.hyp, mod.gr, mod.nc = dict(HYP), 1.0, 80
    m, r = m.cuda().train(), r.train()
    x, t = torch.rand(4, 3, size, size), _targets(4, 40)
    loss_r, _ = ComputeLoss(r)(r(x), t)
    loss_r.backward()
    gr = {k: p.grad.detach() for k, p in r.named_parameters()}
    l16, raws16, g16 = _train_step(m, x.cuda(), t.cuda(), amp=True)
    assert abs(l16 - float(loss_r.detach())) <= (2e-4 if name == "s" else 1e-3) * abs(float(loss_r.detach())), (l16, float(loss_r.detach()))
    direct = _cos(_flat({k: v.cpu() for k, v in g16.items()}), _flat(gr))
    gr, cond = _oracle_grads_at_logits(r, x, t, raws16)
    glob = _cos(_flat({k: v.cpu() for k, v in g16.items()}), _flat(gr))
    worst = min((_cos(g16[k].cpu().flatten().double(), gr[k].flatten().double()), k) for k in gr if gr[k].dim() == 4)
    print("yolov5%s fp16 (BN gain 0.3) vs oracle: whole-gradient cosine %.5f, worst conv weight %.5f (%s); against the oracle's own step "
          "%.5f, loss-gradient cosine per level %s" % (name, glob, worst[0], worst[1], direct, [round(c, 5) for c in cond]))
    assert glob >= thr[0], glob
    assert worst[0] >= thr[1], worst


@pytest.mark.parametrize("name,batch,thr", [("s", 64, (0.975, 0.96, 0.93)), ("l", 16, None)])
def test_full_size_fp16_step_vs_fp32_mode(name, batch, thr):
    """The exact configuration bench.py times (YOLOv5s, batch 64, 640x640, fp16 autocast) against the exact-fp32 mode of
    the same step: loss within 1e-3 relative (measured 2.5e-5) and the gradients' directions (cosine, thresholds from
    measurement, see below) -- every fp16 kernel variant (32x32x16 MFMA, permlane 16-byte stores, 256-pixel tiles, merged C3 siblings) at its full-size
    launch geometry.  Gradients of ~10^6 cancelling terms keep their direction; their norm carries the fp16 rounding.
    Second case: BASELINE cfg 3's model (YOLOv5l, 640x640) at batch 16.  At random initialisation this deeper BatchNorm
    network's gradient is ill-conditioned: in the EXACT-fp32 mode a relative perturbation of 3e-4 of the input image alone
    (fp16's rounding unit is 4.9e-4) already turns the whole gradient to cosine 0.72 (YOLOv5s: 0.993), measured on MI355X
    with tools/grad_conditioning.py.  So the fp16 step is held against that calibration, computed here (measured: fp16
    0.56-0.67 vs calibration 0.72), and YOLOv5l's kernel variants at their full-size launch geometry are pinned where the
    problem is well conditioned: layer by layer in tests/test_gpu_conv.py::test_full_size_layers_fp16_vs_fp32_mode."""
    from ayolov2_amd import YOLOModel
    torch.manual_seed(28)
    m = YOLOModel(os.path.join(CFG, f"yolov5{name}.yaml")).cuda().train()
    m.hyp, m.gr, m.nc = dict(HYP), 1.0, 80
    # trained-like BN statistics / affine so that activations are not at their random-init extremes
    x, t = torch.rand(batch, 3, 640, 640).cuda(), _targets(batch, 29).cuda()
    sd = copy.deepcopy(m.state_dict())
    l16, _, g16 = _train_step(m, x, t, amp=True)
    m.load_state_dict(sd)
    l32, _, g32 = _train_step(m, x, t, amp=False)
    assert abs(l16 - l32) <= 1e-3 * abs(l32), (l16, l32)

    def cos(a, b):
        a, b = a.flatten().double(), b.flatten().double()
        return float((a @ b) / (a.norm() * b.norm() + 1e-300))

    glob = cos(torch.cat([g16[k].flatten() for k in g32]), torch.cat([g32[k].flatten() for k in g32]))
    worst_w, worst_v = (1.0, None), (1.0, None)
    for k in g32:
        if float(g32[k].norm()) < 1e-12:
            continue
        c = cos(g16[k], g32[k])
        if g32[k].dim() == 4:
            worst_w = min(worst_w, (c, k))
        else:
            worst_v = min(worst_v, (c, k))
    print(f"yolov5{name} full-size fp16 vs fp32: loss %.6f / %.6f, gradient cosine: whole model %.6f, worst conv weight %.5f (%s), "
          "worst BN / bias vector %.5f (%s)" % (l16, l32, glob, worst_w[0], worst_w[1], worst_v[0], worst_v[1]))
    # Measured on MI355X: loss agrees to 2.5e-5; cosine 0.988-0.992 for the whole gradient, 0.985-0.991 for the worst conv
    # weight, 0.978-0.988 for the worst BatchNorm vector (run-to-run: fp32 atomics order).  The 0.999 one would like is out of reach for fp16 at random initialisation:
    # every gradient element is a sum of ~10^6 signed terms that cancel to ~1e-3 of their mass, and fp16 rounds each
    # activation to 1e-3.  The thresholds sit just under the measured values so that a broken kernel variant (which moves
    # the cosine to < 0.9) cannot pass.
    if thr is None:
        m.load_state_dict(sd)
        g = torch.Generator(device="cuda").manual_seed(30)
        _, _, gp = _train_step(m, x * (1 + 3e-4 * torch.randn(x.shape, device="cuda", generator=g)), t, amp=False)
        calib = cos(torch.cat([gp[k].flatten() for k in g32]), torch.cat([g32[k].flatten() for k in g32]))
        print(f"yolov5{name}: exact-fp32 gradient under a 3e-4 input perturbation: cosine %.4f" % calib)
        assert glob >= 0.5 * calib and glob >= 0.35, (glob, calib)      # measured 0.56-0.67 vs calibration 0.72-0.86
        return
    assert glob >= thr[0], glob
    assert worst_w[0] >= thr[1], worst_w
    assert worst_v[0] >= thr[2], worst_v


def test_cfg3_model_yolov5l_fp16_train_step_vs_oracle():
    """BASELINE cfg 3's model and dtype on one GPU (the 8-GPU exchange itself is covered by the bucket tests and measured by
    the driver): YOLOv5l, fp16 autocast with a scaled loss, one train step at 4 x 320^2 against the CPU oracle, at a size and
    with BatchNorm gains (0.3, see test_fp16_train_step_well_conditioned_vs_oracle) where the comparison is decided by the kernels:
    the stride-32 level normalises over 400 samples per channel (until round 5 this test ran 2 x 256^2 at random initialisation
    -- 128 samples behind ~100 fp16 layers -- and could only hold the logits to 10 % of their range and the gradient to cosine
    0.80).  Logits within 0.5 % of the range per level (measured 0.05 %), loss within 2e-3, whole-gradient cosine >= 0.99 (measured
    0.9972) against the oracle's Jacobian
    at the fp16 logits (_oracle_grads_at_logits) and >= 0.98 against the oracle's own step."""
    from ayolov2_amd.losses import ComputeLoss
    m, r = _pair("l", seed=33)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.fill_(0.3)
    r.load_state_dict(m.state_dict())
    for mod in (m, r):
        mod.hyp, mod.gr, mod.nc = dict(HYP), 1.0, 80
    m, r = m.cuda().train(), r.train()
    x, t = torch.rand(4, 3, 320, 320), _targets(4, 34)
    raws_r = r(x)
    loss_r, _ = ComputeLoss(r)(raws_r, t)
    loss_r.backward()
    gr_direct = {k: p.grad.detach().clone() for k, p in r.named_parameters()}
    l16, raws16, g16 = _train_step(m, x.cuda(), t.cuda(), amp=True)
    worst = max(float((a.cpu() - b.detach()).abs().max()) / float(b.max() - b.min()) for a, b in zip(raws16, raws_r))
    print("yolov5l fp16 vs oracle: loss %.6f / %.6f, logits within %.4f of the range" % (l16, float(loss_r.detach()), worst))
    assert worst <= 0.005, worst
    assert abs(l16 - float(loss_r.detach())) <= 2e-3 * abs(float(loss_r.detach()))
    gr, cond = _oracle_grads_at_logits(r, x, t, raws16)
    a_ = _flat({k: v.cpu() for k, v in g16.items()})
    cos, direct = _cos(a_, _flat({k: gr[k] for k in g16})), _cos(a_, _flat({k: gr_direct[k] for k in g16}))
    print("yolov5l fp16 vs oracle: whole-gradient cosine %.5f (oracle Jacobian at the fp16 logits), %.5f (oracle's own step), loss-gradient "
          "cosine per level %s" % (cos, direct, [round(c, 5) for c in cond]))
    assert cos >= 0.99, cos
    assert direct >= 0.98, direct


def test_decomposed_model_train_step_fp32():
    """Fine-tuning a Tucker-decomposed model (round 1 raised): 1x1 -> kxk -> 1x1 blocks whose ranks are NOT multiples of the
    vector width (zero-padded channel copies on the per-module path), batch-statistics BatchNorm behind the last conv.
    One train step of decomposed YOLOv5n in the exact-fp32 mode vs the identically decomposed CPU oracle."""
    from ayolov2_amd import decomposition as D
    from ayolov2_amd.losses import ComputeLoss
    from ayolov2_amd.modules import Conv
    m, r = _pair("n", seed=41)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, Conv) and mod.conv.kernel_size != (1, 1):
                w = mod.conv.weight.data
                co, ci, kh, kw = w.shape
                ro, ri = max(co // 4 + 1, 2), max(ci // 4 + 1, 2)                # 5, 9, 17, 33 ...: off the 4 / 8 grid
                std = 1.0 / (ci * kh * kw) ** 0.5
                w.copy_(torch.einsum("abhw,oa,ib->oihw", torch.randn(ro, ri, kh, kw), torch.randn(co, ro), torch.randn(ci, ri))
                        * (std / (ro * ri) ** 0.5) + 0.03 * std * torch.randn_like(w))
    r.load_state_dict(m.state_dict())
    D.decompose_model(m, loss_thr=0.1, prune_step=0.0)
    D.decompose_model(r, loss_thr=0.1, prune_step=0.0)
    ranks = D.decomposed_ranks(m)
    assert len(ranks) >= 10 and any(a % 4 or b % 4 for a, b in ranks.values()), ranks
    for mod in (m, r):
        mod.hyp, mod.gr, mod.nc = dict(HYP), 1.0, 80
    m, r = m.cuda().train(), r.train()
    x, t = torch.rand(2, 3, 128, 160), _targets(2, 42)
    raws_r = r(x)
    loss_r, _ = ComputeLoss(r)(raws_r, t)
    loss_r.backward()
    raws_g = m(x.cuda())
    loss_g, _ = ComputeLoss(m)(raws_g, t.cuda())
    loss_g.backward()
    for a, b in zip(raws_g, raws_r):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-4, atol=2e-4)
    gr = dict(r.named_parameters())
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert float((p.grad.cpu() - gr[k].grad).abs().max()) <= 3e-3 * float(gr[k].grad.abs().max()) + 1e-7, k


def test_full_size_c3_block_vs_cpu_oracle():
    """the bench's FULL-SIZE geometry meets the oracle.  The first three backbone rows of YOLOv5s (6x6 stem,
    3x3 / stride-2 Conv, C3 with its merged cv1 | cv2, Bottleneck and cv3 -- res/configs/model/yolov5s.yaml:21-23) plus a
    one-level YOLOHead at batch 64, 640 x 640 -- the 64 x 160 x 160 maps of the train step, where the training plan runs the
    transform-on-load reader with store-back, BatchNorm statistics over 1.6 M pixels, the grouped weight gradients incl. the
    patch-staged 3x3 kernel and the fused stem backward -- in the exact-fp32 mode against the CPU oracle (plain torch fp32 on the
    box's host cores): logits to 1e-4, every parameter gradient to 2e-3 of its largest element; then the fp16 autocast mode on
    operands pre-rounded to fp16 against the fp32 mode of the same kernels."""
    from ayolov2_amd import YOLOModel
    from oracle.model_ref import RefYOLO
    anchors = [[10, 13, 16, 30, 33, 23]]
    cfg = dict(input_size=[640, 640], input_channel=3, depth_multiple=0.33, width_multiple=0.5, n_classes=3, activation="SiLU",
               anchors=anchors,
               backbone=[[-1, 1, "Conv", [64, 6, 2, 2], {"activation": "SiLU"}], [-1, 1, "Conv", [128, 3, 2], {"activation": "SiLU"}],
                         [-1, 3, "C3", [128], {"activation": "SiLU"}]],
               head=[[[2], 1, "YOLOHead", [3, anchors]]])
    torch.manual_seed(61)
    m = YOLOModel(copy.deepcopy(cfg))
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.6, 1.4)
                mod.bias.uniform_(-0.3, 0.3)
        for p in m.parameters():                    # operands exactly representable in fp16: the fp16 leg below reads the same numbers
            p.copy_(p.half().float())
    r = RefYOLO(copy.deepcopy(cfg))
    r.load_state_dict(m.state_dict())
    m, r = m.cuda().train(), r.train()
    B = 64
    x = torch.rand(B, 3, 640, 640, generator=torch.Generator().manual_seed(62)).half().float()
    wv = torch.linspace(-1.0, 1.0, 8)

    def loss_of(raws):
        o = raws[0].float()
        return 0.5 * (o * o).mean() + (o * wv.to(o.device)).mean()

    def run(mod, xin, amp):
        mod.zero_grad(set_to_none=True)
        scale = 1024.0 if amp else 1.0
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            raws = mod(xin)
            loss = loss_of(raws if isinstance(raws, (list, tuple)) else [raws])
        (loss * scale).backward()
        rr = raws if isinstance(raws, (list, tuple)) else [raws]
        return float(loss.detach()), rr[0].detach().float().cpu(), {k: p.grad.detach().float().cpu() / scale for k, p in mod.named_parameters()}

    assert r(x[:1])[0].shape[1:] == (3, 160, 160, 8)
    l_ref, o_ref, g_ref = run(r, x, False)
    l32, o32, g32 = run(m, x.cuda(), False)
    assert getattr(m, "_plans", None), "the training plan must have taken this model"
    assert abs(l32 - l_ref) <= 1e-5 * abs(l_ref) + 1e-7, (l32, l_ref)
    err = (o32 - o_ref).abs()
    assert bool((err <= 1e-4 + 1e-4 * o_ref.abs()).all()), float(err.max())
    worst = 0.0
    for k, g in g_ref.items():
        e = float((g32[k] - g).abs().max()) / (float(g.abs().max()) + 1e-30)
        worst = max(worst, e)
        assert e <= 2e-3, (k, e)
    print("full-size C3 block, fp32 mode vs CPU oracle: logits max err %.3g, worst gradient error %.3g of its max" % (float(err.max()), worst))
    # ---- the bench's dtype on the same (fp16-exact) operands against the fp32 mode of the same kernels
    l16, o16, g16 = run(m, x.cuda(), True)
    rng = float(o32.abs().max())
    e16 = float((o16 - o32).abs().max())

    def cos(a, b):
        a, b = a.flatten().double(), b.flatten().double()
        return float((a @ b) / (a.norm() * b.norm() + 1e-300))

    glob = cos(torch.cat([g16[k].flatten() for k in g32]), torch.cat([g32[k].flatten() for k in g32]))
    wconv = min(cos(g16[k], g32[k]) for k in g32 if g32[k].dim() == 4 and float(g32[k].norm()) > 1e-12)
    print("full-size C3 block, fp16 vs fp32 mode: loss %.6f / %.6f, logits max err %.3g of range %.3g, gradient cosine %.6f, worst conv weight %.5f"
          % (l16, l32, e16, rng, glob, wconv))
    assert abs(l16 - l32) <= 2e-3 * abs(l32) and e16 <= 2e-2 * rng and glob >= 0.995 and wconv >= 0.99
