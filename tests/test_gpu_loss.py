"""Fused HIP loss (csrc/loss.hip) against the reference-generated golden G6 and against the torch-op path."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fake_model(hyp, anchors, nc=80, device="cuda"):
    class Head(torch.nn.Module):
        pass

    head = Head()
    head.nl, head.na, head.nc = 3, 3, nc
    head.anchors, head.stride = anchors, torch.tensor([8., 16., 32.])

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.model = torch.nn.ModuleList([torch.nn.Identity(), head])
            self.hyp = hyp

    return Fake().to(device)


def test_fused_loss_vs_reference_golden(golden_dir):
    """G6 (values produced by the reference's own ComputeLoss): loss, items and d loss / d preds."""
    from ayolov2_amd.losses import ComputeLoss
    g = np.load(os.path.join(golden_dir, "g6_loss.npz"))
    hyp = dict(box=float(g["hyp_box"]), cls=float(g["hyp_cls"]), obj=float(g["hyp_obj"]), cls_pw=1.0, obj_pw=1.0,
               anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    m = _fake_model(hyp, torch.from_numpy(g["anchors"]).cuda())
    gen = torch.Generator().manual_seed(int(g["pred_seed"]))
    preds = [torch.randn(2, 3, s, s, 85, generator=gen).cuda().requires_grad_(True) for s in (80, 40, 20)]
    targets = torch.from_numpy(g["targets"])
    cl = ComputeLoss(m)
    prepared = cl.prepare(targets, [tuple(p.shape) for p in preds], device=preds[0].device)
    assert cl._fusable(preds)
    loss, items = cl(preds, targets.cuda(), prepared=prepared)
    loss.backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g["loss"], rtol=1e-5)
    np.testing.assert_allclose(items.cpu().numpy(), g["items"], rtol=1e-5)
    for i in range(3):
        gr = preds[i].grad.cpu()
        np.testing.assert_allclose(gr.sum((2, 3)).numpy(), g[f"grad{i}_sum"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(gr.abs().sum().numpy(), g[f"grad{i}_abs_total"], rtol=1e-4)
    # G6b: elementwise -- all 85 logits of every matched cell, objectness logit on a 3 x 3 sub-lattice (the reference's run)
    gb = np.load(os.path.join(golden_dir, "g6b_loss_grads.npz"))
    for i in range(3):
        b, a, gj, gi = (torch.from_numpy(v) for v in g[f"idx{i}"])
        gr = preds[i].grad.cpu()
        scale = float(np.abs(gb[f"rows{i}"]).max())
        np.testing.assert_allclose(gr[b, a, gj, gi].numpy(), gb[f"rows{i}"], rtol=2e-4, atol=2e-6 * scale)
        np.testing.assert_allclose(gr[:, :, ::3, ::3, 4].numpy(), gb[f"obj{i}"], rtol=2e-4, atol=2e-6 * scale)


@pytest.mark.parametrize("variant", ["plain", "smooth_pw_gr", "strided_scaled"])
def test_fused_loss_equals_torch_path(variant):
    """Elementwise gradient equality with the torch-op path run on the CPU, where the duplicate-cell objectness write
    `tobj[b, a, gj, gi] = score` is sequential (last row wins; the same index_put is unordered on a GPU).  Covers
    duplicate cells, label smoothing, pos_weight, gr < 1, NHWC-strided logits as HeadConvFn produces them and a
    GradScaler-like factor on the loss."""
    from ayolov2_amd.losses import ComputeLoss
    torch.manual_seed(1)
    hyp = dict(box=0.05, cls=0.5, obj=1.0, cls_pw=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    if variant == "smooth_pw_gr":
        hyp.update(cls_pw=0.7, obj_pw=1.3, label_smoothing=0.1)
    anchors = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]],
                           dtype=torch.float32).view(3, 3, 2) / torch.tensor([8., 16., 32.]).view(3, 1, 1)
    B = 4
    shapes = [(B, 3, 32, 24, 85), (B, 3, 16, 12, 85), (B, 3, 8, 6, 85)]
    nt = 60
    targets = torch.cat((torch.randint(0, B, (nt, 1)).float(), torch.randint(0, 80, (nt, 1)).float(),
                         torch.rand(nt, 2) * 0.9 + 0.05, torch.rand(nt, 2) * 0.3 + 0.02), 1)
    targets = torch.cat((targets, targets[:10]), 0)          # exact duplicates -> rows sharing a cell
    base = [torch.randn(s) for s in shapes]

    def make(i, fused):
        if not fused:
            return base[i].clone().requires_grad_(True), None
        if variant != "strided_scaled":
            return base[i].clone().cuda().requires_grad_(True), None
        b, na, ny, nx, no = shapes[i]
        buf = torch.zeros(b, ny, nx, 256, device="cuda")
        raw = buf.as_strided((b, na, ny, nx, no), (ny * nx * 256, no, nx * 256, 256, 1))
        raw.copy_(base[i].cuda())
        leaf = buf.requires_grad_(True)
        return leaf.as_strided((b, na, ny, nx, no), (ny * nx * 256, no, nx * 256, 256, 1)), leaf

    outs = {}
    for fused in (True, False):
        dev = "cuda" if fused else "cpu"
        cl = ComputeLoss(_fake_model(hyp, anchors.to(dev), device=dev))
        if variant == "smooth_pw_gr":
            cl.gr = 0.6
        made = [make(i, fused) for i in range(3)]
        preds = [p for p, _ in made]
        prepared = cl.prepare(targets, [tuple(p.shape) for p in preds], device=preds[0].device)
        assert cl._fusable(preds) == fused
        loss, items = cl(preds, targets.to(dev), prepared=prepared)
        scale = 1024.0 if variant == "strided_scaled" else 1.0
        (loss * scale).backward()
        grads = [(leaf.grad.as_strided(p.shape, p.stride()) if leaf is not None else p.grad).detach().cpu() for p, leaf in made]
        outs[fused] = (loss.detach().cpu(), items.cpu(), grads)
    lf, itf, gf = outs[True]
    lt, itt, gt = outs[False]
    torch.testing.assert_close(lf, lt, rtol=2e-5, atol=1e-6)      # the torch path sums 1e6 terms in fp32
    torch.testing.assert_close(itf, itt, rtol=2e-5, atol=1e-6)
    for a, b in zip(gf, gt):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-9, float((a - b).abs().max()) / scale


def _mini_hyp():
    return dict(box=0.05, cls=0.5, obj=1.0, cls_pw=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


@pytest.mark.parametrize("use_plan", [True, False])
def test_packed_head_gradient_end_to_end(use_plan):
    """The loss hands its gradient to the YOLOHead backward directly in the conv's operand layout (no dense fp32
    gradient, no repack): parameter gradients must equal the dense route's, on the plan and on the module path, with
    duplicate-cell rows in the batch."""
    import copy
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.losses import ComputeLoss
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ayolov2_amd", "configs", "yolov5n.yaml")
    torch.manual_seed(4)
    m = YOLOModel(cfg).cuda().train()
    m.hyp, m.gr, m.nc = _mini_hyp(), 1.0, 80
    m.use_plan = use_plan
    m2 = copy.deepcopy(m)
    x = torch.rand(4, 3, 128, 160).cuda()
    nt = 40
    targets = torch.cat((torch.randint(0, 4, (nt, 1)).float(), torch.randint(0, 80, (nt, 1)).float(),
                         torch.rand(nt, 2) * 0.9 + 0.05, torch.rand(nt, 2) * 0.3 + 0.02), 1)
    targets = torch.cat((targets, targets[:8]), 0)
    outs = []
    for mod, packed in ((m, True), (m2, False)):
        cl = ComputeLoss(mod)
        cl.packed_head_grad = packed
        preds = mod(x)
        prepared = cl.prepare(targets, [tuple(p.shape) for p in preds], device=x.device)
        loss, items = cl(preds, targets.cuda(), prepared=prepared)
        (loss * 64.0).backward()
        outs.append((loss.detach().cpu(), {n: p.grad.detach().float().cpu() for n, p in mod.named_parameters()}))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-7)
    for n, g1 in outs[0][1].items():
        g2 = outs[1][1][n]
        scale = float(g2.abs().max()) + 1e-12
        # the dense route rounds the fp32 gradient to the compute dtype in a second pass; same values, same order
        assert float((g1 - g2).abs().max()) <= 2e-3 * scale + 1e-7, (n, float((g1 - g2).abs().max()) / scale)


@pytest.mark.parametrize("use_plan", [True, False])
def test_packed_head_gradient_with_second_consumer(use_plan):
    """The raw logits have a SECOND differentiable consumer (an auxiliary term, as a distillation loss would add): autograd
    then hands the head `zeros placeholder + auxiliary gradient` as a new dense tensor without the payload attribute.
    The packed loss gradient must not be dropped: parameter gradients equal the dense route's (losses.py)."""
    import copy
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.losses import ComputeLoss
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ayolov2_amd", "configs", "yolov5n.yaml")
    torch.manual_seed(5)
    m = YOLOModel(cfg).cuda().train()
    m.hyp, m.gr, m.nc = _mini_hyp(), 1.0, 80
    m.use_plan = use_plan
    m2 = copy.deepcopy(m)
    x = torch.rand(2, 3, 128, 128).cuda()
    nt = 20
    targets = torch.cat((torch.randint(0, 2, (nt, 1)).float(), torch.randint(0, 80, (nt, 1)).float(),
                         torch.rand(nt, 2) * 0.9 + 0.05, torch.rand(nt, 2) * 0.3 + 0.02), 1)
    outs = []
    for mod, packed in ((m, True), (m2, False)):
        cl = ComputeLoss(mod)
        cl.packed_head_grad = packed
        preds = mod(x)
        prepared = cl.prepare(targets, [tuple(p.shape) for p in preds], device=x.device)
        loss, _ = cl(preds, targets.cuda(), prepared=prepared)
        aux = sum((p[..., :5] ** 2).mean() for p in preds)          # second consumer of the same logits
        (loss + 3.0 * aux).backward()
        outs.append({n: p.grad.detach().float().cpu() for n, p in mod.named_parameters()})
    for n, g1 in outs[0].items():
        g2 = outs[1][n]
        scale = float(g2.abs().max()) + 1e-12
        assert float((g1 - g2).abs().max()) <= 2e-3 * scale + 1e-7, (n, float((g1 - g2).abs().max()) / scale)


def test_prepare_staging_is_not_overwritten_by_the_next_step():
    """prepare() copies the assigned targets through pinned staging asynchronously while the GPU is still busy with the
    previous step; consecutive steps with DIFFERENT labels (different row counts per level) must each see their own
    (a single staging buffer let step k train on step k+1's targets)."""
    from ayolov2_amd.losses import ComputeLoss
    torch.manual_seed(6)
    anchors = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]],
                           dtype=torch.float32).view(3, 3, 2) / torch.tensor([8., 16., 32.]).view(3, 1, 1)
    cl = ComputeLoss(_fake_model(_mini_hyp(), anchors.cuda()))
    B = 4
    shapes = [(B, 3, 32, 32, 85), (B, 3, 16, 16, 85), (B, 3, 8, 8, 85)]
    preds = [torch.randn(s, device="cuda") for s in shapes]

    def make_targets(nt):
        return torch.cat((torch.randint(0, B, (nt, 1)).float(), torch.randint(0, 80, (nt, 1)).float(),
                          torch.rand(nt, 2) * 0.9 + 0.05, torch.rand(nt, 2) * 0.3 + 0.02), 1)

    tlist = [make_targets(n) for n in (50, 7, 90, 23, 61)]
    tdev = [t.cuda() for t in tlist]                  # (a pageable upload inside the loop would itself synchronise)
    want = []
    for t, td in zip(tlist, tdev):                    # reference: one step at a time, fully synchronised
        prep = cl.prepare(t, shapes, device="cuda")
        torch.cuda.synchronize()
        want.append(float(cl(preds, td, prepared=prep)[0]))
        torch.cuda.synchronize()
    big = torch.randn(8192, 8192, device="cuda")
    got = []
    for t, td in zip(tlist, tdev):                    # back to back behind a long-running kernel queue: host runs ahead
        for _ in range(4):
            big = big @ big * 1e-4
        prep = cl.prepare(t, shapes, device="cuda")
        got.append(cl(preds, td, prepared=prep)[0])
    torch.cuda.synchronize()
    np.testing.assert_allclose([float(g) for g in got], want, rtol=1e-6)
