"""CPU-only checks of the host side: model construction KATs, state-dict naming, loss vs golden, C-ABI exports."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_DIR = os.path.join(ROOT, "ayolov2_amd", "configs")


@pytest.mark.parametrize("name,want", [("n", 1872157), ("s", 7235389), ("m", 21190557), ("l", 46563709), ("x", 86749405)])
def test_param_count_kat(name, want):
    """K1: README.md:206-211 parameter counts (architecture known-answer test)."""
    from ayolov2_amd import YOLOModel
    m = YOLOModel(os.path.join(CFG_DIR, f"yolov5{name}.yaml"))
    assert sum(p.numel() for p in m.parameters()) == want
    assert m.stride.tolist() == [8.0, 16.0, 32.0]
    head = m.model[-1]
    assert (head.nl, head.na, head.nc) == (3, 3, 80)
    assert all(k.startswith("model.") for k in m.state_dict())
    # 640 -> 25200 proposals, 1280 -> 100800
    assert sum(3 * (640 // int(s)) ** 2 for s in m.stride) == 25200
    assert sum(3 * (1280 // int(s)) ** 2 for s in m.stride) == 100800


def test_oracle_model_shares_state_dict():
    from ayolov2_amd import YOLOModel
    from oracle.model_ref import RefYOLO
    cfg = os.path.join(CFG_DIR, "yolov5s.yaml")
    m, r = YOLOModel(cfg), RefYOLO(cfg)
    assert set(m.state_dict()) == set(r.state_dict())
    r.load_state_dict(m.state_dict())


def test_model_is_picklable_and_decomposable_surface():
    import copy
    import pickle
    from torch import nn
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.modules import Conv
    m = YOLOModel(os.path.join(CFG_DIR, "yolov5n.yaml"))
    m2 = pickle.loads(pickle.dumps(m))
    assert sum(p.numel() for p in m2.parameters()) == 1872157
    copy.deepcopy(m).half().float()
    # every Conv block exposes `.conv: nn.Conv2d` as a direct child (decomposition.py:262-272)
    convs = [mod for mod in m.modules() if isinstance(mod, Conv)]
    assert len(convs) == 57 and all(isinstance(c.conv, nn.Conv2d) for c in convs)
    assert isinstance(m.model[-1].conv, nn.ModuleList) and len(m.model[-1].conv) == 3


def test_product_raises_without_gpu_tensor():
    from ayolov2_amd import YOLOModel, _lib
    from ayolov2_amd.metrics import non_max_suppression
    m = YOLOModel(os.path.join(CFG_DIR, "yolov5n.yaml"))
    with pytest.raises(_lib.AyoloError):
        m(torch.rand(1, 3, 64, 64))
    with pytest.raises(_lib.AyoloError):
        non_max_suppression(torch.rand(1, 10, 85))


def test_loss_vs_golden(golden_dir):
    """G6: ComputeLoss value, items, build_targets indices and d loss / d preds vs the reference's own run."""
    from ayolov2_amd.losses import ComputeLoss
    g = np.load(os.path.join(golden_dir, "g6_loss.npz"))
    hyp = dict(box=float(g["hyp_box"]), cls=float(g["hyp_cls"]), obj=float(g["hyp_obj"]), cls_pw=1.0, obj_pw=1.0,
               anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)

    class Head(torch.nn.Module):
        pass

    head = Head()
    head.nl, head.na, head.nc = 3, 3, 80
    head.anchors, head.stride = torch.from_numpy(g["anchors"]), torch.tensor([8., 16., 32.])

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.model = torch.nn.ModuleList([torch.nn.Identity(), head])
            self.hyp = hyp

    gen = torch.Generator().manual_seed(int(g["pred_seed"]))
    preds = [torch.randn(2, 3, s, s, 85, generator=gen).requires_grad_(True) for s in (80, 40, 20)]
    targets = torch.from_numpy(g["targets"])
    cl = ComputeLoss(Fake())
    loss, items = cl(preds, targets)
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), g["loss"], rtol=1e-5)
    np.testing.assert_allclose(items.numpy(), g["items"], rtol=1e-5)
    tcls, tbox, indices, anch = cl.build_targets(preds, targets)
    for i in range(3):
        np.testing.assert_array_equal(tcls[i].numpy(), g[f"tcls{i}"])
        np.testing.assert_allclose(tbox[i].numpy(), g[f"tbox{i}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(torch.stack(indices[i]).numpy(), g[f"idx{i}"])
        np.testing.assert_allclose(anch[i].numpy(), g[f"anch{i}"])
        np.testing.assert_allclose(preds[i].grad.sum((2, 3)).numpy(), g[f"grad{i}_sum"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(preds[i].grad.abs().sum().numpy(), g[f"grad{i}_abs_total"], rtol=1e-4)
    # G6b: the same run's gradient ELEMENTWISE -- all 85 logits of every matched cell and the objectness logit on a 3 x 3
    # sub-lattice of every level (tools/make_golden.py g6b)
    gb = np.load(os.path.join(golden_dir, "g6b_loss_grads.npz"))
    for i in range(3):
        b, a, gj, gi = (torch.from_numpy(v) for v in g[f"idx{i}"])
        scale = float(np.abs(gb[f"rows{i}"]).max())
        np.testing.assert_allclose(preds[i].grad[b, a, gj, gi].numpy(), gb[f"rows{i}"], rtol=1e-4, atol=1e-6 * scale)
        np.testing.assert_allclose(preds[i].grad[:, :, ::3, ::3, 4].numpy(), gb[f"obj{i}"], rtol=1e-4, atol=1e-6 * scale)


def test_c_abi_exports_every_declared_symbol():
    """libayolo_hip.so loads (no GPU needed) and exports exactly the entry points include/ayolo.h declares."""
    from ayolov2_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ayolo.h")).read()
    declared = sorted(set(re.findall(r"\b(ayolo_[a-z0-9_]+)\s*\(", hdr)))
    assert os.path.exists(_lib.LIB_PATH), "build the library first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ayolo.h but not exported"
    assert sorted(_lib.EXPORTED) == declared
    assert lib.ayolo_version() >= 1


def test_general_helpers_vs_golden(golden_dir):
    from ayolov2_amd import general
    g = np.load(os.path.join(golden_dir, "g3_general.npz"))
    t = torch.from_numpy
    np.testing.assert_array_equal(general.xywh2xyxy(t(g["x"])).numpy(), g["xywh2xyxy"])
    np.testing.assert_array_equal(general.clip_coords(t(g["xyxy"].copy()), (640, 480)).numpy(), g["clip"])
    np.testing.assert_array_equal(general.scale_coords((640, 640), t(g["xyxy"].copy()), (480, 600)).numpy(), g["scale_a"])
    np.testing.assert_array_equal(
        general.scale_coords((640, 640), t(g["xyxy"].copy()), (720, 1280), ratio_pad=((0.5, 0.5), (0.0, 140.0))).numpy(),
        g["scale_b"])


def test_xyxy2xywh_vs_golden(golden_dir):
    """G3b: the reference's own xyxy2xywh (general.py:252-294) on numpy boxes incl. boxes that overhang the unit square;
    the torch-tensor form must agree with the numpy form."""
    from ayolov2_amd import general
    g = np.load(os.path.join(golden_dir, "g3b_xyxy2xywh.npz"))
    np.testing.assert_array_equal(general.xyxy2xywh(g["box"].copy()), g["default"])
    np.testing.assert_array_equal(general.xyxy2xywh(g["box"].copy(), check_validity=False), g["no_check"])
    np.testing.assert_array_equal(general.xyxy2xywh(g["px"].copy(), wh=(640.0, 480.0)), g["sized"])
    np.testing.assert_array_equal(general.xyxy2xywh(g["px"].copy(), wh=(640.0, 480.0), clip_eps=1e-3), g["sized_clip"])
    f32 = general.xyxy2xywh(g["box"].astype(np.float32))
    assert f32.dtype == np.float32
    np.testing.assert_array_equal(f32, g["f32"])
    t = general.xyxy2xywh(torch.from_numpy(g["px"].copy()), wh=(640.0, 480.0))
    np.testing.assert_array_equal(t.numpy(), g["sized"])


def test_grad_pool_never_reuses_a_referenced_buffer():
    """the plan hands gradients out as views of a pooled flat buffer; a buffer is reused only when nothing --
    a .grad, an autograd.grad result, a detached alias, a stashed list -- still shares its storage."""
    from ayolov2_amd.plan import TrainPlan

    class Arena:
        buf = torch.zeros(1000)

    class P:
        gradarena = Arena()
        _storage_refs = staticmethod(TrainPlan._storage_refs)

    pl = P()
    get = lambda: TrainPlan._grad_out_buffer(pl)          # noqa: E731
    f1 = get()[0]
    g = f1[10:20]                                         # a gradient view handed out of buffer 1
    f2 = get()[0]
    assert f2 is not f1
    alias = g.detach()                                    # what AccumulateGrad keeps as p.grad
    del g
    assert get()[0] is f2                                 # buffer 1 still pinned by the alias, buffer 2 idle
    held = f2[0:5]
    f3 = get()[0]
    assert f3 is not f1 and f3 is not f2                  # both pinned: a fresh buffer, not an overwrite
    del alias, held, f3
    assert get()[0] is f1
    entry = get()
    side = torch.empty(5)
    entry[1][7] = side                                    # a side buffer (the stem's compacted gradient) handed out with it
    h = side.as_strided(side.shape, side.stride())
    assert get()[0] is not f1
    del h
    assert get()[0] is f1


def test_bbox_iou_vs_golden(golden_dir):
    from ayolov2_amd.metrics import bbox_iou
    g = np.load(os.path.join(golden_dir, "g2_bbox_iou.npz"))
    p = torch.from_numpy(g["pred"]).requires_grad_(True)
    t = torch.from_numpy(g["target"])
    v = bbox_iou(p.T, t, x1y1x2y2=False, c_iou=True)
    v.sum().backward()
    np.testing.assert_allclose(v.detach().numpy(), g["ciou_0"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(p.grad.numpy(), g["ciou_0_grad"], rtol=1e-4, atol=1e-5)


def test_tucker_product_vs_golden(golden_dir):
    """G7: EVBMF ranks, Sequential shapes and probe loss of the product's Tucker-2 code vs the reference driver."""
    from torch import nn
    from ayolov2_amd import decomposition as D
    g = np.load(os.path.join(golden_dir, "g7_tucker.npz"))
    for name in "abc":
        L, M, r = g[f"shape_{name}"]
        rs = np.random.default_rng(7)
        Y = (rs.standard_normal((L, r)) @ rs.standard_normal((r, M)) / np.sqrt(r) + 0.05 * rs.standard_normal((L, M))).astype(np.float32)
        _, d, _, _ = D.EVBMF(torch.from_numpy(Y))
        assert d.shape[0] == int(g[f"rank_{name}"])
    conv = nn.Conv2d(64, 96, 3, padding=1, bias=False)
    conv.weight.data = torch.from_numpy(g["conv_w"])
    assert D.estimate_ranks(conv) == g["conv_ranks"].tolist()
    xin = torch.rand((64, 64, 3, 3), generator=torch.Generator().manual_seed(5))
    seq, loss = D.decompose_layer_evaluation(conv, xin, conv(xin))
    assert [list(m.weight.shape) for m in seq] == g["conv_shapes"].tolist()
    assert abs(float(loss) - float(g["conv_loss"])) < 1e-4


def test_decompose_model_prune_bisection_vs_golden(golden_dir):
    """G7b: the reference's DEFAULT decompose_model path -- loss_thr 0.1, prune_step 0.01, i.e. the L1-unstructured prune
    bisection of decomposition.py:296-323 (decompose_model.py:63-74) -- on a seeded two-block net: same accepted ranks, the
    same pruned-then-decomposed kernels (compared as ONE dense 3x3 kernel, which is free of the factors' sign / rotation
    ambiguity) and the same network output as the reference driver run by tools/make_golden.py."""
    from torch import nn
    from ayolov2_amd import decomposition as D
    g = np.load(os.path.join(golden_dir, "g7b_decompose_model.npz"))

    class Blk(nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)

        def forward(self, x):
            return self.conv(x)

    net = nn.Sequential(Blk(16, 24), Blk(24, 16))
    for blk, k in zip(net, ("w0", "w1")):
        blk.conv.weight.data = torch.from_numpy(g[k].copy())
    torch.manual_seed(int(g["probe_seed"]))          # the probe inputs come from the global generator, as in the reference
    D.decompose_model(net, loss_thr=0.1, prune_step=0.01)
    for i, blk in enumerate(net):
        seq = blk.conv
        assert isinstance(seq, nn.Sequential)
        assert [list(m.weight.shape) for m in seq] == g[f"shapes{i}"].tolist()
        assert (seq.in_channels, seq.out_channels, tuple(seq.kernel_size)) == (g[f"w{i}"].shape[1], g[f"w{i}"].shape[0], (3, 3))
        first, core, last = (m.weight.detach().double() for m in seq)
        dense = torch.einsum("oa,abhw,bi->oihw", last[:, :, 0, 0], core, first[:, :, 0, 0]).float().numpy()
        # a different prune ratio (one bisection step is 1 / 128 of the weights) moves the kernel by ~1e-2; HOOI by < 1e-4
        np.testing.assert_allclose(dense, g[f"dense{i}"], atol=2e-4, rtol=0)
    x = torch.rand((2, 16, 12, 12), generator=torch.Generator().manual_seed(int(g["x_seed"])))
    with torch.no_grad():
        y = net(x)
    np.testing.assert_allclose(y.numpy(), g["y"], atol=1e-3, rtol=1e-3)


def test_decompose_model_surface():
    """decompose_model swaps `.conv` for a 3-conv Sequential carrying in/out_channels/kernel_size attrs and lowers
    the parameter count (decomposition.py:325-335)."""
    from torch import nn
    from ayolov2_amd import YOLOModel
    from ayolov2_amd import decomposition as D
    from ayolov2_amd.modules import Conv
    torch.manual_seed(0)
    m = YOLOModel(os.path.join(CFG_DIR, "yolov5n.yaml"))
    # give two 3x3 convs an exactly low multilinear rank so the decomposition is accepted
    targets = [m.model[1], m.model[2].m[0].cv2]
    for blk in targets:
        w = blk.conv.weight.data
        co, ci = w.shape[:2]
        core = torch.randn(co // 4, ci // 4, 3, 3)
        w.copy_(torch.einsum("abhw,oa,ib->oihw", core, torch.randn(co, co // 4), torch.randn(ci, ci // 4)) / 8
                + 0.002 * torch.randn_like(w))
    before = sum(p.numel() for p in m.parameters())
    D.decompose_model(m, loss_thr=0.1, prune_step=0.0)
    swapped = [c for c in m.modules() if isinstance(c, Conv) and isinstance(c.conv, nn.Sequential)]
    assert all(blk in swapped for blk in targets)
    for c in swapped:
        assert len(c.conv) == 3 and c.conv.kernel_size == (3, 3) and hasattr(c.conv, "in_channels")
    assert sum(p.numel() for p in m.parameters()) < before


def test_tta_vs_golden(golden_dir):
    """G9: inference_with_tta / scale_img / descale_pred / clip_augmented against the reference's own run
    (tta_utils.py:15-86) around a deterministic YOLO-shaped stand-in model."""
    from ayolov2_amd import tta
    g = np.load(os.path.join(golden_dir, "g9_tta.npz"))

    class Head:
        nl = 3

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = [None, Head()]
            self.stride = torch.tensor([8.0, 16.0, 32.0])

        def forward(self, x):
            B, _, H, W = x.shape
            outs = []
            for s_ in (8, 16, 32):
                ny, nx = H // s_, W // s_
                pooled = torch.nn.functional.adaptive_avg_pool2d(x, (ny, nx))
                yy, xx = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32), indexing="ij")
                rows = []
                for a in range(3):
                    cx = (xx + 0.5) * s_ + pooled[:, 0] * 3
                    cy = (yy + 0.5) * s_ + pooled[:, 1] * 3
                    w_ = torch.full_like(cx, 10.0 * (a + 1)) + pooled[:, 2]
                    h_ = torch.full_like(cx, 7.0 * (a + 1)) + pooled[:, 0]
                    rest = pooled.mean(1, keepdim=True).expand(B, 4, ny, nx).permute(0, 2, 3, 1)
                    rows.append(torch.cat((torch.stack((cx, cy, w_, h_), -1), rest), -1).reshape(B, ny * nx, 8))
                outs.append(torch.cat(rows, 1))
            return torch.cat(outs, 1), None

    x = torch.from_numpy(g["x"])
    flips = [None if f == 0 else int(f) for f in g["flips"]]
    y, none = tta.inference_with_tta(Fake(), x, [float(s) for s in g["scales"]], flips)
    assert none is None
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-6, atol=1e-5)
    # one (scale, flip) pair: both cuts hit the same entry, the second computed from the already shortened row count
    y1, _ = tta.inference_with_tta(Fake(), x, [0.83], [3])
    assert y1.shape == g["y_single"].shape
    np.testing.assert_allclose(y1.numpy(), g["y_single"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(tta.scale_img(x, 0.83, gs=32).numpy(), g["scaled_083"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(tta.scale_img(x, 0.5, same_shape=True, gs=32).numpy(), g["scaled_same"], rtol=1e-6, atol=1e-6)


def test_plan_gradient_buckets_cover_the_arena():
    """The train plan compiles on CPU tensors (no kernel runs): its reverse-layer gradient buckets must tile the flat
    gradient arena exactly, become ready in backward order, and no bucket may be released before the last backward op
    that writes into it (weight-gradient / BatchNorm-apply / head-bias writers recorded at emission)."""
    import torch
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.plan import TrainPlan
    m = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", "yolov5s.yaml")).train()
    pl = TrainPlan(m, (2, 3, 64, 64), torch.float16, torch.device("cpu"))
    b = pl.buckets
    assert 3 <= len(b) <= 8
    assert b[0][2] == pl.gradarena.total and b[-1][1] == 0
    for (r0, lo0, hi0), (r1, lo1, hi1) in zip(b, b[1:]):
        assert lo0 == hi1 and r0 <= r1
    for idx, off, n in pl.grad_done:
        ready = next(r for r, lo, hi in b if lo <= off < hi)
        assert ready >= idx and off + n <= next(hi for r, lo, hi in b if lo <= off < hi)
    covered = sum(n for _, _, n in pl.grad_done)
    assert covered >= sum(p.numel() for p in m.parameters())          # every parameter has a writer (slots are padded)
    # ... and the recorded "complete after op idx" of a range is not EARLIER than an op that writes into it: every gradient-arena
    # pointer a BatchNorm-backward op carries (dgamma / dbeta of the apply pass, of the paired apply pass of a merged cv1 | cv2, the
    # stem's fused op incl. its dw) must lie in a range whose hand-over index is >= that op's index (round 6: the paired pass
    # recorded its ranges two ops early -- a bucket's all-reduce could have started before the pass had written them)
    base, end = pl.gradarena.buf.data_ptr(), pl.gradarena.buf.data_ptr() + 4 * pl.gradarena.total
    slots = {8: (8, 9), 24: (7, 8, 14, 15), 22: (8, 9, 10)}           # op kind -> pointer slots that are gradient destinations
    checked = 0
    for k, o in enumerate(pl.bwd):
        for sl in slots.get(o.kind & 0xff, ()):
            ptr = o.p[sl]
            if not ptr:
                continue
            assert base <= ptr < end, (k, sl)
            off = (ptr - base) // 4
            idxs = [idx for idx, eo, n in pl.grad_done if eo <= off < eo + n]
            assert idxs and max(idxs) >= k, (k, o.kind & 0xff, sl, idxs)
            checked += 1
    assert checked >= 2 * 57 - 10
    # C3's cv1 | cv2 run as one conv: 8 fewer forward / dgrad / wgrad launches than the 60 convolutions of the model
    from collections import Counter
    kinds = Counter(o.kind & 0xff for o in pl.bwd)
    # weight gradients: 51 layers in <= 12 grouped launches (23) + the stem's fused BN-backward + weight gradient (22)
    assert kinds[3] == 0 and kinds[2] == 51 and kinds[22] == 1 and 4 <= kinds[23] <= 12
    assert 51 <= sum(v[2] for v in pl.wgroup_costs.values()) <= 51 + kinds[23] and len(pl.wgroup_costs) == kinds[23]
    gidx = sorted(pl.wgroup_costs)
    assert all((pl.bwd[k].kind & 0xff) == 23 for k in gidx)
    # the tail of backward is cut finer than its body (what the last group still has to do after the main stream is exposed)
    assert pl.wgroup_costs[gidx[-1]][1] < 0.6 * max(v[1] for v in pl.wgroup_costs.values())
    # transform on load: 21 of the 57 BatchNorm + SiLU passes are folded into their single 1x1 reader (18 convs, 7 of them over two
    # input segments); their finalize rides in the reader's launch (p[9]) and the reader's first channel tile writes the activation
    # back for the weight gradient (p[10]), which therefore stays one plain job per layer
    fk = Counter(o.kind & 0xff for o in pl.fwd)
    assert pl.xf_layers == 21 and fk[18] == 36 and fk[5] == 0
    xfc = [o for o in pl.fwd if (o.kind & 0xff) == 1 and o.p[6]]
    assert len(xfc) == 18 and sum(1 for o in xfc if o.p[8]) == 7 and all(o.p[9] and o.p[10] for o in xfc)
    assert len(pl._wjobs) == 51 and not any(j["xf"] for j in pl._wjobs)
    # sync_bn cut points: one per conv launch (forward), one per BatchNorm layer (backward)
    assert len(pl.fwd_sync_idx) == 49 and len(pl.bwd_sync) == 57
    # fp16 plans fold the BatchNorm-backward sums of every layer whose output gradient is last written by a conv dgrad into
    # that dgrad's epilogue (all but the three fed by pool / upsample backward ops): the separate reduce op becomes a no-op
    assert pl.bn_in_dgrad == 54 and kinds[0] >= 54 and kinds[7] == 3


def test_wgrad_group_tables_cover_every_tile_once(monkeypatch):
    """The planner of the grouped weight-gradient launch (ayolo_wgrad_group_size / _build; needs no GPU): every (layer,
    dw tile, pixel split) is exactly one item of its tile class, the gx * gy tiles of one split sit on ONE XCD queue (block
    index % 8) back to back, queues are padded to equal length, the splits of a layer own disjoint workspace slots and a
    layer whose tensors exceed the 2 GiB descriptor range is cut into batch halves that share one reduction."""
    import ctypes
    from ayolov2_amd import _lib, ops
    from ayolov2_amd._lib import WgradJob
    lib = _lib.lib()
    F16 = 0
    monkeypatch.setenv("AYOLO_WGRAD3", "1")      # the patch-staged 3x3 kernel for every stride-1 3x3 layer (retired from the default
                                                 # route in round 6): its planner is part of what is checked here
    shapes = [  # (B, H, W, Cin, Cout, k, s)
        (8, 40, 40, 128, 128, 3, 1), (8, 40, 40, 256, 64, 1, 1), (8, 80, 80, 64, 32, 3, 2), (8, 20, 20, 512, 255 + 1, 1, 1),
        (192, 320, 320, 64, 64, 1, 1)]           # the last: x = dy = 2.5 GB -> two batch halves of 1.26 GB
    arr = (WgradJob * len(shapes))()
    for k, (B, H, W, Ci, Co, kk, st) in enumerate(shapes):
        pd = kk // 2
        Ho, Wo = (H + 2 * pd - kk) // st + 1, (W + 2 * pd - kk) // st + 1
        arr[k].conv = ops.make_desc(torch.float16, B, H, W, Ci, Ci, Co, Co, (kk, kk), (st, st), (pd, pd), Ho, Wo)
        arr[k].x, arr[k].dy, arr[k].dw = 0x10000, (0 if k == 3 else 0x20000), 0x30000 + 0x1000000 * k
        arr[k].alpha, arr[k].dy_slot, arr[k].overwrite = 1.0, (1 if k == 3 else -1), 1
    tb, wb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.check(lib.ayolo_wgrad_group_size(arr, len(shapes), ctypes.byref(tb), ctypes.byref(wb)), "size")
    host = ctypes.create_string_buffer(tb.value)
    _lib.check(lib.ayolo_wgrad_group_build(arr, len(shapes), host, tb.value), "build")
    out = (ctypes.c_int64 * 20)()
    _lib.check(lib.ayolo_wgrad_group_info(host, -1, out, 20), "info")
    njobs, n_items, n_red, ws_floats, njobs3 = out[0], [out[1], out[2], out[3], out[12]], out[4], out[5], out[13]
    # the fp16 3x3 / stride-1 layer is a k_wgrad3 job (item class 3, numbered behind k_wgrad's; stride 2 stays on the generic kernel
    # unless AYOLO_WGRAD3=2); the 320 x 320 layer became two halves
    assert njobs == len(shapes) - 1 + 1 and njobs3 == 1 and ws_floats * 4 == wb.value
    jobs = []
    for j in range(njobs + njobs3):
        _lib.check(lib.ayolo_wgrad_group_info(host, j, out, 20), "info")
        jobs.append(dict(tm=out[6], gx=out[7], gy=out[8], splits=out[9], chunk=out[10], zz0=out[11], TC=out[14], RPS=out[15],
                         strips=out[16], NB=out[17], CB=out[18], stage=out[19]))
        assert jobs[-1]["splits"] >= 1
        if j < njobs:
            assert jobs[-1]["chunk"] % 32 == 0
        else:                                   # virtual rows per item: whole steps; two stages fit the CU's 160 KiB of LDS
            assert jobs[-1]["tm"] == 0 and jobs[-1]["chunk"] % jobs[-1]["RPS"] == 0 and jobs[-1]["TC"] % 4 == 0
            assert jobs[-1]["stage"] <= 64512 and jobs[-1]["NB"] * jobs[-1]["CB"] in (1, 2, 4)
    assert jobs[njobs - 1]["zz0"] == jobs[njobs - 2]["splits"] and jobs[njobs - 2]["zz0"] == 0   # second half's slots follow the first's
    # 128 -> 128 on 40 x 40: 2 x 2 blocks per workgroup, 2 x 2 tiles
    assert (jobs[njobs]["NB"], jobs[njobs]["CB"], jobs[njobs]["gx"], jobs[njobs]["gy"]) == (2, 2, 2, 2)
    it = (ctypes.c_int64 * 3)()
    seen = set()
    for cls, n in enumerate(n_items):
        assert n % 8 == 0
        where = {}
        for i in range(n):
            _lib.check(lib.ayolo_wgrad_group_item(host, cls, i, it), "item")
            if it[0] < 0:
                continue
            job, tile, zz = int(it[0]), int(it[1]), int(it[2])
            assert jobs[job]["tm"] == (32 << cls if cls < 3 else 0) and tile < jobs[job]["gx"] * jobs[job]["gy"] and zz < jobs[job]["splits"]
            assert (job, tile, zz) not in seen
            seen.add((job, tile, zz))
            where.setdefault((job, zz), []).append(i)
        for (job, zz), idxs in where.items():
            assert len({i % 8 for i in idxs}) == 1                          # one XCD
            assert [i // 8 for i in idxs] == list(range(idxs[0] // 8, idxs[0] // 8 + len(idxs)))   # back to back in its queue
    assert len(seen) == sum(j["gx"] * j["gy"] * j["splits"] for j in jobs)
    # reduction blocks: a layer's S partials are split between tpc threads per 16-byte column when S is large (tpc doubles while
    # S > 32 * tpc, up to 64), and a block then covers 1024 / tpc elements instead of 2048
    def red_blocks(nk, S):
        tpc = 1
        while tpc < 64 and S > 32 * tpc:
            tpc *= 2
        chunk = 1024 // tpc if tpc > 1 else 2048
        return -(-nk // chunk)
    # (jobs: the three k_wgrad layers 1, 2, 3, the two halves of layer 4, then the k_wgrad3 layer 0)
    per_shape = {1: jobs[0]["splits"], 2: jobs[1]["splits"], 3: jobs[2]["splits"], 4: jobs[3]["splits"] + jobs[4]["splits"], 0: jobs[5]["splits"]}
    layer_S = [per_shape[k] for k in range(len(shapes))]
    assert n_red == sum(red_blocks(Co * kk * kk * Ci, S) for (_, _, _, Ci, Co, kk, _), S in zip(shapes, layer_S))
    assert max(layer_S) > 32                                                    # the case the split reduction exists for


def test_model_ema_follows_reassigned_tensors_cpu():
    """ModelEMA finds the tensors of a state-dict key through their owner module every step (no state_dict() per step):
    in-place updates, `.data` swaps and re-assigned buffers (model.double().float()) must all be followed; arithmetic =
    the reference's two in-place ops (scripts/utils/torch_utils.py:405-416)."""
    import copy
    import math
    import os
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.trainer import ModelEMA
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.manual_seed(5)
    m = YOLOModel(os.path.join(root, "ayolov2_amd", "configs", "yolov5n.yaml"), verbose=False)
    ema = ModelEMA(m)
    ref = copy.deepcopy(m).eval()
    with torch.no_grad():
        for step in range(3):
            for p in m.parameters():
                p.add_(0.01 * torch.randn_like(p))
            for b in m.buffers():
                if b.dtype.is_floating_point:
                    b.add_(0.01)
            if step == 1:
                m = m.double().float()
            ema.update(m)
            d = 0.9999 * (1 - math.exp(-(step + 1) / 2000))
            msd = m.state_dict()
            for k, v in ref.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= d
                    v += (1 - d) * msd[k].detach()
    for (k, a), b in zip(ema.ema.state_dict().items(), ref.state_dict().values()):
        assert torch.equal(a, b), k


def test_tucker_block_merge_algebra_and_launch_form_cpu():
    """infer_plan._MergedConv multiplies the linear convs of a Tucker block out (decomposition.py:363-424 puts nothing between them):
    first factor into the core, core into the last factor, all three -- each must reproduce the Sequential on the CPU; and
    the per-block cost model (_tucker_form) must keep the factors apart only where that is cheaper: a stride-2 block never keeps
    the rank-r1 intermediate at input resolution, an HBM-bound large-map block with ranks C/2 goes back to the dense conv, a
    small-map block with ranks C/4 keeps a factorised form."""
    import types
    import torch.nn as nn
    import torch.nn.functional as F
    from ayolov2_amd import infer_plan as IP
    torch.manual_seed(0)
    a, b, c = nn.Conv2d(16, 4, 1, bias=False), nn.Conv2d(4, 6, 3, 2, 1, bias=False), nn.Conv2d(6, 20, 1, bias=True)
    x = torch.randn(2, 16, 9, 11)
    with torch.no_grad():
        ref = c(b(a(x)))
        m3 = IP._MergedConv([a, b, c])
        assert float((F.conv2d(x, m3.weight, m3.bias, m3.stride, m3.padding) - ref).abs().max()) < 1e-5
        m_first = IP._MergedConv([a, b])
        assert float((c(F.conv2d(x, m_first.weight, None, m_first.stride, m_first.padding)) - ref).abs().max()) < 1e-5
        m_last = IP._MergedConv([b, c])
        assert float((F.conv2d(a(x), m_last.weight, m_last.bias, m_last.stride, m_last.padding) - ref).abs().max()) < 1e-5
        # a factor update reaches the merged weight on recompute (the plan calls it when a source's version changes)
        a.weight.mul_(-2.0)
        m3.recompute()
        assert float((F.conv2d(x, m3.weight, m3.bias, m3.stride, m3.padding) - c(b(a(x)))).abs().max()) < 1e-5
    with pytest.raises(IP.PlanUnsupported):
        IP._MergedConv([nn.Conv2d(4, 4, 1, bias=True), nn.Conv2d(4, 4, 3, 1, 1, bias=False)])     # inner bias does not commute with padding

    def form(cin, r1, r2, cout, k, s, H, W, B=128, image=False):
        plan = types.SimpleNamespace(B=B, dt=torch.float16, H=H, W=W, _ce=lambda: 8)
        convs = [nn.Conv2d(cin, r1, 1, bias=False), nn.Conv2d(r1, r2, k, s, k // 2, bias=False), nn.Conv2d(r2, cout, 1, bias=True)]
        xt = None if image else torch.empty(B, cin, H, W, device="meta")
        return IP.InferPlan._tucker_form(plan, convs, xt, image)

    assert form(64, 16, 32, 128, 3, 2, 160, 160) in ("first", "dense")            # stride 2: no r1 intermediate at input resolution
    assert form(32, 16, 16, 32, 3, 1, 160, 160) == "dense"                         # large map, ranks C/2: HBM-bound either way
    assert form(256, 64, 64, 256, 3, 1, 20, 20) in ("first", "last", "factors")     # small map, ranks C/4: the factors pay
    assert form(3, 2, 8, 32, 6, 2, 640, 640, image=True) in ("first", "dense")      # the stem


def test_weight_gradient_group_count_follows_the_work():
    """The number of grouped weight-gradient launches is chosen from the layers' total work (32-pixel steps x dw tiles): four
    groups (+ the finer tail) for the benchmark's YOLOv5s at batch 64, eight for YOLOv5l at batch 32 (cfg 3), never fewer than
    four for a toy shape; AYOLO_WGRAD_GROUPS is not set in the test environment."""
    import torch
    from collections import Counter
    from ayolov2_amd import YOLOModel
    from ayolov2_amd import plan as P
    if P.WGRAD_GROUPS > 0:
        pytest.skip("AYOLO_WGRAD_GROUPS pins the count")
    want = {("yolov5s", 64, 640): (4, 7), ("yolov5l", 32, 640): (8, 11), ("yolov5s", 2, 64): (4, 7)}
    for (name, batch, size), (lo, hi) in want.items():
        m = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", f"{name}.yaml")).train()
        pl = P.TrainPlan(m, (batch, 3, size, size), torch.float16, torch.device("cpu"))
        n = Counter(o.kind & 0xff for o in pl.bwd)[P.OP_WGRAD_GROUP]
        assert lo <= n <= hi, (name, batch, n)
        assert sum(v[2] for v in pl.wgroup_costs.values()) >= len({j["off"] for j in pl._wjobs})


def test_kernarg_touch_destinations_survive_until_the_wait():
    """ScalarTouch (csrc/common.h) prefetches a kernel's argument block with inline-asm `s_load_dword` into SGPRs the compiler
    believes are defined at once; tools/check_kernarg_touch.py compiles a source to gfx950 assembly and verifies that nothing
    redefines those registers before the `s_waitcnt lgkmcnt(0)` of done().  Here on elementwise.hip (10 s; conv.hip
    takes 90 s and is checked by tools/final_round.sh: profiles/r06_kernarg_touch_check.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_kernarg_touch.py"),
                          os.path.join(root, "ayolov2_amd", "csrc", "elementwise.hip")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " 0 hazards" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert int(out.stdout.split(" touch sequences")[0].split()[-1]) >= 20      # the parser still finds the inline asm
