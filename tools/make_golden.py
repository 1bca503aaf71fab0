#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON in this container.

Runs only where /root/reference exists (the build container).  Nothing from the reference is copied
into the repo: this script imports the reference's L3 functions (scripts/utils/{general,metrics,nms}.py,
scripts/loss/losses.py, scripts/tensor_decomposition/decomposition.py) with `sys.modules` stubs for the
third-party packages that are not installed here, feeds them seeded synthetic inputs, and stores
inputs + outputs as small .npz fixtures (data only).

Third-party leaves are bound to the oracle's restatements so the fixtures pin the reference's WRAPPER
logic (filtering, offsets, caps, ordering):
  torchvision.ops.nms / ops.boxes.batched_nms -> oracle.ops_ref.tv_nms / tv_batched_nms
  tensorly.base.unfold / decomposition.partial_tucker -> oracle.tucker_ref
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import ops_ref, tucker_ref  # noqa: E402


def _install_stubs():
    for name in ["cv2", "kindle", "kindle.modules", "kindle.model", "kindle.utils", "kindle.utils.torch_utils",
                 "wandb", "seaborn", "albumentations", "p_tqdm", "orjson", "pycocotools", "pycocotools.coco",
                 "pycocotools.cocoeval", "optuna", "onnx", "onnxsim", "tensorly", "tensorly.base",
                 "tensorly.decomposition", "torchvision", "torchvision.ops", "torchvision.ops.boxes",
                 "torchvision.transforms", "matplotlib.pyplot"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock(name=name)
    tv = sys.modules["torchvision"]
    ops = sys.modules["torchvision.ops"]
    boxes = sys.modules["torchvision.ops.boxes"]
    tv.ops = ops
    ops.boxes = boxes

    def nms(b, s, thr):
        return torch.from_numpy(ops_ref.tv_nms(b.detach().numpy(), s.detach().numpy(), thr))

    def batched_nms(b, s, idx, thr):
        return torch.from_numpy(ops_ref.tv_batched_nms(b.detach().numpy(), s.detach().numpy(),
                                                       idx.detach().numpy(), thr))

    ops.nms = nms
    boxes.batched_nms = batched_nms
    ops.batched_nms = batched_nms

    tl = sys.modules["tensorly"]
    tl.base = sys.modules["tensorly.base"]
    tl.decomposition = sys.modules["tensorly.decomposition"]
    tl.base.unfold = lambda t, m: torch.from_numpy(tucker_ref.unfold(t.detach().numpy(), m))

    def partial_tucker(t, modes, rank, init="svd"):
        core, factors = tucker_ref.partial_tucker(t.detach().numpy(), modes, rank)
        return torch.from_numpy(core), [torch.from_numpy(f) for f in factors]

    tl.decomposition.partial_tucker = partial_tucker


def synth_pred(B, N, nc, img, mu_obj, seed):
    """SURVEY.md 8d 'NMS synthetic' distribution (sized so the reference never trips its 10 s limit)."""
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, N, 2, generator=g) * img
    wh = torch.rand(B, N, 2, generator=g) ** 3 * img / 2 + 2
    obj = torch.sigmoid(torch.randn(B, N, 1, generator=g) * 2 + mu_obj)
    cls = torch.sigmoid(torch.randn(B, N, nc, generator=g) * 2 - 4)
    return torch.cat((xy, wh, obj, cls), 2).float()


def g3b_xyxy2xywh():
    """G3b: xyxy2xywh (general.py:252-294) -- numpy boxes the way the label export calls it: normalised coordinates that
    overhang the unit square (the validity shrink), pixel coordinates with an image size, clip_eps, check_validity off."""
    from scripts.utils import general as rg
    rng = np.random.default_rng(33)
    lo = rng.uniform(-0.15, 0.9, (64, 2))
    box = np.concatenate([lo, lo + rng.uniform(0.0, 0.4, (64, 2))], 1)
    box[:4] = [[0.0, 0.0, 1.0, 1.0], [0.2, 0.2, 0.2, 0.2], [-0.2, 0.5, 0.1, 1.3], [0.95, -0.1, 1.2, 0.05]]
    px = box * np.array([640.0, 480.0, 640.0, 480.0])
    out = dict(box=box, px=px,
               default=rg.xyxy2xywh(box.copy()),
               no_check=rg.xyxy2xywh(box.copy(), check_validity=False),
               sized=rg.xyxy2xywh(px.copy(), wh=(640.0, 480.0)),
               sized_clip=rg.xyxy2xywh(px.copy(), wh=(640.0, 480.0), clip_eps=1e-3),
               f32=rg.xyxy2xywh(box.astype(np.float32)))
    np.savez_compressed(os.path.join(OUT, "g3b_xyxy2xywh.npz"), **out)


def g8_validator():
    """G8: YoloValidator.process_batch (train_utils.py:294-333) on seeded detections / labels of several images and
    ap_per_class (metrics.py:476-548) on the stacked statistics -- the reference's own functions, run here."""
    from scripts.utils import metrics as rm
    from scripts.utils import train_utils as rt
    rng = np.random.default_rng(8)
    iouv = torch.linspace(0.5, 0.95, 10)

    class FakeSelf:
        pass

    fs = FakeSelf()
    fs.iouv = iouv
    g8 = {"iouv": iouv.numpy()}
    stats = []
    n_img = 6
    for i in range(n_img):
        m = int(rng.integers(0, 9)) if i != 2 else 0            # image 2: no labels
        n = int(rng.integers(5, 40)) if i != 4 else 0           # image 4: no detections
        lab_xy = rng.uniform(20, 500, (m, 2)).astype(np.float32)
        lab = np.concatenate([rng.integers(0, 5, (m, 1)).astype(np.float32), lab_xy,
                              lab_xy + rng.uniform(20, 140, (m, 2)).astype(np.float32)], 1)
        det = np.zeros((n, 6), np.float32)
        for k in range(n):
            if m and rng.uniform() < 0.7:                        # jittered copy of a label (several per label)
                j = int(rng.integers(0, m))
                det[k, :4] = lab[j, 1:] + rng.normal(0, 6, 4)
                det[k, 5] = lab[j, 0] if rng.uniform() < 0.85 else float(rng.integers(0, 5))
            else:
                xy = rng.uniform(0, 520, 2)
                det[k, :4] = [xy[0], xy[1], xy[0] + rng.uniform(10, 150), xy[1] + rng.uniform(10, 150)]
                det[k, 5] = float(rng.integers(0, 5))
            det[k, 4] = rng.uniform(0.01, 1.0)
        det = det[np.argsort(-det[:, 4])].astype(np.float32)     # NMS output order
        g8[f"det{i}"], g8[f"lab{i}"] = det, lab
        if n == 0:
            continue
        if m:
            correct = rt.YoloValidator.process_batch(fs, torch.from_numpy(det), torch.from_numpy(lab)).numpy()
        else:
            correct = np.zeros((n, 10), bool)
        g8[f"correct{i}"] = correct
        stats.append((correct, det[:, 4], det[:, 5], lab[:, 0]))
    g8["n_img"] = n_img
    tp, conf, pcls, tcls = [np.concatenate(x, 0) for x in zip(*stats)]
    tcls = np.concatenate([g8[f"lab{i}"][:, 0] for i in range(n_img)])   # labels of images without detections count too
    p, r, ap, f1, cls = rm.ap_per_class(tp, conf, pcls, tcls, plot=False)
    g8.update(ap_p=p, ap_r=r, ap=ap, ap_f1=f1, ap_cls=cls, tp=tp, conf=conf, pcls=pcls, tcls=tcls)
    np.savez_compressed(os.path.join(OUT, "g8_validator.npz"), **g8)


def g7b_net(weights=None):
    """Two Conv blocks (`.conv` children, as kindle's Conv exposes them) with planted low-rank 3x3 weights + noise."""
    class Blk(torch.nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False)

        def forward(self, x):
            return self.conv(x)

    net = torch.nn.Sequential(Blk(16, 24), Blk(24, 16))
    if weights is None:
        rs = np.random.default_rng(17)
        weights = []
        for (co, ci, ro, ri) in ((24, 16, 6, 5), (16, 24, 5, 6)):
            core = rs.standard_normal((ro, ri, 3, 3))
            w = np.einsum("abhw,oa,ib->oihw", core, rs.standard_normal((co, ro)), rs.standard_normal((ci, ri))) / 12
            weights.append((w + 0.01 * rs.standard_normal(w.shape)).astype(np.float32))
    for blk, w in zip(net, weights):
        blk.conv.weight.data = torch.from_numpy(np.ascontiguousarray(w))
    return net, weights


def g7b_prune_bisection(rd):
    net, weights = g7b_net()
    torch.manual_seed(123)                       # decompose_model draws its probe inputs from the global generator
    rd.decompose_model(net, loss_thr=0.1, prune_step=0.01)
    x = torch.rand((2, 16, 12, 12), generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        y = net(x)
    out = {"w0": weights[0], "w1": weights[1], "y": y.numpy().astype(np.float32), "x_seed": 9, "probe_seed": 123}
    for i, blk in enumerate(net):
        seq = blk.conv
        assert isinstance(seq, torch.nn.Sequential), "the fixture must exercise a successful decomposition"
        out[f"shapes{i}"] = np.array([list(m.weight.shape) for m in seq])
        # the pruned-then-decomposed kernel the bisection settled on, as ONE dense 3x3 kernel (sign / rotation free)
        first, core, last = (m.weight.detach().double() for m in seq)
        out[f"dense{i}"] = torch.einsum("oa,abhw,bi->oihw", last[:, :, 0, 0], core, first[:, :, 0, 0]).float().numpy()
    return out


def g9_tta():
    """G9: inference_with_tta / scale_img / descale_pred / clip_augmented (tta_utils.py:15-86, torch_utils.py:305-331)
    around a deterministic stand-in model (a YOLO-shaped function of the input), the reference's own code run here."""
    from scripts.utils import tta_utils as rt
    from scripts.utils import torch_utils as rtu

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
                pooled = torch.nn.functional.adaptive_avg_pool2d(x, (ny, nx))          # (B,3,ny,nx)
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

    m = Fake()
    x = torch.rand(1, 3, 96, 128, generator=torch.Generator().manual_seed(9))
    s_, f_ = [1, 0.83, 0.67], [None, 3, None]
    y, _ = rt.inference_with_tta(m, x, s_, f_)
    # ONE augmentation: clip_augmented cuts the tail of y[0], then computes the head cut of y[-1] -- the same, already shortened,
    # entry -- from its NEW row count (tta_utils.py:54-58)
    y1, _ = rt.inference_with_tta(m, x, [0.83], [3])
    g9 = dict(x=x.numpy(), scales=np.array(s_), flips=np.array([0, 3, 0]), y=y.numpy(), y_single=y1.numpy(),
              scaled_083=rtu.scale_img(x, 0.83, gs=32).numpy(), scaled_same=rtu.scale_img(x, 0.5, same_shape=True, gs=32).numpy())
    np.savez_compressed(os.path.join(OUT, "g9_tta.npz"), **g9)


def g6_loss(only_b=False):
    """G6: ComputeLoss value / items / build_targets / gradient sums; G6b: ELEMENTWISE gradients of the same run -- every logit of
    every matched cell, and the objectness logit on a 3 x 3 sub-lattice of every level (the dense part of the gradient)."""
    # ---- G6 ComputeLoss on a fake model (hyp after set_model_params scaling, model_manager.py:252-258)
    from scripts.loss import losses as rl
    import yaml
    hyp = yaml.safe_load(open(os.path.join(REF, "res/configs/cfg/train_config.yaml")))["hyper_params"]
    nl, nc, imgsz = 3, 80, 640
    hyp["box"] *= 3.0 / nl
    hyp["cls"] *= nc / 80.0 * 3.0 / nl
    hyp["obj"] *= (imgsz / 640) ** 2 * 3.0 / nl
    mcfg = yaml.safe_load(open(os.path.join(REF, "res/configs/model/yolov5s.yaml")))
    strides = torch.tensor([8., 16., 32.])
    anchors = torch.tensor(mcfg["anchors"]).float().view(3, 3, 2) / strides.view(-1, 1, 1)

    class Head(torch.nn.Module):
        pass

    head = Head()
    head.nl, head.na, head.nc, head.anchors, head.stride = 3, 3, 80, anchors, strides

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.model = torch.nn.ModuleList([torch.nn.Identity(), head])
            self.hyp = hyp

    rl.is_parallel = lambda m: False
    fake = Fake()
    g = torch.Generator().manual_seed(3)
    preds = [torch.randn(2, 3, s, s, 85, generator=g).requires_grad_(True) for s in (80, 40, 20)]
    # targets from the reference's own test labels (class x y w h normalised)
    lab_dir = os.path.join(REF, "tests/res/datasets/coco/labels/train2017")
    files = sorted(os.listdir(lab_dir))[:2]
    tg = []
    for bi, f in enumerate(files):
        arr = np.loadtxt(os.path.join(lab_dir, f), ndmin=2).astype(np.float32)[:, :5]
        tg.append(np.concatenate([np.full((arr.shape[0], 1), bi, np.float32), arr], 1))
    targets = torch.from_numpy(np.concatenate(tg, 0))
    # torch-1.9 -> 2.x compat shim (the reference pins torch 1.9.1, environment.yml:27): losses.py:385 clamps a
    # LongTensor in place with 0-dim FLOAT tensor bounds, which torch 1.9 accepted (bound cast to the self
    # dtype) and torch 2.x rejects.  Reproduce the 1.9 behaviour without touching the reference.
    _orig_clamp_ = torch.Tensor.clamp_

    def _clamp_compat(self, min=None, max=None):
        if not self.is_floating_point():
            min = int(min) if torch.is_tensor(min) else min
            max = int(max) if torch.is_tensor(max) else max
        return _orig_clamp_(self, min, max)

    torch.Tensor.clamp_ = _clamp_compat
    cl = rl.ComputeLoss(fake)
    loss, items = cl(preds, targets)
    loss.backward()
    tcls, tbox, indices, anch = cl.build_targets(preds, targets)
    g6 = dict(targets=targets.numpy(), loss=loss.detach().numpy(), items=items.numpy(),
              hyp_box=hyp["box"], hyp_cls=hyp["cls"], hyp_obj=hyp["obj"], anchors=anchors.numpy())
    for i in range(3):
        g6[f"pred{i}"] = preds[i].detach().numpy().astype(np.float32)
        g6[f"grad{i}_sum"] = preds[i].grad.sum((2, 3)).numpy()
        nz = preds[i].grad.abs().sum(-1) > 1e-3 * preds[i].grad.abs().max()
        g6[f"grad{i}_abs_total"] = preds[i].grad.abs().sum().numpy()
        g6[f"tcls{i}"] = tcls[i].numpy()
        g6[f"tbox{i}"] = tbox[i].numpy()
        g6[f"idx{i}"] = torch.stack(indices[i]).numpy()
        g6[f"anch{i}"] = anch[i].numpy()
    # predictions are regenerated from the seed in the test (too large to store): keep only a checksum
    for i in range(3):
        del g6[f"pred{i}"]
    g6["pred_seed"] = 3
    g6b = {}
    for i in range(3):
        b, a, gj, gi = indices[i]
        g6b[f"rows{i}"] = preds[i].grad[b, a, gj, gi].numpy()
        g6b[f"obj{i}"] = preds[i].grad[:, :, ::3, ::3, 4].numpy()
    np.savez_compressed(os.path.join(OUT, "g6b_loss_grads.npz"), **g6b)
    if not only_b:
        np.savez_compressed(os.path.join(OUT, "g6_loss.npz"), **g6)


def main():
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    sys.path.insert(0, REF)
    if len(sys.argv) > 1 and sys.argv[1] == "g3b":
        g3b_xyxy2xywh()
        print("g3b written")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        g8_validator()
        print("g8 written")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        g9_tta()
        print("g9 written")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g6b":
        g6_loss(only_b=True)
        print("g6b written")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g7b":
        from scripts.tensor_decomposition import decomposition as rd
        np.savez_compressed(os.path.join(OUT, "g7b_decompose_model.npz"), **g7b_prune_bisection(rd))
        print("g7b written")
        return
    from scripts.utils import general as rg
    from scripts.utils import metrics as rm
    from scripts.utils import nms as rn

    rng = np.random.default_rng(0)

    # ---- G1 box_iou (incl. degenerate / identical / zero-area boxes)
    a = rng.uniform(0, 600, (64, 2)).astype(np.float32)
    a = np.concatenate([a, a + rng.uniform(0, 200, (64, 2)).astype(np.float32)], 1)
    b = rng.uniform(0, 600, (48, 2)).astype(np.float32)
    b = np.concatenate([b, b + rng.uniform(0, 200, (48, 2)).astype(np.float32)], 1)
    b[:8] = a[:8]                       # identical boxes
    a[60] = [10, 10, 10, 50]            # zero-area
    b[40] = [10, 10, 10, 50]            # zero-area vs zero-area -> 0/0 = nan
    b[41] = [300, 300, 250, 250]        # inverted
    np.savez_compressed(os.path.join(OUT, "g1_box_iou.npz"), box1=a, box2=b,
                        iou=rm.box_iou(torch.from_numpy(a), torch.from_numpy(b)).numpy())

    # ---- G2 bbox_iou CIoU values and grads
    p = torch.from_numpy(rng.uniform(0.1, 8, (200, 4)).astype(np.float32)).requires_grad_(True)
    t = torch.from_numpy(rng.uniform(0.1, 8, (200, 4)).astype(np.float32))
    out = {}
    for tag, kw in {"iou": {}, "giou": {"g_iou": True}, "diou": {"d_iou": True}, "ciou": {"c_iou": True}}.items():
        for fmt in (False, True):
            pp = p.detach().clone().requires_grad_(True)
            if fmt:  # make valid xyxy
                q = torch.cat((pp[:, :2], pp[:, :2] + pp[:, 2:]), 1)
                tt = torch.cat((t[:, :2], t[:, :2] + t[:, 2:]), 1)
            else:
                q, tt = pp, t
            v = rm.bbox_iou(q.T, tt, x1y1x2y2=fmt, **kw)
            v.sum().backward()
            out[f"{tag}_{int(fmt)}"] = v.detach().numpy()
            out[f"{tag}_{int(fmt)}_grad"] = pp.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g2_bbox_iou.npz"), pred=p.detach().numpy(), target=t.numpy(), **out)

    # ---- G3 general.py helpers
    x = rng.uniform(0, 640, (100, 4)).astype(np.float32)
    xyxy = np.concatenate([x[:, :2], x[:, :2] + x[:, 2:]], 1).astype(np.float32) - 100
    g3 = dict(x=x, xywh2xyxy=rg.xywh2xyxy(torch.from_numpy(x)).numpy(),
              xywh2xyxy_r=rg.xywh2xyxy(torch.from_numpy(x / 640), ratio=(0.5, 0.75), wh=(640, 480), pad=(3.0, 7.0)).numpy(),
              xyxy=xyxy,
              clip=rg.clip_coords(torch.from_numpy(xyxy.copy()), (640, 480)).numpy(),
              scale_a=rg.scale_coords((640, 640), torch.from_numpy(xyxy.copy()), (480, 600)).numpy(),
              scale_b=rg.scale_coords((640, 640), torch.from_numpy(xyxy.copy()), (720, 1280),
                                      ratio_pad=((0.5, 0.5), (0.0, 140.0))).numpy())
    np.savez_compressed(os.path.join(OUT, "g3_general.npz"), **g3)

    # ---- G4 non_max_suppression: 5 nms types x agnostic x multi_label on (2, 3000, 85)
    pred = synth_pred(2, 1000, 80, 640, -7.0, seed=1)
    # float32 and tie-free: equal objectness/conf values would expose torch's unstable argsort tie order
    assert len(np.unique(pred[..., 4].numpy())) == pred[..., 4].numel()
    g4 = {"pred": pred.numpy()}
    for nms_type in ["nms", "batched_nms", "fast_nms", "matrix_nms", "merge_nms"]:
        for agn in (False, True):
            for ml in (False, True):
                res = rm.non_max_suppression(pred.clone(), conf_thres=0.001, iou_thres=0.65, multi_label=ml,
                                             agnostic=agn, nms_type=nms_type)
                for bi, r in enumerate(res):
                    g4[f"{nms_type}_a{int(agn)}_m{int(ml)}_{bi}"] = r.numpy()
    # a conf/iou variation + classes filter + hybrid labels
    res = rm.non_max_suppression(pred.clone(), conf_thres=0.25, iou_thres=0.45, classes=[0, 3, 17])
    for bi, r in enumerate(res):
        g4[f"cls_filter_{bi}"] = r.numpy()
    lab = [torch.tensor([[3, 100., 120., 40., 60.], [7, 300., 320., 80., 30.]]), torch.zeros((0, 5))]
    res = rm.non_max_suppression(pred.clone(), conf_thres=0.1, iou_thres=0.6, labels=lab, multi_label=True)
    g4["hybrid_labels_0"] = lab[0].numpy()
    for bi, r in enumerate(res):
        g4[f"hybrid_{bi}"] = r.numpy()
    np.savez_compressed(os.path.join(OUT, "g4_nms.npz"), **g4)

    # ---- G5 nms.py::batched_nms
    g5 = {}   # input = g4_nms.npz["pred"]
    for nms_type in ["nms", "batched_nms", "fast_nms", "matrix_nms", "merge_nms"]:
        for agn in (False, True):
            for nb in (500, 1000):
                res = rn.batched_nms(pred.clone(), conf_thres=0.001, iou_thres=0.65, nms_box=nb, agnostic=agn,
                                     nms_type=nms_type)
                for bi, r in enumerate(res):
                    g5[f"{nms_type}_a{int(agn)}_n{nb}_{bi}"] = r.numpy()
    np.savez_compressed(os.path.join(OUT, "g5_batched_nms.npz"), **g5)

    g6_loss()

    # ---- G7 EVBMF ranks + Tucker layer loss through the reference driver
    from scripts.tensor_decomposition import decomposition as rd
    g7 = {}
    for name, (L, M, r) in {"a": (64, 576, 12), "b": (128, 1152, 30), "c": (256, 2304, 40)}.items():
        rs = np.random.default_rng(7)
        Y = (rs.standard_normal((L, r)) @ rs.standard_normal((r, M)) / np.sqrt(r) + 0.05 * rs.standard_normal((L, M))).astype(np.float32)
        _, d, _, _ = rd.EVBMF(torch.from_numpy(Y))
        g7[f"Y_{name}_seed"] = 7
        g7[f"rank_{name}"] = d.shape[0]
        g7[f"shape_{name}"] = np.array([L, M, r])
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(64, 96, 3, padding=1, bias=False)
    rs = np.random.default_rng(11)
    core = rs.standard_normal((20, 16, 3, 3)).astype(np.float32)
    U0 = rs.standard_normal((96, 20)).astype(np.float32)
    U1 = rs.standard_normal((64, 16)).astype(np.float32)
    W = np.einsum("abhw,oa,ib->oihw", core, U0, U1) / 10 + 0.01 * rs.standard_normal((96, 64, 3, 3))
    conv.weight.data = torch.from_numpy(W.astype(np.float32))
    ranks = rd.estimate_ranks(conv)
    xin = torch.rand((64, 64, 3, 3), generator=torch.Generator().manual_seed(5))
    seq, loss = rd.decompose_layer_evaluation(conv, xin, conv(xin))
    g7["conv_w"] = conv.weight.data.numpy().astype(np.float32)
    g7["conv_ranks"] = np.array(ranks)
    g7["conv_loss"] = float(loss)
    g7["conv_shapes"] = np.array([list(m.weight.shape) for m in seq])
    np.savez_compressed(os.path.join(OUT, "g7_tucker.npz"), **g7)

    # ---- G7b: the reference's decompose_model() driver on its DEFAULT path (decompose_model.py:63-74: loss_thr 0.1,
    # prune_step 0.01 -> the L1-prune bisection of decomposition.py:296-323) over a seeded two-block net
    g7b = g7b_prune_bisection(rd)
    np.savez_compressed(os.path.join(OUT, "g7b_decompose_model.npz"), **g7b)

    g8_validator()
    g9_tta()
    g3b_xyxy2xywh()

    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
