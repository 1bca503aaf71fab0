"""Glue rows of SURVEY.md 8a/8f on the device: coordinate helpers (general.py) on CUDA tensors against the reference-generated
golden G3, test-time augmentation composed with the HIP eval forward, and the COCO result writer (8f.4)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_general_helpers_on_device(golden_dir):
    """xywh2xyxy / clip_coords / scale_coords (scripts/utils/general.py:203-230, 297-358) on CUDA tensors: the same float32
    operations as on the CPU against the reference-generated golden values (exact, except where torch's device kernel
    replaces a division by a reciprocal multiply)."""
    from ayolov2_amd import general
    g = np.load(os.path.join(golden_dir, "g3_general.npz"))
    t = lambda a: torch.from_numpy(np.array(a, copy=True)).cuda()
    np.testing.assert_array_equal(general.xywh2xyxy(t(g["x"])).cpu().numpy(), g["xywh2xyxy"])
    np.testing.assert_array_equal(general.clip_coords(t(g["xyxy"]), (640, 480)).cpu().numpy(), g["clip"])
    # scale_coords divides by the gain: torch's CUDA kernel multiplies by the reciprocal of a scalar divisor, one ulp away
    # from the CPU's true division that produced the golden values
    np.testing.assert_allclose(general.scale_coords((640, 640), t(g["xyxy"]), (480, 600)).cpu().numpy(), g["scale_a"], rtol=2.5e-7)
    np.testing.assert_allclose(
        general.scale_coords((640, 640), t(g["xyxy"]), (720, 1280), ratio_pad=((0.5, 0.5), (0.0, 140.0))).cpu().numpy(), g["scale_b"],
        rtol=2.5e-7)


def test_tta_through_the_hip_forward():
    """inference_with_tta (scripts/utils/tta_utils.py:62-86; glue pinned on the CPU by golden G9) around the HIP eval
    forward: three forwards at scales 1 / 0.83 / 0.67 with a left-right flip, each on its own cached inference plan, against
    the same procedure around the CPU oracle network.  fp32 mode: 1e-4 of the logit-derived values, 2e-3 px on coordinates."""
    from ayolov2_amd import YOLOModel, tta
    from oracle.model_ref import RefYOLO
    torch.manual_seed(31)
    cfg = os.path.join(ROOT, "ayolov2_amd", "configs", "yolov5n.yaml")
    m = YOLOModel(cfg)
    r = RefYOLO(cfg)
    r.load_state_dict(m.state_dict())
    r.stride = torch.tensor([8.0, 16.0, 32.0])
    m, r = m.cuda().eval(), r.eval()
    x = torch.rand(2, 3, 128, 192)
    s, f = [1, 0.83, 0.67], [None, 3, None]
    with torch.no_grad():
        yg, _ = tta.inference_with_tta(m, x.cuda(), s, f)
        yr, _ = tta.inference_with_tta(r, x, s, f)
        # a second call reuses the three cached plans (static buffers) and must give the same answer
        yg2, _ = tta.inference_with_tta(m, x.cuda(), s, f)
    assert sum(1 for k, v in m._plans.items() if v is not False and k[0] == "eval") == 3
    assert yg.shape == yr.shape
    np.testing.assert_allclose(yg.cpu().numpy(), yr.numpy(), rtol=1e-4, atol=2e-3)
    assert torch.equal(yg, yg2)
    # the decode-store route (ayolo_head_decode_aug: de-scale / de-flip / tail window inside the decode kernels, one merged
    # tensor) against the tensor-op route around the SAME HIP forwards, incl. the up-down flip; CPU arithmetic of the
    # tensor-op route (true division) so that the comparison is exact
    s2, f2 = [1, 0.83, 0.67, 0.83], [None, 3, 2, 2]
    with torch.no_grad():
        fused, _ = tta.inference_with_tta(m, x.cuda(), s2, f2)

        class HostSide(torch.nn.Module):                  # same HIP forwards, predictions post-processed on the host
            def __init__(self):
                super().__init__()
                self.model, self.stride = m.model, m.stride

            def forward(self, xi):
                return m(xi)[0].cpu(), None

        plain, _ = tta.inference_with_tta(HostSide(), x.cuda(), s2, f2)      # same (GPU-resized) inputs as the fused route
    assert fused.shape == plain.shape
    np.testing.assert_array_equal(fused.cpu().numpy(), plain.numpy())


def test_result_writer_rows(tmp_path):
    """ResultWriter (multi_queue.py:204-305): un-letterbox + clip + [x, y, w, h] + COCO category id for a whole batch in one
    launch, against the numpy restatement; images seen twice are ignored, None / empty outputs give no objects, shapes=None
    leaves the boxes untouched."""
    from ayolov2_amd.result_writer import ResultWriter
    g = torch.Generator().manual_seed(5)
    img_size = (640, 640)
    names = ["000000000139.jpg", "17.jpg", "000000000285.jpg", "42.jpg"]
    shapes = [((426, 640), ((1.0, 1.0), (0.0, 107.0))), ((480, 600), ((1.0, 1.0), (20.0, 0.0))), ((640, 640), ((1.0, 1.0), (0.0, 0.0))),
              ((500, 375), ((1.0, 1.0), (0.0, 0.0)))]
    outs = []
    for n in (37, 0, 5, 12):
        if n == 0:
            outs.append(None)
            continue
        xy = torch.rand(n, 2, generator=g) * 700 - 30                 # some boxes stick out of the image: clipping
        wh = torch.rand(n, 2, generator=g) * 200 + 1
        outs.append(torch.cat((xy, xy + wh, torch.rand(n, 1, generator=g), torch.randint(0, 80, (n, 1), generator=g).float()), 1))
    w = ResultWriter(str(tmp_path / "r.json"))
    w.add_outputs(names, [o.cuda() if o is not None else None for o in outs], img_size, shapes)
    w.add_outputs(names[:2], [outs[0].cuda() * 2, None], img_size, shapes[:2])      # already seen: ignored
    want = ops_ref.coco_rows(names, [o.numpy() if o is not None else None for o in outs], img_size, shapes)
    got = w.objects()
    assert len(got) == len(want) == 54
    for a, b in zip(got, want):
        assert a["image_id"] == b["image_id"] and a["category_id"] == b["category_id"]
        assert a["bbox"] == b["bbox"] and a["score"] == b["score"], (a, b)
    w.close()
    assert json.load(open(tmp_path / "r.json")) == got
    w2 = ResultWriter(str(tmp_path / "r2.json"))
    w2.add_outputs(names[:1], [outs[0].cuda()], img_size, None)
    want2 = ops_ref.coco_rows(names[:1], [outs[0].numpy()], img_size, None)
    assert [o["bbox"] for o in w2.objects()] == [o["bbox"] for o in want2]
