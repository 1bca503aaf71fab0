"""Device-side validator matching (ayolo_match_detections) vs the reference-generated golden G8 and the oracle,
and the validator end to end against a host-side replay of the reference procedure."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_matching_vs_reference_golden(golden_dir):
    from ayolov2_amd.validator import match_batch, process_batch
    g = np.load(os.path.join(golden_dir, "g8_validator.npz"))
    iouv = torch.from_numpy(g["iouv"]).cuda()
    n = int(g["n_img"])
    dets = [torch.from_numpy(g[f"det{i}"]).cuda() for i in range(n)]
    labs = [torch.from_numpy(g[f"lab{i}"]).cuda().reshape(-1, 5) for i in range(n)]
    out = match_batch(dets, labs, iouv)                      # all images in one call (incl. empty label / detection sets)
    for i in range(n):
        want = g[f"correct{i}"] if f"correct{i}" in g.files else np.zeros((dets[i].shape[0], 10), bool)
        np.testing.assert_array_equal(out[i].cpu().numpy(), want)
        if dets[i].shape[0]:
            np.testing.assert_array_equal(process_batch(dets[i], labs[i], iouv).cpu().numpy(), want)


def test_matching_vs_oracle_random_large():
    """Crowded scenes (hundreds of detections per image, many per label), 16 images in one call."""
    from oracle import ops_ref
    from ayolov2_amd.validator import match_batch
    rng = np.random.default_rng(21)
    iouv = torch.linspace(0.5, 0.95, 10)
    dets, labs = [], []
    for i in range(16):
        m, n = int(rng.integers(1, 40)), int(rng.integers(50, 400))
        xy = rng.uniform(0, 560, (m, 2)).astype(np.float32)
        lab = np.concatenate([rng.integers(0, 4, (m, 1)).astype(np.float32), xy, xy + rng.uniform(15, 120, (m, 2)).astype(np.float32)], 1)
        j = rng.integers(0, m, n)
        det = np.concatenate([lab[j, 1:] + rng.normal(0, 5, (n, 4)).astype(np.float32),
                              rng.uniform(0.01, 1, (n, 1)).astype(np.float32), lab[j, 0:1]], 1).astype(np.float32)
        flip = rng.uniform(size=n) < 0.2
        det[flip, 5] = rng.integers(0, 4, int(flip.sum()))
        det = det[np.argsort(-det[:, 4])]
        dets.append(det); labs.append(lab)
    out = match_batch([torch.from_numpy(d).cuda() for d in dets], [torch.from_numpy(l).cuda() for l in labs], iouv.cuda())
    total = 0
    for d, l, o in zip(dets, labs, out):
        want = ops_ref.process_batch(d, l, iouv.numpy())
        np.testing.assert_array_equal(o.cpu().numpy(), want)
        total += int(want[:, 0].sum())
    assert total > 50


def test_validator_end_to_end_matches_host_replay():
    """YoloValidator (HIP model -> HIP NMS -> device matching -> AP) against the same detections pushed through the
    oracle's process_batch / ap_per_class on the host."""
    from oracle import ops_ref
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.validator import YoloValidator
    from ayolov2_amd.general import scale_coords, xywh2xyxy
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ayolov2_amd", "configs", "yolov5n.yaml")
    torch.manual_seed(2)
    m = YOLOModel(cfg).cuda().eval()
    with torch.no_grad():                                     # make the random head fire: raise objectness / a few classes
        for conv in m.model[-1].conv:
            b = conv.bias.view(3, -1)
            b[:, 4] += 6.0          # the YOLO bias init keeps objectness ~0.01 and classes ~0.007
            b[:, 5:9] += 5.5
    val = YoloValidator(m, torch.device("cuda"), {"conf_t": 0.25, "iou_t": 0.6})
    B = 3
    imgs = torch.rand(B, 3, 128, 160)
    targets = torch.tensor([[0, 0, 0.3, 0.3, 0.3, 0.4], [0, 2, 0.7, 0.6, 0.2, 0.3], [2, 1, 0.5, 0.5, 0.5, 0.5], [2, 3, 0.2, 0.7, 0.2, 0.2]])
    shapes = [((128, 160), ((1.0, 1.0), (0.0, 0.0)))] * B
    val.validation_step((imgs, targets, ["a", "b", "c"], shapes))
    res = val.compute_statistics()
    assert val.seen == B
    # host replay from the validator's own NMS output statistics: recompute `correct` with the oracle
    stats = val.statistics["stats"]
    assert len(stats) >= 2
    conf_all = np.concatenate([s[1] for s in stats])
    assert conf_all.size > 0 and np.all(np.diff(np.concatenate([s[1] for s in stats[:1]])) <= 1e-6)
    p, r, ap, f1, cls = ops_ref.ap_per_class(*[np.concatenate(x, 0) for x in zip(*stats)])
    np.testing.assert_allclose(res["map50"], ap[:, 0].mean(), rtol=1e-12)
    np.testing.assert_allclose(res["map"], ap.mean(1).mean(), rtol=1e-12)
    np.testing.assert_allclose(res["mp"], p.mean(), rtol=1e-12)
    # the reference's stage timers (train_utils.py:420-470): pre-process / inference / NMS seconds, event-timed on the stream
    dt = res["dt"]
    assert len(dt) == 3 and all(t > 0 for t in dt) and val.statistics["dt"] == dt
