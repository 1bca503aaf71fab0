"""Fixed-shape batched NMS (TensorRT BatchedNMS_TRT contract) vs the CPU restatement of the plugin.
Everything is compared bit for bit: the kept set is decided by IEEE single-precision arithmetic in the plugin's order."""
import numpy as np
import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu


def synth(B, N, nc, img, seed, xyxy):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, N, 2, generator=g) * img
    wh = torch.rand(B, N, 2, generator=g) ** 2 * img / 3 + 2
    obj = torch.sigmoid(torch.randn(B, N, 1, generator=g) * 2 - 1)
    cls = torch.sigmoid(torch.randn(B, N, nc, generator=g) * 2 - 2)
    box = torch.cat((xy - wh / 2, xy + wh / 2), 2) if xyxy else torch.cat((xy, wh), 2)
    return torch.cat((box, obj, cls), 2).float()


def oracle_from_pred(pred, xyxy, **kw):
    p = pred.numpy()
    boxes = p[..., :4] if xyxy else ops_ref.xywh2xyxy(p[..., :4].reshape(-1, 4)).reshape(p.shape[0], -1, 4)
    scores = p[..., 5:] * p[..., 4:5]
    return ops_ref.batched_nms_trt(boxes, scores, **kw)


def _same(got, want, what):
    names = ("num_detections", "nmsed_boxes", "nmsed_scores", "nmsed_classes")
    for g, w, n in zip(got, want, names):
        g = g.cpu().numpy()
        assert g.shape == w.shape and g.dtype == w.dtype, f"{what} {n}: {g.shape} {g.dtype} vs {w.shape} {w.dtype}"
        np.testing.assert_array_equal(g, w, err_msg=f"{what} {n}")


@pytest.mark.parametrize("xyxy", [True, False])
@pytest.mark.parametrize("B,N,nc,top_k,keep", [(2, 1500, 12, 96, 40), (3, 700, 5, 200, 100), (1, 64, 3, 8, 8)])
def test_fixed_nms_vs_oracle(B, N, nc, top_k, keep, xyxy):
    from ayolov2_amd.fixed_nms import BatchedNMS
    pred = synth(B, N, nc, 320, seed=B * 100 + N, xyxy=xyxy)
    kw = dict(top_k=top_k, keep_top_k=keep, score_threshold=0.05, iou_threshold=0.5)
    want = oracle_from_pred(pred, xyxy, **kw)
    nms = BatchedNMS(nc, capacity=B * N * nc, **kw)     # worst case: the plugin itself never drops a pair
    got = nms.from_prediction(pred.cuda(), box_xyxy=xyxy)
    _same(got, want, f"xyxy={xyxy}")
    assert not bool(nms.overflow)
    assert int(got[0].max()) <= keep and int(got[0].min()) >= 0


def test_fixed_nms_ties_duplicates_and_empty():
    """Equal scores (stable order: class, then box index), identical boxes, an image with nothing above the threshold,
    inverted boxes (bboxSize 0) and boxes far apart (the plugin's unit-box intersection of disjoint boxes)."""
    from ayolov2_amd.fixed_nms import BatchedNMS
    pred = synth(3, 256, 4, 100, seed=9, xyxy=True)
    pred[0, :, 4] = 0.5
    pred[0, :, 5:] = (torch.arange(256)[:, None] % 7 + 1).float() / 8        # many equal scores across boxes and classes
    pred[0, 10:20, :4] = pred[0, 10, :4]                                      # identical boxes
    pred[0, 30, :4] = torch.tensor([50., 50., 40., 45.])                      # inverted
    pred[1, :, 4] = 0                                                         # empty image
    pred[2, :128, :4] = torch.tensor([0., 0., 0.5, 0.5])                      # tiny boxes: (w+1)(h+1) dominates
    kw = dict(top_k=64, keep_top_k=32, score_threshold=0.1, iou_threshold=0.4)
    want = oracle_from_pred(pred, True, **kw)
    got = BatchedNMS(4, capacity=3 * 256 * 4, **kw).from_prediction(pred.cuda())
    _same(got, want, "ties")
    assert int(got[0][1]) == 0 and float(got[3][1].max()) == -1.0


def test_fixed_nms_plugin_inputs_and_convert():
    """The plugin's own (boxes (B,N,1,4), scores (B,N,nc)) call, and YoloValidator.convert_trt_out on its outputs."""
    from ayolov2_amd.fixed_nms import BatchedNMS, convert_trt_out
    pred = synth(2, 900, 6, 256, seed=4, xyxy=True)
    boxes, scores = pred[..., :4].reshape(2, 900, 1, 4), pred[..., 5:] * pred[..., 4:5]
    kw = dict(top_k=128, keep_top_k=50, score_threshold=0.02, iou_threshold=0.6)
    want = ops_ref.batched_nms_trt(boxes.reshape(2, 900, 4).numpy(), scores.numpy(), **kw)
    got = BatchedNMS(6, capacity=2 * 900 * 6, **kw)(boxes.cuda(), scores.cuda())
    _same(got, want, "plugin inputs")
    rows = convert_trt_out(*got)
    for b, r in enumerate(rows):
        n = int(want[0][b, 0])
        assert r.shape == (n, 6)
        np.testing.assert_array_equal(r.cpu().numpy(), np.concatenate((want[1][b, :n], want[2][b, :n, None], want[3][b, :n, None]), 1))


def test_fixed_nms_overflow_flag_and_properties_at_full_size():
    """BASELINE.json config-5 proposal count (8 x 100 800 x 85): no read-back, fixed shapes, outputs sorted by score,
    every kept box of a class clear of the better kept boxes of that class; a too small capacity raises the flag."""
    from ayolov2_amd.fixed_nms import BatchedNMS
    g = torch.Generator().manual_seed(0)
    B, N, nc, img = 8, 100800, 80, 1280
    xy, wh = torch.rand(B, N, 2, generator=g) * img, torch.rand(B, N, 2, generator=g) ** 3 * img / 2 + 2
    pred = torch.cat((xy - wh / 2, xy + wh / 2, torch.sigmoid(torch.randn(B, N, 1, generator=g) * 2 - 9.5),
                      torch.sigmoid(torch.randn(B, N, nc, generator=g) * 2 - 4)), 2).cuda()
    nms = BatchedNMS(nc, top_k=512, keep_top_k=100, score_threshold=0.001, iou_threshold=0.65)
    num, boxes, scores, classes = nms.from_prediction(pred)
    assert not bool(nms.overflow)
    assert boxes.shape == (B, 100, 4) and scores.shape == (B, 100) and classes.shape == (B, 100) and num.shape == (B, 1)
    s, c, bx, n = scores.cpu().numpy(), classes.cpu().numpy(), boxes.cpu().numpy(), num.cpu().numpy().ravel()
    for b in range(B):
        assert 0 < n[b] <= 100
        assert (np.diff(s[b, :n[b]]) <= 0).all() and (s[b, :n[b]] > 0.001).all()
        assert (c[b, n[b]:] == -1).all() and (s[b, n[b]:] == 0).all()
        for i in range(n[b]):
            for j in range(i):
                if c[b, i] == c[b, j]:
                    assert not ops_ref._trt_jaccard(bx[b, j], bx[b, i]) > np.float32(0.65)
    small = BatchedNMS(nc, top_k=512, keep_top_k=100, score_threshold=0.001, iou_threshold=0.65, capacity=1000)
    small.from_prediction(pred)
    assert bool(small.overflow)


def test_fixed_nms_graph_capture_replays():
    """The sequence has no host read-back: capture it once in a hipGraph, replay it on new data."""
    from ayolov2_amd.fixed_nms import BatchedNMS
    kw = dict(top_k=64, keep_top_k=32, score_threshold=0.05, iou_threshold=0.5)
    nms = BatchedNMS(8, capacity=2 * 2048 * 8, **kw)
    static = synth(2, 2048, 8, 320, seed=1, xyxy=True).cuda()
    nms.from_prediction(static)                       # allocates the work buffers outside the capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            outs = nms.from_prediction(static)
    for seed in (2, 3):
        fresh = synth(2, 2048, 8, 320, seed=seed, xyxy=True)
        static.copy_(fresh)
        graph.replay()
        torch.cuda.synchronize()
        _same(outs, oracle_from_pred(fresh, True, **kw), f"replay seed {seed}")


def test_validator_accepts_nms_engine():
    """A model with the fixed-shape NMS appended looks like the reference's TensorRT wrapper to YoloValidator: the second
    output is a tensor of counts and the ragged rows come from convert_trt_out (train_utils.py:456-457, 262-283)."""
    import os
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.fixed_nms import BatchedNMS, NMSEngine
    from ayolov2_amd.validator import YoloValidator
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ayolov2_amd", "configs", "yolov5n.yaml")
    torch.manual_seed(2)
    m = YOLOModel(cfg).cuda().eval()
    with torch.no_grad():                                     # make the random head fire (see test_gpu_validator.py)
        for conv in m.model[-1].conv:
            b = conv.bias.view(3, -1)
            b[:, 4] += 6.0
            b[:, 5:9] += 5.5
    nc = int(m.model[-1].nc)
    B = 3
    imgs = torch.rand(B, 3, 128, 160)
    kw = dict(top_k=128, keep_top_k=50, score_threshold=0.25, iou_threshold=0.6)
    eng = NMSEngine(m, BatchedNMS(nc, capacity=B * 1260 * nc, **kw), box_xyxy=False)
    out, n_objs = eng(imgs.cuda())
    assert out.shape == (B, 50, 6) and n_objs.shape == (B,) and int(n_objs.sum()) > 0
    with torch.no_grad():
        pred = m(imgs.cuda())[0].float().cpu()
    want = oracle_from_pred(pred, False, **kw)
    np.testing.assert_array_equal(n_objs.cpu().numpy(), want[0].ravel())
    np.testing.assert_array_equal(out.cpu().numpy(), np.concatenate((want[1], want[2][..., None], want[3][..., None]), 2))
    val = YoloValidator(eng, torch.device("cuda"), {"conf_t": 0.25, "iou_t": 0.6})
    targets = torch.tensor([[0, 0, 0.3, 0.3, 0.3, 0.4], [0, 2, 0.7, 0.6, 0.2, 0.3], [2, 1, 0.5, 0.5, 0.5, 0.5]])
    shapes = [((128, 160), ((1.0, 1.0), (0.0, 0.0)))] * B
    val.validation_step((imgs, targets, ["a", "b", "c"], shapes))
    res = val.compute_statistics()
    assert val.seen == B and sum(len(s[1]) for s in val.statistics["stats"]) == int(n_objs.sum())
    assert 0.0 <= res["map50"] <= 1.0
