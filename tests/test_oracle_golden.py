"""The CPU oracle (oracle/) against the golden vectors generated from the reference's own Python
(tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import ops_ref, tucker_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_box_iou(golden_dir):
    g = _load(golden_dir, "g1_box_iou.npz")
    got = ops_ref.box_iou(g["box1"], g["box2"])
    np.testing.assert_array_equal(got, g["iou"])          # bit-exact, NaN positions included


def test_g2_bbox_iou(golden_dir):
    g = _load(golden_dir, "g2_bbox_iou.npz")
    p, t = g["pred"], g["target"]
    for tag, kw in {"iou": {}, "giou": {"g_iou": True}, "diou": {"d_iou": True}, "ciou": {"c_iou": True}}.items():
        for fmt in (0, 1):
            if fmt:
                q = np.concatenate((p[:, :2], p[:, :2] + p[:, 2:]), 1)
                tt = np.concatenate((t[:, :2], t[:, :2] + t[:, 2:]), 1)
            else:
                q, tt = p, t
            got = ops_ref.bbox_iou(q.T, tt, x1y1x2y2=bool(fmt), **kw)
            np.testing.assert_allclose(got, g[f"{tag}_{fmt}"], rtol=2e-6, atol=2e-6)   # atan differs in ulps


def test_g3_general(golden_dir):
    g = _load(golden_dir, "g3_general.npz")
    np.testing.assert_array_equal(ops_ref.xywh2xyxy(g["x"]), g["xywh2xyxy"])
    np.testing.assert_array_equal(
        ops_ref.xywh2xyxy(g["x"] / np.float32(640), ratio=(0.5, 0.75), wh=(640, 480), pad=(3.0, 7.0)), g["xywh2xyxy_r"])
    np.testing.assert_array_equal(ops_ref.clip_coords(g["xyxy"].copy(), (640, 480)), g["clip"])
    np.testing.assert_array_equal(ops_ref.scale_coords((640, 640), g["xyxy"].copy(), (480, 600)), g["scale_a"])
    np.testing.assert_array_equal(
        ops_ref.scale_coords((640, 640), g["xyxy"].copy(), (720, 1280), ratio_pad=((0.5, 0.5), (0.0, 140.0))), g["scale_b"])


NMS_TYPES = ["nms", "batched_nms", "fast_nms", "matrix_nms", "merge_nms"]


def _cmp(got, want, nms_type):
    assert got.shape == want.shape
    if nms_type in ("matrix_nms", "merge_nms"):   # exp / matmul summation order: values to tolerance
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-4)
    else:
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("nms_type", NMS_TYPES)
def test_g4_non_max_suppression(golden_dir, nms_type):
    g = _load(golden_dir, "g4_nms.npz")
    pred = g["pred"].astype(np.float32)
    for agn in (0, 1):
        for ml in (0, 1):
            res = ops_ref.non_max_suppression(pred, conf_thres=0.001, iou_thres=0.65, multi_label=bool(ml),
                                              agnostic=bool(agn), nms_type=nms_type)
            for bi, r in enumerate(res):
                _cmp(r, g[f"{nms_type}_a{agn}_m{ml}_{bi}"], nms_type)


def test_g4_variants(golden_dir):
    g = _load(golden_dir, "g4_nms.npz")
    pred = g["pred"].astype(np.float32)
    res = ops_ref.non_max_suppression(pred, conf_thres=0.25, iou_thres=0.45, classes=[0, 3, 17])
    for bi, r in enumerate(res):
        _cmp(r, g[f"cls_filter_{bi}"], "nms")
    lab = [g["hybrid_labels_0"], np.zeros((0, 5), np.float32)]
    res = ops_ref.non_max_suppression(pred, conf_thres=0.1, iou_thres=0.6, labels=lab, multi_label=True)
    for bi, r in enumerate(res):
        _cmp(r, g[f"hybrid_{bi}"], "nms")


@pytest.mark.parametrize("nms_type", NMS_TYPES)
def test_g5_batched_nms(golden_dir, nms_type):
    g = _load(golden_dir, "g5_batched_nms.npz")
    pred = _load(golden_dir, "g4_nms.npz")["pred"].astype(np.float32)
    for agn in (0, 1):
        for nb in (500, 1000):
            res = ops_ref.batched_nms(pred, conf_thres=0.001, iou_thres=0.65, nms_box=nb, agnostic=bool(agn),
                                      nms_type=nms_type)
            for bi, r in enumerate(res):
                _cmp(r, g[f"{nms_type}_a{agn}_n{nb}_{bi}"], nms_type)


def test_k2_greedy_hand_cases():
    # two identical boxes -> second suppressed
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 10]], np.float32)
    assert ops_ref.tv_nms(b, np.array([0.9, 0.8], np.float32), 0.5).tolist() == [0]
    # equal scores: stable -> lower index first
    assert ops_ref.tv_nms(b, np.array([0.9, 0.9], np.float32), 0.5).tolist() == [0]
    # IoU exactly == thr is kept (strict >): boxes [0,0,2,1] and [1,0,3,1]: inter 1, union 3 -> 1/3
    b = np.array([[0, 0, 2, 1], [1, 0, 3, 1]], np.float32)
    thr = float(np.float32(1) / np.float32(3))
    assert ops_ref.tv_nms(b, np.array([0.9, 0.8], np.float32), thr).tolist() == [0, 1]
    assert ops_ref.tv_nms(b, np.array([0.9, 0.8], np.float32), np.nextafter(np.float32(thr), np.float32(0))).tolist() == [0]
    # empty / single
    assert ops_ref.tv_nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).tolist() == []
    assert ops_ref.tv_nms(b[:1], np.array([0.1], np.float32), 0.5).tolist() == [0]
    # chain: A^B, B^C overlap, A and C do not: A kept, B suppressed, C kept
    b = np.array([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], np.float32)
    assert ops_ref.tv_nms(b, np.array([0.9, 0.8, 0.7], np.float32), 0.4).tolist() == [0, 2]


def test_g7_tucker(golden_dir):
    g = _load(golden_dir, "g7_tucker.npz")
    for name in "abc":
        L, M, r = g[f"shape_{name}"]
        rs = np.random.default_rng(7)
        Y = (rs.standard_normal((L, r)) @ rs.standard_normal((r, M)) / np.sqrt(r) + 0.05 * rs.standard_normal((L, M))).astype(np.float32)
        assert tucker_ref.evbmf_rank(Y) == int(g[f"rank_{name}"])
    w = g["conv_w"]
    ranks = tucker_ref.estimate_ranks(w)
    assert ranks == g["conv_ranks"].tolist()
    first, core, last = tucker_ref.tucker2_conv_weights(w, ranks)
    assert [list(first.shape), list(core.shape), list(last.shape)] == g["conv_shapes"].tolist()
    # reconstruction error small (synthetic low multilinear rank + noise)
    rec = np.einsum("abhw,oa,ib->oihw", core, last[:, :, 0, 0], first[:, :, 0, 0].T)
    assert np.abs(rec - w).mean() < 0.02


def test_validator_matching_and_ap_vs_golden(golden_dir):
    """G8: the reference's own YoloValidator.process_batch / ap_per_class outputs (train_utils.py:294-333,
    metrics.py:476-548) vs the oracle restatement and vs the product's host-side AP."""
    import numpy as np
    from oracle import ops_ref
    from ayolov2_amd.validator import ap_per_class
    g = np.load(os.path.join(golden_dir, "g8_validator.npz"))
    for i in range(int(g["n_img"])):
        if f"correct{i}" in g.files:
            np.testing.assert_array_equal(ops_ref.process_batch(g[f"det{i}"], g[f"lab{i}"], g["iouv"]), g[f"correct{i}"])
    for fn in (ops_ref.ap_per_class, ap_per_class):
        p, r, ap, f1, cls = fn(g["tp"], g["conf"], g["pcls"], g["tcls"])
        np.testing.assert_allclose(p, g["ap_p"], rtol=1e-12)
        np.testing.assert_allclose(r, g["ap_r"], rtol=1e-12)
        np.testing.assert_allclose(ap, g["ap"], rtol=1e-12)
        np.testing.assert_allclose(f1, g["ap_f1"], rtol=1e-12)
        np.testing.assert_array_equal(cls, g["ap_cls"])


def test_trt_batched_nms_restatement_hand_cases():
    """Known answers for the BatchedNMS_TRT restatement (third-party plugin: parity unpinned, so the arithmetic is at
    least pinned by hand): +1 extents, strict '>' on the IoU, the unit-box quirk of disjoint boxes, padding values."""
    a, b = np.float32([0, 0, 9, 9]), np.float32([5, 0, 14, 9])                 # 10x10 boxes (with the +1), overlap 5x10
    assert ops_ref._trt_bbox_size(a) == np.float32(100)
    assert ops_ref._trt_jaccard(a, b) == np.float32(50) / np.float32(150)
    assert ops_ref._trt_jaccard(a, np.float32([100, 100, 109, 109])) == np.float32(1) / np.float32(199)
    assert ops_ref._trt_bbox_size(np.float32([5, 5, 4, 9])) == 0
    boxes = np.float32([[[0, 0, 9, 9], [5, 0, 14, 9], [0, 0, 9, 9], [50, 50, 59, 59]]])
    scores = np.float32([[[0.9, 0.0], [0.8, 0.7], [0.9, 0.05], [0.2, 0.6]]])
    thr = float(np.float32(50) / np.float32(150))
    num, ob, osc, ocl = ops_ref.batched_nms_trt(boxes, scores, top_k=4, keep_top_k=5, score_threshold=0.1, iou_threshold=thr)
    # class 0: box 0 (0.9) keeps, box 2 is identical (suppressed), box 1 has IoU == thr exactly (kept: strict '>'), box 3 kept
    # class 1: box 1 (0.7), box 3 (0.6); box 2's 0.05 is under the score threshold
    assert num[0, 0] == 5
    np.testing.assert_array_equal(osc[0], np.float32([0.9, 0.8, 0.7, 0.6, 0.2]))
    np.testing.assert_array_equal(ocl[0], np.float32([0, 0, 1, 1, 0]))
    np.testing.assert_array_equal(ob[0, 0], boxes[0, 0])
    num, ob, osc, ocl = ops_ref.batched_nms_trt(boxes, scores, top_k=4, keep_top_k=5, score_threshold=0.95, iou_threshold=0.5)
    assert num[0, 0] == 0 and (ocl == -1).all() and (osc == 0).all() and (ob == 0).all()
