"""ctypes binding of libayolo_hip.so (the C ABI declared in include/ayolo.h).

There is deliberately NO fallback: if the shared library is missing or a call fails the product path raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# AYOLO_LIB selects another build of the same ABI (A/B timing of two kernel versions on one box: tools/ab_bench.sh)
LIB_PATH = os.environ.get("AYOLO_LIB") or os.path.join(_HERE, "libayolo_hip.so")

F16, F32 = 0, 1
EPI_NONE, EPI_AFFINE, EPI_AFFINE_SILU, EPI_HEAD, EPI_AFFINE_RES, EPI_AFFINE_SILU_RES = 0, 1, 2, 3, 4, 5


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("dtype", "B", "H", "W", "Cin", "ldx", "Cout", "ldy", "kh", "kw", "sh", "sw",
                                     "ph", "pw", "Ho", "Wo")]


class BnSeg(ctypes.Structure):
    """ayolo_bn_seg (include/ayolo.h): one Conv-BN-act block whose BatchNorm-backward sums a dgrad epilogue accumulates."""
    _fields_ = [("z", c_void_p), ("mean_invstd", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("sums", c_void_p),
                ("ldz", c_int), ("c0", c_int), ("C", c_int), ("reserved", c_int)]


class BnApplySeg(ctypes.Structure):
    """ayolo_bn_apply_seg (include/ayolo.h): one of the two blocks of an ayolo_bn_act_bwd_apply2 launch."""
    _fields_ = [("da", c_void_p), ("save_mean", c_void_p), ("save_invstd", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("sums", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("ldda", c_int), ("C", c_int)]


class WgradJob(ctypes.Structure):
    """ayolo_wgrad_job (include/ayolo.h): one layer's weight gradient inside a grouped launch."""
    _fields_ = [("conv", ConvDesc), ("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("alpha", c_float),
                ("dy_slot", c_int), ("overwrite", c_int), ("xact", c_int), ("xscale", c_void_p), ("xshift", c_void_p), ("dw_ld", c_int), ("reserved", c_int)]


class XfSeg(ctypes.Structure):
    """ayolo_xf_seg (include/ayolo.h): one input segment of a transform-on-load conv."""
    _fields_ = [("x", c_void_p), ("ld", c_int), ("C", c_int), ("act", c_int), ("virt", c_int)]


class XfFin(ctypes.Structure):
    """ayolo_xf_fin (include/ayolo.h): BatchNorm finalize of one virtual input segment inside the consumer's launch."""
    _fields_ = [("stats", c_void_p), ("reps", c_int), ("sld", c_int), ("C", c_int), ("c0", c_int), ("count", c_double),
                ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("momentum", c_float),
                ("running_mean", c_void_p), ("running_var", c_void_p), ("save_mean", c_void_p), ("save_invstd", c_void_p)]


class LossLevel(ctypes.Structure):
    _fields_ = ([("pred", c_void_p)] + [(n, c_int64) for n in ("sb", "sa", "sy", "sx")]
                + [(n, c_int) for n in ("B", "na", "ny", "nx", "no", "n")]
                + [(n, c_void_p) for n in ("b", "a", "gj", "gi", "tcls", "tbox", "anch", "own", "score")]
                + [("balance", c_float), ("grad", c_void_p), ("head", c_void_p), ("next", c_void_p), ("rowbox", c_void_p),
                   ("dz", c_void_p), ("ldz", c_int), ("dz_dtype", c_int), ("dbias", c_void_p)])


class AyoloError(RuntimeError):
    pass


_P = c_void_p

# name -> argtypes (restype is int unless noted).  Mirrors include/ayolo.h one to one.
_SIGNATURES = {
    "ayolo_conv_fwd": [POINTER(ConvDesc), _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, _P],
    "ayolo_conv_fwd_xf": [POINTER(ConvDesc), POINTER(XfSeg), c_int, _P, _P, POINTER(XfFin), c_int, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, _P],
    "ayolo_conv_dgrad": [POINTER(ConvDesc), _P, _P, _P, c_int, _P],
    "ayolo_conv_dgrad_bn": [POINTER(ConvDesc), _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P],
    "ayolo_conv_wgrad": [POINTER(ConvDesc), _P, _P, _P, c_float, _P, c_size_t, _P],
    "ayolo_wgrad_group_size": [POINTER(WgradJob), c_int, POINTER(c_size_t), POINTER(c_size_t)],
    "ayolo_wgrad_group_build": [POINTER(WgradJob), c_int, _P, c_size_t],
    "ayolo_wgrad_group_run": [_P, _P, _P, c_size_t, POINTER(c_void_p), c_int, _P],
    "ayolo_wgrad_group_info": [_P, c_int, POINTER(c_int64), c_int],
    "ayolo_wgrad_group_item": [_P, c_int, c_int64, POINTER(c_int64)],
    "ayolo_wgrad3_geometry": [POINTER(ConvDesc), POINTER(c_int64), c_int],
    "ayolo_stem_bn_wgrad": [POINTER(ConvDesc), _P, _P, c_int, _P, _P, _P, _P, _P, c_int, _P, c_int, _P, _P, _P, c_float, c_float, _P],
    "ayolo_cast_weight": [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P],
    "ayolo_ema_update": [_P, c_int, c_float, _P],
    "ayolo_cast_weights": [_P, c_int, c_int, _P],
    "ayolo_sgd_step": [_P, c_int, _P, _P, _P, _P],
    "ayolo_coco_rows": [_P, _P, c_int64, _P, _P, c_int, _P, _P],
    "ayolo_bn_finalize": [_P, c_int, c_int, c_double, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P],
    "ayolo_bn_finalize_ld": [_P, c_int, c_int, c_int, c_double, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P],
    "ayolo_affine_act": [c_int, _P, c_int, _P, c_int, c_int64, c_int, _P, _P, c_int, _P],
    "ayolo_bn_train_act": [c_int, _P, c_int, _P, c_int, c_int64, c_int, _P, c_int, c_int, c_double, _P, _P, c_float, c_float, _P, _P,
                           _P, _P, c_int, _P, c_int, _P],
    "ayolo_bn_act_bwd_reduce": [c_int, _P, c_int, _P, c_int, c_int64, c_int, _P, _P, _P, _P, c_int, _P, c_int, _P],
    "ayolo_bn_act_bwd_apply": [c_int, _P, c_int, _P, c_int, _P, c_int, c_int64, c_int, _P, _P, _P, _P, c_int, _P, c_int,
                               _P, _P, c_float, _P],
    "ayolo_bn_act_bwd_apply_res": [c_int, _P, c_int, _P, c_int, _P, c_int, c_int64, c_int, _P, _P, _P, _P, c_int, _P, c_int,
                                   _P, _P, c_float, _P, c_int, c_int, _P],
    "ayolo_bn_act_bwd_apply2": [c_int, _P, c_int, _P, c_int, c_int64, POINTER(BnApplySeg), POINTER(BnApplySeg), c_int, c_int, c_float, _P],
    "ayolo_sppf_pool_fwd": [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P],
    "ayolo_sppf_pool_bwd": [c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "ayolo_sppf_pool_supported": [c_int, c_int, c_int, c_int],
    "ayolo_maxpool_fwd": [c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "ayolo_maxpool_bwd": [c_int, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "ayolo_upsample2x_fwd": [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "ayolo_upsample2x_bwd": [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "ayolo_pack_input": [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "ayolo_head_grad_pack": [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P],
    "ayolo_copy2d": [c_int, _P, c_int, _P, c_int, c_int64, c_int, c_int, _P],
    "ayolo_yolo_loss_fwd": [POINTER(LossLevel), c_int] + [c_float] * 8 + [_P, _P, _P],
    "ayolo_yolo_loss_bwd": [POINTER(LossLevel), c_int] + [c_float] * 8 + [_P, _P],
    "ayolo_yolo_loss_bwd_packed": [POINTER(LossLevel), c_int] + [c_float] * 8 + [_P, _P],
    "ayolo_match_detections": [_P, _P, c_int64, _P, _P, c_int64, _P, c_int, _P, _P, _P, _P, _P],
    "ayolo_head_decode": [_P, POINTER(c_int64), c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, c_int64, c_int64, _P],
    "ayolo_head_decode_aug": [_P, POINTER(c_int64), c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, c_int64, c_int64, c_float, c_int,
                              c_float, c_int64, c_int64, _P],
    "ayolo_nms_candidates": [_P, c_int, c_int, c_int, c_float, c_int, c_int, _P, _P, c_int, _P, _P, _P, c_uint32, c_int,
                             _P],
    "ayolo_nms_key_bits": [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)],
    "ayolo_iota_u32": [_P, c_uint32, _P],
    "ayolo_seg_max_coord": [_P, _P, _P, c_int, _P, _P],
    "ayolo_sort_pairs_u64": [_P, _P, _P, _P, c_uint32, c_int, c_int, _P, POINTER(c_size_t), _P],
    "ayolo_nms_obj_keys": [_P, c_int, c_int, c_int, _P, _P, _P],
    "ayolo_gather_rows": [_P, _P, _P, c_uint32, c_int, _P],
    "ayolo_nms_mask": [_P, _P, _P, _P, c_int, c_uint32, c_float, c_float, _P, c_int, _P, _P],
    "ayolo_nms_reduce": [_P, _P, _P, _P, _P, c_int, c_uint32, _P, _P, _P, c_uint32, _P],
    "ayolo_nms_class_keys": [_P, _P, _P, c_int, c_int, c_uint32, _P, _P, _P, _P, _P],
    "ayolo_nms_class_layout": [_P, c_uint32, c_int, _P, _P, _P, _P, _P],
    "ayolo_nms_class_merge": [_P, _P, _P, _P, _P, c_int, c_uint32, _P, c_int, c_uint32, c_uint32, _P, _P, _P, _P, _P,
                              POINTER(c_size_t), _P],
    "ayolo_nms_class_fast": [_P, c_int, c_int, c_int, c_float, c_int, _P, c_float, c_uint32, c_uint32, c_uint32, _P, POINTER(c_size_t), _P, _P, _P],
    "ayolo_trt_nms_key_bits": [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)],
    "ayolo_trt_nms_candidates": [_P, c_int, c_int, c_int, c_float, c_int, _P, _P, _P, c_uint32, _P],
    "ayolo_trt_nms_layout": [_P, c_uint32, c_int, c_int, c_int, c_uint32, _P, _P, _P, _P],
    "ayolo_trt_nms_mask": [_P, _P, _P, _P, c_int, c_uint32, c_float, _P, _P],
    "ayolo_trt_nms_final_keys": [_P, _P, c_int, c_int, c_uint32, _P, _P, POINTER(c_int), _P],
    "ayolo_trt_nms_emit": [_P, _P, _P, c_int, c_int, c_uint32, c_uint32, _P, _P, _P, _P, _P],
    "ayolo_box_iou": [_P, c_int64, _P, c_int64, _P, _P],
    "ayolo_iou_colmax": [_P, _P, c_float, c_uint32, _P, _P],
    "ayolo_matrix_nms_decay": [_P, _P, c_float, c_uint32, _P, _P, _P],
    "ayolo_merge_boxes": [_P, c_uint32, c_float, _P, c_uint32, c_float, _P, _P, _P],
    "ayolo_affine_act_res": [c_int, _P, c_int, _P, c_int, c_int64, c_int, _P, _P, c_int, _P, c_int, _P],
    "ayolo_bn_eval_affine": [_P, _P, _P, _P, _P, c_float, c_int, _P, _P, _P],
    "ayolo_run_ops": [_P, c_int, _P],
    "ayolo_run_ops_ex": [_P, c_int, _P, c_int],
    "ayolo_side_stream_join": [_P],
    "ayolo_run_ops_timed": [_P, c_int, _P, _P],
    "ayolo_fill_zero": [_P, c_size_t, _P],
    "ayolo_release_thread_state": [],
}

EXPORTED = sorted(list(_SIGNATURES) + ["ayolo_version", "ayolo_last_error", "ayolo_conv_wgrad_workspace"])

_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AyoloError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(or `make -C ayolov2_amd/csrc`). There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = c_int
        l.ayolo_version.restype = c_int
        l.ayolo_conv_wgrad_workspace.argtypes = [POINTER(ConvDesc)]
        l.ayolo_conv_wgrad_workspace.restype = c_size_t
        l.ayolo_last_error.restype = c_char_p
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ayolo_last_error()
        raise AyoloError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args), name)


def bump_versions(tensors) -> None:
    """Kernels write parameters / buffers through raw pointers, which autograd's version counters do not see.  Everything
    that caches derived data keyed on ``tensor._version`` (the inference executor's folded BatchNorm + fp16 weight copies,
    functional._WeightCache) relies on the writers calling this after such a write (optim.SGD.step, ModelEMA.update, the
    BatchNorm running statistics of a training forward)."""
    import torch
    ts = [t for t in tensors if t is not None]
    if ts:
        with torch.no_grad():
            torch._C._increment_version(ts)


class _Roctx:
    """roctx ranges (rocprofv3 --marker-trace) around the stages of the train / validation step: the reference's `dt`
    stage timers and logger lines (scripts/utils/train_utils.py:420-470) as profiler markers.  A no-op
    when libroctx64 is not on the loader path."""

    def __init__(self) -> None:
        self._lib = None
        for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
            try:
                self._lib = ctypes.CDLL(name)
                self._lib.roctxRangePushA.argtypes = [c_char_p]
                break
            except OSError:
                continue

    def push(self, name: str) -> None:
        if self._lib is not None:
            self._lib.roctxRangePushA(name.encode())

    def pop(self) -> None:
        if self._lib is not None:
            self._lib.roctxRangePop()


_roctx: Optional[_Roctx] = None


class roctx_range:
    def __init__(self, name: str) -> None:
        self.name = name

    def __enter__(self):
        global _roctx
        if _roctx is None:
            _roctx = _Roctx()
        _roctx.push(self.name)
        return self

    def __exit__(self, *exc):
        _roctx.pop()
        return False

