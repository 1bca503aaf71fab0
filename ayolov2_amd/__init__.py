"""ayolov2_amd -- MI355X-native hot path of AYolov2 (conv stack fwd/bwd, head decode + NMS, Tucker-2)."""
from .model import YOLOModel  # noqa: F401
from .modules import C3, SPPF, Bottleneck, Concat, Conv, UpSample, YOLOHead  # noqa: F401

__all__ = ["YOLOModel", "Conv", "C3", "Bottleneck", "SPPF", "UpSample", "Concat", "YOLOHead"]
