"""COCO-format result writer with the reference's interface (scripts/utils/multi_queue.py:60-339 ``ResultWriterTorch``:
``add_outputs(names, outputs, img_size, shapes)``, ``to_json``, ``close``).

The reference ships every image's detections to a consumer process that un-letterboxes, converts and appends them one image
at a time in numpy.  Here the per-detection arithmetic of a whole batch is ONE HIP launch (``ayolo_coco_rows``: scale_coords
-> clip -> [x, y, w, h] -> category id) followed by one asynchronous device->host copy into pinned memory; rows are turned
into json objects only when the file is written.  Same float32 arithmetic, so the numbers equal the reference's."""
from __future__ import annotations

import json
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

# YOLO class index -> COCO category id (the 80-of-91 table, multi_queue.py:78-159)
COCO80_TO_91 = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28, 31, 32, 33, 34, 35, 36,
                37, 38, 39, 40, 41, 42, 43, 44, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 67,
                70, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 84, 85, 86, 87, 88, 89, 90]


class ResultWriter:
    label_fixer = COCO80_TO_91

    def __init__(self, file_name: str) -> None:
        self.file_name = file_name
        self.seen_paths = set()
        self._pending: List[tuple] = []         # (image ids per row (host), pinned rows, event)
        self._cat = None

    def add_outputs(self, names: Sequence[str], outputs: Sequence[Optional[torch.Tensor]], img_size: Tuple[int, int],
                    shapes: Optional[Sequence[Tuple]] = None) -> None:
        """names: image paths; outputs: per image (n, 6) [x1, y1, x2, y2, conf, cls] or None; img_size: batched image size
        (h, w); shapes: per image (original (h, w), ((ratio_w, ratio_h), (pad_w, pad_h))) or None (boxes stay xyxy)."""
        keep = []
        for i, name in enumerate(names):
            if name in self.seen_paths:
                continue
            self.seen_paths.add(name)
            if outputs[i] is not None and outputs[i].shape[0]:
                keep.append(i)
        if not keep:
            return
        dev = outputs[keep[0]].device
        if dev.type != "cuda":
            raise _lib.AyoloError("ResultWriter: detections must live on the GPU (there is no CPU fallback)")
        det = torch.cat([outputs[i].float() for i in keep], 0).contiguous()
        counts = [int(outputs[i].shape[0]) for i in keep]
        img_idx = np.repeat(np.arange(len(keep), dtype=np.int32), counts)
        lb = np.zeros((len(keep), 6), dtype=np.float32)
        for k, i in enumerate(keep):
            if shapes is not None:
                h0, w0 = shapes[i][0]
                gain = min(img_size[0] / h0, img_size[1] / w0)                     # scale_coords with ratio_pad=None
                lb[k] = (gain, (img_size[1] - w0 * gain) / 2, (img_size[0] - h0 * gain) / 2, w0, h0, 1.0)
        if self._cat is None or self._cat.device != dev:
            self._cat = torch.tensor(self.label_fixer, dtype=torch.int32, device=dev)
        meta = torch.from_numpy(np.concatenate((img_idx.view(np.float32), lb.reshape(-1)))).pin_memory().to(dev, non_blocking=True)
        n = det.shape[0]
        out = torch.empty((n, 6), dtype=torch.float32, device=dev)
        _lib.call("ayolo_coco_rows", det.data_ptr(), meta.data_ptr(), n, meta.data_ptr() + 4 * n, self._cat.data_ptr(),
                  len(self.label_fixer), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        host = torch.empty((n, 6), dtype=torch.float32).pin_memory()
        host.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ids = np.repeat(np.array([int(Path(names[i]).stem) for i in keep], dtype=np.int64), counts)
        self._pending.append((ids, host, ev, (det, meta, out)))

    def objects(self) -> List[dict]:
        objs = []
        for ids, host, ev, _ in self._pending:
            ev.synchronize()
            rows = host.numpy()
            objs.extend({"image_id": int(i), "category_id": int(r[5]), "bbox": [float(v) for v in r[:4]], "score": float(r[4])}
                        for i, r in zip(ids, rows))
        return objs

    def to_json(self, filepath: Optional[str] = None) -> None:
        with open(filepath or self.file_name, "w") as f:
            json.dump(self.objects(), f)

    def close(self) -> None:
        self.to_json(self.file_name)
