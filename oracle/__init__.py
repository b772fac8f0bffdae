"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / the reported CPU baseline.  The product (yolo_amd/)
never imports it and has no CPU fallback.

Contents
  decode_nms.c / decode_nms()   plain-C restatement of tiny_yolov1.py:82-140 (pinned by goldens)
  decode_nms_np.py              NumPy restatement of the same lines (small cases; cross-check)
  model_ref.py                  torch-CPU fp32 restatement of models/model_tiny_yolov1.py:14-99
                                ("parity unpinned" at the TensorFlow boundary: SURVEY 8c)
  loss_ref.py                   restatement of yolo/yolo.py:6-107 (+ closed-form gradients),
                                pinned by tests/golden/loss_*.npz
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so with gcc (strict IEEE flags, see Makefile)."""
    subprocess.check_call(['make', '-s', '-C', _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.oracle_decode_nms.restype = ctypes.c_int
        lib.oracle_decode_nms.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def decode_nms(pred, B=2, C=20, img_size=448, iou_thr=0.2, want_trace=False):
    """pred (N,S,S,C+5B) float32 -> dict(boxes f64 (N,S*S*B,4), score f32, cls i32, keep u8[, trace])."""
    pred = np.ascontiguousarray(pred, np.float32)
    if pred.ndim == 3:
        pred = pred[None]
    N, S = pred.shape[0], pred.shape[1]
    assert pred.shape[2] == S and pred.shape[3] == C + 5 * B, pred.shape
    slots = S * S * B
    boxes = np.empty((N, slots, 4), np.float64)
    score = np.empty((N, slots), np.float32)
    cls = np.empty((N, slots), np.int32)
    keep = np.empty((N, slots), np.uint8)
    cap = N * slots * slots if want_trace else 0
    trace = np.empty(max(cap, 1), np.float64)
    tlen = ctypes.c_int64(0)
    rc = _lib().oracle_decode_nms(
        pred.ctypes.data, N, S, B, C, int(img_size), float(iou_thr),
        boxes.ctypes.data, score.ctypes.data, cls.ctypes.data, keep.ctypes.data,
        trace.ctypes.data if want_trace else None, cap, ctypes.byref(tlen))
    if rc != 0:
        raise ValueError('oracle_decode_nms: bad arguments')
    out = dict(boxes=boxes, score=score, cls=cls, keep=keep)
    if want_trace:
        out['trace'] = trace[:tlen.value].copy()
    return out
