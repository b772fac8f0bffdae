#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference; never on the GPU box).
The reference is imported unmodified; the packages it needs but that are absent
here (cv2, tensorflow/keras) are replaced by recording stubs, and for
yolo/yolo.py the 16 `K.*` backend functions it uses are mapped one-to-one onto
torch (CPU, fp32) so that `loss.backward()` gives dL/dy_pred.

Outputs (small .npz files next to this script):
  nms_c20.npz    prediction (M,7,7,30) f32  -> drawn rectangles / labels / iou trace
                 of tiny_yolov1._main            (/root/reference/tiny_yolov1.py:75-161)
  nms_c1.npz     prediction (M,7,7,11) f32  -> rectangles / str(conf) / iou trace
                 of faces.testFace._main        (/root/reference/faces/testFace.py:45-166)
  head_np.npz    YOLO_head / iou NumPy twins     (/root/reference/tiny_yolov1.py:36-72)
  loss_c20.npz   (y_true, y_pred) -> loss, dL/dy_pred, yolo_head, iou of yolo/yolo.py
                 (/root/reference/yolo/yolo.py:6-107)
  loss_c1.npz    same for faces.trainFace.yolo_loss (/root/reference/faces/trainFace.py:66-126)
  head_c20.npz   Dense output z (n,1470) -> ReshapeYOLO((7,7,30)).call(z)
                 (/root/reference/models/model_tiny_yolov1.py:14-34)
  head_c1.npz    z (n,539) -> ReshapeYOLO((7,7,11)).call(z)  (/root/reference/faces/trainFace.py:30-63)

Usage: python tests/golden/make_golden.py
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------- stubs


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        return _Dummy()


class _StubModule(types.ModuleType):
    """Module whose every unknown attribute is a dummy class (so `class X(Layer)` parses)."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return type(name, (_Dummy,), {})


class _CvRecorder(_StubModule):
    """cv2 stand-in that records the drawing calls the reference makes."""
    COLOR_BGR2RGB = 4
    CAP_DSHOW = 700

    def __init__(self):
        super().__init__('cv2')
        self.calls = []

    def reset(self):
        self.calls = []

    def imread(self, path):
        return np.zeros((448, 448, 3), np.uint8)

    def resize(self, img, size):
        return img

    def cvtColor(self, img, code):
        return img

    def rectangle(self, img, p0, p1, color, *a):
        self.calls.append(('rect', tuple(p0), tuple(p1)))

    def putText(self, img, text, org, *a):
        self.calls.append(('text', text, tuple(org)))

    def imshow(self, *a):
        pass

    def imwrite(self, *a):
        pass

    def waitKey(self, *a):
        return 0


def _k_shim():
    """torch (CPU fp32) equivalents of the Keras backend functions yolo/yolo.py uses."""
    K = types.ModuleType('tensorflow.keras.backend')

    def _t(x):
        return x if isinstance(x, torch.Tensor) else torch.as_tensor(x)

    K.maximum = lambda a, b: torch.maximum(_t(a), _t(b).to(_t(a).dtype) if not isinstance(b, torch.Tensor) else b)
    K.minimum = lambda a, b: torch.minimum(_t(a), _t(b))
    K.shape = lambda x: tuple(int(s) for s in x.shape)
    K.arange = lambda start, stop=None: torch.arange(start, stop)
    K.tile = lambda x, n: _t(x).repeat(*[int(v) for v in n])
    K.expand_dims = lambda x, axis=-1: _t(x).unsqueeze(axis)
    K.flatten = lambda x: _t(x).reshape(-1)
    K.transpose = lambda x: _t(x).t()
    K.stack = lambda xs, axis=0: torch.stack([_t(v) for v in xs], dim=axis)
    K.reshape = lambda x, shape: _t(x).reshape([int(v) for v in shape])
    K.cast = lambda x, dtype: _t(x).to(dtype)
    K.dtype = lambda x: x.dtype
    K.max = lambda x, axis=None, keepdims=False: (torch.amax(x, dim=axis, keepdim=keepdims)
                                                   if axis is not None else torch.max(x))
    K.square = lambda x: x * x
    K.sum = lambda x, axis=None, keepdims=False: (torch.sum(x) if axis is None
                                                   else torch.sum(x, dim=axis, keepdim=keepdims))
    # Keras backend sqrt clips its argument to [0, inf) before tf.sqrt.
    K.sqrt = lambda x: torch.sqrt(torch.clamp(x, min=0.0))
    # ReshapeYOLO.call (models/model_tiny_yolov1.py:14-34) needs three more; Keras defaults: axis=-1
    K.softmax = lambda x, axis=-1: torch.softmax(_t(x), dim=axis)
    K.sigmoid = lambda x: torch.sigmoid(_t(x))
    K.concatenate = lambda xs, axis=-1: torch.cat([_t(v) for v in xs], dim=axis)
    return K


def install_stubs():
    cv = _CvRecorder()
    sys.modules['cv2'] = cv
    tf = _StubModule('tensorflow')
    keras = _StubModule('tensorflow.keras')
    tf.keras = keras
    sys.modules['tensorflow'] = tf
    sys.modules['tensorflow.keras'] = keras
    K = _k_shim()
    keras.backend = K
    sys.modules['tensorflow.keras.backend'] = K
    for sub in ('layers', 'regularizers', 'callbacks', 'optimizers', 'utils'):
        m = _StubModule('tensorflow.keras.' + sub)
        setattr(keras, sub, m)
        sys.modules['tensorflow.keras.' + sub] = m
    sys.path.insert(0, REF)
    return cv


# --------------------------------------------------------------------------- inputs


def softmax(x, axis=-1):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def make_predictions(C, rng, n_random=24):
    """A mix of typical, stress and adversarial head outputs, layout [cls C | conf 2 | box 8]."""
    D = C + 10
    cases = []

    def base(conf_scale=1.0, box_lo=0.0, box_hi=1.0, cls_sharp=4.0):
        p = np.empty((7, 7, D), np.float32)
        p[..., :C] = softmax(cls_sharp * rng.random((7, 7, C))).astype(np.float32) if C > 1 else 1.0
        p[..., C:C + 2] = (rng.random((7, 7, 2)) * conf_scale).astype(np.float32)
        p[..., C + 2:] = (box_lo + (box_hi - box_lo) * rng.random((7, 7, 8))).astype(np.float32)
        return p

    # 1. SURVEY 8(d) "NMS-only stress": class softmax(4U), conf U, box U
    for _ in range(n_random):
        cases.append(base())
    # 2. dense small boxes: almost nothing suppressed (many kept)
    for _ in range(8):
        p = base(box_lo=0.0, box_hi=1.0)
        p[..., C + 2:] = p[..., C + 2:] * np.array([1, 1, .12, .12, 1, 1, .12, .12], np.float32)
        p[..., C:C + 2] = (0.5 + 0.5 * rng.random((7, 7, 2))).astype(np.float32)
        cases.append(p)
    # 3. typical: a few confident cells, rest near zero
    for _ in range(8):
        p = base(conf_scale=0.05)
        for _k in range(rng.integers(1, 6)):
            r, c, k = rng.integers(0, 7), rng.integers(0, 7), rng.integers(0, 2)
            p[r, c, C + k] = np.float32(0.6 + 0.4 * rng.random())
        cases.append(p)
    # 4. exact confidence ties between many slots (Q4), some overlapping
    for _ in range(6):
        p = base()
        vals = np.array([0.9, 0.75, 0.5, 0.25], np.float32)
        p[..., C:C + 2] = vals[rng.integers(0, 4, (7, 7, 2))]
        p[..., C + 2:] *= np.array([1, 1, .5, .5, 1, 1, .5, .5], np.float32)
        cases.append(p)
    # 5. both boxes of a cell tie on class score (box_mask keeps both), identical boxes
    for _ in range(4):
        p = base()
        p[..., C + 1] = p[..., C]
        if rng.random() < 0.5:
            p[..., C + 6:] = p[..., C + 2:C + 6]
        cases.append(p)
    # 6. near-uniform scores: every cell passes max/2 (random-init stress)
    for _ in range(4):
        p = base()
        p[..., :C] = np.float32(1.0 / C)
        p[..., C:C + 2] = (0.5 + 1e-3 * rng.standard_normal((7, 7, 2))).astype(np.float32)
        p[..., C + 2:] = (0.5 + 0.05 * rng.standard_normal((7, 7, 8))).astype(np.float32)
        cases.append(p)
    # 7. zero-area boxes (0/0 -> NaN -> not suppressed), zero confidences, all-zero input
    p = base()
    p[..., C + 4:C + 6] = 0
    p[..., C + 8:C + 10] = 0
    cases.append(p)
    p = base()
    p[..., C + 2:] = 0
    cases.append(p)
    p = base()
    p[..., C:C + 2] = 0
    cases.append(p)
    cases.append(np.zeros((7, 7, D), np.float32))
    # 8. IoU straddling the 0.2 threshold: concentric boxes with area ratio ~0.2
    for eps in (-1e-6, 0.0, 1e-6, -1e-3, 1e-3):
        p = np.zeros((7, 7, D), np.float32)
        p[..., :C] = np.float32(1.0 / C) if C > 1 else 1.0
        p[3, 3, C] = 0.9
        p[3, 3, C + 2:C + 6] = [0.5, 0.5, 0.5, 0.5]
        s = np.float32(0.5 * np.sqrt(0.2 + eps))
        p[3, 4, C] = 0.8
        p[3, 4, C + 2:C + 6] = [-0.5, 0.5, s, s]      # same centre as the cell-(3,3) box
        p[2, 3, C + 1] = 0.7
        p[2, 3, C + 6:C + 10] = [0.5, 1.5, s, s]
        cases.append(p)
    # 9. raw unnormalised values incl. negatives (algorithm is agnostic to the ranges)
    for _ in range(4):
        p = rng.standard_normal((7, 7, D)).astype(np.float32)
        cases.append(p)
    return np.stack(cases).astype(np.float32)


def make_loss_inputs(C, rng, N):
    """y_true per data_sequence.py:54-86 (first writer wins), y_pred = head activations."""
    D_t, D_p = C + 5, C + 10
    y_true = np.zeros((N, 7, 7, D_t), np.float32)
    for n in range(N):
        for _ in range(3):
            r, c, cls = rng.integers(0, 7), rng.integers(0, 7), rng.integers(0, C)
            if y_true[n, r, c, C + 4] == 0:
                y_true[n, r, c, cls] = 1
                y_true[n, r, c, C:C + 4] = rng.random(4)
                y_true[n, r, c, C + 4] = 1
    z = rng.standard_normal((N, 7, 7, D_p)).astype(np.float32)
    y_pred = np.empty_like(z)
    y_pred[..., :C] = softmax(z[..., :C])
    y_pred[..., C:] = 1.0 / (1.0 + np.exp(-z[..., C:]))
    # adversarial cells: identical predicted boxes (IoU tie -> both responsible)
    y_pred[0, :, :, C + 6:] = y_pred[0, :, :, C + 2:C + 6]
    # a labelled cell whose predicted boxes cannot overlap the label (IoU 0 for both)
    y_true[1, 0, 0, :] = 0
    y_true[1, 0, 0, 0] = 1
    y_true[1, 0, 0, C:C + 4] = [0.5, 0.5, 0.01, 0.01]
    y_true[1, 0, 0, C + 4] = 1
    y_pred[1, 0, 0, C + 2:] = [0.99, 0.99, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01]
    # a labelled cell with a zero-area label AND a zero-area predicted box 0: iou_0 = 0/0 = NaN, K.max propagates it,
    # `>=` is False for both boxes -> neither is responsible (yolo/yolo.py:79-82); the sqrt of the zero width has an
    # infinite derivative, so autodiff gives 0 * inf = NaN for dL/dw, dL/dh of that box
    y_true[2, 3, 3, :] = 0
    y_true[2, 3, 3, 0] = 1
    y_true[2, 3, 3, C:C + 4] = [0.5, 0.5, 0.0, 0.0]
    y_true[2, 3, 3, C + 4] = 1
    y_pred[2, 3, 3, C + 2:C + 6] = [0.25, 0.75, 0.0, 0.0]
    # a labelled cell (ordinary label) whose predicted box 0 has zero width: iou_0 = 0, box 1 responsible
    y_true[3, 4, 2, :] = 0
    y_true[3, 4, 2, 0] = 1
    y_true[3, 4, 2, C:C + 4] = [0.5, 0.5, 0.3, 0.4]
    y_true[3, 4, 2, C + 4] = 1
    y_pred[3, 4, 2, C + 2:C + 6] = [0.5, 0.5, 0.0, 0.2]
    y_pred[3, 4, 2, C + 6:C + 10] = [0.4, 0.6, 0.25, 0.35]
    return y_true, y_pred.astype(np.float32)


# --------------------------------------------------------------------------- drivers


def _pack_ragged(rows, dtype):
    off = np.zeros(len(rows) + 1, np.int64)
    for i, r in enumerate(rows):
        off[i + 1] = off[i] + len(r)
    flat = np.concatenate([np.asarray(r, dtype).reshape(len(r), -1) for r in rows]
                          + [np.zeros((0, 1), dtype)][:0]) if any(len(r) for r in rows) else np.zeros((0, 1), dtype)
    return flat, off


def run_nms_c20(cv, preds):
    import tiny_yolov1 as ref
    traces = []
    real_iou = ref.iou

    def rec_iou(*a):
        v = real_iou(*a)
        traces[-1].append(float(v))
        return v

    ref.iou = rec_iou
    rects, labels = [], []
    args = types.SimpleNamespace(weights_path='w', image_path='img.jpg')
    for p in preds:
        cv.reset()
        traces.append([])
        ref.TinyYOLOv1.predict = lambda self, _p=p: _p[None].copy()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            with np.errstate(all='ignore'):
                ref._main(args)
        rects.append([c[1] + c[2] for c in cv.calls if c[0] == 'rect'])
        labels.append([ref.classes_name.index(c[1]) for c in cv.calls if c[0] == 'text'])
    ref.iou = real_iou
    return rects, labels, traces


def run_nms_c1(cv, preds):
    from faces import testFace as ref
    traces = []
    real_iou = ref.iou

    def rec_iou(*a):
        v = real_iou(*a)
        traces[-1].append(float(v))
        return v

    ref.iou = rec_iou
    rects, confs = [], []

    class Fake:
        def __init__(self, p):
            self.p = p

        def predict_path(self, path):
            return self.p[None].copy()

    for p in preds:
        cv.reset()
        traces.append([])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            with np.errstate(all='ignore'):
                ref._main('img.jpg', Fake(p))
        rects.append([c[1] + c[2] for c in cv.calls if c[0] == 'rect'])
        confs.append([c[1] for c in cv.calls if c[0] == 'text'])
    ref.iou = real_iou
    return rects, confs, traces


def save_nms(path, preds, rects, second, traces, second_name, second_dtype):
    n = len(preds)
    roff = np.zeros(n + 1, np.int64)
    toff = np.zeros(n + 1, np.int64)
    for i in range(n):
        roff[i + 1] = roff[i] + len(rects[i])
        toff[i + 1] = toff[i] + len(traces[i])
    rflat = np.array([r for rr in rects for r in rr], np.int64).reshape(-1, 4)
    sflat = np.array([s for ss in second for s in ss], second_dtype)
    tflat = np.array([t for tt in traces for t in tt], np.float64)
    np.savez_compressed(path, prediction=preds, rect_off=roff, rects=rflat,
                        iou_off=toff, iou_trace=tflat, **{second_name: sflat})
    print(f'{os.path.basename(path)}: {n} cases, {len(rflat)} kept boxes, {len(tflat)} iou calls')


def run_head_np(rng):
    import tiny_yolov1 as ref
    from utils import X_Y_W_H_To_Min_Max
    feats = rng.random((7, 7, 2, 4)).astype(np.float32)
    xy, wh = ref.YOLO_head(feats)
    xy224, wh224 = ref.YOLO_head(feats, img_size=224)
    mn, mx = X_Y_W_H_To_Min_Max(xy, wh)
    a_min = rng.random((50, 2)) * 200
    a_max = a_min + rng.random((50, 2)) * 200
    b_min = rng.random((50, 2)) * 200
    b_max = b_min + rng.random((50, 2)) * 200
    with np.errstate(all='ignore'):
        ious = ref.iou(a_min, a_max, b_min, b_max)
    assert xy.dtype == np.float64 and wh.dtype == np.float32
    np.savez_compressed(os.path.join(OUT, 'head_np.npz'), feats=feats, xy=xy, wh=wh, xy224=xy224,
                        wh224=wh224, xy_min=mn, xy_max=mx, a_min=a_min, a_max=a_max,
                        b_min=b_min, b_max=b_max, iou=ious)
    print('head_np.npz: YOLO_head xy', xy.dtype, 'wh', wh.dtype)


def run_loss(module_loss, head_fn, iou_fn, C, rng, N, path, img_size):
    y_true, y_pred = make_loss_inputs(C, rng, N)
    yt = torch.from_numpy(y_true)
    yp = torch.from_numpy(y_pred).requires_grad_(True)
    loss = module_loss(yt, yp)
    loss.backward()
    per_image = []
    for n in range(N):
        per_image.append(float(module_loss(yt[n:n + 1], yp[n:n + 1].detach())))
    # yolo_head / iou of the K version on the predicted boxes
    boxes = torch.from_numpy(y_pred[..., C + 2:]).reshape(-1, 7, 7, 2, 4)
    hxy, hwh = head_fn(boxes, img_size=img_size)
    mins, maxes = hxy - hwh / 2, hxy + hwh / 2
    ious = iou_fn(mins[:, :, :, 0], maxes[:, :, :, 0], mins[:, :, :, 1], maxes[:, :, :, 1])
    np.savez_compressed(path, y_true=y_true, y_pred=y_pred, loss=np.float32(loss.item()),
                        loss_per_image=np.array(per_image, np.float32),
                        grad=yp.grad.numpy(), head_xy=hxy.numpy(), head_wh=hwh.numpy(),
                        iou_b0_b1=ious.numpy(), img_size=np.int32(img_size))
    print(f'{os.path.basename(path)}: N={N} loss={loss.item():.6f} |grad|max={yp.grad.abs().max():.4f}')


def run_head(layer_cls, target_shape, C, path, seed):
    """The reference's own ReshapeYOLO.call on seeded Dense outputs (planar logits), incl. saturating extremes."""
    rng = np.random.default_rng(seed)
    S, D = target_shape[0], target_shape[2]
    units = S * S * D
    z = (3.0 * rng.standard_normal((10, units))).astype(np.float32)
    z[1] *= 10.0                                   # |z| up to ~100: softmax needs the max-subtraction
    z[2, ::3] = 30.0
    z[2, 1::3] = -30.0
    z[3] = 0.0
    z[4, :S * S * C] = 88.0 + rng.random(S * S * C).astype(np.float32)      # exp() of these overflows fp32 without it
    z[5] = -104.0 + rng.random(units).astype(np.float32)                   # sigmoid underflows towards 0
    z[6] = np.float32(1e-4) * rng.standard_normal(units).astype(np.float32)
    layer = layer_cls(target_shape)
    with torch.no_grad():
        y = layer.call(torch.from_numpy(z))
    assert tuple(y.shape) == (10,) + tuple(target_shape)
    assert layer.compute_output_shape((None, units)) == (None,) + tuple(target_shape)
    np.savez_compressed(path, z=z, y=y.numpy(), C=np.int32(C))
    print(f'{os.path.basename(path)}: ReshapeYOLO{tuple(target_shape)}.call on {z.shape}')


def main():
    cv = install_stubs()
    rng = np.random.default_rng(20260101)

    preds = make_predictions(20, rng)
    rects, labels, traces = run_nms_c20(cv, preds)
    save_nms(os.path.join(OUT, 'nms_c20.npz'), preds, rects, labels, traces, 'labels', np.int64)

    preds1 = make_predictions(1, rng)
    rects1, confs1, traces1 = run_nms_c1(cv, preds1)
    save_nms(os.path.join(OUT, 'nms_c1.npz'), preds1, rects1, confs1, traces1, 'conf_str', 'U32')

    run_head_np(rng)

    from yolo import yolo as ref_yolo
    run_loss(ref_yolo.yolo_loss, ref_yolo.yolo_head, ref_yolo.iou, 20, rng, 6,
             os.path.join(OUT, 'loss_c20.npz'), 448)
    from faces import trainFace
    run_loss(trainFace.yolo_loss, ref_yolo.yolo_head, ref_yolo.iou, 1, rng, 6,
             os.path.join(OUT, 'loss_c1.npz'), 224)

    # head activations: the model files' own layer (own rng: the streams of the fixtures above stay as they were)
    from models import model_tiny_yolov1 as ref_model
    run_head(ref_model.ReshapeYOLO, (7, 7, 30), 20, os.path.join(OUT, 'head_c20.npz'), 20260202)
    run_head(trainFace.ReshapeYOLO, (7, 7, 11), 1, os.path.join(OUT, 'head_c1.npz'), 20260203)


if __name__ == '__main__':
    main()
