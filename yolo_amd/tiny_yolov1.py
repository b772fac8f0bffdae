"""Drop-in for /root/reference/tiny_yolov1.py: TinyYOLOv1.predict(), the NumPy-twin helpers
YOLO_head / iou, and detect(image) -- the reference's inline post-processing of _main (:82-140)
turned into a callable.  Drawing (:142-161) is out of scope; detections are returned.
"""
import numpy as np
import torch

from . import ops
from .keras_facade import Model
from .models.model_tiny_yolov1 import model_tiny_YOLOv1
from .utils import load_img, load_img_device, X_Y_W_H_To_Min_Max  # noqa: F401

classes_name = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable',
                'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


class TinyYOLOv1(object):
    """Same constructor as the reference; the model is built ONCE (the reference rebuilds the graph
    and reloads the weights on every predict() call: tiny_yolov1.py:25-27)."""

    def __init__(self, weights_path=None, input_path=None, device='cuda:0'):
        self.weights_path, self.input_path = weights_path, input_path
        self.classes_name = list(classes_name)
        outputs, inputs = model_tiny_YOLOv1()
        self.inputs = inputs
        self.model = Model(inputs=inputs, outputs=outputs, device=device)
        if weights_path:
            self.model.load_weights(weights_path, by_name=True)

    def predict(self, image=None):
        """-> np.ndarray (1,7,7,30) float32 for the image at input_path (or an HxWx3 array in [0,1])."""
        if image is None:
            image, _, _ = load_img(path=self.input_path, shape=self.inputs.shape[1:])
        image = np.expand_dims(image, axis=0) if np.ndim(image) == 3 else image
        return self.model.predict(image, batch_size=1)

    def detect(self, image=None, iou_threshold=0.2):
        """Full path for one image or a batch: preprocessing -> model -> decode -> threshold -> NMS.
        Returns a list (one entry per image) of dicts with `boxes` (k,4) float64 pixels in the
        448x448 frame, `corners` (k,4) int (the reference's int() truncation), `classes`, `names`,
        `scores`, in the reference's raster order."""
        if image is None or isinstance(image, str):
            # file -> host decode -> uint8 upload -> resize + /255 on the GPU (same pixels as load_img, fp32)
            x, _, _ = load_img_device(image or self.input_path, tuple(self.inputs.shape[1:]), self.model.device)
            image = x[None]
        if isinstance(image, np.ndarray) and image.ndim == 3:
            image = image[None]
        x = self.model._to_device(image)
        boxes, score, cls, keep = self.model.engine.detect(x, iou_threshold)
        boxes, score, cls, keep = boxes.cpu().numpy(), score.cpu().numpy(), cls.cpu().numpy(), keep.cpu().numpy()
        out = []
        for n in range(boxes.shape[0]):
            idx = np.nonzero(keep[n])[0]
            out.append(dict(boxes=boxes[n, idx], corners=boxes[n, idx].astype(np.int64), classes=cls[n, idx],
                            names=[self.classes_name[c] for c in cls[n, idx]], scores=score[n, idx], slots=idx))
        return out


def _dev(x, dtype):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to('cuda:0', dtype).contiguous()


def YOLO_head(feats, img_size=448):
    """tiny_yolov1.py:36-55 (no batch axis): feats (S,S,B,4) float32 -> box_xy float64, box_wh float32."""
    xy, wh = ops.yolo_head_np(_dev(feats, torch.float32), img_size)
    return (xy.cpu().numpy(), wh.cpu().numpy()) if isinstance(feats, np.ndarray) else (xy, wh)


def iou(pred_mins, pred_maxes, true_mins, true_maxes):
    """tiny_yolov1.py:58-72 in float64."""
    out = ops.iou_f64(*(_dev(t, torch.float64) for t in (pred_mins, pred_maxes, true_mins, true_maxes)))
    return out.cpu().numpy() if isinstance(pred_mins, np.ndarray) else out
