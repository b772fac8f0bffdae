"""Drop-in for /root/reference/yolo/yolo.py: iou, yolo_head, yolo_loss with the reference's
signatures, plus yolo_eval, which the reference only has inline (tiny_yolov1.py:82-140).
Operands are torch ROCm tensors (NumPy arrays are uploaded); every function is one libty1.so call.
"""
import numpy as np
import torch

from .. import ops


def _dev(x, device='cuda:0'):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, np.float32))
    return x.to(device, torch.float32).contiguous() if not x.is_cuda else x.to(torch.float32).contiguous()


def iou(pred_mins, pred_maxes, true_mins, true_maxes):
    """yolo.py:6-20: broadcasting IoU over the leading axes; the last axis is (x, y)."""
    return ops.iou(*(_dev(t) for t in (pred_mins, pred_maxes, true_mins, true_maxes)))


def yolo_head(feats, img_size=448):
    """yolo.py:23-45: feats (N,S,S,B,4) -> box_xy, box_wh in pixels (float32)."""
    return ops.yolo_head(_dev(feats), img_size)


class _YoloLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_true, y_pred, B, C, img_size):
        loss, _, grad = ops.yolo_loss_fwd_bwd(y_true, y_pred, B, C, img_size, want_grad=True)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return None, grad * g, None, None, None


def yolo_loss(y_true, y_pred, B=2, img_size=448):
    """yolo.py:48-107: scalar batch-SUM loss, differentiable w.r.t. y_pred (closed-form backward
    computed by the same kernel).  C is taken from the label width (C+5), so the C=1 twins
    (faces/trainFace.py:66-126, img_size=224) are the same call."""
    y_true, y_pred = _dev(y_true), _dev(y_pred)
    C = y_true.shape[-1] - 5
    return _YoloLoss.apply(y_true, y_pred, B, C, img_size)


def yolo_eval(prediction, B=2, C=None, img_size=448, iou_threshold=0.2):
    """The reference's inline decode + adaptive threshold + class-agnostic NMS (tiny_yolov1.py:82-140)
    over a batch.  prediction (N,S,S,C+5B) -> (boxes (N,S*S*B,4) float64 [x0,y0,x1,y1] pixels,
    scores (N,S*S*B) f32, classes (N,S*S*B) i32, keep (N,S*S*B) u8), slots in raster (row, col, box)
    order exactly as the reference scans them."""
    prediction = _dev(prediction)
    if prediction.dim() == 3:
        prediction = prediction.unsqueeze(0)
    if C is None:
        C = prediction.shape[-1] - 5 * B
    return ops.decode_nms(prediction, B, C, img_size, iou_threshold)
