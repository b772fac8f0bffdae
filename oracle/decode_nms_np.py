"""ORACLE (test infrastructure only): NumPy restatement of the reference's decode +
threshold + NMS for ONE image, for small cases and as a cross-check of decode_nms.c.

Follows /root/reference/tiny_yolov1.py:82-140 (and the C=1 twin faces/testFace.py:51-110)
with the same NumPy dtypes (scores/conf float32, geometry float64); the only
change is that C, B, S and img_size are parameters and results are returned, not drawn.
"""
import numpy as np


def yolo_head_np(feats, img_size=448):
    """tiny_yolov1.py:36-55.  feats (S,S,B,4) f32 -> xy float64, wh float32."""
    S0, S1 = feats.shape[0:2]
    col = np.arange(S1).reshape(1, S1, 1)            # x offset = column index   
    row = np.arange(S0).reshape(S0, 1, 1)            # y offset = row index
    conv_index = np.stack(np.broadcast_arrays(col, row), axis=-1)      # (S,S,1,2) int64
    conv_dims = np.array([S0, S1]).reshape(1, 1, 1, 2)
    box_xy = (feats[..., :2] + conv_index) / conv_dims * img_size
    box_wh = feats[..., 2:4] * img_size
    return box_xy, box_wh


def x_y_w_h_to_min_max(xy, wh):
    """utils.py:19-23"""
    return xy - wh / 2, xy + wh / 2


def iou_np(pred_mins, pred_maxes, true_mins, true_maxes):
    """tiny_yolov1.py:58-72"""
    intersect_wh = np.maximum(np.minimum(pred_maxes, true_maxes) - np.maximum(pred_mins, true_mins), 0.)
    inter = intersect_wh[..., 0] * intersect_wh[..., 1]
    pred_wh = pred_maxes - pred_mins
    true_wh = true_maxes - true_mins
    union = pred_wh[..., 0] * pred_wh[..., 1] + true_wh[..., 0] * true_wh[..., 1] - inter
    return inter / union


def decode_nms_one(prediction, B=2, C=20, img_size=448, iou_thr=0.2):
    """prediction (S,S,C+5B) f32 -> boxes (S*S*B,4) f64, score f32, cls i64, keep bool, iou trace."""
    S = prediction.shape[0]
    p = np.asarray(prediction, np.float32)
    predict_class = np.reshape(p[..., :C], [S, S, 1, C]).copy()
    predict_trust = np.reshape(p[..., C:C + B], [S, S, B, 1]).copy()
    predict_box = np.reshape(p[..., C + B:], [S, S, B, 4]).copy()
    with np.errstate(all='ignore'):
        predict_scores = predict_class * predict_trust
        box_classes = np.argmax(predict_scores, axis=-1)
        box_class_scores = np.max(predict_scores, axis=-1)
        best = np.max(box_class_scores, axis=-1, keepdims=True)
        box_mask = box_class_scores >= best
        filter_mask = box_class_scores >= np.max(box_class_scores) / 2
        filter_mask = (filter_mask * box_mask)[..., None]
        predict_box *= filter_mask
        box_classes = box_classes[..., None] * filter_mask
        box_xy, box_wh = yolo_head_np(predict_box, img_size)
        mins, maxs = x_y_w_h_to_min_max(box_xy, box_wh)
        predict_trust *= filter_mask
        score = predict_trust.copy()
        nms_mask = np.zeros_like(filter_mask)
        trace = []
        tmax = np.max(predict_trust)
        last = (0, 0, 0)
        while tmax > 0:
            for idx in zip(*np.nonzero(predict_trust[..., 0] == tmax)):      # raster order
                nms_mask[idx] = True
                filter_mask[idx] = False
                last = idx
            for idx in zip(*np.nonzero(filter_mask[..., 0])):
                v = iou_np(mins[last], maxs[last], mins[idx], maxs[idx])
                trace.append(float(v))
                if v > iou_thr:
                    filter_mask[idx] = False
            predict_trust *= filter_mask
            tmax = np.max(predict_trust)
        mins = mins * nms_mask
        maxs = maxs * nms_mask
    boxes = np.concatenate([mins, maxs], axis=-1).reshape(S * S * B, 4)
    return (boxes, score.reshape(-1), box_classes.reshape(-1), nms_mask.reshape(-1),
            np.array(trace, np.float64))
