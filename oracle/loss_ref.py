"""ORACLE (test infrastructure only): restatement of the reference's yolo_head / iou /
yolo_loss in torch-CPU fp32, plus the closed-form gradient in NumPy.

Follows /root/reference/yolo/yolo.py:6-107 (C=20, img_size 448) and its C=1 twin
/root/reference/faces/trainFace.py:66-126 (img_size 224; the divisor 448/224 in the box
terms is the image size in both).  Pinned by tests/golden/loss_c20.npz / loss_c1.npz,
which hold (loss, dL/dy_pred) computed by the reference's own source run against a torch
mapping of the Keras backend (tests/golden/make_golden.py).
"""
import numpy as np
import torch


def yolo_head_ref(feats, img_size=448):
    """yolo/yolo.py:23-45.  feats (N,S,S,B,4) -> box_xy, box_wh (both feats.dtype)."""
    S0, S1 = feats.shape[1:3]
    col = torch.arange(S1, dtype=feats.dtype).reshape(1, 1, S1, 1)
    row = torch.arange(S0, dtype=feats.dtype).reshape(1, S0, 1, 1)
    conv_index = torch.stack(torch.broadcast_tensors(col, row), dim=-1)     # (1,S,S,1,2): x+=c, y+=r
    conv_dims = torch.tensor([S0, S1], dtype=feats.dtype).reshape(1, 1, 1, 1, 2)
    box_xy = (feats[..., :2] + conv_index) / conv_dims * img_size
    box_wh = feats[..., 2:4] * img_size
    return box_xy, box_wh


def iou_ref(pred_mins, pred_maxes, true_mins, true_maxes):
    """yolo/yolo.py:6-20"""
    intersect_mins = torch.maximum(pred_mins, true_mins)
    intersect_maxes = torch.minimum(pred_maxes, true_maxes)
    intersect_wh = torch.clamp(intersect_maxes - intersect_mins, min=0.)
    inter = intersect_wh[..., 0] * intersect_wh[..., 1]
    pred_wh = pred_maxes - pred_mins
    true_wh = true_maxes - true_mins
    union = pred_wh[..., 0] * pred_wh[..., 1] + true_wh[..., 0] * true_wh[..., 1] - inter
    return inter / union


def yolo_loss_ref(y_true, y_pred, C=20, B=2, img_size=448):
    """yolo/yolo.py:48-107.  y_true (N,S,S,C+5), y_pred (N,S,S,C+5B) -> scalar (batch SUM, Q12)."""
    S = y_true.shape[1]
    label_class = y_true[..., :C]
    label_box = y_true[..., C:C + 4]
    response_mask = y_true[..., C + 4].unsqueeze(-1)
    predict_class = y_pred[..., :C]
    predict_trust = y_pred[..., C:C + B]
    predict_box = y_pred[..., C + B:]
    _label_box = label_box.reshape(-1, S, S, 1, 4)
    _predict_box = predict_box.reshape(-1, S, S, B, 4)

    label_xy, label_wh = yolo_head_ref(_label_box, img_size)
    label_xy_e, label_wh_e = label_xy.unsqueeze(3), label_wh.unsqueeze(3)
    l_min, l_max = label_xy_e - label_wh_e / 2, label_xy_e + label_wh_e / 2
    predict_xy, predict_wh = yolo_head_ref(_predict_box, img_size)
    p_xy_e, p_wh_e = predict_xy.unsqueeze(4), predict_wh.unsqueeze(4)
    p_min, p_max = p_xy_e - p_wh_e / 2, p_xy_e + p_wh_e / 2
    iou_scores = iou_ref(p_min, p_max, l_min, l_max)
    best_ious = torch.amax(iou_scores, dim=4)
    best_box = torch.amax(best_ious, dim=3, keepdim=True)
    box_mask = (best_ious >= best_box).to(best_ious.dtype)

    no_object_loss = 0.5 * (1 - box_mask * response_mask) * (0 - predict_trust) ** 2
    object_loss = box_mask * response_mask * (1 - predict_trust) ** 2
    confidence_loss = torch.sum(no_object_loss + object_loss)
    class_loss = torch.sum(response_mask * (label_class - predict_class) ** 2)

    box_mask = box_mask.unsqueeze(-1)
    response_mask = response_mask.unsqueeze(-1)
    box_loss = 5 * box_mask * response_mask * ((label_xy - predict_xy) / img_size) ** 2
    box_loss = box_loss + 5 * box_mask * response_mask * (
        (torch.sqrt(torch.clamp(label_wh, min=0)) - torch.sqrt(torch.clamp(predict_wh, min=0))) / img_size) ** 2
    return confidence_loss + class_loss + torch.sum(box_loss)


def yolo_loss_and_grad_ref(y_true, y_pred, C=20, B=2, img_size=448):
    """NumPy in -> (loss f32 scalar, per-image loss (N,), grad (N,S,S,C+5B)) via torch autograd."""
    yt = torch.from_numpy(np.ascontiguousarray(y_true, np.float32))
    yp = torch.from_numpy(np.ascontiguousarray(y_pred, np.float32)).requires_grad_(True)
    loss = yolo_loss_ref(yt, yp, C, B, img_size)
    loss.backward()
    per = np.array([float(yolo_loss_ref(yt[n:n + 1], yp[n:n + 1].detach(), C, B, img_size))
                    for n in range(yt.shape[0])], np.float32)
    return np.float32(loss.item()), per, yp.grad.numpy()


def yolo_loss_closed_form(y_true, y_pred, C=20, B=2, img_size=448):
    """SURVEY 8a row A8 closed form, float64 NumPy: (loss, grad).  Independent of autograd."""
    yt = np.asarray(y_true, np.float64)
    yp = np.asarray(y_pred, np.float64)
    N, S = yt.shape[0], yt.shape[1]
    R = yt[..., C + 4]
    l = yt[..., C:C + 4]
    col = np.arange(S).reshape(1, 1, S)
    row = np.arange(S).reshape(1, S, 1)
    box = yp[..., C + B:].reshape(N, S, S, B, 4)
    conf = yp[..., C:C + B]

    def corners(x, y, w, h, col, row):
        cx = (x + col) / S * img_size
        cy = (y + row) / S * img_size
        pw, ph = w * img_size, h * img_size
        return cx - pw / 2, cy - ph / 2, cx + pw / 2, cy + ph / 2

    lx0, ly0, lx1, ly1 = corners(l[..., 0], l[..., 1], l[..., 2], l[..., 3], col, row)
    ious = np.empty((N, S, S, B))
    with np.errstate(all='ignore'):
        for k in range(B):
            px0, py0, px1, py1 = corners(box[..., k, 0], box[..., k, 1], box[..., k, 2], box[..., k, 3], col, row)
            iw = np.maximum(np.minimum(px1, lx1) - np.maximum(px0, lx0), 0)
            ih = np.maximum(np.minimum(py1, ly1) - np.maximum(py0, ly0), 0)
            inter = iw * ih
            ious[..., k] = inter / ((px1 - px0) * (py1 - py0) + (lx1 - lx0) * (ly1 - ly0) - inter)
        M = (ious >= ious.max(axis=-1, keepdims=True)).astype(np.float64)
    MR = M * R[..., None]
    grad = np.zeros_like(yp)
    loss = np.sum(0.5 * (1 - MR) * conf ** 2 + MR * (1 - conf) ** 2)
    grad[..., C:C + B] = (1 - MR) * conf - 2 * MR * (1 - conf)
    loss += np.sum(R[..., None] * (yt[..., :C] - yp[..., :C]) ** 2)
    grad[..., :C] = 2 * R[..., None] * (yp[..., :C] - yt[..., :C])
    gbox = np.zeros((N, S, S, B, 4))
    for k in range(B):
        dx = l[..., 0] - box[..., k, 0]
        dy = l[..., 1] - box[..., k, 1]
        sw = np.sqrt(l[..., 2]) - np.sqrt(box[..., k, 2])
        sh = np.sqrt(l[..., 3]) - np.sqrt(box[..., k, 3])
        loss += np.sum(5 * MR[..., k] * ((dx / S) ** 2 + (dy / S) ** 2 + (sw ** 2 + sh ** 2) / img_size))
        gbox[..., k, 0] = -10 * MR[..., k] * dx / (S * S)
        gbox[..., k, 1] = -10 * MR[..., k] * dy / (S * S)
        with np.errstate(all='ignore'):
            for j in (2, 3):
                # autodiff through K.sqrt = sqrt(clip(., 0)): no gradient for w < 0; at w == 0 the chain rule gives
                # 0 * inf = NaN even for a box that is not responsible (golden cells [2,3,3] and [3,4,2])
                b_, l_ = box[..., k, j], np.maximum(l[..., j], 0)
                g_ = np.where(MR[..., k] > 0, 5 * MR[..., k] * (1 - np.sqrt(l_ / b_)) / img_size, np.where(b_ == 0, np.nan, 0))
                gbox[..., k, j] = np.where(b_ < 0, 0, g_)
    grad[..., C + B:] = gbox.reshape(N, S, S, 4 * B)
    return loss, grad
