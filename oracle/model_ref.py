"""ORACLE (test infrastructure only): torch-CPU fp32 restatement of the reference's Keras
model, layer by layer.

Follows /root/reference/models/model_tiny_yolov1.py:
  :42-90  original_backbone_network  8 x [Conv2D 3x3 'same' no bias -> BatchNormalization
          (inference, Keras default epsilon 1e-3: SURVEY Q9) -> LeakyReLU(0.1)], MaxPooling2D 2x2/2
          after the first six, Flatten in HWC order (Q10)
  :96     Dense(1470, linear)
  :14-34  ReshapeYOLO.call: planar split -> softmax(C) | sigmoid(B) | sigmoid(4B) -> concat

PARITY UNPINNED at the TensorFlow boundary: TensorFlow/Keras is not installable here and
the reference holds no tests, fixtures or weights for the model, so this
restatement is pinned only by an independent naive NumPy convolution
(tests/test_oracle_model.py), not by reference outputs.

`params` is the dict made by yolo_amd.params (NumPy arrays, Keras layouts):
  conv{i}_w (3,3,Cin,Cout) HWIO, bn{i}_gamma/beta/mean/var (Cout,), dense_w (12544,D*49), dense_b.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
N_CONV = 8
N_POOL = 6


def conv_block_ref(x_nchw, w_hwio, gamma, beta, mean, var, pool, alpha=0.1):
    w = torch.from_numpy(np.ascontiguousarray(w_hwio)).permute(3, 2, 0, 1).contiguous()   # OIHW
    y = F.conv2d(x_nchw, w, bias=None, stride=1, padding=1)
    g, b, m, v = (torch.from_numpy(np.asarray(a, np.float32)).reshape(1, -1, 1, 1) for a in (gamma, beta, mean, var))
    y = (y - m) / torch.sqrt(v + BN_EPS) * g + b          # Keras BatchNormalization, inference
    y = torch.where(y > 0, y, y * alpha)                  # LeakyReLU(alpha); the model's own slope is 0.1
    if pool:
        y = F.max_pool2d(y, 2, 2)
    return y


def _bf16(t):
    """round to the nearest bf16 (ties to even), kept in fp32"""
    return t.to(torch.bfloat16).to(torch.float32)


def backbone_ref(x_nhwc, params, return_all=False, bf16=False):
    """x (N,448,448,3) f32 NHWC -> (N,12544) flattened in HWC order.
    bf16=True restates the build's bf16 network (BASELINE config 3; the reference itself is fp32 only): the image,
    the conv weights and every layer's output are rounded to bf16, everything is accumulated in fp32."""
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc, np.float32)).permute(0, 3, 1, 2).contiguous()
    if bf16:
        x = _bf16(x)
    acts = []
    for i in range(N_CONV):
        w = params[f'conv{i}_w']
        if bf16:
            w = _bf16(torch.from_numpy(np.ascontiguousarray(w, np.float32))).numpy()
        x = conv_block_ref(x, w, params[f'bn{i}_gamma'], params[f'bn{i}_beta'],
                           params[f'bn{i}_mean'], params[f'bn{i}_var'], pool=i < N_POOL)
        if bf16:
            x = _bf16(x)
        if return_all:
            acts.append(x.permute(0, 2, 3, 1).contiguous().numpy())
    flat = x.permute(0, 2, 3, 1).contiguous().reshape(x.shape[0], -1)
    return (flat.numpy(), acts) if return_all else flat.numpy()


def dense_ref(x, w, b, bf16=False):
    """bf16=True: features and weights rounded to bf16 (the build's bf16 Dense), fp32 accumulation and bias."""
    xt, wt = torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(w))
    if bf16:
        xt, wt = _bf16(xt), _bf16(wt)
    return (xt @ wt + torch.from_numpy(b)).numpy()


def head_activate_ref(z, S=7, B=2, C=20):
    """ReshapeYOLO.call: z (N, S*S*(C+5B)) planar -> (N,S,S,C+5B) [cls | conf | box]."""
    z = torch.from_numpy(np.ascontiguousarray(z, np.float32))
    n = z.shape[0]
    i1 = S * S * C
    i2 = i1 + S * S * B
    cls = torch.softmax(z[:, :i1].reshape(n, S, S, C), dim=-1)
    conf = torch.sigmoid(z[:, i1:i2].reshape(n, S, S, B))
    box = torch.sigmoid(z[:, i2:].reshape(n, S, S, 4 * B))
    return torch.cat([cls, conf, box], dim=-1).numpy()


def model_ref(x_nhwc, params, S=7, B=2, C=20, bf16=False):
    feats = backbone_ref(x_nhwc, params, bf16=bf16)
    z = dense_ref(feats, params['dense_w'], params['dense_b'], bf16=bf16)
    return head_activate_ref(z, S, B, C)


def conv3x3_naive_np(x_hwc, w_hwio):
    """Independent check: direct 3x3 'same' convolution in float64 NumPy on one small image."""
    H, W, _ = x_hwc.shape
    xp = np.zeros((H + 2, W + 2, x_hwc.shape[2]), np.float64)
    xp[1:-1, 1:-1] = x_hwc
    out = np.zeros((H, W, w_hwio.shape[3]), np.float64)
    for kh in range(3):
        for kw in range(3):
            out += xp[kh:kh + H, kw:kw + W, :] @ w_hwio[kh, kw].astype(np.float64)
    return out
