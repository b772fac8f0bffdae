"""The compiled hot path: frozen conv backbone -> Dense -> head activations [-> decode + NMS], run
as a fixed sequence of libty1.so kernel launches on one HIP stream with all buffers preallocated.

Replaces what Keras' Model.predict executes for /root/reference/models/model_tiny_yolov1.py:42-99
and the NumPy post-processing of /root/reference/tiny_yolov1.py:82-140.
"""
import numpy as np
import torch

from . import ops
from .params import CONV_CHANNELS, N_POOL, BN_EPS, LEAKY_ALPHA, hwio_to_ohwi, dense_to_device_layout


class DeviceModel:
    def __init__(self, params, S=7, B=2, C=20, img_size=448, device='cuda:0'):
        self.S, self.B, self.C, self.img_size = S, B, C, img_size
        self.device = torch.device(device)
        self.units = S * S * (C + 5 * B)
        self.conv = []
        for i, (cin, cout) in enumerate(CONV_CHANNELS):
            scale, shift = ops.fold_bn(params[f'bn{i}_gamma'], params[f'bn{i}_beta'], params[f'bn{i}_mean'],
                                       params[f'bn{i}_var'], BN_EPS)
            self.conv.append(dict(
                w=torch.from_numpy(hwio_to_ohwi(params[f'conv{i}_w'])).to(self.device),
                scale=torch.from_numpy(scale).to(self.device), shift=torch.from_numpy(shift).to(self.device),
                pool=i < N_POOL))
        assert params['dense_w'].shape[1] == self.units, (params['dense_w'].shape, self.units)
        self.dense_wt = torch.from_numpy(dense_to_device_layout(params['dense_w'])).to(self.device)
        self.dense_b = torch.from_numpy(np.ascontiguousarray(params['dense_b'], np.float32)).to(self.device)
        self.K = self.dense_wt.shape[1]
        self._ws = {}

    # ------------------------------------------------------------------ workspaces
    def workspace(self, N):
        ws = self._ws.get(N)
        if ws is None:
            dev, f32 = self.device, torch.float32
            acts, h = [], self.img_size
            for i, (_, cout) in enumerate(CONV_CHANNELS):
                if i < N_POOL:
                    h //= 2
                acts.append(torch.empty((N, h, h, cout), dtype=f32, device=dev))
            splits = ops.pick_splits(self.K, N, self.units)
            ldz = ops.round_up(self.units, 4)
            slots = self.S * self.S * self.B
            ws = dict(acts=acts, part=torch.empty((splits, N, ldz), dtype=f32, device=dev),
                      y=torch.empty((N, self.S, self.S, self.C + 5 * self.B), dtype=f32, device=dev),
                      det=(torch.empty((N, slots, 4), dtype=torch.float64, device=dev),
                           torch.empty((N, slots), dtype=f32, device=dev),
                           torch.empty((N, slots), dtype=torch.int32, device=dev),
                           torch.empty((N, slots), dtype=torch.uint8, device=dev)))
            self._ws[N] = ws
        return ws

    # ------------------------------------------------------------------ forward pieces
    def backbone(self, x, ws=None):
        """x (N,H,W,3) f32 NHWC on the device -> features (N, 12544) (HWC flatten)."""
        ws = ws or self.workspace(x.shape[0])
        for layer, out in zip(self.conv, ws['acts']):
            x = ops.conv3x3_bn_leaky(x, layer['w'], layer['scale'], layer['shift'], layer['pool'], LEAKY_ALPHA, out=out)
        return x.reshape(x.shape[0], -1)

    def head(self, feats, ws=None):
        ws = ws or self.workspace(feats.shape[0])
        part = ops.dense_splitk(feats, self.dense_wt, self.units, part=ws['part'])
        return ops.head_activate(part, self.dense_b, self.S, self.B, self.C, out=ws['y'])

    def predict(self, x):
        """-> y_pred (N,S,S,C+5B) on the device (valid until the next call with the same N)."""
        ws = self.workspace(x.shape[0])
        return self.head(self.backbone(x, ws), ws)

    def detect(self, x, iou_thr=0.2):
        """-> (boxes f64 (N,S*S*B,4), score, cls, keep) on the device."""
        ws = self.workspace(x.shape[0])
        y = self.head(self.backbone(x, ws), ws)
        return ops.decode_nms(y, self.B, self.C, self.img_size, iou_thr, out=ws['det'])
