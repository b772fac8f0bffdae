"""Thin host wrappers: torch ROCm tensors are used only as device buffers; every op below is one
call into libty1.so (hand-written gfx950 kernels) on the current torch stream."""
import torch

from . import _lib

DENSE_K_ALIGN = 32


def _chk(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f'{name}: expected a tensor on the GPU (ROCm device)')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: must be contiguous')
    return t


def _chk_out(t, dtype, shape, name):
    """A caller-provided output buffer: its raw pointer goes to a kernel, so type, device, layout and shape must fit."""
    _chk(t, dtype, name)
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f'{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}')
    return t


def round_up(v, m):
    return (v + m - 1) // m * m


def fold_bn(gamma, beta, mean, var, eps=1e-3):
    """NumPy (host) in/out: scale, shift of an inference BatchNormalization (Keras eps 1e-3)."""
    import numpy as np
    g, b, m, v = (np.ascontiguousarray(a, np.float32) for a in (gamma, beta, mean, var))
    scale = np.empty_like(g)
    shift = np.empty_like(g)
    _lib.call('ty1_fold_bn', g.ctypes.data, b.ctypes.data, m.ctypes.data, v.ctypes.data,
              float(eps), int(g.shape[0]), scale.ctypes.data, shift.ctypes.data)
    return scale, shift


def conv3x3_workspace_bytes(N, H, W, Cin, Cout, pool):
    """Scratch bytes ty1_conv3x3_bn_leaky_ws_f32 wants for this problem (0: it would not split its tail)."""
    import ctypes
    n = ctypes.c_size_t(0)
    _lib.call('ty1_conv3x3_workspace_bytes', N, H, W, Cin, Cout, int(bool(pool)), ctypes.addressof(n))
    return int(n.value)


def conv3x3_bn_leaky(x, w_ohwi, scale, shift, pool, alpha=0.1, out=None, workspace=None):
    """x (N,H,W,Cin) NHWC f32 -> (N,H/2,W/2,Cout) if pool else (N,H,W,Cout).
    workspace: optional 1-D uint8/float32 device tensor used as scratch for the K-split tail."""
    _chk(x, torch.float32, 'x'); _chk(w_ohwi, torch.float32, 'w')
    _chk(scale, torch.float32, 'scale'); _chk(shift, torch.float32, 'shift')
    N, H, W, Cin = x.shape
    Cout = w_ohwi.shape[0]
    assert tuple(w_ohwi.shape) == (Cout, 3, 3, Cin), w_ohwi.shape
    oshape = (N, H // 2, W // 2, Cout) if pool else (N, H, W, Cout)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    else:
        _chk(out, torch.float32, 'out')
        assert tuple(out.shape) == oshape
    if workspace is None:
        _lib.call('ty1_conv3x3_bn_leaky_f32', x.data_ptr(), w_ohwi.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  out.data_ptr(), N, H, W, Cin, Cout, int(bool(pool)), float(alpha), _lib.current_stream())
    else:
        if not workspace.is_cuda or not workspace.is_contiguous():
            raise ValueError('workspace must be a contiguous tensor on the GPU')
        _lib.call('ty1_conv3x3_bn_leaky_ws_f32', x.data_ptr(), w_ohwi.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  out.data_ptr(), N, H, W, Cin, Cout, int(bool(pool)), float(alpha), workspace.data_ptr(),
                  workspace.numel() * workspace.element_size(), _lib.current_stream())
    return out


def static_scratch_bytes():
    return int(_lib.lib().ty1_static_scratch_bytes())


def static_scratch_init():
    """Resolve the static scratch array's address on the current device (once per device, outside graph capture)."""
    _lib.call('ty1_static_scratch_init')


def set_reserved_cus(k):
    """Persistent grids leave k CUs free from now on (0: fill the chip).  Results do not depend on k."""
    _lib.call('ty1_set_reserved_cus', int(k))


def get_reserved_cus():
    import ctypes
    k = ctypes.c_int(0)
    _lib.call('ty1_get_reserved_cus', ctypes.byref(k))
    return int(k.value)


def conv3x3_bn_leaky_static(x, w_ohwi, scale, shift, pool, alpha, out):
    """The conv block through the entry point that allocates nothing (ty1_conv3x3_bn_leaky_static_f32): the caller
    brings every buffer, the K-split scratch is the library's static device array."""
    _chk(x, torch.float32, 'x'); _chk(w_ohwi, torch.float32, 'w')
    _chk(scale, torch.float32, 'scale'); _chk(shift, torch.float32, 'shift')
    N, H, W, Cin = x.shape
    Cout = w_ohwi.shape[0]
    assert tuple(w_ohwi.shape) == (Cout, 3, 3, Cin), w_ohwi.shape
    _chk_out(out, torch.float32, (N, H // 2, W // 2, Cout) if pool else (N, H, W, Cout), 'out')
    _lib.call('ty1_conv3x3_bn_leaky_static_f32', x.data_ptr(), w_ohwi.data_ptr(), scale.data_ptr(), shift.data_ptr(),
              out.data_ptr(), N, H, W, Cin, Cout, int(bool(pool)), float(alpha), _lib.current_stream())
    return out


def conv3x3_workspace_bytes_bf16(N, H, W, Cin, Cout, pool):
    """Scratch bytes ty1_conv3x3_bn_leaky_bf16 wants for this problem."""
    import ctypes
    n = ctypes.c_size_t(0)
    _lib.call('ty1_conv3x3_workspace_bytes_bf16', N, H, W, Cin, Cout, int(bool(pool)), ctypes.addressof(n))
    return int(n.value)


def cast_bf16(x, out=None):
    """f32 device tensor -> bf16 (round to nearest even) with the library's own kernel."""
    _chk(x, torch.float32, 'x')
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    else:
        _chk(out, torch.bfloat16, 'out')
        assert out.shape == x.shape
    _lib.call('ty1_cast_f32_to_bf16', x.data_ptr(), out.data_ptr(), x.numel(), _lib.current_stream())
    return out


def conv3x3_bn_leaky_bf16(x, w_ohwi, scale, shift, pool, alpha=0.1, out_f32=False, out=None, workspace=None):
    """The conv block of the bf16 network.  Cin == 3: x and w are f32 (first layer), otherwise bf16; the result is
    bf16, or f32 with out_f32 (unpooled layers only).  fp32 accumulation and epilogue."""
    N, H, W, Cin = x.shape
    in_dt = torch.float32 if Cin == 3 else torch.bfloat16
    _chk(x, in_dt, 'x'); _chk(w_ohwi, in_dt, 'w')
    _chk(scale, torch.float32, 'scale'); _chk(shift, torch.float32, 'shift')
    Cout = w_ohwi.shape[0]
    assert tuple(w_ohwi.shape) == (Cout, 3, 3, Cin), w_ohwi.shape
    oshape = (N, H // 2, W // 2, Cout) if pool else (N, H, W, Cout)
    out_dt = torch.float32 if out_f32 else torch.bfloat16
    if out is None:
        out = torch.empty(oshape, dtype=out_dt, device=x.device)
    else:
        _chk(out, out_dt, 'out')
        assert tuple(out.shape) == oshape
    if workspace is not None and (not workspace.is_cuda or not workspace.is_contiguous()):
        raise ValueError('workspace must be a contiguous tensor on the GPU')
    _lib.call('ty1_conv3x3_bn_leaky_bf16', x.data_ptr(), w_ohwi.data_ptr(), scale.data_ptr(), shift.data_ptr(),
              out.data_ptr(), N, H, W, Cin, Cout, int(bool(pool)), float(alpha), int(bool(out_f32)),
              workspace.data_ptr() if workspace is not None else None,
              workspace.numel() * workspace.element_size() if workspace is not None else 0, _lib.current_stream())
    return out


def conv3x3_x2_c3_bf16(x, w0, scale0, shift0, w1, scale1, shift1, alpha=0.1, out=None):
    """The first two conv blocks of the bf16 network in ONE launch (ty1_conv3x3_x2_c3_bf16): x (N,H,W,3) f32, w0 (16,3,3,3)
    f32, w1 (32,3,3,16) bf16 -> (N,H/4,W/4,32) bf16; the 16-channel tensor between the blocks stays in LDS."""
    _chk(x, torch.float32, 'x'); _chk(w0, torch.float32, 'w0'); _chk(w1, torch.bfloat16, 'w1')
    for name, t in (('scale0', scale0), ('shift0', shift0), ('scale1', scale1), ('shift1', shift1)):
        _chk(t, torch.float32, name)
    N, H, W, cin = x.shape
    if cin != 3 or tuple(w0.shape) != (16, 3, 3, 3) or tuple(w1.shape) != (32, 3, 3, 16):
        raise ValueError(f'conv3x3_x2_c3_bf16: x {tuple(x.shape)}, w0 {tuple(w0.shape)}, w1 {tuple(w1.shape)} are not the 3 -> 16 -> 32 blocks')
    if scale0.numel() != 16 or shift0.numel() != 16 or scale1.numel() != 32 or shift1.numel() != 32:
        raise ValueError('conv3x3_x2_c3_bf16: scale / shift sizes must be 16 and 32')
    oshape = (N, H // 4, W // 4, 32)
    if out is None:
        out = torch.empty(oshape, dtype=torch.bfloat16, device=x.device)
    else:
        _chk_out(out, torch.bfloat16, oshape, 'out')
    _lib.call('ty1_conv3x3_x2_c3_bf16', x.data_ptr(), w0.data_ptr(), scale0.data_ptr(), shift0.data_ptr(), w1.data_ptr(),
              scale1.data_ptr(), shift1.data_ptr(), out.data_ptr(), N, H, W, float(alpha), _lib.current_stream())
    return out


def preprocess_u8(img_u8, out_hw=(448, 448), swap_rb=False, out=None):
    """uint8 (H0,W0,3) image on the device -> fp32 (out_h,out_w,3) in [0,1]: cv2.resize INTER_LINEAR + /255
    (utils.py:11-15).  `out` may be one slot x[n] of the network's input batch."""
    _chk(img_u8, torch.uint8, 'img')
    H0, W0, c = img_u8.shape
    assert c == 3
    if out is None:
        out = torch.empty((out_hw[0], out_hw[1], 3), dtype=torch.float32, device=img_u8.device)
    else:
        _chk(out, torch.float32, 'out')
        assert tuple(out.shape) == (out_hw[0], out_hw[1], 3)
    _lib.call('ty1_preprocess_u8', img_u8.data_ptr(), H0, W0, int(bool(swap_rb)), out.data_ptr(), out_hw[0], out_hw[1],
              _lib.current_stream())
    return out


def preprocess_batch_u8(imgs_u8, out_hw=(448, 448), swap_rb=False, out=None):
    """uint8 (n,H0,W0,3) equal-sized frames on the device -> fp32 (n,out_h,out_w,3) in [0,1] in ONE launch; frame by frame
    the bits of preprocess_u8 (utils.py:11-15 over the batch loop of data_sequence.py:94-99, or over a camera's frames)."""
    _chk(imgs_u8, torch.uint8, 'imgs')
    if imgs_u8.dim() != 4 or imgs_u8.shape[3] != 3:
        raise ValueError(f'imgs: expected (n,H0,W0,3) uint8, got {tuple(imgs_u8.shape)}')
    n, H0, W0, _ = imgs_u8.shape
    shape = (n, out_hw[0], out_hw[1], 3)
    out = torch.empty(shape, dtype=torch.float32, device=imgs_u8.device) if out is None else _chk_out(out, torch.float32, shape, 'out')
    _lib.call('ty1_preprocess_batch_u8', imgs_u8.data_ptr(), n, H0, W0, int(bool(swap_rb)), out.data_ptr(), out_hw[0], out_hw[1],
              _lib.current_stream())
    return out


def pick_splits(K, n_rows, units, slots=None, chunk=DENSE_K_ALIGN):
    """Split-K factor for the Dense GEMM: the number of blocks (tiles * splits) that best fills ONE round of `slots`
    resident blocks, with at least 4 chunks (32 floats / 64 bf16 of K) per slice.  slots: 2 per CU for the 64 x 64 tiles
    of batches above 32, 1 per CU for the 32 x 128 tiles below (every extra split is one more partial for the kernel
    behind the GEMM to reduce: measured with tools/experiments/dense_splits.py, GEMM + tail at 32 images 52 -> 34 us
    with 21 splits instead of 85, at one image 49 -> 40 us)."""
    chunks = K // chunk
    tiles = ((n_rows + 63) // 64) * ((units + 63) // 64) if n_rows > 32 else (units + 127) // 128
    if slots is None:
        slots = 512 if n_rows > 32 else 256
    best, best_eff = 1, 0.0
    for s in range(1, max(1, chunks // 4) + 1):
        blocks = tiles * s
        if blocks > slots:
            break
        eff = blocks / (-(-blocks // slots) * slots)
        if eff > best_eff + 1e-9:
            best, best_eff = s, eff
    return best


def dense_splitk(x, wt, units, splits=None, part=None):
    """x (N,K), wt (rows>=units, ldk) transposed Dense weight, both f32 or both bf16 -> f32 partials (splits,N,ldz)."""
    bf = x.dtype == torch.bfloat16
    _chk(x, torch.bfloat16 if bf else torch.float32, 'x'); _chk(wt, x.dtype, 'wt')
    N, K = x.shape
    ldk = wt.shape[1]
    if splits is None:
        splits = pick_splits(K, N, units, chunk=64 if bf else DENSE_K_ALIGN)
    ldz = round_up(units, 4)
    if part is None:
        part = torch.empty((splits, N, ldz), dtype=torch.float32, device=x.device)
    else:
        _chk(part, torch.float32, 'part')
        if part.dim() != 3 or part.shape[1] != N or part.shape[2] < units:
            raise ValueError(f'part: expected (splits, {N}, >= {units}), got {tuple(part.shape)}')
        splits, _, ldz = part.shape
    if wt.shape[0] < units or ldk < K:
        raise ValueError(f'wt: expected (>= {units}, >= {K}), got {tuple(wt.shape)}')
    _lib.call('ty1_dense_splitk_bf16' if bf else 'ty1_dense_splitk_f32', x.data_ptr(), wt.data_ptr(), part.data_ptr(), N, K, units, ldk, ldz,
              splits, _lib.current_stream())
    return part


def head_activate(part, bias, S=7, B=2, C=20, out=None):
    """partials (splits,N,ldz) [+ bias (units)] -> y (N,S,S,C+5B) = [softmax C | sigmoid B | sigmoid 4B]."""
    _chk(part, torch.float32, 'part')
    if part.dim() == 2:
        part = part.unsqueeze(0)
    splits, N, ldz = part.shape
    if bias is not None:
        _chk(bias, torch.float32, 'bias')
        if bias.numel() < S * S * (C + 5 * B):
            raise ValueError(f'bias: expected {S * S * (C + 5 * B)} values, got {bias.numel()}')
    if ldz < S * S * (C + 5 * B):
        raise ValueError(f'part: last dimension {ldz} is smaller than the {S * S * (C + 5 * B)} Dense units')
    if out is None:
        out = torch.empty((N, S, S, C + 5 * B), dtype=torch.float32, device=part.device)
    else:
        _chk_out(out, torch.float32, (N, S, S, C + 5 * B), 'out')
    _lib.call('ty1_head_activate_f32', part.data_ptr(), _lib.ptr(bias), out.data_ptr(), N, S, B, C, ldz, splits,
              _lib.current_stream())
    return out


def decode_nms(pred, B=2, C=20, img_size=448, iou_thr=0.2, out=None):
    """pred (N,S,S,C+5B) f32 -> boxes (N,S*S*B,4) f64, score (N,S*S*B) f32, cls i32, keep u8."""
    _chk(pred, torch.float32, 'pred')
    if pred.dim() == 3:
        pred = pred.unsqueeze(0)
    N, S = pred.shape[0], pred.shape[1]
    if pred.shape[2] != S or pred.shape[3] != C + 5 * B:
        raise ValueError(f'pred shape {tuple(pred.shape)} does not match S,S,C+5B with B={B}, C={C}')
    slots = S * S * B
    if out is None:
        dev = pred.device
        out = (torch.empty((N, slots, 4), dtype=torch.float64, device=dev),
               torch.empty((N, slots), dtype=torch.float32, device=dev),
               torch.empty((N, slots), dtype=torch.int32, device=dev),
               torch.empty((N, slots), dtype=torch.uint8, device=dev))
    boxes, score, cls, keep = out
    _chk_out(boxes, torch.float64, (N, slots, 4), 'boxes'); _chk_out(score, torch.float32, (N, slots), 'score')
    _chk_out(cls, torch.int32, (N, slots), 'cls'); _chk_out(keep, torch.uint8, (N, slots), 'keep')
    _lib.call('ty1_decode_nms', pred.data_ptr(), N, S, B, C, int(img_size), float(iou_thr),
              boxes.data_ptr(), score.data_ptr(), cls.data_ptr(), keep.data_ptr(), _lib.current_stream())
    return boxes, score, cls, keep


def head_decode_nms(part, bias, S=7, B=2, C=20, img_size=448, iou_thr=0.2, y=None, out=None):
    """Dense partials (splits,N,ldz) [+ bias] -> detections in ONE launch (head_activate + decode_nms, same bits).
    y: optional (N,S,S,C+5B) f32 tensor that also receives the head tensor."""
    _chk(part, torch.float32, 'part')
    if part.dim() == 2:
        part = part.unsqueeze(0)
    splits, N, ldz = part.shape
    units, slots = S * S * (C + 5 * B), S * S * B
    if ldz < units:
        raise ValueError(f'part: last dimension {ldz} is smaller than the {units} Dense units')
    if bias is not None:
        _chk(bias, torch.float32, 'bias')
        if bias.numel() < units:
            raise ValueError(f'bias: expected {units} values, got {bias.numel()}')
    if y is not None:
        _chk_out(y, torch.float32, (N, S, S, C + 5 * B), 'y')
    if out is None:
        dev = part.device
        out = (torch.empty((N, slots, 4), dtype=torch.float64, device=dev), torch.empty((N, slots), dtype=torch.float32, device=dev),
               torch.empty((N, slots), dtype=torch.int32, device=dev), torch.empty((N, slots), dtype=torch.uint8, device=dev))
    boxes, score, cls, keep = out
    _chk_out(boxes, torch.float64, (N, slots, 4), 'boxes'); _chk_out(score, torch.float32, (N, slots), 'score')
    _chk_out(cls, torch.int32, (N, slots), 'cls'); _chk_out(keep, torch.uint8, (N, slots), 'keep')
    _lib.call('ty1_head_decode_nms_f32', part.data_ptr(), _lib.ptr(bias), _lib.ptr(y), N, S, B, C, ldz, splits, int(img_size),
              float(iou_thr), boxes.data_ptr(), score.data_ptr(), cls.data_ptr(), keep.data_ptr(), _lib.current_stream())
    return boxes, score, cls, keep


def pack_detections(boxes, score, cls, keep, out=None):
    """-> (N, slots, 6) f64 records x0,y0,x1,y1,score,class (-1 where not kept): the unit of the detection gather."""
    _chk(boxes, torch.float64, 'boxes'); _chk(score, torch.float32, 'score'); _chk(cls, torch.int32, 'cls'); _chk(keep, torch.uint8, 'keep')
    N, slots = score.shape
    for name, t, shp in (('boxes', boxes, (N, slots, 4)), ('cls', cls, (N, slots)), ('keep', keep, (N, slots))):
        if tuple(t.shape) != shp:
            raise ValueError(f'{name}: expected shape {shp}, got {tuple(t.shape)}')
    if out is None:
        out = torch.empty((N, slots, 6), dtype=torch.float64, device=boxes.device)
    else:
        _chk_out(out, torch.float64, (N, slots, 6), 'out')
    _lib.call('ty1_pack_detections', boxes.data_ptr(), score.data_ptr(), cls.data_ptr(), keep.data_ptr(), out.data_ptr(),
              N * slots, _lib.current_stream())
    return out


def yolo_head(feats, img_size=448):
    _chk(feats, torch.float32, 'feats')
    if feats.dim() != 5 or feats.shape[-1] != 4 or feats.shape[1] != feats.shape[2]:
        raise ValueError(f'feats must be (N,S,S,B,4), got {tuple(feats.shape)}')
    N, S, _, B, _ = feats.shape
    xy = torch.empty((N, S, S, B, 2), dtype=torch.float32, device=feats.device)
    wh = torch.empty_like(xy)
    _lib.call('ty1_yolo_head_f32', feats.data_ptr(), xy.data_ptr(), wh.data_ptr(), N, S, B, int(img_size),
              _lib.current_stream())
    return xy, wh


def iou(pred_mins, pred_maxes, true_mins, true_maxes):
    shape = torch.broadcast_shapes(pred_mins.shape, pred_maxes.shape, true_mins.shape, true_maxes.shape)
    ops = [_chk(t.expand(shape).contiguous(), torch.float32, 'iou operand')
           for t in (pred_mins, pred_maxes, true_mins, true_maxes)]
    if shape[-1] != 2:
        raise ValueError('last axis must be (x, y)')
    out = torch.empty(shape[:-1], dtype=torch.float32, device=ops[0].device)
    _lib.call('ty1_iou_f32', *(o.data_ptr() for o in ops), out.data_ptr(), out.numel(), _lib.current_stream())
    return out


def yolo_head_np(feats, img_size=448):
    """NumPy twin: feats (...,S,S,B,4) f32 -> xy float64, wh float32 (tiny_yolov1.py:36-55)."""
    _chk(feats, torch.float32, 'feats')
    shape = feats.shape
    S, B = shape[-3], shape[-2]
    if shape[-1] != 4 or shape[-4] != S:
        raise ValueError(f'feats must be (...,S,S,B,4), got {tuple(shape)}')
    N = feats.numel() // (S * S * B * 4)
    xy = torch.empty(shape[:-1] + (2,), dtype=torch.float64, device=feats.device)
    wh = torch.empty(shape[:-1] + (2,), dtype=torch.float32, device=feats.device)
    _lib.call('ty1_yolo_head_np_f64', feats.data_ptr(), xy.data_ptr(), wh.data_ptr(), N, S, B, int(img_size),
              _lib.current_stream())
    return xy, wh


def iou_f64(pred_mins, pred_maxes, true_mins, true_maxes):
    shape = torch.broadcast_shapes(pred_mins.shape, pred_maxes.shape, true_mins.shape, true_maxes.shape)
    ops_ = [_chk(t.to(torch.float64).expand(shape).contiguous(), torch.float64, 'iou operand')
            for t in (pred_mins, pred_maxes, true_mins, true_maxes)]
    out = torch.empty(shape[:-1], dtype=torch.float64, device=ops_[0].device)
    _lib.call('ty1_iou_f64', *(o.data_ptr() for o in ops_), out.data_ptr(), out.numel(), _lib.current_stream())
    return out


def yolo_loss_fwd_bwd(y_true, y_pred, B=2, C=20, img_size=448, want_grad=True, loss_out=None):
    """-> loss (1,) f32 batch SUM, loss_per_image (N,), grad (N,S,S,C+5B) or None.
    loss_out: optional (1,) f32 device tensor the batch loss is written to (e.g. a slot of the gradient bucket)."""
    _chk(y_true, torch.float32, 'y_true'); _chk(y_pred, torch.float32, 'y_pred')
    N, S = y_pred.shape[0], y_pred.shape[1]
    if tuple(y_true.shape) != (N, S, S, C + 5) or tuple(y_pred.shape) != (N, S, S, C + 5 * B):
        raise ValueError(f'shapes {tuple(y_true.shape)} / {tuple(y_pred.shape)} do not match B={B}, C={C}')
    dev = y_pred.device
    per = torch.empty((max(N, 1),), dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev) if loss_out is None else _chk_out(loss_out, torch.float32, (1,), 'loss_out')
    grad = torch.empty_like(y_pred) if want_grad else None
    _lib.call('ty1_yolo_loss_fwd_bwd', y_true.data_ptr(), y_pred.data_ptr(), N, S, B, C, int(img_size),
              per.data_ptr(), loss.data_ptr(), _lib.ptr(grad), _lib.current_stream())
    return loss, per[:N], grad


def train_tail(part, bias, y_true, ldz, S=7, B=2, C=20, img_size=448, y=None, dz=None, loss_out=None):
    """Dense partials (splits,N,ldz_part) [+ bias], labels (N,S,S,C+5) -> loss (1,) batch SUM, loss_per_image (N,), dz (N,ldz)
    in ONE launch + the batch sum (head_activate + yolo_loss_fwd_bwd + head_activate_bwd, same bits).
    y: optional (N,S,S,C+5B) tensor that also receives the head tensor; loss_out: optional (1,) slot for the loss."""
    _chk(part, torch.float32, 'part'); _chk(y_true, torch.float32, 'y_true')
    if part.dim() == 2:
        part = part.unsqueeze(0)
    splits, N, ldz_part = part.shape
    units = S * S * (C + 5 * B)
    if ldz_part < units or ldz < units:
        raise ValueError(f'train_tail: leading dimensions {ldz_part} / {ldz} are smaller than the {units} Dense units')
    if tuple(y_true.shape) != (N, S, S, C + 5):
        raise ValueError(f'y_true {tuple(y_true.shape)} does not match N={N}, S={S}, C={C}')
    if bias is not None:
        _chk(bias, torch.float32, 'bias')
        if bias.numel() < units:
            raise ValueError(f'bias: expected {units} values, got {bias.numel()}')
    if y is not None:
        _chk_out(y, torch.float32, (N, S, S, C + 5 * B), 'y')
    dev = part.device
    dz = torch.empty((N, ldz), dtype=torch.float32, device=dev) if dz is None else _chk_out(dz, torch.float32, (N, ldz), 'dz')
    per = torch.empty((max(N, 1),), dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev) if loss_out is None else _chk_out(loss_out, torch.float32, (1,), 'loss_out')
    _lib.call('ty1_train_tail_f32', part.data_ptr(), _lib.ptr(bias), y_true.data_ptr(), _lib.ptr(y), dz.data_ptr(), N, S, B, C,
              ldz_part, splits, int(img_size), ldz, per.data_ptr(), loss.data_ptr(), _lib.current_stream())
    return loss, per[:N], dz


def head_activate_bwd(y, grad_y, ldz, B=2, C=20, out=None):
    _chk(y, torch.float32, 'y'); _chk(grad_y, torch.float32, 'grad_y')
    N, S = y.shape[0], y.shape[1]
    if tuple(grad_y.shape) != tuple(y.shape) or y.shape[-1] != C + 5 * B or ldz < y[0].numel():
        raise ValueError(f'head_activate_bwd: y {tuple(y.shape)}, grad_y {tuple(grad_y.shape)}, ldz {ldz} do not fit B={B}, C={C}')
    if out is None:
        out = torch.empty((N, ldz), dtype=torch.float32, device=y.device)
    else:
        _chk_out(out, torch.float32, (N, ldz), 'out')
    _lib.call('ty1_head_activate_bwd_f32', y.data_ptr(), grad_y.data_ptr(), out.data_ptr(), N, S, B, C, ldz,
              _lib.current_stream())
    return out


def dense_wgrad(dz, x, units, dwt, db):
    _chk(dz, torch.float32, 'dz'); _chk(x, torch.float32, 'x'); _chk(dwt, torch.float32, 'dwt'); _chk(db, torch.float32, 'db')
    N, K = x.shape
    if dz.shape[0] != N or dz.shape[1] < units or dwt.shape[0] < units or dwt.shape[1] < K or db.numel() < units:
        raise ValueError(f'dense_wgrad: dz {tuple(dz.shape)}, x {tuple(x.shape)}, dwt {tuple(dwt.shape)}, db {tuple(db.shape)} '
                         f'do not fit units={units}')
    _lib.call('ty1_dense_wgrad_f32', dz.data_ptr(), x.data_ptr(), dwt.data_ptr(), db.data_ptr(), N, K, units,
              dwt.shape[1], dz.shape[1], _lib.current_stream())
    return dwt, db


def adam_step(p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
    for name, a in (('p', p), ('g', g), ('m', m), ('v', v)):
        _chk(a, torch.float32, name)
    _lib.call('ty1_adam_step_f32', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
              float(lr), float(beta1), float(beta2), float(eps), int(t), _lib.current_stream())


class KernelTimer:
    """HIP-event timing of every library launch on its own stream (bench.py roofline leg)."""

    def __init__(self, only=None, every=1):
        """only: time just the launches whose name starts with this prefix (two event records per launch cost a few
        per cent of a step when every launch is bracketed); every: of those, only every n-th one."""
        self.only, self.every = only, int(every)

    def __enter__(self):
        _lib.call('ty1_prof_reset')
        _lib.call('ty1_prof_filter', self.only.encode() if self.only else b'')
        _lib.call('ty1_prof_stride', self.every)
        _lib.call('ty1_prof_enable', 1)
        return self

    def __exit__(self, *exc):
        _lib.call('ty1_prof_enable', 0)
        _lib.call('ty1_prof_filter', b'')
        _lib.call('ty1_prof_stride', 1)

    @staticmethod
    def table():
        import ctypes
        buf = ctypes.create_string_buffer(16384)
        _lib.call('ty1_prof_dump', buf, len(buf))
        rows = {}
        for line in buf.value.decode().splitlines():
            name, n, ms, flops, nbytes = line.split()
            rows[name] = dict(launches=int(n), ms=float(ms), flops=float(flops), bytes=float(nbytes))
        return rows
