"""Drop-in for /root/reference/utils.py."""
import numpy as np


def X_Y_W_H_To_Min_Max(xy, wh):
    """utils.py:19-23 (works on NumPy arrays and torch tensors alike)."""
    xy_min = xy - wh / 2
    xy_max = xy + wh / 2
    return xy_min, xy_max


def load_img(path, shape=(448, 448, 3)):
    """utils.py:5-16: read -> RGB -> bilinear resize -> /255. (float64), returns (image, h, w).
    OpenCV is not available here, so decoding uses PIL and the resize is OpenCV's 8-bit INTER_LINEAR
    arithmetic restated in NumPy (resize_bilinear_u8).  PARITY UNPINNED: cv2 is absent and the reference
    holds no resized image."""
    from PIL import Image
    if type(shape) is not tuple:
        shape = tuple(shape)
    if len(shape) < 3:
        shape = shape + (3,)
    image = np.asarray(Image.open(path).convert('RGB'))
    image_h, image_w = image.shape[0:2]
    image = resize_bilinear_u8(image, shape[1], shape[0])     # cv2.resize takes (width, height)
    image = np.reshape(image, shape)
    image = image / 255.
    return image, image_h, image_w


def load_img_device(path_or_array, shape=(448, 448, 3), device='cuda:0', out=None, bgr=False):
    """load_img with the resize and normalisation on the GPU (ty1_preprocess_u8): decode on the host (PIL) or take a
    decoded uint8 HxWx3 array, upload the uint8 pixels (3 bytes per source pixel instead of 12 per output pixel),
    resize + /255 into `out` (e.g. one slot of the input batch).  -> (fp32 (H,W,3) device tensor, image_h, image_w);
    equals np.float32(load_img(...)[0]) bit for bit."""
    import torch
    from . import ops
    if isinstance(path_or_array, str):
        from PIL import Image
        image = np.array(Image.open(path_or_array).convert('RGB'))      # (a writable copy: torch.from_numpy wants one)
    else:
        image = np.ascontiguousarray(path_or_array, np.uint8)
    image_h, image_w = image.shape[0:2]
    d = torch.from_numpy(image).to(device)
    return ops.preprocess_u8(d, (shape[0], shape[1]), swap_rb=bgr, out=out), image_h, image_w


def _linear_taps(n_out, n_in, x_axis):
    """Source indices and 11-bit integer weights of one axis (OpenCV: float32 coordinates, half-pixel centres)."""
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (np.float64(n_in) / np.float64(n_out)) - 0.5).astype(np.float32)
    lo = np.floor(f)
    t = (f - lo).astype(np.float32)
    lo = lo.astype(np.int64)
    if x_axis:          # the x pass snaps out-of-range taps to the border pixel; the y pass clamps rows and keeps t
        t[(lo < 0) | (lo >= n_in - 1)] = 0
    w_hi = np.rint(t * np.float32(2048)).astype(np.int32)
    w_lo = np.rint((np.float32(1) - t) * np.float32(2048)).astype(np.int32)
    return np.clip(lo, 0, n_in - 1), np.clip(lo + 1, 0, n_in - 1), w_lo, w_hi


def resize_bilinear_u8(img, out_w, out_h):
    """cv2.resize(img, (out_w, out_h)) INTER_LINEAR on uint8 as OpenCV's generic 8-bit path computes it: a horizontal
    pass into int32 with 11-bit weights, a vertical pass that truncates (>> 4, >> 16) and rounds once (+2 >> 2); an
    exact 2x downscale takes OpenCV's 2x2 area average.  PARITY UNPINNED (cv2 absent): may differ from a given cv2
    build (IPP, SIMD variants) by one count.  ty1_preprocess_u8 is this arithmetic on the GPU."""
    h, w = img.shape[:2]
    px = img.astype(np.int32)
    if h == 2 * out_h and w == 2 * out_w:
        return ((px[0::2, 0::2] + px[0::2, 1::2] + px[1::2, 0::2] + px[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xl, xr, wl, wr = _linear_taps(out_w, w, True)
    yt, yb, wt, wb = _linear_taps(out_h, h, False)
    wl, wr, wt, wb = wl[None, :, None], wr[None, :, None], wt[:, None, None], wb[:, None, None]
    top = (px[yt][:, xl] * wl + px[yt][:, xr] * wr) >> 4
    bot = (px[yb][:, xl] * wl + px[yb][:, xr] * wr) >> 4
    return ((((wt * top) >> 16) + ((wb * bot) >> 16) + 2) >> 2).astype(np.uint8)
