"""Drop-in for /root/reference/utils.py."""
import numpy as np


def X_Y_W_H_To_Min_Max(xy, wh):
    """utils.py:19-23 (works on NumPy arrays and torch tensors alike)."""
    xy_min = xy - wh / 2
    xy_max = xy + wh / 2
    return xy_min, xy_max


def load_img(path, shape=(448, 448, 3)):
    """utils.py:5-16: read -> RGB -> bilinear resize -> /255. (float64), returns (image, h, w).
    OpenCV is not available here, so decoding uses PIL and the resize is OpenCV's INTER_LINEAR
    definition (half-pixel centres, no antialiasing) restated in NumPy.  PARITY UNPINNED: cv2 is
    absent, so this is checked against the stated definition only."""
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


def resize_bilinear_u8(img, out_w, out_h):
    """cv2.resize(img, (out_w, out_h)) INTER_LINEAR on uint8: half-pixel centres, edge clamp,
    11-bit fixed-point coefficients and round-half-up as OpenCV's 8-bit path does."""
    h, w = img.shape[:2]

    def coeffs(n_out, n_in):
        f = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(f).astype(np.int64)
        frac = f - i0
        frac = np.where(i0 < 0, 0.0, frac)
        i0c = np.clip(i0, 0, n_in - 1)
        i1c = np.clip(i0 + 1, 0, n_in - 1)
        frac = np.where(i0 >= n_in - 1, 0.0, frac)
        c1 = np.rint(frac * 2048).astype(np.int64)
        return i0c, i1c, 2048 - c1, c1

    y0, y1, wy0, wy1 = coeffs(out_h, h)
    x0, x1, wx0, wx1 = coeffs(out_w, w)
    src = img.astype(np.int64)
    rows = src[:, x0] * wx0[None, :, None] + src[:, x1] * wx1[None, :, None]
    out = rows[y0] * wy0[:, None, None] + rows[y1] * wy1[:, None, None]
    return ((out + (1 << 21)) >> 22).astype(np.uint8)
