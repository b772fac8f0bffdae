"""ORACLE (test infrastructure only): restatement of OpenCV's 8-bit INTER_LINEAR resize, the arithmetic behind
`cv2.resize(image, (w, h))` in /root/reference/utils.py:13, followed by `image / 255.` (:15).

PARITY UNPINNED: cv2 is not installed in this image and the reference holds no resized image, so nothing from the
reference pins this file.  It restates OpenCV's published generic (non-IPP) code path as two passes, in OpenCV's own
integer widths:
  coordinates   f = float32((d + 0.5) * (n_in / n_out) - 0.5); s = floor(f); frac = f - s        (float32)
                x axis only: s < 0 -> (s, frac) = (0, 0);  s >= n_in - 1 -> (n_in - 1, 0)
                y axis: rows s and s + 1 are clamped into the image, frac is kept
  coefficients  short(round_half_even((1 - frac) * 2048)), short(round_half_even(frac * 2048))      (11 bits)
  horizontal    R[x] = S[x0] * a0 + S[x1] * a1                                      (int32, <= 255 * 2048)
  vertical      D = (((b0 * (R0 >> 4)) >> 16) + ((b1 * (R1 >> 4)) >> 16) + 2) >> 2   (truncating, as OpenCV's uchar path)
  exact 2x2 downscale: OpenCV switches INTER_LINEAR to its fast area path: (a + b + c + d + 2) >> 2
"""
import numpy as np


def _axis(n_out, n_in, zero_frac_at_edges):
    scale = np.float64(n_in) / np.float64(n_out)
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f)
    frac = (f - s).astype(np.float32)
    s = s.astype(np.int64)
    if zero_frac_at_edges:
        frac = np.where(s < 0, np.float32(0), frac)
        frac = np.where(s >= n_in - 1, np.float32(0), frac)
    i0 = np.clip(s, 0, n_in - 1)
    i1 = np.clip(s + 1, 0, n_in - 1)
    c0 = np.rint((np.float32(1) - frac) * np.float32(2048)).astype(np.int32)
    c1 = np.rint(frac * np.float32(2048)).astype(np.int32)
    return i0, i1, c0, c1


def resize_linear_u8_ref(img, out_w, out_h):
    img = np.asarray(img, np.uint8)
    h, w = img.shape[:2]
    if h == 2 * out_h and w == 2 * out_w:
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _axis(out_w, w, True)
    y0, y1, b0, b1 = _axis(out_h, h, False)
    src = img.astype(np.int32)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # horizontal pass, every source row
    top, bot = rows[y0] >> 4, rows[y1] >> 4                                          # vertical pass
    out = (((b0[:, None, None] * top) >> 16) + ((b1[:, None, None] * bot) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def load_img_ref(image_rgb_u8, shape=(448, 448, 3)):
    """utils.py:11-15 after decoding: resize to (shape[1], shape[0]) and divide by 255. (float64)."""
    return resize_linear_u8_ref(image_rgb_u8, shape[1], shape[0]).reshape(shape) / 255.
