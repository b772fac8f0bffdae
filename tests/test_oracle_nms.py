"""The oracle for decode+threshold+NMS against the reference-generated golden vectors
(tests/golden/make_golden.py ran /root/reference/tiny_yolov1.py:_main and
faces/testFace.py:_main).  CPU only."""
import numpy as np
import pytest

import oracle
from oracle.decode_nms_np import decode_nms_one, yolo_head_np, x_y_w_h_to_min_max, iou_np
from conftest import load_golden

CASES = [('nms_c20.npz', 20, 448), ('nms_c1.npz', 1, 224)]


def _kept_rects(boxes, keep):
    idx = np.nonzero(keep)[0]                       # raster order (row, col, box)
    return idx, np.array([[int(v) for v in boxes[i]] for i in idx], np.int64).reshape(-1, 4)


def _same_f64(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64)) or \
        (a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b))
         and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]))


@pytest.mark.parametrize('name,C,img', CASES)
def test_c_oracle_matches_reference_goldens(name, C, img):
    g = load_golden(name)
    pred = g['prediction']
    out = oracle.decode_nms(pred, B=2, C=C, img_size=img, want_trace=True)
    toff = 0
    for i in range(pred.shape[0]):
        idx, rects = _kept_rects(out['boxes'][i], out['keep'][i])
        want = g['rects'][g['rect_off'][i]:g['rect_off'][i + 1]]
        assert np.array_equal(rects, want), f'case {i}: kept rectangles differ'
        if C == 20:
            assert np.array_equal(out['cls'][i][idx], g['labels'][g['rect_off'][i]:g['rect_off'][i + 1]])
        else:
            got = [str(np.float32(v)) for v in out['score'][i][idx]]
            assert got == list(g['conf_str'][g['rect_off'][i]:g['rect_off'][i + 1]])
        n = g['iou_off'][i + 1] - g['iou_off'][i]
        assert _same_f64(out['trace'][toff:toff + n], g['iou_trace'][g['iou_off'][i]:g['iou_off'][i + 1]]), \
            f'case {i}: IoU trace differs'
        toff += n
    assert toff == len(out['trace'])


@pytest.mark.parametrize('name,C,img', CASES)
def test_numpy_restatement_agrees_with_c_oracle(name, C, img):
    g = load_golden(name)
    pred = g['prediction']
    out = oracle.decode_nms(pred, B=2, C=C, img_size=img, want_trace=True)
    toff = 0
    for i in range(pred.shape[0]):
        boxes, score, cls, keep, trace = decode_nms_one(pred[i], 2, C, img)
        assert np.array_equal(keep.astype(np.uint8), out['keep'][i])
        assert _same_f64(boxes, out['boxes'][i])
        assert np.array_equal(score.view(np.uint32), out['score'][i].view(np.uint32))
        assert np.array_equal(cls, out['cls'][i])
        assert _same_f64(trace, out['trace'][toff:toff + len(trace)])
        toff += len(trace)


def test_numpy_twins_match_reference():
    g = load_golden('head_np.npz')
    xy, wh = yolo_head_np(g['feats'])
    assert xy.dtype == np.float64 and wh.dtype == np.float32
    assert np.array_equal(xy, g['xy']) and np.array_equal(wh, g['wh'])
    xy2, wh2 = yolo_head_np(g['feats'], 224)
    assert np.array_equal(xy2, g['xy224']) and np.array_equal(wh2, g['wh224'])
    mn, mx = x_y_w_h_to_min_max(xy, wh)
    assert np.array_equal(mn, g['xy_min']) and np.array_equal(mx, g['xy_max'])
    assert np.array_equal(iou_np(g['a_min'], g['a_max'], g['b_min'], g['b_max']), g['iou'])


def test_oracle_edge_cases():
    # empty batch, all-zero image (nothing kept), NaN confidence (loop never starts)
    out = oracle.decode_nms(np.zeros((0, 7, 7, 30), np.float32))
    assert out['keep'].shape == (0, 98)
    out = oracle.decode_nms(np.zeros((1, 7, 7, 30), np.float32))
    assert out['keep'].sum() == 0 and not out['boxes'].any()
    p = np.random.default_rng(0).random((1, 7, 7, 30)).astype(np.float32)
    p[0, 3, 3, 20] = np.nan
    out = oracle.decode_nms(p)
    b, s, c, k, t = decode_nms_one(p[0])
    assert np.array_equal(out['keep'][0], k.astype(np.uint8))
