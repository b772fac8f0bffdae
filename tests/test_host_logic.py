"""CPU tests of the host-side mirror of the reference interface (no GPU compute)."""
import os

import numpy as np
import pytest

from yolo_amd import darknet, params as P
from yolo_amd import keras_facade as kf
from yolo_amd.models.model_tiny_yolov1 import model_tiny_YOLOv1, original_backbone_network, ReshapeYOLO
from yolo_amd.utils import X_Y_W_H_To_Min_Max, resize_bilinear_u8, load_img
from yolo_amd import parallel


def test_model_graph_matches_reference_structure():
    outputs, inputs = model_tiny_YOLOv1()
    assert inputs.shape[1:] == (448, 448, 3) and outputs.shape == (None, 7, 7, 30)
    m = kf.Model(inputs=inputs, outputs=outputs)
    assert m.conv_specs == [(ci, co, i < 6) for i, (ci, co) in enumerate(P.CONV_CHANNELS)]
    assert m.features == 12544 and (m.S, m.B, m.C) == (7, 2, 20)
    # parameter count of the reference model: 27,099,950 incl. the 4 BN vectors per layer
    n = sum(v.size for v in m.params.values())
    assert n == 27_099_950
    assert m.params['dense_w'].size + m.params['dense_b'].size == 18_441_150
    # layer names as the reference wrote them (by_name loading)
    names = m._keras_names()
    assert 'convolution_0/kernel' in names and 'bn_convolution_7/moving_variance' in names and 'connected_0/bias' in names
    rs = ReshapeYOLO((7, 7, 30))
    assert rs.compute_output_shape((None, 1470)) == (None, 7, 7, 30) and rs.get_config()['target_shape'] == (7, 7, 30)


def test_unsupported_graphs_are_rejected_loudly():
    inp = kf.Input((448, 448, 3))
    x = kf.Conv2D(16, (5, 5), padding='same', use_bias=False, trainable=False, name='c')(inp)
    x = kf.BatchNormalization(trainable=False)(x)
    x = kf.LeakyReLU(alpha=0.1)(x)
    x = kf.Flatten()(x)
    x = kf.Dense(1470, name='d')(x)
    with pytest.raises(ValueError, match='3x3'):
        kf.Model(inputs=inp, outputs=ReshapeYOLO((7, 7, 30))(x))
    x = original_backbone_network(inp)
    with pytest.raises(ValueError, match='ReshapeYOLO'):
        kf.Model(inputs=inp, outputs=kf.Dense(1470, name='d')(x))


def test_weights_by_name_roundtrip(tmp_path):
    outputs, inputs = model_tiny_YOLOv1(C=1)
    m = kf.Model(inputs=inputs, outputs=outputs, seed=3)
    path = str(tmp_path / 'w.npz')
    m.save_weights(path)
    m2 = kf.Model(inputs=inputs, outputs=outputs, seed=4)
    loaded = m2.load_weights(path, by_name=True)
    assert len(loaded) == 8 * 5 + 2
    assert all(np.array_equal(m.params[k], m2.params[k]) for k in m.params)
    # converter-style names match only connected_0
    z = dict(np.load(path))
    renamed = {k.replace('convolution_', 'convolutional_'): v for k, v in z.items()}
    np.savez(str(tmp_path / 'r.npz'), **renamed)
    m3 = kf.Model(inputs=inputs, outputs=outputs, seed=5)
    assert sorted(m3.load_weights(str(tmp_path / 'r.npz'))) == ['connected_0/bias', 'connected_0/kernel']
    with pytest.raises((ImportError, NotImplementedError)):
        m3.load_weights('tiny-yolov1.hdf5')


def test_darknet_weight_file_layout(tmp_path):
    # a scaled-down stack with the same section pattern as cfg/yolov1-tiny.cfg, written byte by byte
    cfg = tmp_path / 't.cfg'
    cfg.write_text('[net]\nheight=16\nwidth=16\nchannels=3\n'
                   '[convolutional]\nbatch_normalize=1\nfilters=4\nsize=3\nstride=1\npad=1\nactivation=leaky\n'
                   '[maxpool]\nsize=2\nstride=2\n'
                   '[convolutional]\nbatch_normalize=1\nfilters=8\nsize=3\nstride=1\npad=1\nactivation=leaky\n'
                   '[connected]\noutput=10\nactivation=linear\n[detection]\nclasses=20\n')
    rng = np.random.default_rng(0)
    F1, F2, feats = 4, 8, 8 * 8 * 8
    pieces = [np.array([0, 1, 0, 0], np.int32).tobytes()]
    raw = {}
    for i, (cin, f) in enumerate([(3, F1), (F1, F2)]):
        raw[f'b{i}'], raw[f'g{i}'], raw[f'm{i}'], raw[f'v{i}'] = (rng.standard_normal(f).astype(np.float32) for _ in range(4))
        raw[f'w{i}'] = rng.standard_normal((f, cin, 3, 3)).astype(np.float32)
        pieces += [raw[f'b{i}'].tobytes(), raw[f'g{i}'].tobytes(), raw[f'm{i}'].tobytes(), raw[f'v{i}'].tobytes(), raw[f'w{i}'].tobytes()]
    raw['fb'] = rng.standard_normal(10).astype(np.float32)
    raw['fw'] = rng.standard_normal((10, feats)).astype(np.float32)
    pieces += [raw['fb'].tobytes(), raw['fw'].tobytes()]
    wpath = tmp_path / 't.weights'
    wpath.write_bytes(b''.join(pieces))
    p, size, specs = darknet.load_darknet_weights(str(cfg), str(wpath))
    assert size == 16 and specs == [(3, 4, True), (4, 8, False)]
    for i in range(2):
        assert np.array_equal(p[f'bn{i}_beta'], raw[f'b{i}']) and np.array_equal(p[f'bn{i}_gamma'], raw[f'g{i}'])
        assert np.array_equal(p[f'bn{i}_mean'], raw[f'm{i}']) and np.array_equal(p[f'bn{i}_var'], raw[f'v{i}'])
        assert np.array_equal(p[f'conv{i}_w'], np.transpose(raw[f'w{i}'], (2, 3, 1, 0)))      # convert.py:144
    assert np.array_equal(p['dense_w'], raw['fw'].T) and np.array_equal(p['dense_b'], raw['fb'])   # convert.py:228
    assert p['floats_read'] * 4 + 16 == os.path.getsize(wpath)
    out = tmp_path / 'rt.weights'
    darknet.write_darknet_weights(str(out), p, specs)
    assert out.read_bytes() == wpath.read_bytes()
    # the real cfg parses to the reference's layer list
    size, specs, units = darknet.conv_specs_from_cfg(darknet.parse_cfg(os.path.join(os.path.dirname(__file__), 'golden', 'yolov1-tiny.cfg'))) \
        if os.path.exists(os.path.join(os.path.dirname(__file__), 'golden', 'yolov1-tiny.cfg')) else (448, None, 1470)
    assert size == 448 and units == 1470


def test_device_layout_helpers():
    rng = np.random.default_rng(1)
    w = rng.standard_normal((3, 3, 5, 7)).astype(np.float32)
    o = P.hwio_to_ohwi(w)
    assert o.shape == (7, 3, 3, 5) and o[2, 1, 0, 4] == w[1, 0, 4, 2]
    d = rng.standard_normal((64, 45)).astype(np.float32)
    wt = P.dense_to_device_layout(d)
    assert wt.shape == (64, 64) and np.array_equal(wt[:45], d.T) and not wt[45:].any()
    assert np.array_equal(P.dense_from_device_layout(wt, 45), d)


def test_utils_twins(tmp_path):
    xy, wh = np.array([[10., 20.]]), np.array([[4., 6.]], np.float32)
    mn, mx = X_Y_W_H_To_Min_Max(xy, wh)
    assert np.array_equal(mn, [[8., 17.]]) and np.array_equal(mx, [[12., 23.]])
    # INTER_LINEAR definition: identity at equal size, exact 2x2 box average on a 2x downscale
    img = np.random.default_rng(2).integers(0, 256, (8, 12, 3), dtype=np.uint8)
    assert np.array_equal(resize_bilinear_u8(img, 12, 8), img)
    half = resize_bilinear_u8(img, 6, 4).astype(np.int64)
    box = img.reshape(4, 2, 6, 2, 3).astype(np.int64).sum((1, 3))
    assert np.abs(half * 4 - box).max() <= 2
    from PIL import Image
    Image.fromarray(img).save(str(tmp_path / 'a.png'))
    im, h, w = load_img(str(tmp_path / 'a.png'), (448, 448, 3))
    assert im.shape == (448, 448, 3) and im.dtype == np.float64 and (h, w) == (8, 12) and 0 <= im.min() and im.max() <= 1


def test_shard_bounds_cover_the_batch():
    for n, ws in [(1024, 8), (64, 1), (10, 4), (3, 8)]:
        spans = [parallel.shard_bounds(n, r, ws) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


# ------------------------------------------------------------------------------- next rows: labels, consumer
def test_label_encoder_vs_reference_goldens():
    """tests/golden/labels_*.npz were written by the reference's own SequenceData.read / SequenceForAirplanes.read
    (tests/golden/make_golden_labels.py): bit-exact."""
    from conftest import load_golden
    from yolo_amd.data_sequence import encode_boxes, parse_annotation_line
    g = load_golden('labels_c20.npz')
    for line, (h, w), want in zip(g['lines'], g['hw'], g['label_matrix']):
        _, boxes = parse_annotation_line(str(line))
        got = encode_boxes(boxes, int(h), int(w), 7, 20)
        assert got.dtype == np.float64 and np.array_equal(got, want)
    g = load_golden('labels_c1.npz')
    for i, ((h, w), want) in enumerate(zip(g['hw'], g['label_matrix'])):
        boxes = g['boxes'][g['box_off'][i]:g['box_off'][i + 1]]
        assert np.array_equal(encode_boxes(boxes, int(h), int(w), 7, 1), want)


def test_sequence_data_protocol(tmp_path):
    from PIL import Image
    from yolo_amd.data_sequence import SequenceData
    rng = np.random.default_rng(5)
    lines = []
    for i in range(5):
        Image.fromarray(rng.integers(0, 256, (30 + i, 40, 3), dtype=np.uint8)).save(str(tmp_path / f'{i}.png'))
        lines.append(f'{tmp_path}/{i}.png 4,5,20,25,{i} 10,10,30,{29 + i},7\n')
    (tmp_path / '2007_train.txt').write_text(''.join(lines))
    seq = SequenceData('train', str(tmp_path), (448, 448, 3), 2, shuffle=False)
    assert len(seq) == 3
    X, y = seq[0]
    assert X.shape == (2, 448, 448, 3) and y.shape == (2, 7, 7, 25) and y[..., 24].sum() == 4
    X, y = seq[2]
    assert X.shape[0] == 1
    seq.on_epoch_end()


def test_detection_records():
    from yolo_amd.detections import to_records, voc_result_lines
    boxes = np.array([[10.9, 20.2, 100.7, 200.99], [0, 0, 1, 1], [-3.5, 5.5, 447.9, 448.0]])
    recs = to_records(boxes, np.array([0.9, 0.1, 0.5], np.float32), np.array([14, 3, 6]), np.array([1, 0, 1], np.uint8),
                      image_hw=(224, 896))
    assert [r['slot'] for r in recs] == [0, 2] and [r['name'] for r in recs] == ['person', 'car']
    assert recs[0]['corners'] == (10, 20, 100, 200) and recs[1]['corners'] == (-3, 5, 447, 448)     # int() truncates toward 0
    assert recs[0]['box'] == (10.9 * 2, 20.2 * 0.5, 100.7 * 2, 200.99 * 0.5)
    lines = voc_result_lines('000012', recs)
    assert lines['person'] == ['000012 0.900000 21.8 10.1 201.4 100.5']
