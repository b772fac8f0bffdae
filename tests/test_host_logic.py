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


REFERENCE_CFG = '/root/reference/cfg/yolov1-tiny.cfg'
MODEL_SPECS = [(ci, co, i < P.N_POOL) for i, (ci, co) in enumerate(P.CONV_CHANNELS)]


def test_darknet_cfg_of_the_model(tmp_path):
    """The cfg the reference ships (/root/reference/cfg/yolov1-tiny.cfg:1-129, present in the build container only)
    and a cfg generated here with the same section pattern both parse to the layer list the model file transcribes."""
    gen = str(tmp_path / 'gen.cfg')
    darknet.write_cfg(gen, MODEL_SPECS, 448, 1470)
    paths = [gen] + ([REFERENCE_CFG] if os.path.exists(REFERENCE_CFG) else [])
    for path in paths:
        sections = darknet.parse_cfg(path)
        assert [t for t, _ in sections].count('convolutional') == 8 and sections[-1][0] == 'detection'
        size, specs, units = darknet.conv_specs_from_cfg(sections)
        assert (size, specs, units) == (448, MODEL_SPECS, 1470), path
        det = sections[-1][1]
        assert (int(det['classes']), int(det['side']), int(det['num'])) == (20, 7, 2)
    # a cfg the fused kernels do not cover is refused
    bad = tmp_path / 'bad.cfg'
    bad.write_text(open(gen).read().replace('size=3', 'size=5', 1))
    with pytest.raises(ValueError, match='3x3'):
        darknet.conv_specs_from_cfg(darknet.parse_cfg(str(bad)))


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
    # INTER_LINEAR as OpenCV computes it on 8-bit images: identity at equal size, 2x2 box average on an exact 2x downscale
    img = np.random.default_rng(2).integers(0, 256, (8, 12, 3), dtype=np.uint8)
    assert np.array_equal(resize_bilinear_u8(img, 12, 8), img)
    half = resize_bilinear_u8(img, 6, 4).astype(np.int64)
    box = img.reshape(4, 2, 6, 2, 3).astype(np.int64).sum((1, 3))
    assert np.array_equal(half, (box + 2) >> 2)
    # the product's host resize against the checker's restatement (oracle/resize_ref.py; parity with cv2 itself is
    # unpinned: OpenCV is absent), and against plain float bilinear interpolation within the fixed-point error
    from oracle.resize_ref import resize_linear_u8_ref
    rng = np.random.default_rng(3)
    for h, w, oh, ow in [(375, 500, 448, 448), (37, 100, 448, 448), (600, 450, 224, 224), (2, 3, 448, 448), (449, 447, 448, 448)]:
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = resize_bilinear_u8(im, ow, oh)
        assert np.array_equal(got, resize_linear_u8_ref(im, ow, oh))
        fy = np.clip((np.arange(oh) + 0.5) * h / oh - 0.5, 0, h - 1); fx = np.clip((np.arange(ow) + 0.5) * w / ow - 0.5, 0, w - 1)
        y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
        y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
        ty, tx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
        f = im.astype(np.float64)
        ideal = (f[y0][:, x0] * (1 - tx) + f[y0][:, x1] * tx) * (1 - ty) + (f[y1][:, x0] * (1 - tx) + f[y1][:, x1] * tx) * ty
        assert np.abs(got - ideal).max() <= 1.0
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


def test_label_encoder_single_class_twins_vs_reference_goldens(tmp_path):
    """SequenceForFaces.read (/root/reference/data_sequence.py:248-287: xml corners, float, shifted by -1) and
    SequenceForCelebA.read (:353-395: xywh -> corners, -1, swap on negative sizes), goldens written by the reference's
    own methods (tests/golden/make_golden_labels.py): bit-exact."""
    from conftest import load_golden
    from yolo_amd.data_sequence import encode_boxes, voc_xml_boxes, celeba_box
    g = load_golden('labels_faces.npz')
    for i, ((h, w), want) in enumerate(zip(g['hw'], g['label_matrix'])):
        rows = g['xml_corners'][g['box_off'][i]:g['box_off'][i + 1]]
        xml = tmp_path / f'{i}.xml'
        xml.write_text('<annotation><size><width>%d</width></size>' % w + ''.join(
            '<object><name>face</name><bndbox><xmin>%r</xmin><ymin>%r</ymin><xmax>%r</xmax><ymax>%r</ymax></bndbox></object>'
            % tuple(float(v) for v in r) for r in rows) + '</annotation>')
        got = encode_boxes(voc_xml_boxes(str(xml)), int(h), int(w), 7, 1)
        assert got.shape == (7, 7, 6) and np.array_equal(got, want), i
    g = load_golden('labels_celeba.npz')
    swapped = 0
    for xywh, (h, w), want in zip(g['xywh'], g['hw'], g['label_matrix']):
        box = celeba_box([int(v) for v in xywh])
        swapped += int(xywh[2] < 0 or xywh[3] < 0)
        assert box[0] <= box[2] and box[1] <= box[3]
        assert np.array_equal(encode_boxes([box], int(h), int(w), 7, 1), want)
    assert swapped >= 5                                   # the fixture exercises the swap branch


def test_single_class_sequences_protocol(tmp_path):
    """Constructors and record discovery of the three single-class loaders (data_sequence.py:111-129, :211-227,
    :310-332) on a tiny synthetic directory tree."""
    from PIL import Image
    from yolo_amd.data_sequence import SequenceForAirplanes, SequenceForFaces, SequenceForCelebA
    rng = np.random.default_rng(6)
    img_dir, ann = tmp_path / 'img', tmp_path / 'ann'
    (ann / 'train').mkdir(parents=True); (ann / 'val').mkdir(); (img_dir / 'a' / 'b' / 'c' / 'd').mkdir(parents=True)

    def save(path, h=40, w=60):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(str(path))
    # airplanes: csv + jpg with the same stem; names starting with 'airplane' train, with '4' validate
    for n in ('airplane_001', 'airplane_002', '428483'):
        (ann / f'{n}.csv').write_text('2\n5 6 30 20\n10 10 50 38\n')
        save(img_dir / f'{n}.jpg')
    seq = SequenceForAirplanes('train', str(ann), str(img_dir), (224, 224, 3), 2, shuffle=False)
    assert len(seq.data_sets) == 2 and len(seq) == 1
    X, y = seq[0]
    assert X.shape == (2, 224, 224, 3) and y.shape == (2, 7, 7, 6) and y[..., 5].sum() == 4 and (y[..., 0] == y[..., 5]).all()
    assert len(SequenceForAirplanes('val', str(ann), str(img_dir), (224, 224, 3), 2).data_sets) == 1
    # faces: the first four '_' of the xml name are directory separators of the image path
    (ann / 'train' / 'a_b_c_d_img_1.xml').write_text(
        '<annotation><object><bndbox><xmin>11</xmin><ymin>6</ymin><xmax>31</xmax><ymax>26</ymax></bndbox></object></annotation>')
    save(img_dir / 'a' / 'b' / 'c' / 'd' / 'img_1.jpg')
    seq = SequenceForFaces('train', str(ann), str(img_dir), (224, 224, 3), 4, shuffle=False)
    X, y = seq[0]
    assert X.shape == (1, 224, 224, 3) and y[0, 2, 2, 5] == 1            # centre (20, 15) of a 60x40 image -> cell (2, 2)
    assert np.allclose(y[0, 2, 2, 1:5], [7 * 20 / 60 - 2, 7 * 15 / 40 - 2, 20 / 60, 20 / 40])
    # CelebA: two header lines, 80 / 20 split in file order
    save(img_dir / '000001.jpg')
    lst = tmp_path / 'list_bbox_celeba.txt'
    lst.write_text('5\nimage_id x_1 y_1 width height\n' + ''.join(f'000001.jpg   {5 + i}  4 20  30\n' for i in range(5)))
    tr = SequenceForCelebA('train', str(lst), str(img_dir), (224, 224, 3), 2)
    va = SequenceForCelebA('val', str(lst), str(img_dir), (224, 224, 3), 2, shuffle=False)
    assert len(tr.data_sets) == 4 and len(va.data_sets) == 1 and va.data_sets[0]['xywh'] == [9, 4, 20, 30]
    X, y = va[0]
    assert y.shape == (1, 7, 7, 6) and y[..., 5].sum() == 1


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


def test_dense_split_choice():
    """ops.pick_splits (host logic, no GPU): one block per CU for the 32 x 128 tiles of batches up to 32, two per CU for the
    64 x 64 tiles above, never a K slice shorter than four chunks, never more blocks than one round of the chip."""
    from yolo_amd import ops
    K, units = P.FEATURES, 1470
    for n in (1, 8, 32):
        assert ops.pick_splits(K, n, units) == 21 and ops.pick_splits(K, n, units, chunk=64) == 21
    assert ops.pick_splits(K, 64, units) == 22 and ops.pick_splits(K, 128, units, chunk=64) == 11
    for n in (1, 33, 64, 65, 128, 200, 512, 4096):
        for chunk in (ops.DENSE_K_ALIGN, 64):
            s = ops.pick_splits(K, n, units, chunk=chunk)
            tiles = ((n + 63) // 64) * ((units + 63) // 64) if n > 32 else (units + 127) // 128
            assert s >= 1 and (s == 1 or (K // chunk) // s >= 4)
            assert s == 1 or tiles * s <= (512 if n > 32 else 256)
