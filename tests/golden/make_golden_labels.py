#!/usr/bin/env python3
"""Golden vectors for the label encoder, produced by the REFERENCE's own code (build container only).

  labels_c20.npz  annotation lines -> (7,7,25) matrices of SequenceData.read        (/root/reference/data_sequence.py:45-88)
  labels_c1.npz   box lists        -> (7,7,6)  matrices of SequenceForAirplanes.read (/root/reference/data_sequence.py:150-188)
  labels_faces.npz  VOC xml corner values (1-based, as written in the xml) -> (7,7,6) matrices of SequenceForFaces.read
                  (/root/reference/data_sequence.py:248-287)
  labels_celeba.npz 'x y w h' records (incl. negative sizes) -> (7,7,6) matrices of SequenceForCelebA.read
                  (/root/reference/data_sequence.py:353-395)

The reference module is imported unmodified behind the same cv2 / tensorflow stubs as make_golden.py.  Two things it
needs that this image lacks are supplied: `np.int` (an alias NumPy removed in 1.24; the reference predates that) is set
to `int`, and `load_img` is replaced by a stand-in that returns the image size the test case names (no image files).
"""
import os
import sys
import tempfile
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    make_golden.install_stubs()
    if not hasattr(np, 'int'):
        np.int = int
    import data_sequence as ref
    rng = np.random.default_rng(20260202)
    sizes = {}
    ref.load_img = lambda path, shape: (None,) + sizes[path]

    lines, hw, mats = [], [], []
    for i in range(40):
        h, w = int(rng.integers(120, 700)), int(rng.integers(120, 700))
        path = f'img{i:03d}.jpg'
        sizes[path] = (h, w)
        objs = []
        for _ in range(int(rng.integers(0, 7))):
            x0, y0 = int(rng.integers(0, w - 2)), int(rng.integers(0, h - 2))
            x1, y1 = int(rng.integers(x0 + 1, w)), int(rng.integers(y0 + 1, h))      # centre stays inside the image
            objs.append(f'{x0},{y0},{x1},{y1},{int(rng.integers(0, 20))}')
        if i % 8 == 0 and objs:                 # two objects in one cell: first writer wins (:79)
            objs.append(objs[0].rsplit(',', 1)[0] + f',{int(rng.integers(0, 20))}')
        line = ' '.join([path] + objs) + '\n'
        _, m = ref.SequenceData.read(types.SimpleNamespace(image_size=(448, 448)), line)
        lines.append(line); hw.append((h, w)); mats.append(m)
    np.savez_compressed(os.path.join(OUT, 'labels_c20.npz'), lines=np.array(lines), hw=np.array(hw, np.int64),
                        label_matrix=np.array(mats))
    print('labels_c20.npz:', len(lines), 'lines,', int(sum(m[..., 24].sum() for m in mats)), 'responsible cells')

    boxes, off, hw1, mats1 = [], [0], [], []
    with tempfile.TemporaryDirectory() as td:
        for i in range(24):
            h, w = int(rng.integers(120, 700)), int(rng.integers(120, 700))
            jpg = f'air{i:03d}.jpg'
            sizes[jpg] = (h, w)
            rows = []
            for _ in range(int(rng.integers(0, 6))):
                x0, y0 = int(rng.integers(0, w - 2)), int(rng.integers(0, h - 2))
                rows.append((x0, y0, int(rng.integers(x0 + 1, w)), int(rng.integers(y0 + 1, h))))
            csv = os.path.join(td, f'{i}.csv')
            with open(csv, 'w') as f:
                f.write(f'{len(rows)}\n' + ''.join('%d %d %d %d\n' % r for r in rows))
            _, m = ref.SequenceForAirplanes.read(types.SimpleNamespace(image_size=(224, 224)), {'csv': csv, 'jpg': jpg})
            boxes += rows; off.append(len(boxes)); hw1.append((h, w)); mats1.append(m)
    np.savez_compressed(os.path.join(OUT, 'labels_c1.npz'), boxes=np.array(boxes, np.int64).reshape(-1, 4),
                        box_off=np.array(off, np.int64), hw=np.array(hw1, np.int64), label_matrix=np.array(mats1))
    print('labels_c1.npz:', len(mats1), 'images,', len(boxes), 'boxes')

    # SequenceForFaces.read: xml per image, float corners - 1
    vals, off, hw2, mats2 = [], [0], [], []
    with tempfile.TemporaryDirectory() as td:
        for i in range(24):
            h, w = int(rng.integers(120, 700)), int(rng.integers(120, 700))
            jpg = f'face{i:03d}.jpg'
            sizes[jpg] = (h, w)
            rows = []
            for _ in range(int(rng.integers(0, 6))):
                x0, y0 = float(rng.integers(1, w - 2)), float(rng.integers(1, h - 2))
                x1, y1 = float(rng.integers(x0 + 1, w)), float(rng.integers(y0 + 1, h))
                if i % 3 == 0:                      # FDDB2XML writes fractional corners too
                    x0, y0, x1, y1 = (round(v + float(rng.random()) * 0.9, 3) for v in (x0, y0, x1 - 1, y1 - 1))
                rows.append((x0, y0, x1, y1))
            xml = os.path.join(td, f'{i}.xml')
            with open(xml, 'w') as f:
                f.write('<annotation>' + ''.join(
                    '<object><name>face</name><bndbox><xmin>%r</xmin><ymin>%r</ymin><xmax>%r</xmax><ymax>%r</ymax></bndbox></object>' % r
                    for r in rows) + '</annotation>')
            _, m = ref.SequenceForFaces.read(types.SimpleNamespace(image_size=(224, 224)), {'xml': xml, 'jpg': jpg})
            vals += rows; off.append(len(vals)); hw2.append((h, w)); mats2.append(m)
    np.savez_compressed(os.path.join(OUT, 'labels_faces.npz'), xml_corners=np.array(vals, np.float64).reshape(-1, 4),
                        box_off=np.array(off, np.int64), hw=np.array(hw2, np.int64), label_matrix=np.array(mats2))
    print('labels_faces.npz:', len(mats2), 'images,', len(vals), 'boxes')

    # SequenceForCelebA.read: one 'x y w h' record per image; some sizes negative (the swap at :364-371)
    xywh, hw3, mats3 = [], [], []
    for i in range(32):
        h, w = int(rng.integers(120, 700)), int(rng.integers(120, 700))
        jpg = f'celeb{i:03d}.jpg'
        sizes[jpg] = (h, w)
        bw, bh = int(rng.integers(2, w // 2)), int(rng.integers(2, h // 2))
        x, y = int(rng.integers(1, w - bw)), int(rng.integers(1, h - bh))
        if i % 5 == 1:
            x, bw = x + bw, -bw
        if i % 7 == 2:
            y, bh = y + bh, -bh
        _, m = ref.SequenceForCelebA.read(types.SimpleNamespace(image_size=(224, 224)), {'xywh': [x, y, bw, bh], 'jpg': jpg})
        xywh.append((x, y, bw, bh)); hw3.append((h, w)); mats3.append(m)
    np.savez_compressed(os.path.join(OUT, 'labels_celeba.npz'), xywh=np.array(xywh, np.int64), hw=np.array(hw3, np.int64),
                        label_matrix=np.array(mats3))
    print('labels_celeba.npz:', len(mats3), 'records')


if __name__ == '__main__':
    main()
