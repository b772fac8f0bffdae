"""Drop-in for the label side of /root/reference/data_sequence.py: annotation record -> y_true.

The reference has four loaders whose `read` methods repeat one grid encoder and differ only in where the boxes come
from and how they are pre-processed:

  SequenceData          :10-107   VOC text lines 'path x0,y0,x1,y1,cls ...', integer corners, 20 classes
  SequenceForAirplanes  :110-207  one csv per image (first line = row count, then 'x0 y0 x1 y1'), integer corners, 1 class
  SequenceForFaces      :210-306  VOC xml per image, float corners shifted by -1 (:260-263), 1 class
  SequenceForCelebA     :309-414  'name x y w h' lines, corner = xy - 1, far corner = corner + wh, swapped into
                                  order when w or h is negative (:360-371), 1 class

Here that is one encoder (`encode_boxes`), four box readers, and one Sequence-protocol class that the four loader
names specialise.  Images come from yolo_amd.utils.load_img.  The reference needs `np.int` and compares strings with
`is`; neither is reproduced.
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from .utils import load_img


def encode_boxes(boxes, image_h, image_w, S=7, C=20):
    """boxes: iterable of (xmin, ymin, xmax, ymax[, cls]) in pixels of the ORIGINAL image (ints or floats, used as
    given).  -> (S, S, C+5) float64: [one-hot C | x, y offset in the cell | w, h fraction of the image | objectness].
    The FIRST box whose centre lands in a cell owns it (data_sequence.py:79)."""
    grid = np.zeros([S, S, C + 5])
    for b in boxes:
        xmin, ymin, xmax, ymax = b[0], b[1], b[2], b[3]
        cls = int(b[4]) if len(b) > 4 else 0
        gx = S * ((xmin + xmax) / 2 / image_w)          # centre in grid units
        gy = S * ((ymin + ymax) / 2 / image_h)
        row, col = int(gy), int(gx)
        if grid[row, col, C + 4] == 0:
            grid[row, col, cls] = 1
            grid[row, col, C:C + 4] = [gx - col, gy - row, (xmax - xmin) / image_w, (ymax - ymin) / image_h]
            grid[row, col, C + 4] = 1
    return grid


# ----------------------------------------------------------------------------------------------- box readers
def parse_annotation_line(line):
    """'path x0,y0,x1,y1,cls x0,y0,x1,y1,cls ...' (voc_annotation.py output) -> (path, [(x0,y0,x1,y1,cls), ...])."""
    parts = line.strip().split()
    return parts[0], [tuple(int(v) for v in p.split(',')) for p in parts[1:]]


def airplane_csv_boxes(csv_path):
    """data_sequence.py:158-164: pandas reads the first line as the header, every later line is 'x0 y0 x1 y1'."""
    with open(csv_path) as f:
        rows = [ln.strip() for ln in f.read().splitlines()[1:]]
    return [tuple(int(v) for v in r.split(' ')[:4]) for r in rows if r]


def voc_xml_boxes(xml_path):
    """data_sequence.py:256-263: every <object>/<bndbox>, corners as floats shifted by -1 (VOC pixels are 1-based)."""
    out = []
    for obj in ET.parse(xml_path).getroot().iter('object'):
        bb = obj.find('bndbox')
        out.append(tuple(float(bb.find(k).text) - 1 for k in ('xmin', 'ymin', 'xmax', 'ymax')))
    return out


def celeba_box(xywh):
    """data_sequence.py:360-371: x, y are 1-based; the far corner is corner + size; a negative size swaps the pair."""
    x0, y0 = float(xywh[0]) - 1, float(xywh[1]) - 1
    x1, y1 = float(xywh[2]) + x0, float(xywh[3]) + y0
    return (min(x0, x1), min(y0, y1), max(x0, x1), max(y0, y1))


# ----------------------------------------------------------------------------------------------- Sequence protocol
class _GridSequence:
    """What keras.utils.Sequence asks for (__len__, __getitem__, on_epoch_end) over a list of records; a subclass says
    how a record becomes (image path, boxes).  `read` and `data_generation` keep the reference's method names."""
    S, C = 7, 20

    def __init__(self, records, target_size, batch_size, shuffle=True):
        self.data_sets = list(records)
        self.image_size = target_size[0:2]
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.indexes = np.arange(len(self.data_sets))

    def __len__(self):
        return math.ceil(len(self.data_sets) / float(self.batch_size))

    def __getitem__(self, idx):
        picked = self.indexes[idx * self.batch_size:(idx + 1) * self.batch_size]
        return self.data_generation([self.data_sets[k] for k in picked])

    def on_epoch_end(self):
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def boxes_of(self, record):
        raise NotImplementedError

    def read(self, record):
        path, boxes = self.boxes_of(record)
        image, image_h, image_w = load_img(path=path, shape=self.image_size)
        return image, encode_boxes(boxes, image_h, image_w, self.S, self.C)

    def data_generation(self, records):
        pairs = [self.read(r) for r in records]
        return np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])


class SequenceData(_GridSequence):
    """data_sequence.py:10-107.  `model` is 'train' or 'val' (2007_train.txt / 2007_val.txt in `dir`)."""

    def __init__(self, model, dir, target_size, batch_size, shuffle=True):
        self.model = model
        lines = []
        if model in ('train', 'val'):
            with open(os.path.join(dir, f'2007_{model}.txt'), 'r') as f:
                lines = f.readlines()
        super().__init__(lines, target_size, batch_size, shuffle)

    def boxes_of(self, record):
        return parse_annotation_line(record)


class SequenceForAirplanes(_GridSequence):
    """data_sequence.py:110-207: csv files named airplane* train, 4* validate; the image has the csv's name."""
    C = 1

    def __init__(self, model, annotation, img_dir, target_size, batch_size, shuffle=True):
        self.model = model
        prefix = {'train': 'airplane', 'val': '4'}.get(model)
        names = [n for n in os.listdir(annotation) if n.startswith(prefix)] if prefix else []
        super().__init__([{'csv': os.path.join(annotation, n), 'jpg': os.path.join(img_dir, n.replace('csv', 'jpg'))}
                          for n in names], target_size, batch_size, shuffle)

    def boxes_of(self, record):
        return record['jpg'], airplane_csv_boxes(record['csv'])


class SequenceForFaces(_GridSequence):
    """data_sequence.py:210-306: xml files under annotation/train or /val; the first four '_' of a file name are
    directory separators of the image path (:217)."""
    C = 1

    def __init__(self, model, annotation, img_dir, target_size, batch_size, shuffle=True):
        self.model = model
        recs = []
        if model in ('train', 'val'):
            sub = os.path.join(annotation, model)
            recs = [{'xml': os.path.join(sub, n), 'jpg': os.path.join(img_dir, n.replace('_', '/', 4).replace('xml', 'jpg'))}
                    for n in os.listdir(sub)]
        super().__init__(recs, target_size, batch_size, shuffle)

    def boxes_of(self, record):
        return record['jpg'], voc_xml_boxes(record['xml'])


class SequenceForCelebA(_GridSequence):
    """data_sequence.py:309-414: list_bbox_celeba.txt (two header lines), first 80 % train, rest validate."""
    C = 1

    def __init__(self, model, annotation, img_dir, target_size, batch_size, shuffle=True):
        self.model = model
        with open(annotation, 'r') as f:
            rows = f.readlines()[2:]
        cut = int(0.8 * len(rows))
        rows = rows[:cut] if model == 'train' else rows[cut:] if model == 'val' else rows
        recs = []
        for row in rows:
            name, x, y, w, h = row.split()[:5]
            recs.append({'xywh': [int(x), int(y), int(w), int(h)], 'jpg': os.path.join(img_dir, name)})
        super().__init__(recs, target_size, batch_size, shuffle)

    def boxes_of(self, record):
        return record['jpg'], [celeba_box(record['xywh'])]
