"""Drop-in for the label side of /root/reference/data_sequence.py: annotation line -> y_true.

`encode_boxes` is the grid encoder every `read` of the reference repeats (:55-86 for VOC, :157-186 / :255-285 /
:360-393 for the single-class sets): centre -> responsible cell, offsets within the cell, size as a fraction of the
image, one-hot class, objectness; the FIRST object that lands in a cell owns it (:79).  `SequenceData` mirrors the
reference class (same constructor and Sequence protocol) on top of it; images come from yolo_amd.utils.load_img.
The reference needs `np.int`, which NumPy removed; plain `int` is the same type.
"""
import math
import os

import numpy as np

from .utils import load_img


def encode_boxes(boxes, image_h, image_w, S=7, C=20):
    """boxes: iterable of (xmin, ymin, xmax, ymax[, cls]) integers in pixels of the ORIGINAL image.
    -> (S, S, C+5) float64: [one-hot C | x, y offset in the cell | w, h fraction of the image | objectness]."""
    label_matrix = np.zeros([S, S, C + 5])
    for b in boxes:
        xmin, ymin, xmax, ymax = (int(v) for v in b[:4])
        cls = int(b[4]) if len(b) > 4 else 0
        x = (xmin + xmax) / 2 / image_w
        y = (ymin + ymax) / 2 / image_h
        w = (xmax - xmin) / image_w
        h = (ymax - ymin) / image_h
        loc = [S * x, S * y]
        loc_i, loc_j = int(loc[1]), int(loc[0])
        y = loc[1] - loc_i
        x = loc[0] - loc_j
        if label_matrix[loc_i, loc_j, C + 4] == 0:
            label_matrix[loc_i, loc_j, cls] = 1
            label_matrix[loc_i, loc_j, C:C + 4] = [x, y, w, h]
            label_matrix[loc_i, loc_j, C + 4] = 1
    return label_matrix


def parse_annotation_line(line):
    """'path x0,y0,x1,y1,cls x0,y0,x1,y1,cls ...' (voc_annotation.py output) -> (path, [(x0,y0,x1,y1,cls), ...])."""
    parts = line.strip().split()
    return parts[0], [tuple(int(v) for v in p.split(',')) for p in parts[1:]]


class SequenceData:
    """data_sequence.py:10-105.  `model` is 'train' or 'val' (reads 2007_train.txt / 2007_val.txt in `dir`)."""

    def __init__(self, model, dir, target_size, batch_size, shuffle=True):
        self.model = model
        self.data_sets = []
        name = {'train': '2007_train.txt', 'val': '2007_val.txt'}.get(model)
        if name:
            with open(os.path.join(dir, name), 'r') as f:
                self.data_sets = self.data_sets + f.readlines()
        self.image_size = target_size[0:2]
        self.batch_size = batch_size
        self.indexes = np.arange(len(self.data_sets))
        self.shuffle = shuffle

    def __len__(self):
        return math.ceil(len(self.data_sets) / float(self.batch_size))

    def __getitem__(self, idx):
        batch_indexs = self.indexes[idx * self.batch_size:(idx + 1) * self.batch_size]
        return self.data_generation([self.data_sets[k] for k in batch_indexs])

    def on_epoch_end(self):
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def read(self, dataset):
        image_path, boxes = parse_annotation_line(dataset)
        image, image_h, image_w = load_img(path=image_path, shape=self.image_size)
        return image, encode_boxes(boxes, image_h, image_w, 7, 20)

    def data_generation(self, batch_datasets):
        images, labels = [], []
        for dataset in batch_datasets:
            image, label = self.read(dataset)
            images.append(image)
            labels.append(label)
        return np.array(images), np.array(labels)
