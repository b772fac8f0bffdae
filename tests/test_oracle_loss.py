"""The loss oracle against the reference-generated golden vectors (yolo/yolo.py and
faces/trainFace.py run through a torch mapping of the Keras backend).  CPU only."""
import numpy as np
import pytest
import torch

from oracle.loss_ref import (yolo_head_ref, iou_ref, yolo_loss_and_grad_ref, yolo_loss_closed_form)
from conftest import load_golden

CASES = [('loss_c20.npz', 20), ('loss_c1.npz', 1)]


@pytest.mark.parametrize('name,C', CASES)
def test_loss_restatement_matches_reference(name, C):
    g = load_golden(name)
    img = int(g['img_size'])
    loss, per, grad = yolo_loss_and_grad_ref(g['y_true'], g['y_pred'], C, 2, img)
    assert abs(loss - g['loss']) <= 1e-5 * abs(g['loss'])
    np.testing.assert_allclose(per, g['loss_per_image'], rtol=1e-5)
    np.testing.assert_allclose(grad, g['grad'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('name,C', CASES)
def test_closed_form_gradient_matches_reference(name, C):
    g = load_golden(name)
    img = int(g['img_size'])
    loss, grad = yolo_loss_closed_form(g['y_true'], g['y_pred'], C, 2, img)
    assert abs(loss - g['loss']) <= 2e-6 * abs(g['loss'])
    np.testing.assert_allclose(grad, g['grad'], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize('name,C', CASES)
def test_head_and_iou_match_reference(name, C):
    g = load_golden(name)
    img = int(g['img_size'])
    boxes = torch.from_numpy(g['y_pred'][..., C + 2:]).reshape(-1, 7, 7, 2, 4)
    xy, wh = yolo_head_ref(boxes, img)
    assert np.array_equal(xy.numpy(), g['head_xy']) and np.array_equal(wh.numpy(), g['head_wh'])
    mins, maxes = xy - wh / 2, xy + wh / 2
    ious = iou_ref(mins[:, :, :, 0], maxes[:, :, :, 0], mins[:, :, :, 1], maxes[:, :, :, 1])
    assert np.array_equal(ious.numpy(), g['iou_b0_b1'])
    # grid convention: position [r=2, c=5] adds (x += 5, y += 2)
    z = torch.zeros(1, 7, 7, 1, 4)
    xy0, _ = yolo_head_ref(z, 448)
    assert xy0[0, 2, 5, 0, 0] == 320 and xy0[0, 2, 5, 0, 1] == 128
