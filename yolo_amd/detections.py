"""Detection consumer: what comes after decode + NMS.

The reference only draws (tiny_yolov1.py:142-160): for every kept box it truncates the corners with int() and writes
`classes_name[box_cls]` next to the rectangle, all in the 448-pixel frame of the resized image.  This module emits
the same information as records and, for evaluation, maps the boxes back to the original image (the resize of
utils.load_img stretches both axes independently, so the inverse is a per-axis scale)."""
import numpy as np

VOC_CLASSES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable',
               'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


def to_records(boxes, scores, classes, keep, image_hw=None, img_size=448, names=VOC_CLASSES):
    """One image.  boxes (slots,4) float64 (x0,y0,x1,y1) in the img_size frame, scores (slots,), classes (slots,),
    keep (slots,) as returned by yolo_eval / ty1_decode_nms.  -> list of dicts in the reference's raster order:
    `corners` = the int()-truncated corners the reference draws (:151-154), `box` = the same box in pixels of the
    original image when image_hw = (h, w) is given (float64, not truncated), `class_id`, `name`, `score`."""
    boxes, scores, classes, keep = (np.asarray(a) for a in (boxes, scores, classes, keep))
    sx = sy = 1.0
    if image_hw is not None:
        sy, sx = image_hw[0] / float(img_size), image_hw[1] / float(img_size)
    out = []
    for i in np.nonzero(keep)[0]:
        x0, y0, x1, y1 = (float(v) for v in boxes[i])
        c = int(classes[i])
        out.append(dict(slot=int(i), class_id=c, name=names[c] if names is not None and c < len(names) else str(c),
                        score=float(scores[i]), corners=(int(x0), int(y0), int(x1), int(y1)),
                        box=(x0 * sx, y0 * sy, x1 * sx, y1 * sy)))
    return out


def batch_records(det, image_hws=None, img_size=448, names=VOC_CLASSES):
    """det = (boxes (N,slots,4), scores, classes, keep) tensors or arrays -> list (per image) of record lists."""
    arrs = [a.cpu().numpy() if hasattr(a, 'cpu') else np.asarray(a) for a in det]
    n = arrs[0].shape[0]
    return [to_records(arrs[0][i], arrs[1][i], arrs[2][i], arrs[3][i], None if image_hws is None else image_hws[i],
                       img_size, names) for i in range(n)]


def voc_result_lines(image_id, records):
    """PASCAL VOC detection-result format, one line per record: '<image id> <score> <xmin> <ymin> <xmax> <ymax>',
    grouped per class name -> {name: [lines]} (what comp4_det_test_<class>.txt files hold)."""
    out = {}
    for r in records:
        x0, y0, x1, y1 = r['box']
        out.setdefault(r['name'], []).append(f'{image_id} {r["score"]:.6f} {x0:.1f} {y0:.1f} {x1:.1f} {y1:.1f}')
    return out
