"""Oracle: SCRFD post-processing.  Follows reference models/scrfd.py:89-207 and
utils/helpers.py:62-107 statement by statement (numpy, fp32).  Test infrastructure only."""
import numpy as np

STRIDES = (8, 16, 32)
NUM_ANCHORS = 2


def anchor_centers(height, width, stride, num_anchors=NUM_ANCHORS):
    # scrfd.py:102-105
    c = np.stack(np.mgrid[:height, :width][::-1], axis=-1).astype(np.float32)
    c = (c * stride).reshape((-1, 2))
    if num_anchors > 1:
        c = np.stack([c] * num_anchors, axis=1).reshape((-1, 2))
    return c


def distance2bbox(points, distance):
    # helpers.py:74-83
    x1 = points[:, 0] - distance[:, 0]
    y1 = points[:, 1] - distance[:, 1]
    x2 = points[:, 0] + distance[:, 2]
    y2 = points[:, 1] + distance[:, 3]
    return np.stack([x1, y1, x2, y2], axis=-1)


def distance2kps(points, distance):
    # helpers.py:98-107
    preds = []
    for i in range(0, distance.shape[1], 2):
        preds.append(points[:, i % 2] + distance[:, i])
        preds.append(points[:, i % 2 + 1] + distance[:, i + 1])
    return np.stack(preds, axis=-1)


def decode_heads(outputs, input_hw, threshold):
    """scrfd.py:89-119.  outputs: 9 arrays (scores x3, bbox x3, kps x3)."""
    H, W = input_hw
    scores_list, bboxes_list, kpss_list = [], [], []
    for idx, stride in enumerate(STRIDES):
        scores = outputs[idx]
        bbox_preds = outputs[idx + 3] * stride
        kps_preds = outputs[idx + 6] * stride
        centers = anchor_centers(H // stride, W // stride, stride)
        pos = np.where(scores >= threshold)[0]
        bboxes = distance2bbox(centers, bbox_preds)
        scores_list.append(scores[pos])
        bboxes_list.append(bboxes[pos])
        kpss = distance2kps(centers, kps_preds)
        kpss = kpss.reshape((kpss.shape[0], -1, 2))
        kpss_list.append(kpss[pos])
    return scores_list, bboxes_list, kpss_list


def nms(dets, iou_thres):
    """scrfd.py:180-207.  Returns indices into dets (list of np.int64).
    Tie order: the reference uses the default (unstable) argsort; the oracle fixes it to a
    stable sort so results are deterministic -- identical on tie-free input."""
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort(kind="stable")[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(0.0, xx2 - xx1 + 1)
        h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[np.where(ovr <= iou_thres)[0] + 1]
    return keep


def letterbox_geometry(img_h, img_w, input_size=(640, 640)):
    """scrfd.py:123-134 (python float / int() truncation semantics)."""
    width, height = int(input_size[0]), int(input_size[1])
    img_h, img_w = int(img_h), int(img_w)      # image.shape entries are python ints in the reference
    im_ratio = float(img_h) / img_w
    model_ratio = height / width
    if im_ratio > model_ratio:
        new_height = height
        new_width = int(new_height / im_ratio)
    else:
        new_width = width
        new_height = int(new_width * im_ratio)
    det_scale = float(new_height) / img_h
    return new_width, new_height, det_scale


def detect_from_heads(outputs, img_hw, input_size=(640, 640), conf_thres=0.5, iou_thres=0.4,
                      max_num=0, metric="max"):
    """scrfd.py:122-178 with session.run's result given: (det[K,5] f32, kpss[K,5,2] f32)."""
    _, _, det_scale = letterbox_geometry(img_hw[0], img_hw[1], input_size)
    s, b, k = decode_heads(outputs, (input_size[1], input_size[0]), conf_thres)
    scores = np.vstack(s)
    order = scores.ravel().argsort(kind="stable")[::-1]
    bboxes = np.vstack(b) / det_scale
    kpss = np.vstack(k) / det_scale
    pre_det = np.hstack((bboxes, scores)).astype(np.float32, copy=False)
    pre_det = pre_det[order, :]
    keep = nms(pre_det, iou_thres)
    det = pre_det[keep, :]
    kpss = kpss[order, :, :][keep, :, :]
    if 0 < max_num < det.shape[0]:
        area = (det[:, 2] - det[:, 0]) * (det[:, 3] - det[:, 1])
        center = int(img_hw[0]) // 2, int(img_hw[1]) // 2
        offsets = np.vstack([(det[:, 0] + det[:, 2]) / 2 - center[1],
                             (det[:, 1] + det[:, 3]) / 2 - center[0]])
        dist2 = np.sum(np.power(offsets, 2.0), 0)
        values = area if metric == "max" else (area - dist2 * 2.0)
        bindex = np.argsort(values, kind="stable")[::-1][0:max_num]
        det = det[bindex, :]
        kpss = kpss[bindex, :]
    return det.astype(np.float32), kpss.astype(np.float32)
