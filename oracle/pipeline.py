"""Oracle: the whole reference path on the CPU, frame by frame, face by face -- the structure of
reference main.py:78-150 (build_targets / frame_processor) over the oracle pieces.
Test infrastructure only (tests, smoke(), bench.py's cpu_baseline leg)."""
import numpy as np

from . import align, match, nets
from . import postprocess as pp


def detect(image, det_net, det_P, input_size=(640, 640), conf_thres=0.5, iou_thres=0.4, max_num=0, metric="max"):
    """SCRFD.detect (scrfd.py:122-178) with the net evaluated by the fp32 torch-CPU oracle."""
    det_img, _ = align.letterbox(image, input_size)
    blob = align.blob_from_images([det_img], det_net.in_scale, det_net.in_mean)
    outs = nets.scrfd_session_outputs(det_net, det_P, blob)
    return pp.detect_from_heads(outs, image.shape[:2], input_size, conf_thres, iou_thres, max_num, metric), outs


def embed(image, kps, rec_net, rec_P):
    """ArcFace.__call__ (arcface.py:54-57)."""
    crop = align.norm_crop_image(image, kps)
    blob = align.blob_from_images([crop], rec_net.in_scale, rec_net.in_mean)
    return nets.run_net(rec_net, rec_P, blob)[rec_net.outputs[0]].reshape(-1), crop


def process_frame(frame, det_net, det_P, rec_net, rec_P, gallery, max_num=0, similarity_thresh=0.4, **det_kw):
    """frame_processor (main.py:108-150) without the drawing: [(bbox, score, kps, gallery_index, similarity)]"""
    (det, kpss), _ = detect(frame, det_net, det_P, max_num=max_num, **det_kw)
    out = []
    for bbox, kps in zip(det, kpss):
        emb, _ = embed(frame, kps, rec_net, rec_P)
        j, s = match.gallery_scan(emb, gallery, similarity_thresh)
        out.append((bbox[:4], float(bbox[4]), kps, j, float(s), emb))
    return out
