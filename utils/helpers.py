"""Drop-in for the reference's top-level `utils.helpers` module (reference main.py:12:
`from utils.helpers import compute_similarity, draw_bbox_info, draw_bbox`)."""
from scrfd_arcface_facerecognition_amd.utils.helpers import (  # noqa: F401
    compute_similarity, distance2bbox, distance2kps, draw_bbox, draw_bbox_info, estimate_norm, match_gallery,
    norm_crop_image, reference_alignment)
