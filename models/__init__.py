"""Drop-in for the reference's top-level `models` package (reference main.py:11:
`from models import SCRFD, ArcFace`)."""
from scrfd_arcface_facerecognition_amd.models import SCRFD, ArcFace

__all__ = ["SCRFD", "ArcFace"]
