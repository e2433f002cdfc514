from .arcface import ArcFace
from .scrfd import SCRFD

__all__ = ["ArcFace", "SCRFD"]
