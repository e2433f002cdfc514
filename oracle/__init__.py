"""CPU oracle for the det -> align -> embed -> match hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / torch-CPU restatement of the reference's algorithm
(reference models/scrfd.py, models/arcface.py, utils/helpers.py, main.py:78-150) plus the
third-party numerics those files call into (OpenCV resize / warpAffine / blobFromImage,
scikit-image Umeyama, onnxruntime conv nets), per SURVEY.md Appendix A/B.

It is the CHECKER: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.  Nothing under scrfd_arcface_facerecognition_amd/ (the product) imports it, and the
product has no CPU fallback: it raises when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  pinned by reference-generated goldens (tests/golden, tools/gen_golden*.py):
      decode, threshold, sort, NMS, max_num selection, cosine, gallery scan, estimate_norm
  PARITY UNPINNED (no OpenCV / onnxruntime / weights exist offline; restated from their
  documented algorithms, self-consistency only):
      cv2.resize, cv2.warpAffine, cv2.dnn.blobFromImage(s), the conv nets themselves
"""
