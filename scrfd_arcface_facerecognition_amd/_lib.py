"""ctypes binding of libfaceid.so (include/faceid.h).  There is NO fallback: if the shared library
is missing or a call fails, this raises -- the product never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FID_LIB names another build of the same library in this directory (libfaceid_asan.so from `make asan`); never a CPU fallback
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("FID_LIB") or "libfaceid.so"))

c_void_pp = C.POINTER(C.c_void_p)
c_int_p = C.POINTER(C.c_int)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
c_f32_p = C.POINTER(C.c_float)
c_f64_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); every symbol include/faceid.h declares
class GateConfig(C.Structure):
    """fid_gate_config (include/faceid.h): the reference's config.json blocks face_quality / side_face_detection / face_detection, flattened;
    the defaults are the reference's own values"""
    _fields_ = ([("size_normalization", C.c_float)]
                + [(n, C.c_float) for n in ("w_detection", "w_size", "w_blur", "w_pose", "w_lighting",
                                            "ar_extreme_profile", "ar_very_strong_profile", "ar_strong_profile", "ar_very_wide", "ar_wide", "ar_moderately_wide",
                                            "area_extremely_small", "area_very_small", "area_small", "area_very_large", "area_large",
                                            "compactness_very_low", "compactness_low", "confidence_very_low", "confidence_low", "edge_position_threshold")]
                + [("decision_threshold", C.c_int32)]
                + [(n, C.c_float) for n in ("yaw_threshold", "pitch_threshold", "confidence_threshold", "min_quality_threshold")])

    DEFAULTS = dict(size_normalization=10000.0, w_detection=0.4, w_size=0.2, w_blur=0.2, w_pose=0.1, w_lighting=0.1,
                    ar_extreme_profile=0.2, ar_very_strong_profile=0.3, ar_strong_profile=0.5, ar_very_wide=2.5, ar_wide=2.0, ar_moderately_wide=1.6,
                    area_extremely_small=1200.0, area_very_small=1800.0, area_small=2500.0, area_very_large=400000.0, area_large=300000.0,
                    compactness_very_low=0.10, compactness_low=0.6, confidence_very_low=0.15, confidence_low=0.7,
                    edge_position_threshold=30.0, decision_threshold=4, yaw_threshold=35.0, pitch_threshold=35.0,
                    confidence_threshold=0.6, min_quality_threshold=0.05)

    def __init__(self, **kw):
        super().__init__()
        vals = dict(self.DEFAULTS)
        unknown = set(kw) - set(vals)
        if unknown:
            raise TypeError(f"unknown gate thresholds: {sorted(unknown)}")
        vals.update(kw)
        for k, v in vals.items():
            setattr(self, k, v)

    @classmethod
    def from_reference_json(cls, cfg: dict) -> "GateConfig":
        """the reference's config.json (the three blocks the gates read; missing keys keep the defaults)"""
        q, s, d = cfg.get("face_quality", {}), cfg.get("side_face_detection", {}), cfg.get("face_detection", {})
        kw = {}
        if "size_normalization" in q:
            kw["size_normalization"] = float(q["size_normalization"])
        for src, dst in (("detection_score", "w_detection"), ("size_score", "w_size"), ("blur_score", "w_blur"), ("pose_score", "w_pose"),
                         ("lighting_score", "w_lighting")):
            if src in q.get("weights", {}):
                kw[dst] = float(q["weights"][src])
        for block, prefix in (("aspect_ratio_thresholds", "ar_"), ("area_thresholds", "area_"), ("compactness_thresholds", "compactness_"),
                              ("confidence_thresholds", "confidence_")):
            for k, v in s.get(block, {}).items():
                if prefix + k in cls.DEFAULTS:
                    kw[prefix + k] = float(v)
        for k in ("edge_position_threshold", "decision_threshold"):
            if k in s:
                kw[k] = int(s[k]) if k == "decision_threshold" else float(s[k])
        for k in ("yaw_threshold", "pitch_threshold", "confidence_threshold", "min_quality_threshold"):
            if k in d:
                kw[k] = float(d[k])
        return cls(**kw)


SIGNATURES = {
    "fid_abi_version": (C.c_int, []),
    "fid_last_error": (C.c_char_p, []),
    "fid_device_count": (C.c_int, [c_int_p]),
    "fid_ctx_create": (C.c_int, [C.c_int, C.c_void_p, c_void_pp]),
    "fid_ctx_destroy": (C.c_int, [C.c_void_p]),
    "fid_sync": (C.c_int, [C.c_void_p]),
    "fid_device_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "fid_malloc": (C.c_int, [C.c_void_p, C.c_size_t, c_void_pp]),
    "fid_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fid_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fid_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fid_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "fid_pinned_alloc": (C.c_int, [C.c_void_p, C.c_size_t, c_void_pp]),
    "fid_pinned_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fid_upload_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fid_upload_wait": (C.c_int, [C.c_void_p]),
    "fid_upload_release": (C.c_int, [C.c_void_p, C.c_int]),
    "fid_upload_async_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fid_upload_wait_slot": (C.c_int, [C.c_void_p, C.c_int]),
    "fid_event_record": (C.c_int, [C.c_void_p, C.c_int]),
    "fid_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_f32_p]),
    "fid_net_create": (C.c_int, [C.c_void_p, c_i32_p, C.c_int, c_i32_p, C.c_int, C.c_void_p, C.c_size_t,
                                 C.c_int, C.c_int, C.c_int, c_void_pp]),
    "fid_net_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fid_net_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "fid_net_set_sub_batch": (C.c_int, [C.c_void_p, C.c_int]),
    "fid_net_tensor": (C.c_int, [C.c_void_p, C.c_int, c_void_pp, c_int_p, c_int_p]),
    "fid_net_run_profiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_f32_p]),
    "fid_net_plan_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "fid_net_plan_load": (C.c_int, [C.c_void_p, C.c_char_p, c_int_p]),
    "fid_net_macs": (C.c_int, [C.c_void_p, c_f64_p]),
    "fid_letterbox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                c_f64_p]),
    "fid_scrfd_postprocess": (C.c_int, [C.c_void_p, c_void_pp, c_i32_p, c_i32_p, c_i64_p, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "fid_scrfd_set_candidate_capacity": (C.c_int, [C.c_void_p, C.c_int]),
    "fid_scrfd_check": (C.c_int, [C.c_void_p, c_int_p]),
    "fid_scrfd_decode": (C.c_int, [C.c_void_p, c_void_pp, c_i32_p, c_i32_p, c_i64_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "fid_distance2bbox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fid_distance2kps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fid_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "fid_align_crops": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fid_l2_normalize_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fid_l2_normalize_f16_slots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "fid_face_gates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "fid_gallery_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_void_pp]),
    "fid_gallery_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fid_gallery_info": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p]),
    "fid_gallery_data": (C.c_int, [C.c_void_p, c_void_pp]),
    "fid_gallery_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "fid_gallery_set_rows": (C.c_int, [C.c_void_p, C.c_void_p, c_i32_p, C.c_void_p, C.c_int]),
    "fid_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "fid_match_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fid_match_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "fid_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "fid_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, c_void_pp]),
    "fid_comm_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fid_comm_info": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "fid_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "fid_cosine_matrix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


class FaceIdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfaceid error {code}: {msg}")
        self.code = code


def load():
    """dlopen libfaceid.so and attach the prototypes.  Raises if it has not been built
    (python __graft_entry__.py / make -C scrfd_arcface_facerecognition_amd/csrc)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise FileNotFoundError(
                    f"{LIB_PATH} not built: run `make -C {os.path.join(_HERE, 'csrc')}`; there is no CPU fallback")
            if os.environ.get("FID_LIB"):                 # never silently: a forgotten FID_LIB would test another build's kernels
                print(f"[faceid] FID_LIB={os.environ['FID_LIB']}: loading {LIB_PATH} instead of libfaceid.so", file=sys.stderr, flush=True)
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)      # AttributeError if the library lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            if lib.fid_abi_version() != 2:
                raise RuntimeError("libfaceid ABI version mismatch")
            _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise FaceIdError(rc, load().fid_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(load().fid_device_count(C.byref(n)))
    return n.value


def _ptr(x):
    """device pointer (int / c_void_p / DeviceBuffer / torch tensor) -> c_void_p"""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, DeviceBuffer):
        return C.c_void_p(x.ptr)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return x


class DeviceBuffer:
    """A typed view of device memory owned by the library's allocator (or borrowed)."""

    def __init__(self, ctx: "Context", shape, dtype, ptr=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.owned = ptr is None
        if ptr is None:
            p = C.c_void_p()
            check(ctx.lib.fid_malloc(ctx.handle, max(self.nbytes, 16), C.byref(p)))
            ptr = p.value
        self.ptr = int(ptr)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes, (arr.shape, self.shape)
        check(self.ctx.lib.fid_memcpy_h2d(self.ctx.handle, C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), self.nbytes))
        self.ctx._keepalive = arr      # the copy is asynchronous with respect to the host
        self.ctx.sync()
        return self

    def download(self, count_bytes=None):
        out = np.empty(self.shape, dtype=self.dtype)
        n = self.nbytes if count_bytes is None else count_bytes
        check(self.ctx.lib.fid_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), n))
        return out

    def zero(self):
        check(self.ctx.lib.fid_memset(self.ctx.handle, C.c_void_p(self.ptr), 0, self.nbytes))
        return self

    def free(self):
        if self.owned and self.ptr:
            self.ctx.lib.fid_free(self.ctx.handle, C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            if self.ctx.handle:
                self.free()
        except Exception:
            pass


class Context:
    """fid_ctx: one HIP device + stream.  `stream` may be a raw hipStream_t (e.g.
    torch.cuda.current_stream().cuda_stream) so library work is ordered with torch/RCCL work."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.fid_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self._keepalive = None

    def sync(self):
        check(self.lib.fid_sync(self.handle))

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        check(self.lib.fid_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def empty(self, shape, dtype) -> DeviceBuffer:
        return DeviceBuffer(self, shape, dtype)

    def to_device(self, arr) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.shape, arr.dtype).upload(arr)

    def borrow(self, ptr, shape, dtype) -> DeviceBuffer:
        return DeviceBuffer(self, shape, dtype, ptr=ptr)

    def event_record(self, slot: int):
        check(self.lib.fid_event_record(self.handle, slot))

    def elapsed_ms(self, a: int, b: int) -> float:
        ms = C.c_float()
        check(self.lib.fid_event_elapsed_ms(self.handle, a, b, C.byref(ms)))
        return ms.value

    def close(self):
        if self.handle:
            self.lib.fid_ctx_destroy(self.handle)
            self.handle = None


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    """Process-wide context per device (the reference keeps one global ORT session per model)."""
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
