"""``YOLO`` facade over libmi355yolo.so with the call surface the reference uses.

Drop-in for ``from ultralytics import YOLO`` on the reference's hot path
(``/root/reference/model.py:5,18,38-40``): ``YOLO(path)``, ``model(frame, conf=, iou=, classes=,
max_det=, imgsz=)``, ``model.predict(...)`` and ``model.track(...)`` return ``list[Results]`` whose
``.boxes`` / ``.keypoints`` behave like Ultralytics' (results.py).  All per-frame arithmetic runs in the
HIP engine through the C ABI (include/mi355_yolo.h); nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .results import Results
from .weights import build_from_state_dict, from_bytes

COCO_NAMES = ["person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
              "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow",
              "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
              "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
              "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple",
              "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
              "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard",
              "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors",
              "teddy bear", "hair drier", "toothbrush"]


def default_names(nc: int) -> Dict[int, str]:
    if nc == 80:
        return dict(enumerate(COCO_NAMES))
    if nc == 1:
        return {0: "person"}
    return {i: f"class{i}" for i in range(nc)}


class YOLO:
    """MI355X-native stand-in for ``ultralytics.YOLO`` (detect and pose tasks).

    ``model`` may be a ``.mi355w`` path, a ``.pt`` Ultralytics checkpoint (converted on the fly by
    ``cvsd_amd.convert``), or the raw bytes of a ``.mi355w`` image (e.g. received by broadcast).
    """

    def __init__(self, model: Union[str, bytes, os.PathLike], task: Optional[str] = None, device: int = 0,
                 batch_chunk: int = 0, verbose: bool = False, half: bool = False, fast_act: bool = False, autotune: int = 0,
                 streams: int = 0, flags: int = 0, plan_dir: Optional[str] = None, plan_cache_dir: Optional[str] = None):
        """Engine options are ``mi355_opts`` (include/mi355_yolo.h): ``fast_act`` = the v_exp / v_rcp SiLU on the fp32 path (tolerance
        mode; the default is the canonical, bit-reproducible arithmetic), ``autotune`` / ``streams`` / ``flags`` (``_lib.OPT_*``) as the
        header documents them, ``plan_dir`` = directory of shipped launch-plan files (default: the package's ``plans/``; "" = none),
        ``plan_cache_dir`` = where freshly timed choices are kept (default ~/.cache/mi355yolo; "" = not persisted)."""
        lib = _lib.lib()
        self._lock = threading.Lock()          # Ultralytics serialises predict() with a per-predictor lock
        self._h = C.c_void_p()
        self._h_other = C.c_void_p()           # engine of the other precision, created by the first predict(half=...) that needs it
        self.half = bool(half)                 # precision of the primary engine (predict(half=None) uses it)
        self.device = int(device)
        self._batch_chunk = int(batch_chunk)
        self.fast_act = bool(fast_act)
        pd = _lib.PLAN_DIR if plan_dir is None else plan_dir
        self._opt_kw = dict(batch_chunk=int(batch_chunk), fast_act=int(self.fast_act), autotune=int(autotune), streams=int(streams), flags=int(flags),
                            plan_dir=os.fsencode(pd) if pd else None,
                            plan_cache_dir=None if plan_cache_dir is None else os.fsencode(plan_cache_dir))
        opts = _lib.Opts(struct_size=C.sizeof(_lib.Opts), half=int(self.half), **self._opt_kw)
        if isinstance(model, (bytes, bytearray, memoryview)):
            blob = bytes(model)
            self.ckpt_path = None
        else:
            path = os.fspath(model)
            self.ckpt_path = path
            if not os.path.exists(path):
                raise FileNotFoundError(f"weights file not found: {path!r} (no network: nothing is downloaded)")
            if path.endswith(".pt"):
                from .convert import convert_pt
                blob = convert_pt(path)
            else:
                with open(path, "rb") as f:
                    blob = f.read()
        self._blob_meta = from_bytes(blob)[2] if blob[:8] == b"MI355YW1" else {}
        self._blob = blob
        _lib.check(lib.mi355_yolo_create_from_memory(blob, len(blob), self.device, C.byref(opts), C.byref(self._h)))
        info = _lib.ModelInfo()
        _lib.check(lib.mi355_yolo_info(self._h, C.byref(info)))
        self.info_struct = info
        self.task = "pose" if info.task == 1 else "detect"
        if task is not None and task != self.task:
            raise ValueError(f"checkpoint is a {self.task!r} model, not {task!r}")
        self.nc = info.nc
        self.kpt_shape = (info.nkpt, info.kdim)
        names = self._blob_meta.get("names")
        self.names = {int(k): v for k, v in names.items()} if names else default_names(self.nc)
        self._tracker = None

    # ------------------------------------------------------------------------------------------ construction
    @classmethod
    def from_state_dict(cls, name: str, state_dict: Dict[str, np.ndarray], nc: Optional[int] = None, **kw) -> "YOLO":
        """Build from an unfused Ultralytics-named state dict (what a ``.pt`` holds) for model ``name``."""
        return cls(build_from_state_dict(name, state_dict, nc=nc), **kw)

    def __del__(self):
        for name in ("_h", "_h_other"):
            h = getattr(self, name, None)
            if h is not None and h.value:
                try:
                    _lib.lib().mi355_yolo_destroy(h)
                except Exception:
                    pass
                setattr(self, name, C.c_void_p())

    def _handle(self, half: Optional[bool]):
        """Engine handle for the requested precision (``half=True`` = Ultralytics' predictor argument: fp16 storage)."""
        if half is None or bool(half) == self.half:
            return self._h
        if not self._h_other.value:
            opts = _lib.Opts(struct_size=C.sizeof(_lib.Opts), half=int(bool(half)), **self._opt_kw)
            _lib.check(_lib.lib().mi355_yolo_create_from_memory(self._blob, len(self._blob), self.device, C.byref(opts),
                                                                C.byref(self._h_other)))
        return self._h_other

    def info(self, detailed: bool = False, verbose: bool = False):
        i = self.info_struct
        return i.n_convs, int(i.n_params), 0, 2.0 * i.macs_640 / 1e9     # (layers, params, gradients, GFLOPs)

    # ------------------------------------------------------------------------------------------ inference
    @staticmethod
    def _as_batch(source):
        """-> (array-or-tensor [N,H,W,3] uint8, list of originals or None)"""
        if isinstance(source, torch.Tensor):
            if source.dtype != torch.uint8 or source.ndim != 4 or source.shape[-1] != 3:
                raise ValueError("tensor sources must be uint8 [N,H,W,3] BGR frames")
            return source.contiguous(), None
        if isinstance(source, np.ndarray):
            if source.ndim == 3:
                source = source[None]
            if source.dtype != np.uint8 or source.ndim != 4 or source.shape[-1] != 3:
                raise ValueError("frames must be uint8 arrays of shape [H,W,3] (BGR, as cv2 delivers them)")
            return np.ascontiguousarray(source), list(source)
        if isinstance(source, (list, tuple)):
            if not source:
                raise ValueError("empty source")
            shapes = {tuple(np.shape(f)) for f in source}
            if len(shapes) != 1:
                raise ValueError("frames of different shapes in one call are not supported; call once per shape")
            arr = np.ascontiguousarray(np.stack([np.asarray(f) for f in source]))
            return YOLO._as_batch(arr)[0], list(source)
        raise TypeError(f"unsupported source type {type(source).__name__}: pass decoded BGR uint8 frames "
                        f"(frame decode stays on the host, as in the reference's cv2.VideoCapture loop)")

    class _DeviceFrames:
        """n dense BGR uint8 frames that already sit on the engine's GPU, known by raw pointer (YOLO.track: the copy the tracker's motion
        compensation uploaded)."""
        def __init__(self, ptr: int, n: int, h: int, w: int):
            self.ptr, self.shape = int(ptr), (int(n), int(h), int(w), 3)

    def _infer_rows(self, batch, conf, iou, classes, max_det, imgsz, half=None):
        lib = _lib.lib()
        hnd = self._handle(half)
        self._last_handle = hnd
        n, h, w = int(batch.shape[0]), int(batch.shape[1]), int(batch.shape[2])
        rows = np.empty((n, max_det, _lib.DET_WORDS), dtype=np.float32)    # only rows[i, :counts[i]] are written / meaningful
        counts = np.zeros(n, dtype=np.int32)
        cls_arr = None
        ncls = 0
        if classes is not None:
            cl = [int(classes)] if np.isscalar(classes) else [int(c) for c in classes]
            cls_arr = (C.c_int * len(cl))(*cl)
            ncls = len(cl)
        cp = counts.ctypes.data_as(C.POINTER(C.c_int))
        with self._lock:
            if isinstance(batch, YOLO._DeviceFrames):
                _lib.check(lib.mi355_yolo_infer_device(hnd, batch.ptr, n, h, w, conf, iou, cls_arr, ncls, max_det, imgsz, rows.ctypes.data, max_det, cp))
                return rows, counts, (h, w)
            if isinstance(batch, torch.Tensor):
                if not batch.is_cuda:
                    batch = batch.numpy()
                else:
                    if batch.device.index != self.device:
                        raise ValueError("frames live on a different GPU than the engine")
                    torch.cuda.current_stream(batch.device).synchronize()   # engine runs on its own stream
                    _lib.check(lib.mi355_yolo_infer_device(hnd, batch.data_ptr(), n, h, w, conf, iou, cls_arr, ncls,
                                                           max_det, imgsz, rows.ctypes.data, max_det, cp))
                    return rows, counts, (h, w)
            _lib.check(lib.mi355_yolo_infer(hnd, batch.ctypes.data, n, h, w, 0, conf, iou, cls_arr, ncls, max_det,
                                            imgsz, rows.ctypes.data, max_det, cp))
        return rows, counts, (h, w)

    # ---- device-resident results (multi-GPU pipelines): nothing crosses PCIe, nothing blocks --------------------------
    def new_device_rows(self, n: int, max_det: int = 300):
        """Output buffers for :meth:`infer_async`: (rows [n*max_det, 58] f32, counts [n] i32, total [1] i32) on the engine's GPU."""
        dev = torch.device("cuda", self.device)
        return (torch.empty((n * max_det, _lib.DET_WORDS), dtype=torch.float32, device=dev),
                torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))

    def infer_async(self, frames: torch.Tensor, out, conf: float = 0.25, iou: float = 0.7, classes=None, max_det: int = 300,
                    imgsz: int = 640, half=None) -> None:
        """Enqueue one batch of CUDA-resident uint8 frames [n,H,W,3]; the packed post-NMS rows (frame order), the per-frame
        counts and their sum land in ``out`` (from :meth:`new_device_rows`) -- all in HBM.  Returns at once; order other
        streams behind :attr:`stream` (or call :meth:`sync`) before reading ``out``."""
        rows, counts, total = out
        n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        if not frames.is_cuda or frames.dtype != torch.uint8 or frames.device.index != self.device:
            raise ValueError("infer_async needs uint8 frames on the engine's GPU")
        if rows.shape[0] < n * max_det or counts.numel() < n:
            raise ValueError("output buffers are too small for n * max_det rows")
        cls_arr, ncls = None, 0
        if classes is not None:
            cl = [int(classes)] if np.isscalar(classes) else [int(c) for c in classes]
            cls_arr, ncls = (C.c_int * len(cl))(*cl), len(cl)
        hnd = self._handle(half)
        self._async_handle = hnd              # `stream` / `sync()` refer to the engine that received the last asynchronous call
        with self._lock:
            torch.cuda.current_stream(frames.device).synchronize()       # the frames must be complete; the engine has its own stream
            _lib.check(_lib.lib().mi355_yolo_infer_device_async(hnd, frames.data_ptr(), n, h, w, float(conf), float(iou), cls_arr, ncls,
                                                                int(max_det), int(imgsz), rows.data_ptr(), counts.data_ptr(),
                                                                total.data_ptr()))

    @property
    def stream(self) -> "torch.cuda.ExternalStream":
        """The HIP stream of the engine that received the last :meth:`infer_async` call (the primary engine before any), as
        a torch stream: ``torch.cuda.current_stream().wait_stream(model.stream)``.  A ``half=`` override runs on the second
        engine, which has a stream of its own -- this property follows it."""
        hnd = getattr(self, "_async_handle", None) or self._h
        cache = self.__dict__.setdefault("_ext_streams", {})
        if hnd.value not in cache:
            cache[hnd.value] = torch.cuda.ExternalStream(int(_lib.lib().mi355_yolo_stream(hnd)), device=torch.device("cuda", self.device))
        return cache[hnd.value]

    def sync(self) -> None:
        """Wait for every asynchronous call enqueued so far (both engines, when a ``half=`` override created the second)."""
        for hnd in (self._h, self._h_other):
            if hnd.value:
                _lib.check(_lib.lib().mi355_yolo_sync(hnd))

    def predict(self, source=None, conf: Optional[float] = None, iou: float = 0.7, classes=None, max_det: int = 300,
                imgsz: int = 640, half: Optional[bool] = None, verbose: bool = False, stream: bool = False, **kwargs
                ) -> List[Results]:
        """``model.predict`` / ``model(...)``: ultralytics/engine/model.py:Model.predict.
        ``half=True`` runs the fp16-storage engine (fp32 accumulate; results differ from fp32 by fp16 rounding);
        ``None`` = the precision the model was constructed with (fp32 unless ``YOLO(..., half=True)``)."""
        if isinstance(imgsz, (list, tuple)):
            imgsz = int(max(imgsz))
        conf = 0.25 if conf is None else float(conf)
        batch, originals = self._as_batch(source)
        return self._predict_batch(batch, originals, conf, iou, classes, max_det, imgsz, half)

    def detect_rows(self, batch, conf: float = 0.25, iou: float = 0.7, classes=None, max_det: int = 300, imgsz: int = 640, half=None):
        """``Results.boxes.data`` of every frame of ``batch`` -- float32 [M, 6] rows x1, y1, x2, y2, conf, cls -- and the frames' (h, w),
        without building the Results objects (a sweep reads nothing else: cvsd_amd/sweep.py).  ``batch``: what :meth:`predict` takes
        after stacking, or frames already on the engine's GPU (:class:`_DeviceFrames`)."""
        rows, counts, shape = self._infer_rows(batch, float(conf), float(iou), classes, int(max_det), int(imgsz), half)
        out = []
        for i in range(len(counts)):
            r = rows[i, :counts[i]]
            d = np.empty((len(r), 6), np.float32)
            d[:, :5] = r[:, :5]
            d[:, 5] = r[:, 5:6].view(np.int32)[:, 0]
            out.append(d)
        return out, shape

    def _predict_batch(self, batch, originals, conf, iou, classes, max_det, imgsz, half) -> List[Results]:
        rows, counts, shape = self._infer_rows(batch, conf, float(iou), classes, int(max_det), int(imgsz), half)
        t = _lib.Timing()
        _lib.lib().mi355_yolo_last_timing(self._last_handle, C.byref(t))
        per_img_ms = t.total_ms / max(1, int(batch.shape[0]))
        out = []
        for i in range(len(counts)):
            r = rows[i, :counts[i]]
            ints = r[:, 5:7].view(np.int32)
            data = np.concatenate([r[:, :5], ints[:, :1].astype(np.float32)], axis=1)       # x1,y1,x2,y2,conf,cls
            kp = None
            if self.task == "pose":
                nk = self.kpt_shape[0] * self.kpt_shape[1]
                kp = torch.from_numpy(r[:, 7:7 + nk].reshape(len(r), *self.kpt_shape).copy())
            res = Results(originals[i] if originals is not None else None, f"image{i}.jpg", self.names,
                          boxes=torch.from_numpy(data), keypoints=kp.clone() if kp is not None else None, orig_shape=shape,
                          speed={"preprocess": 0.0, "inference": per_img_ms, "postprocess": 0.0},
                          anchor_idx=ints[:, 1].copy())
            if kp is not None:
                res.keypoints_raw = kp.numpy()          # before Keypoints() zeroes x,y of joints with conf < 0.5
            out.append(res)
        return out

    __call__ = predict

    def track(self, source=None, persist: bool = False, show: bool = False, conf: Optional[float] = None, **kwargs
              ) -> List[Results]:
        """``model.track`` (``/root/reference/model.py:38``): predict at conf 0.1, then the tracker callback adds
        ids.  ultralytics/engine/model.py:Model.track forces batch 1; frames are handled one at a time, in order."""
        from .tracker import BYTETracker
        kwargs.pop("batch", None)
        conf = 0.1 if conf is None else conf
        if self._tracker is None or not persist:
            # the frame preparation and the optical flow of its motion compensation run on the engine's GPU (csrc/gmc_kernels.hip)
            self._tracker = BYTETracker(gmc_device=getattr(self, "device", None))
        batch, originals = self._as_batch(source)
        results = []
        for i in range(int(batch.shape[0])):
            frame = originals[i] if originals is not None else batch[i].cpu().numpy()
            # the tracker's motion compensation for this frame (frame preparation + optical flow, a stream of its own on the GPU) is
            # enqueued before the detector pass and collected inside tracker.update: the two share the GPU instead of queueing
            self._tracker.gmc.begin(frame)
            # ... and the detector pass reads the copy of the frame that step has just put on the GPU: one upload, and no second host -> device
            # copy queueing behind the step's device -> host copies (which wait for its Lucas-Kanade launch -- the two used to run one after
            # the other, tools/track_timeline.py)
            share = os.environ.get("MI355_TRACK_SHARED_FRAME", "1") != "0"            # A/B and tests only
            dev = self._tracker.gmc.pending_device_frame() if share and isinstance(frame, np.ndarray) and frame.ndim == 3 else None
            if dev is not None and dev[1:] == tuple(frame.shape[:2]) and not kwargs.get("stream"):
                imgsz = kwargs.get("imgsz", 640)
                res = self._predict_batch(YOLO._DeviceFrames(dev[0], 1, dev[1], dev[2]), None, float(conf), kwargs.get("iou", 0.7), kwargs.get("classes"),
                                          kwargs.get("max_det", 300), int(max(imgsz)) if isinstance(imgsz, (list, tuple)) else int(imgsz), kwargs.get("half"))[0]
            else:
                res = self.predict(batch[i:i + 1], conf=conf, **kwargs)[0]
            if originals is not None:
                res.orig_img = originals[i]
            # trackers/track.py:on_predict_postprocess_end: the tracker steps on EVERY frame (an empty frame still advances
            # frame_id, ages lost tracks against track_buffer and runs the Kalman predict); only the rewrite of the
            # result is skipped when no track comes back
            tracks = self._tracker.update(res.boxes.data.numpy(), frame)   # trackers/track.py: tracker.update(det, im0)
            if len(tracks):
                idx = tracks[:, -1].astype(int)
                res = res[idx]
                res.update(boxes=torch.as_tensor(tracks[:, :-1], dtype=torch.float32))
            results.append(res)
        return results

    # ------------------------------------------------------------------------------------------ test hooks
    def raw_head(self, source, imgsz: int = 640, half: Optional[bool] = None) -> np.ndarray:
        """Pre-NMS head tensor ``[N, 4+nc+nk, A]`` exactly as ``Detect/Pose.forward`` returns it."""
        lib = _lib.lib()
        hnd = self._handle(half)
        batch, _ = self._as_batch(source)
        if isinstance(batch, torch.Tensor):
            batch = batch.cpu().numpy()
        n, h, w = batch.shape[:3]
        ch, an = C.c_int(), C.c_int()
        _lib.check(lib.mi355_yolo_raw_head(hnd, None, n, h, w, 0, imgsz, None, C.byref(ch), C.byref(an)))
        out = np.empty((n, ch.value, an.value), dtype=np.float32)
        with self._lock:
            _lib.check(lib.mi355_yolo_raw_head(hnd, batch.ctypes.data, n, h, w, 0, imgsz, out.ctypes.data,
                                               C.byref(ch), C.byref(an)))
        return out

    def plan_info(self) -> dict:
        """Launch plans of the shape last run: hash (candidates + choices), where the choices came from, launches per pass and
        the activation footprint in bytes."""
        hsh, src, nl, ab, au = C.c_ulonglong(), C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong()
        _lib.check(_lib.lib().mi355_yolo_plan_info(getattr(self, "_last_handle", self._h), C.byref(hsh), C.byref(src), C.byref(nl), C.byref(ab),
                                                   C.byref(au)))
        return {"plan_hash": f"{hsh.value:016x}", "plan_source": ("static", "memory", "cache", "tuned", "file")[src.value] if 0 <= src.value <= 4 else str(src.value),
                "launches_per_pass": nl.value, "activation_bytes": ab.value, "activation_bytes_unshared": au.value}

    def set_profiling(self, on: bool = True) -> None:
        _lib.check(_lib.lib().mi355_yolo_set_profiling(self._h, int(on)))
        if self._h_other.value:
            _lib.check(_lib.lib().mi355_yolo_set_profiling(self._h_other, int(on)))

    def last_timing(self) -> dict:
        t = _lib.Timing()
        _lib.check(_lib.lib().mi355_yolo_last_timing(getattr(self, "_last_handle", self._h), C.byref(t)))
        return {k: getattr(t, k) for k, _ in _lib.Timing._fields_}
