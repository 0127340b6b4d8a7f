"""ctypes binding of include/mi355_yolo.h (libmi355yolo.so).

There is no CPU fallback: if the shared library has not been built (``python -m cvsd_amd.build``)
importing this module's symbols raises, and every engine call needs a visible MI355X.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_YOLO_LIB: load another build of the same ABI (kernel experiments: tools/ab_build.sh); the product never sets it
LIB_PATH = os.environ.get("MI355_YOLO_LIB") or os.path.join(HERE, "libmi355yolo.so")
MAX_KPT_FLOATS = 51


class Mi355Error(RuntimeError):
    pass


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("batch_chunk", C.c_int), ("half", C.c_int), ("fast_act", C.c_int), ("autotune", C.c_int),
                ("streams", C.c_int), ("flags", C.c_int), ("reserved", C.c_int), ("plan_dir", C.c_char_p), ("plan_cache_dir", C.c_char_p)]


# mi355_opts.flags (include/mi355_yolo.h)
OPT_NO_FUSE_UPSAMPLE, OPT_NO_FUSE_1X1, OPT_NO_FUSE_TAIL, OPT_NO_GROUPS = 0x01, 0x02, 0x04, 0x08
OPT_NO_MEM_REUSE, OPT_HIP_GRAPH, OPT_NO_DIRECT_ROWS, OPT_NO_PASS_TUNE = 0x10, 0x20, 0x40, 0x80
# launch-plan files shipped with the package: the tuned choices of the benchmarked workloads on MI355X (plans/README.md)
PLAN_DIR = os.path.join(HERE, "plans")


class Det(C.Structure):
    _fields_ = [("x1", C.c_float), ("y1", C.c_float), ("x2", C.c_float), ("y2", C.c_float), ("conf", C.c_float),
                ("cls", C.c_int), ("anchor_idx", C.c_int), ("kpt", C.c_float * MAX_KPT_FLOATS)]


class ModelInfo(C.Structure):
    _fields_ = [("task", C.c_int), ("nc", C.c_int), ("nkpt", C.c_int), ("kdim", C.c_int), ("reg_max", C.c_int),
                ("n_levels", C.c_int), ("strides", C.c_int * 4), ("n_convs", C.c_int), ("n_ops", C.c_int),
                ("n_buffers", C.c_int), ("n_params", C.c_longlong), ("macs_640", C.c_longlong),
                ("family", C.c_char * 8), ("scale", C.c_char), ("pad_", C.c_char * 7)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("conv_ms", C.c_float), ("stem_ms", C.c_float), ("pool_ms", C.c_float),
                ("upsample_ms", C.c_float), ("letterbox_ms", C.c_float), ("decode_ms", C.c_float),
                ("nms_ms", C.c_float), ("conv_launches", C.c_int), ("frames", C.c_int)]


DET_WORDS = C.sizeof(Det) // 4          # 58 32-bit words per row
_P = C.POINTER
_u8p, _f32p, _i32p = _P(C.c_uint8), _P(C.c_float), _P(C.c_int)

# symbol -> (restype, argtypes); every function declared in include/mi355_yolo.h
SIGNATURES = {
    "mi355_last_error": (C.c_char_p, []),
    "mi355_yolo_create": (C.c_int, [C.c_char_p, C.c_int, _P(Opts), _P(C.c_void_p)]),
    "mi355_yolo_create_from_memory": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, _P(Opts), _P(C.c_void_p)]),
    "mi355_yolo_destroy": (None, [C.c_void_p]),
    "mi355_yolo_info": (C.c_int, [C.c_void_p, _P(ModelInfo)]),
    "mi355_yolo_infer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                   _i32p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, _i32p]),
    "mi355_yolo_infer_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                          _i32p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, _i32p]),
    "mi355_yolo_infer_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                                _i32p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_yolo_stream": (C.c_void_p, [C.c_void_p]),
    "mi355_yolo_sync": (C.c_int, [C.c_void_p]),
    "mi355_yolo_raw_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      _i32p, _i32p]),
    "mi355_yolo_plan_info": (C.c_int, [C.c_void_p, _P(C.c_ulonglong), _i32p, _i32p, _P(C.c_longlong), _P(C.c_longlong)]),
    "mi355_memory_plan": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P(C.c_longlong),
                                    _P(C.c_longlong), C.c_int, _i32p, _P(C.c_longlong), _P(C.c_longlong)]),
    "mi355_yolo_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mi355_yolo_last_timing": (C.c_int, [C.c_void_p, _P(Timing)]),
    "mi355_op_conv2d": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, _i32p]),
    "mi355_op_conv2d_f16": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i32p]),
    "mi355_op_conv1x1_upcat": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_void_p, C.c_int, _i32p]),
    "mi355_op_conv1x1_upcat_f16": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_void_p, C.c_int, _i32p]),
    "mi355_op_conv2d_fused": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, _i32p]),
    "mi355_op_c2f_tail": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, _i32p]),
    "mi355_op_conv2d_fused_f16": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, _i32p]),
    "mi355_op_conv2d_group": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i32p, _i32p,
                                        C.c_void_p, C.c_void_p, C.c_int]),
    "mi355_bench_conv2d": (C.c_int, [C.c_int] * 12 + [_f32p, _i32p, C.c_char_p, C.c_int]),
    "mi355_bench_conv2d_f16": (C.c_int, [C.c_int] * 12 + [_f32p, _i32p, C.c_char_p, C.c_int]),
    "mi355_plan_query": (C.c_int, [C.c_int] * 13 + [_i32p, C.c_int, _i32p]),
    "mi355_gmc_pyr_lk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                   C.c_double, C.c_void_p, C.c_void_p]),
    "mi355_gmc_pyr_lk_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "mi355_gmc_prepare_device": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_gmc_order_corners": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mi355_gmc_affine_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_ulonglong, C.c_void_p,
                                           C.c_void_p]),
    "mi355_gmc_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mi355_gmc_destroy": (None, [C.c_void_p]),
    "mi355_gmc_step_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]),
    "mi355_gmc_step_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_gmc_pending_frame": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _i32p, _i32p]),
    "mi355_tracker_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mi355_tracker_destroy": (None, [C.c_void_p]),
    "mi355_tracker_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "mi355_tracker_last_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mi355_tracker_state": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i32p]),
    "mi355_tracker_tracks": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "mi355_kalman_initiate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_kalman_predict": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_kalman_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_kalman_warp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_lapjv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]),
    "mi355_gmc_prepare_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "mi355_gmc_track_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mi355_gmc_track_finish": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_gmc_track_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mi355_gmc_batch_frames": (C.c_int, [C.c_void_p, C.c_ulonglong, C.c_int, C.POINTER(C.c_void_p), _i32p, _i32p, _i32p, C.POINTER(C.c_longlong)]),
    "mi355_gmc_batch_seq": (C.c_ulonglong, [C.c_void_p]),
    "mi355_gmc_track_reset": (C.c_int, [C.c_void_p]),
    "mi355_gmc_track_state": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, C.c_void_p, C.c_void_p, C.c_int]),
    "mi355_op_stem": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_int, C.c_int, C.c_void_p]),
    "mi355_op_stem_f16": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mi355_letterbox_shape": (C.c_int, [C.c_int, C.c_int, C.c_int, _i32p, _i32p]),
    "mi355_op_letterbox": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mi355_op_nms": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _i32p,
                               C.c_int, C.c_int, C.c_void_p, C.c_int, _i32p]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libmi355yolo.so (once). Raises Mi355Error if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355Error(f"{LIB_PATH} is missing: build the HIP extension first "
                             f"(python -m cvsd_amd.build, or __graft_entry__.build()). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().mi355_last_error().decode(errors="replace")
        if rc == -2:
            raise FileNotFoundError(msg)
        if rc in (-1, -3):
            raise ValueError(msg)
        raise Mi355Error(f"mi355 error {rc}: {msg}")
