"""Shared helpers for the ME parity tests (oracle side + HIP side)."""
import ctypes as C

import numpy as np

from conftest import load_package, ptr

pkg = load_package()
synth = __import__("importlib").import_module("svt_av1_amd.synth")


class OrcSbSearch(C.Structure):
    _fields_ = [("sb_x", C.c_int32), ("sb_y", C.c_int32), ("x_origin", C.c_int16), ("y_origin", C.c_int16),
                ("width", C.c_int16), ("height", C.c_int16)]


class OrcSearchWindow(C.Structure):
    _fields_ = [("x_origin", C.c_int16), ("y_origin", C.c_int16), ("width", C.c_int16), ("height", C.c_int16)]


def windows(orc, width, height, sa_w, sa_h, centers=None):
    """Per-SB search windows through the oracle's restatement of integer_search_sb's clamp."""
    orc.orc_me_search_window.restype = OrcSearchWindow
    orc.orc_me_search_window.argtypes = [C.c_int] * 8
    sbs = synth.sb_grid(width, height)
    arr = (OrcSbSearch * len(sbs))()
    for i, (x, y) in enumerate(sbs):
        cx, cy = centers[i] if centers is not None else (0, 0)
        v = orc.orc_me_search_window(x, y, int(cx), int(cy), sa_w, sa_h, width, height)
        arr[i] = OrcSbSearch(x, y, v.x_origin, v.y_origin, v.width, v.height)
    return arr


def windows_product(L, width, height, sa_w, sa_h, centers=None):
    """The same windows through the product's own host function (svt_hip_me_search_window, include/svt_hip.h): what a caller of the ABI uses."""
    sbs = synth.sb_grid(width, height)
    arr = (OrcSbSearch * len(sbs))()      # same record layout as SvtHipSbSearch
    for i, (x, y) in enumerate(sbs):
        cx, cy = centers[i] if centers is not None else (0, 0)
        v = L.svt_hip_me_search_window(x, y, int(cx), int(cy), sa_w, sa_h, width, height)
        arr[i] = OrcSbSearch(x, y, v.x_origin, v.y_origin, v.width, v.height)
    return arr


def oracle_frame(orc, cur_p, ref_p, stride, pad, sbs, sub_sad, begin=0, end=None):
    n = len(sbs)
    end = n if end is None else end
    sad = np.zeros((n, 85), np.uint32)
    mv = np.zeros((n, 85), np.uint32)
    orc.orc_me_fullpel_frame(ptr(cur_p), ptr(ref_p), stride, pad, pad, sbs, n, sub_sad, ptr(sad), ptr(mv), begin, end)
    return sad, mv


def hip_frame(hip, cur_p, ref_p, stride, pad, sbs, sub_sad):
    n = len(sbs)
    sad = np.zeros((n, 85), np.uint32)
    mv = np.zeros((n, 85), np.uint32)
    hip.check(hip.L.svt_hip_me_fullpel_frame(hip.h, ptr(cur_p), ptr(ref_p), stride, cur_p.shape[0], pad, pad,
                                            C.cast(sbs, C.c_void_p), n, sub_sad, ptr(sad), ptr(mv)), "me_fullpel_frame")
    return sad, mv
