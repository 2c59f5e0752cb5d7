"""ctypes binding of libctpn_hip.so (C ABI declared in include/ctpn_hip.h).

`cffi` is what BASELINE.json names but it is not installed in this image (and there is no network), so the
binding uses `ctypes`; the declarations below are a 1:1 transcription of the header and
tests/test_abi.py checks that every symbol the header declares is exported.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CTPN_OK = 0
CTPN_ERR_UNSUPPORTED = -6
PREC_FP32, PREC_BF16, PREC_FP16, PREC_SPLIT = 0, 1, 2, 3
PRECISIONS = {"fp32": PREC_FP32, "f32": PREC_FP32, "bf16": PREC_BF16, "fp16": PREC_FP16, "f16": PREC_FP16, "split": PREC_SPLIT}


def precision_code(p):
    if isinstance(p, int) and p in (PREC_FP32, PREC_BF16, PREC_FP16, PREC_SPLIT):
        return p
    try:
        return PRECISIONS[p]
    except KeyError:
        raise ValueError("unknown precision %r (fp32 | bf16 | fp16 | split)" % (p,))


# Per-ctx options of the C ABI (ctpn_set_option). The library itself reads none of them from the environment; for command-line use the
# BINDING maps these variables onto the option of every Context it creates (explicit options= win):
OPTION_ENV = {"keep_acts": "CTPN_KEEP_ACTS", "conv1_kernel": "CTPN_CONV1_MFMA", "conv1_fuse": "CTPN_CONV1_FUSE", "lstm_split": "CTPN_LSTM_SPLIT", "nms_columns": "CTPN_NMS_COLUMNS",
              "nms_check": "CTPN_NMS_CHECK", "connect_device": "CTPN_CONNECT_DEVICE", "tail_overlap": "CTPN_TAIL_OVERLAP", "conv_p64": "CTPN_CONV_P64",
              "tail_confine": "CTPN_TAIL_CONFINE", "nms_prefix": "CTPN_NMS_PREFIX", "debug_hog": "CTPN_DEBUG_HOG", "debug_nms": "CTPN_DEBUG_NMS", "split_edge": "CTPN_SPLIT_EDGE"}
MODE_H, MODE_O = 0, 1
KIND_NAMES = ["conv_first", "conv_gemm", "pool", "gemm", "bilstm", "decode", "sort", "nms"]


class CtpnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libctpn_hip error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    # CTPN_LIB_PATH: an alternative build of the same library (kernel A/B experiments); the default is the in-tree build
    return os.environ.get("CTPN_LIB_PATH") or os.path.join(_HERE, "libctpn_hip.so")


def _declare(lib):
    u8p, f32p, f64p, i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int))
    vp = C.c_void_p
    sig = {
        "ctpn_abi_version": (C.c_int, []),
        "ctpn_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
        "ctpn_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
        "ctpn_option_count": (C.c_int, []),
        "ctpn_option_name": (C.c_char_p, [C.c_int]),
        "ctpn_last_error": (C.c_char_p, []),
        "ctpn_device_count": (C.c_int, []),
        "ctpn_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ctpn_create_postproc": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]),
        "ctpn_host_thread_budget": (C.c_int, [C.c_int, C.c_int, C.c_int]),
        "ctpn_host_threads": (C.c_int, [vp, i32p]),
        "ctpn_proposal_anchors": (C.c_int, [vp, i32p, C.c_int]),
        "ctpn_debug_connect": (C.c_int, [C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, f64p, C.c_int, i32p]),
        "ctpn_destroy": (C.c_int, [vp]),
        "ctpn_sync": (C.c_int, [vp]),
        "ctpn_stream": (C.c_int, [vp, C.POINTER(vp)]),
        "ctpn_weight_manifest": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), i32p, i32p, C.POINTER(C.c_size_t)]),
        "ctpn_weight_count": (C.c_int, []),
        "ctpn_load_weights_host": (C.c_int, [vp, f32p]),
        "ctpn_load_weights_device": (C.c_int, [vp, vp]),
        "ctpn_broadcast_weights": (C.c_int, [C.POINTER(vp), C.c_int]),
        "ctpn_comm_unique_id": (C.c_int, [C.c_char_p, C.c_size_t]),
        "ctpn_broadcast_weights_rank": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int, C.c_int]),
        "ctpn_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ctpn_forward_blob": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ctpn_feat_shape": (C.c_int, [vp, i32p, i32p, i32p]),
        "ctpn_get_tensor": (C.c_int, [vp, C.c_char_p, f32p, C.c_size_t, i32p]),
        "ctpn_proposals": (C.c_int, [vp, f32p, C.c_int, C.c_int, C.c_float, C.c_float, f32p, i32p]),
        "ctpn_proposals_from_host": (C.c_int, [vp, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int,
                                               C.c_float, C.c_float, f32p, i32p]),
        "ctpn_nms": (C.c_int, [i32p, i32p, f32p, C.c_int, C.c_int, C.c_float, C.c_int]),
        "ctpn_resize_dims": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, i32p, i32p]),
        "ctpn_resize": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp, C.c_int,
                                  C.c_longlong, i32p, i32p]),
        "ctpn_result_text": (C.c_int, [f64p, C.c_int, C.c_double, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), i32p]),
        "ctpn_write_result_file": (C.c_int, [C.c_char_p, f64p, C.c_int, C.c_double, i32p]),
        "ctpn_draw_boxes": (C.c_int, [u8p, C.c_int, C.c_int, f64p, C.c_int]),
        "ctpn_text_lines": (C.c_int, [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f64p, C.c_int, i32p]),
        "ctpn_connector_constants": (C.c_int, [f64p]),
        "ctpn_detect": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f64p, C.c_int, i32p,
                                  f32p, i32p]),
        "ctpn_detect_submit": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int]),
        "ctpn_detect_collect": (C.c_int, [vp, C.c_int, C.c_int, f64p, C.c_int, i32p, f32p, i32p]),
        "ctpn_debug_cvt_bf16": (C.c_int, [C.c_int, f32p, C.POINTER(C.c_uint16), C.c_int, C.c_int]),
        "ctpn_debug_lds_dma": (C.c_int, [C.c_int, u8p, C.c_size_t, u8p, u8p]),
        "ctpn_debug_conv3x3": (C.c_int, [C.c_int, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, f32p, f32p]),
        "ctpn_jpeg_probe": (C.c_int, [u8p, C.c_size_t, i32p, i32p, i32p, i32p]),
        "ctpn_jpeg_coef_capacity": (C.c_size_t, [C.c_int, C.c_int]),
        "ctpn_jpeg_entropy_decode": (C.c_int, [u8p, C.c_size_t, C.POINTER(C.c_int16), C.c_size_t, C.POINTER(C.c_uint16), i32p]),
        "ctpn_decode_jpeg_batch": (C.c_int, [vp, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                             C.POINTER(vp), i32p, i32p]),
        "ctpn_jpeg_batch_fetch": (C.c_int, [vp, vp, u8p, C.c_size_t]),
        "ctpn_decode_jpeg_files": (C.c_int, [vp, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(vp), i32p, i32p]),
        "ctpn_jpeg_probe_files": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, i32p, C.c_int]),
        "ctpn_debug_png_backend": (C.c_int, [C.c_int]),
        "ctpn_png_probe": (C.c_int, [u8p, C.c_size_t, i32p, i32p, i32p, i32p]),
        "ctpn_png_decode": (C.c_int, [u8p, C.c_size_t, u8p, C.c_size_t]),
        "ctpn_png_probe_files": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, i32p, C.c_int]),
        "ctpn_decode_png_files": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, u8p, C.c_int]),
        "ctpn_profile_enable": (C.c_int, [vp, C.c_int]),
        "ctpn_profile_reset": (C.c_int, [vp]),
        "ctpn_profile_read": (C.c_int, [vp, C.c_int, f64p, C.POINTER(C.c_longlong), f64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sig


def load_library():
    """Load libctpn_hip.so once. Raises CtpnError if it has not been built (no silent fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise CtpnError(-5, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    # torch bundles its own libamdhip64 (soname libamdhip64.so.7). If torch is (going to be) in this process,
    # it must be loaded first so that both share ONE HIP runtime; see DESIGN.md "HIP runtime sharing".
    if "torch" not in sys.modules and os.environ.get("CTPN_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    _LIB = C.CDLL(path, mode=C.RTLD_GLOBAL)
    _declare(_LIB)
    return _LIB


def _check(rc):
    if rc != CTPN_OK:
        raise CtpnError(rc, load_library().ctpn_last_error().decode("utf-8", "replace"))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def device_count():
    return load_library().ctpn_device_count()


def option_names():
    lib = load_library()
    return [lib.ctpn_option_name(i).decode() for i in range(lib.ctpn_option_count())]


def manifest_from_library():
    lib = load_library()
    out = []
    for i in range(lib.ctpn_weight_count()):
        name = C.c_char_p()
        rank = C.c_int()
        shape = (C.c_int * 4)()
        off = C.c_size_t()
        _check(lib.ctpn_weight_manifest(i, C.byref(name), C.byref(rank), shape, C.byref(off)))
        out.append((name.value.decode(), tuple(shape[: rank.value]), off.value))
    return out


def nms_sorted(boxes_sorted, thresh, device_id=0):
    """B1 seam: same contract as the reference `_nms` (lib/utils/gpu_nms.hpp:1-2) on score-sorted rows."""
    lib = load_library()
    b = _f32(boxes_sorted)
    n = int(b.shape[0])
    if n == 0:
        return np.zeros((0,), np.int32)
    keep = np.zeros((n,), np.int32)
    num = C.c_int(0)
    _check(lib.ctpn_nms(_ptr(keep, C.c_int), C.byref(num), _ptr(b, C.c_float), n, int(b.shape[1]), float(thresh),
                        int(device_id)))
    return keep[: num.value]


def resize_dims(h, w, fx, fy):
    """Output size of cv2.resize(fx, fy): round-half-even(src * f). Host arithmetic of the library, needs no GPU."""
    lib = load_library()
    oh, ow = C.c_int(0), C.c_int(0)
    _check(lib.ctpn_resize_dims(int(h), int(w), float(fx), float(fy), C.byref(oh), C.byref(ow)))
    return oh.value, ow.value


def resize_linear(im, fx, fy, device_id=0):
    """cv2.resize(im, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) on the GPU (ctpn_resize): im is (h,w,3) or
    (n,h,w,3), uint8 or float32; returns the same rank and dtype."""
    lib = load_library()
    a = np.ascontiguousarray(im)
    if a.dtype not in (np.uint8, np.float32):
        a = a.astype(np.float32)
    single = a.ndim == 3
    if single:
        a = a[None]
    if a.ndim != 4 or a.shape[3] != 3:
        raise ValueError("resize_linear wants (h,w,3) or (n,h,w,3)")
    n, h, w, _ = a.shape
    oh, ow = C.c_int(0), C.c_int(0)
    _check(lib.ctpn_resize_dims(h, w, float(fx), float(fy), C.byref(oh), C.byref(ow)))
    out = np.empty((n, oh.value, ow.value, 3), a.dtype)
    _check(lib.ctpn_resize(int(device_id), a.ctypes.data_as(C.c_void_p), 1 if a.dtype == np.float32 else 0, 0, n, h, w, float(fx), float(fy),
                           out.ctypes.data_as(C.c_void_p), 0, out.size, C.byref(oh), C.byref(ow)))
    return out[0] if single else out


def _bytes_ptr(data):
    """bytes / bytearray / uint8 array -> (keep-alive object, POINTER(c_uint8), length) without copying bytes objects."""
    if isinstance(data, bytes):
        return data, C.cast(C.c_char_p(data), C.POINTER(C.c_uint8)), len(data)
    a = np.ascontiguousarray(np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size


def jpeg_probe(data):
    """(h, w, components, layout) of one JPEG file's bytes (ctpn_jpeg_probe; host only). h, w: the size cv2.imread returns (an EXIF
    orientation 5 .. 8 swaps the stored ones). layout & 0xff = luma sampling: 1 (4:4:4, gray), 2 (4:2:0), 0x21 (4:2:2: 2 horizontally, 1
    vertically), 0x12 (4:4:0); layout >> 8 = EXIF orientation - 1. CtpnError with code CTPN_ERR_UNSUPPORTED for well-formed files the
    device decoder does not take (CMYK, 4:1:1, arithmetic coding, 12-bit ...)."""
    lib = load_library()
    keep, ptr, n = _bytes_ptr(data)
    h, w, nc, hs = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    _check(lib.ctpn_jpeg_probe(ptr, n, C.byref(h), C.byref(w), C.byref(nc), C.byref(hs)))
    return h.value, w.value, nc.value, hs.value


def _path_array(paths):
    enc = [os.fsencode(p) for p in paths]
    return enc, (C.c_char_p * len(enc))(*enc)


def jpeg_probe_files(paths, threads=0):
    """Header scan of many files in one call (ctpn_jpeg_probe_files, C++ threads): (n, 4) int32 rows (h, w, components, luma sampling);
    h = 0 for files the device decoder does not take."""
    lib = load_library()
    paths = list(paths)
    keep, arr = _path_array(paths)
    out = np.zeros((len(paths), 4), np.int32)
    _check(lib.ctpn_jpeg_probe_files(arr, len(paths), _ptr(out, C.c_int), int(threads)))
    return out


def png_probe(data):
    """(h, w, colour type, bit depth) of one PNG file's bytes (ctpn_png_probe; host only). CtpnError(CTPN_ERR_UNSUPPORTED) for 16-bit files."""
    lib = load_library()
    keep, ptr, n = _bytes_ptr(data)
    h, w, ct, bd = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    _check(lib.ctpn_png_probe(ptr, n, C.byref(h), C.byref(w), C.byref(ct), C.byref(bd)))
    return h.value, w.value, ct.value, bd.value


def png_backend(zlib_only=-1):
    """'libdeflate' or 'zlib': the DEFLATE back end ctpn_png_decode uses; zlib_only = 1 / 0 forces zlib / lifts that (test hook)."""
    return "libdeflate" if load_library().ctpn_debug_png_backend(int(zlib_only)) else "zlib"


def png_decode(data):
    """cv2.imread(IMREAD_COLOR) of one PNG file's bytes: (h, w, 3) BGR uint8 (ctpn_png_decode; host only by the nature of the format)."""
    lib = load_library()
    h, w, _, _ = png_probe(data)
    keep, ptr, n = _bytes_ptr(data)
    out = np.zeros((h, w, 3), np.uint8)
    _check(lib.ctpn_png_decode(ptr, n, _ptr(out, C.c_uint8), out.size))
    return out


def png_probe_files(paths, threads=0):
    """Header scan of many PNG files in one call (ctpn_png_probe_files): (n, 4) int32 rows (h, w, colour type, bit depth); h = 0 for files
    ctpn_decode_png_files does not take."""
    lib = load_library()
    paths = list(paths)
    keep, arr = _path_array(paths)
    out = np.zeros((len(paths), 4), np.int32)
    _check(lib.ctpn_png_probe_files(arr, len(paths), _ptr(out, C.c_int), int(threads)))
    return out


def decode_png_files(paths, h, w, threads=0, out=None):
    """n PNG files of one size -> (n, h, w, 3) BGR uint8 on the host, one file per C++ thread (ctpn_decode_png_files); `out` may be a
    caller's (page-locked) batch buffer."""
    lib = load_library()
    paths = list(paths)
    keep, arr = _path_array(paths)
    if out is None:
        out = np.empty((len(paths), int(h), int(w), 3), np.uint8)
    assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.shape == (len(paths), int(h), int(w), 3)
    _check(lib.ctpn_decode_png_files(arr, len(paths), int(h), int(w), _ptr(out, C.c_uint8), int(threads)))
    return out


def jpeg_entropy_decode(data):
    """The host half of the JPEG decoder alone (ctpn_jpeg_entropy_decode; no device): returns (planes, qt, layout) with planes = one
    (block rows, block columns, 64) int16 array of quantised coefficients in natural order per component, qt = (3, 64) uint16 and
    layout = dict(h, w, ncomp, hs, vs, orientation); h, w are the STORED size."""
    lib = load_library()
    keep, ptr, n = _bytes_ptr(data)
    h, w, nc, hs = jpeg_probe(data)
    cap = int(lib.ctpn_jpeg_coef_capacity(h, w))
    coef = np.zeros((cap,), np.int16)
    qt = np.zeros((3, 64), np.uint16)
    l8 = np.zeros((8,), np.int32)
    _check(lib.ctpn_jpeg_entropy_decode(ptr, n, _ptr(coef, C.c_int16), cap, _ptr(qt, C.c_uint16), _ptr(l8, C.c_int)))
    h, w, nc, hs, bw0, bw1, bh0, bh1 = (int(v) for v in l8)       # h, w: as STORED (the probe's are the turned image's)
    hs, orient = hs & 0xff, (hs >> 8) + 1
    planes, off = [], 0
    for c in range(nc):
        bw, bh = (bw0, bh0) if c == 0 else (bw1, bh1)
        planes.append(coef[off: off + bw * bh * 64].reshape(bh, bw, 64))
        off += bw * bh * 64
    return planes, qt, {"h": h, "w": w, "ncomp": nc, "hs": hs, "vs": (bh0 // bh1 if nc == 3 else 1), "orientation": orient}


def text_lines(boxes, scores, size, mode="H", device_id=0, capacity=4096):
    """B4 seam: TextDetector.detect (reference lib/text_connector/detectors.py:19-35)."""
    lib = load_library()
    b = _f32(boxes).reshape(-1, 4)
    s = _f32(scores).reshape(-1)
    recs = np.zeros((capacity, 9), np.float64)
    cnt = C.c_int(0)
    m = MODE_O if str(mode).upper().startswith("O") else MODE_H
    _check(lib.ctpn_text_lines(_ptr(b, C.c_float), _ptr(s, C.c_float), int(b.shape[0]), int(size[0]), int(size[1]), m,
                               int(device_id), _ptr(recs, C.c_double), capacity, C.byref(cnt)))
    return recs[: cnt.value].copy()


CONNECTOR_CONSTANT_NAMES = ("TEXT_PROPOSALS_WIDTH * MIN_NUM_PROPOSALS", "MIN_RATIO", "LINE_MIN_SCORE", "MAX_HORIZONTAL_GAP", "TEXT_PROPOSALS_MIN_SCORE",
                            "TEXT_PROPOSALS_NMS_THRESH", "MIN_V_OVERLAPS", "MIN_SIZE_SIM")


def connector_constants():
    """{name: value} of the connector constants compiled into the library (ctpn_connector_constants)."""
    out = np.zeros((8,), np.float64)
    _check(load_library().ctpn_connector_constants(_ptr(out, C.c_double)))
    return dict(zip(CONNECTOR_CONSTANT_NAMES, out.tolist()))


def result_text(recs, scale):
    """bytes of res_<stem>.txt for (M,9) records (ctpn_result_text; reference ctpn/demo.py:28-49). Host C++, needs no GPU."""
    lib = load_library()
    r = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, 9)
    n = C.c_size_t(0)
    _check(lib.ctpn_result_text(_ptr(r, C.c_double), int(r.shape[0]), float(scale), None, 0, C.byref(n), None))
    buf = C.create_string_buffer(max(int(n.value), 1))
    _check(lib.ctpn_result_text(_ptr(r, C.c_double), int(r.shape[0]), float(scale), buf, len(buf), C.byref(n), None))
    return buf.raw[: n.value]


def write_result_file(path, recs, scale):
    lib = load_library()
    r = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, 9)
    cnt = C.c_int(0)
    _check(lib.ctpn_write_result_file(str(path).encode(), _ptr(r, C.c_double), int(r.shape[0]), float(scale), C.byref(cnt)))
    return cnt.value


def draw_boxes(img_bgr, recs):
    """Outlines of the text lines into a (h,w,3) uint8 BGR image, in place (ctpn_draw_boxes)."""
    lib = load_library()
    if img_bgr.dtype != np.uint8 or img_bgr.ndim != 3 or img_bgr.shape[2] != 3 or not img_bgr.flags["C_CONTIGUOUS"]:
        raise ValueError("draw_boxes wants a C-contiguous (h,w,3) uint8 image")
    r = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, 9)
    _check(lib.ctpn_draw_boxes(_ptr(img_bgr, C.c_uint8), int(img_bgr.shape[0]), int(img_bgr.shape[1]), _ptr(r, C.c_double), int(r.shape[0])))
    return img_bgr


def debug_connect(rois, size, mode="H", scale=1.0, device_id=0, capacity=512):
    """TextDetector.detect on the device connector (lines_prep -> nms 0.2 -> connect_kernel) for one image's rois (R,5)."""
    lib = load_library()
    r = _f32(rois).reshape(-1, 5)
    recs = np.zeros((capacity, 9), np.float64)
    cnt = C.c_int(0)
    m = MODE_O if str(mode).upper().startswith("O") else MODE_H
    _check(lib.ctpn_debug_connect(int(device_id), _ptr(r, C.c_float), int(r.shape[0]), int(size[0]), int(size[1]), float(scale), m,
                                  _ptr(recs, C.c_double), capacity, C.byref(cnt)))
    return recs[: cnt.value].copy()


COMM_ID_BYTES = 128


def comm_unique_id():
    """RCCL unique id (ctpn_comm_unique_id): created by the root, handed to the other ranks through any side channel."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().ctpn_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


def broadcast_weights(contexts):
    """One process, one ctx per GPU: contexts[0]'s loaded weights reach every other ctx over RCCL (ctpn_broadcast_weights)."""
    arr = (C.c_void_p * len(contexts))(*[c._h.value for c in contexts])
    _check(load_library().ctpn_broadcast_weights(arr, len(contexts)))


def host_thread_budget(cpu_count, local_world_size=1, requested=0):
    """ctpn_host_thread_budget: host workers per ctx (pure function of its arguments, no GPU needed)."""
    return int(load_library().ctpn_host_thread_budget(int(cpu_count), int(local_world_size), int(requested)))


def debug_cvt_bf16(x, use_hw=True, device_id=0):
    lib = load_library()
    x = _f32(x).reshape(-1)
    out = np.zeros((x.size,), np.uint16)
    _check(lib.ctpn_debug_cvt_bf16(int(device_id), _ptr(x, C.c_float), _ptr(out, C.c_uint16), int(x.size), 1 if use_hw else 0))
    return out


def debug_lds_dma(src, device_id=0):
    """ctpn_debug_lds_dma: (bytes through the m0-clobber form, bytes through the save / restore form) of the kernels' LDS-DMA helper."""
    lib = load_library()
    src = np.ascontiguousarray(src, np.uint8).reshape(-1)
    a, b = np.zeros_like(src), np.zeros_like(src)
    _check(lib.ctpn_debug_lds_dma(int(device_id), _ptr(src, C.c_uint8), src.size, _ptr(a, C.c_uint8), _ptr(b, C.c_uint8)))
    return a, b


def debug_conv3x3(x, w_hwio, bias, precision="fp32", impl=1, fuse_pool=False, want_full=True, device_id=0):
    """Unit-test hook (ctpn_debug_conv3x3): returns (full or None, pooled or None) as dense fp32 NHWC."""
    lib = load_library()
    x = _f32(x); w_hwio = _f32(w_hwio); bias = _f32(bias)
    n, h, w, ci = x.shape
    co = w_hwio.shape[3]
    full = np.zeros((n, h, w, co), np.float32) if want_full else None
    pooled = np.zeros((n, h // 2, w // 2, co), np.float32) if fuse_pool else None
    prec = precision_code(precision)
    _check(lib.ctpn_debug_conv3x3(int(device_id), _ptr(x, C.c_float), _ptr(w_hwio, C.c_float), _ptr(bias, C.c_float), n, h, w, ci,
                                  co, prec, int(impl), 1 if fuse_pool else 0,
                                  _ptr(full, C.c_float) if want_full else None, _ptr(pooled, C.c_float) if fuse_pool else None))
    return full, pooled


class Context:
    """One ctpn_ctx: a GPU, a stream and the HBM arena for up to max_batch images of max_h x max_w."""

    def __init__(self, device_id=0, max_batch=1, max_h=600, max_w=900, precision="bf16", postproc_only=False, options=None):
        """precision: "fp32" | "bf16" | "fp16" | "split" (CTPN_PREC_*). options: {name: int} for ctpn_set_option (see OPTION_ENV).
        postproc_only: ctpn_create_postproc -- proposal-layer buffers for max_h//16 x max_w//16 feature maps, no network."""
        self._lib = load_library()
        self._h = C.c_void_p()
        self.precision = precision
        self.postproc_only = bool(postproc_only)
        prec = precision_code(precision)
        if postproc_only:
            _check(self._lib.ctpn_create_postproc(C.byref(self._h), int(device_id), int(max_batch), int(max_h) // 16, int(max_w) // 16))
        else:
            _check(self._lib.ctpn_create(C.byref(self._h), int(device_id), int(max_batch), int(max_h), int(max_w), prec))
        self.device_id = device_id
        self.max_batch, self.max_h, self.max_w = max_batch, max_h, max_w
        opts = {k: int(os.environ[e]) for k, e in OPTION_ENV.items() if os.environ.get(e, "") != ""}
        opts.update(options or {})
        for k, v in opts.items():
            self.set_option(k, v)

    def set_option(self, key, value):
        _check(self._lib.ctpn_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int(0)
        _check(self._lib.ctpn_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def host_threads(self):
        n = C.c_int(0)
        _check(self._lib.ctpn_host_threads(self._h, C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ctpn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- weights
    def load_weights(self, arena):
        a = _f32(arena).reshape(-1)
        from .weights import WEIGHT_FLOATS
        if a.size != WEIGHT_FLOATS:
            raise ValueError("weight arena must hold %d floats, got %d" % (WEIGHT_FLOATS, a.size))
        _check(self._lib.ctpn_load_weights_host(self._h, _ptr(a, C.c_float)))

    def load_weights_device(self, dev_ptr):
        _check(self._lib.ctpn_load_weights_device(self._h, C.c_void_p(int(dev_ptr))))

    def broadcast_weights_rank(self, unique_id, rank, world, root=0):
        """One process per GPU: collective over RCCL (every rank calls it with the root's comm_unique_id()); the root has its
        weights loaded, every other rank receives the arena into its HBM and packs it (ctpn_broadcast_weights_rank)."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % COMM_ID_BYTES)
        _check(self._lib.ctpn_broadcast_weights_rank(self._h, C.c_char_p(bytes(unique_id)), int(rank), int(world), int(root)))

    # ---- forward
    def forward(self, images, device_ptr=None, shape=None):
        """images: (n,h,w,3) uint8 BGR host array, or device_ptr + shape for HBM-resident input."""
        if device_ptr is not None:
            n, h, w = shape
            _check(self._lib.ctpn_forward(self._h, C.c_void_p(int(device_ptr)), 1, int(n), int(h), int(w)))
            return
        im = np.ascontiguousarray(images, dtype=np.uint8)
        if im.ndim == 3:
            im = im[None]
        n, h, w, c = im.shape
        assert c == 3
        self._keepalive = im
        _check(self._lib.ctpn_forward(self._h, im.ctypes.data_as(C.c_void_p), 0, n, h, w))

    def forward_blob(self, blob, device_ptr=None, shape=None):
        """blob: (n,h,w,3) float32 BGR with PIXEL_MEANS already subtracted (the reference's net.data feed)."""
        if device_ptr is not None:
            n, h, w = shape
            _check(self._lib.ctpn_forward_blob(self._h, C.c_void_p(int(device_ptr)), 1, int(n), int(h), int(w)))
            return
        b = np.ascontiguousarray(blob, dtype=np.float32)
        if b.ndim == 3:
            b = b[None]
        n, h, w, c = b.shape
        assert c == 3
        self._keepalive = b
        _check(self._lib.ctpn_forward_blob(self._h, b.ctypes.data_as(C.c_void_p), 0, n, h, w))

    def sync(self):
        _check(self._lib.ctpn_sync(self._h))

    def stream(self):
        s = C.c_void_p()
        _check(self._lib.ctpn_stream(self._h, C.byref(s)))
        return s.value

    def feat_shape(self):
        n, hf, wf = C.c_int(), C.c_int(), C.c_int()
        _check(self._lib.ctpn_feat_shape(self._h, C.byref(n), C.byref(hf), C.byref(wf)))
        return n.value, hf.value, wf.value

    def get_tensor(self, name, capacity=None):
        shape = (C.c_int * 4)()
        if capacity is None:
            probe = np.zeros((1,), np.float32)
            rc = self._lib.ctpn_get_tensor(self._h, name.encode(), _ptr(probe, C.c_float), 0, shape)
            if rc not in (CTPN_OK, -4):
                _check(rc)
            capacity = int(np.prod([shape[i] for i in range(4)]))
        out = np.zeros((max(capacity, 1),), np.float32)
        _check(self._lib.ctpn_get_tensor(self._h, name.encode(), _ptr(out, C.c_float), out.size, shape))
        shp = tuple(shape[i] for i in range(4))
        return out[: int(np.prod(shp))].reshape(shp)

    # ---- proposals
    def _anchors(self, n, post_nms_topn, counts):
        a = np.zeros((n, post_nms_topn), np.int32)
        _check(self._lib.ctpn_proposal_anchors(self._h, _ptr(a, C.c_int), int(post_nms_topn)))
        return [a[i, : counts[i]].copy() for i in range(n)]

    def proposals(self, im_info, pre_nms_topn=12000, post_nms_topn=1000, nms_thresh=0.7, min_size=8.0, want_anchors=False):
        """-> list of rois (R,5) per image; with want_anchors also the anchor index (y*wf + x)*10 + a of every roi."""
        info = _f32(im_info).reshape(-1, 3)
        n = info.shape[0]
        rois = np.zeros((n, post_nms_topn, 5), np.float32)
        counts = np.zeros((n,), np.int32)
        _check(self._lib.ctpn_proposals(self._h, _ptr(info, C.c_float), int(pre_nms_topn), int(post_nms_topn),
                                        float(nms_thresh), float(min_size), _ptr(rois, C.c_float), _ptr(counts, C.c_int)))
        out = [rois[i, : counts[i]].copy() for i in range(n)]
        return (out, self._anchors(n, post_nms_topn, counts)) if want_anchors else out

    def proposals_from_host(self, cls_prob, bbox_pred, im_info, pre_nms_topn=12000, post_nms_topn=1000,
                            nms_thresh=0.7, min_size=8.0, want_anchors=False):
        cp = _f32(cls_prob)
        bp = _f32(bbox_pred)
        n, hf, wf, _ = cp.shape
        info = _f32(im_info).reshape(-1, 3)
        rois = np.zeros((n, post_nms_topn, 5), np.float32)
        counts = np.zeros((n,), np.int32)
        _check(self._lib.ctpn_proposals_from_host(self._h, _ptr(cp, C.c_float), _ptr(bp, C.c_float), n, hf, wf,
                                                  _ptr(info, C.c_float), int(pre_nms_topn), int(post_nms_topn),
                                                  float(nms_thresh), float(min_size), _ptr(rois, C.c_float),
                                                  _ptr(counts, C.c_int)))
        out = [rois[i, : counts[i]].copy() for i in range(n)]
        return (out, self._anchors(n, post_nms_topn, counts)) if want_anchors else out

    # ---- whole path
    def detect(self, images=None, scales=None, mode="H", line_capacity=512, device_ptr=None, shape=None,
               want_rois=False):
        if device_ptr is not None:
            n, h, w = shape
            ptr, on_dev = C.c_void_p(int(device_ptr)), 1
        else:
            im = np.ascontiguousarray(images, dtype=np.uint8)
            if im.ndim == 3:
                im = im[None]
            n, h, w, _ = im.shape
            self._keepalive = im
            ptr, on_dev = im.ctypes.data_as(C.c_void_p), 0
        sc = _f32(scales if scales is not None else np.ones((n,), np.float32)).reshape(-1)
        recs = np.zeros((n, line_capacity, 9), np.float64)
        lcnt = np.zeros((n,), np.int32)
        rois = np.zeros((n, 1000, 5), np.float32) if want_rois else None
        rcnt = np.zeros((n,), np.int32) if want_rois else None
        m = MODE_O if str(mode).upper().startswith("O") else MODE_H
        _check(self._lib.ctpn_detect(self._h, ptr, on_dev, int(n), int(h), int(w), _ptr(sc, C.c_float), m,
                                     _ptr(recs, C.c_double), int(line_capacity), _ptr(lcnt, C.c_int),
                                     _ptr(rois, C.c_float) if want_rois else None,
                                     _ptr(rcnt, C.c_int) if want_rois else None))
        lines = [recs[i, : lcnt[i]].copy() for i in range(n)]
        if want_rois:
            return lines, [rois[i, : rcnt[i]].copy() for i in range(n)]
        return lines

    def decode_jpeg_batch(self, files, h=None, w=None, fx=1.0, fy=1.0):
        """resize_im(cv2.imread(f)) of n JPEG files of one size on the device (ctpn_decode_jpeg_batch): Huffman decoding on the ctx's host
        pool, IDCT / upsampling / colour conversion / cv2.resize(fx, fy) as HIP kernels. Returns (device pointer, (n, out_h, out_w)) for
        forward / detect / detect_submit (device_ptr=, shape=); the buffer stays valid until the second-next call.
        CtpnError(code CTPN_ERR_UNSUPPORTED) for CMYK / 4:1:1 / arithmetic-coded / incomplete files: decode those on the host."""
        files = list(files)
        if h is None or w is None:
            h, w = jpeg_probe(files[0])[:2]
        keeps = [_bytes_ptr(f) for f in files]
        n = len(keeps)
        ptrs = (C.POINTER(C.c_uint8) * n)(*[k[1] for k in keeps])
        sizes = (C.c_size_t * n)(*[k[2] for k in keeps])
        out, oh, ow = C.c_void_p(0), C.c_int(0), C.c_int(0)
        _check(self._lib.ctpn_decode_jpeg_batch(self._h, ptrs, sizes, n, int(h), int(w), float(fx), float(fy), C.byref(out), C.byref(oh), C.byref(ow)))
        return out.value, (n, oh.value, ow.value)

    def decode_jpeg_files(self, paths, h, w, fx=1.0, fy=1.0):
        """decode_jpeg_batch from paths (ctpn_decode_jpeg_files): the library's worker threads read the files themselves."""
        paths = list(paths)
        keep, arr = _path_array(paths)
        out, oh, ow = C.c_void_p(0), C.c_int(0), C.c_int(0)
        _check(self._lib.ctpn_decode_jpeg_files(self._h, arr, len(paths), int(h), int(w), float(fx), float(fy), C.byref(out), C.byref(oh), C.byref(ow)))
        return out.value, (len(paths), oh.value, ow.value)

    def jpeg_batch_fetch(self, device_ptr, shape):
        """The decoded batch as an (n, h, w, 3) uint8 array on the host (ctpn_jpeg_batch_fetch)."""
        n, h, w = shape
        out = np.empty((n, h, w, 3), np.uint8)
        _check(self._lib.ctpn_jpeg_batch_fetch(self._h, C.c_void_p(int(device_ptr)), _ptr(out, C.c_uint8), out.size))
        return out

    def detect_submit(self, images=None, slot=0, scales=None, device_ptr=None, shape=None):
        """Asynchronous detect, part 1 (ctpn_detect_submit). Returns immediately."""
        if device_ptr is not None:
            n, h, w = shape
            ptr, on_dev = C.c_void_p(int(device_ptr)), 1
        else:
            im = np.ascontiguousarray(images, dtype=np.uint8)
            if im.ndim == 3:
                im = im[None]
            n, h, w, _ = im.shape
            self._keep_slot = getattr(self, "_keep_slot", {})
            self._keep_slot[slot] = im
            ptr, on_dev = im.ctypes.data_as(C.c_void_p), 0
        sc = _f32(scales if scales is not None else np.ones((n,), np.float32)).reshape(-1)
        _check(self._lib.ctpn_detect_submit(self._h, ptr, on_dev, int(n), int(h), int(w), _ptr(sc, C.c_float), int(slot)))
        self._slot_n = getattr(self, "_slot_n", {})
        self._slot_n[slot] = n

    def detect_collect(self, slot=0, mode="H", line_capacity=512, want_rois=False):
        """Asynchronous detect, part 2 (ctpn_detect_collect): waits for the slot and returns its text lines."""
        n = self._slot_n[slot]
        recs = np.zeros((n, line_capacity, 9), np.float64)
        lcnt = np.zeros((n,), np.int32)
        rois = np.zeros((n, 1000, 5), np.float32) if want_rois else None
        rcnt = np.zeros((n,), np.int32) if want_rois else None
        m = MODE_O if str(mode).upper().startswith("O") else MODE_H
        _check(self._lib.ctpn_detect_collect(self._h, int(slot), m, _ptr(recs, C.c_double), int(line_capacity), _ptr(lcnt, C.c_int),
                                             _ptr(rois, C.c_float) if want_rois else None, _ptr(rcnt, C.c_int) if want_rois else None))
        lines = [recs[i, : lcnt[i]].copy() for i in range(n)]
        if want_rois:
            return lines, [rois[i, : rcnt[i]].copy() for i in range(n)]
        return lines

    # ---- measurement
    def profile_enable(self, on=True):
        """on: False / True (an event pair around every stage) / 2 (ONE pair around the 13 conv3x3 launches of a forward)."""
        _check(self._lib.ctpn_profile_enable(self._h, 2 if on == 2 else (1 if on else 0)))

    def profile_reset(self):
        _check(self._lib.ctpn_profile_reset(self._h))

    def profile_read(self):
        out = {}
        for k, name in enumerate(KIND_NAMES):
            ms, n, work = C.c_double(), C.c_longlong(), C.c_double()
            _check(self._lib.ctpn_profile_read(self._h, k, C.byref(ms), C.byref(n), C.byref(work)))
            out[name] = {"ms": ms.value, "launches": n.value, "work": work.value}
        return out
