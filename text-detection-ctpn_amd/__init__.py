"""MI355X-native CTPN inference hot path (drop-in for eragonruan/text-detection-ctpn's demo path).

Layout mirrors the reference tree for the modules on the hot path:
  ctpn/demo.py, lib/fast_rcnn/{config,test,nms_wrapper,bbox_transform}.py, lib/rpn_msr/*, lib/utils/*,
  lib/networks/*, lib/text_connector/*  -- Python host code over the C ABI of libctpn_hip.so (include/ctpn_hip.h).
There is no CPU fallback: importing works without a GPU (so the build and the host logic can be checked),
but every compute entry point raises CtpnError when the HIP library or a gfx950 device is missing.
"""
from ._binding import CtpnError, lib_path, load_library, Context  # noqa: F401
from .weights import MANIFEST, WEIGHT_FLOATS, make_synthetic_arena, arena_views  # noqa: F401

__all__ = ["CtpnError", "lib_path", "load_library", "Context", "MANIFEST", "WEIGHT_FLOATS",
           "make_synthetic_arena", "arena_views"]
