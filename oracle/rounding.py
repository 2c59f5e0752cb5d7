"""ORACLE (test infrastructure only -- never imported by the product path).

The two 16-bit roundings the device's throughput modes apply to MFMA operands and stored activations, as numpy functions."""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest-even bf16 -> fp32: the rounding of v_cvt_pk_bf16_f32."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def fp16_round(x):
    """fp32 -> nearest-even IEEE fp16 -> fp32: the rounding of v_cvt_pk_f16_f32."""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
