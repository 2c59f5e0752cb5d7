"""GPU (-m gpu), round 4: conv1_1 inside conv1_2's window stage (VERDICT r3 "next round" item 3).

The uint8 feed of the 16-bit modes goes bytes -> q-image (image_to_q_kernel) and then EITHER
  * option conv1_fuse = 1 (default, production path): conv3x3_wr_kernel<FUSE> computes conv1_1 for every window it needs, in LDS, or
  * conv1_fuse = 0 / keep_acts = 1: conv_first_p_kernel stores conv1_1 (the 69 MB per image the fused form never writes) and the
    un-fused conv1_2 reads it back.
Both run the same MFMA sequence on the same operands, so everything downstream must agree BIT FOR BIT -- that equality is the test; what
the arithmetic itself is worth is checked where it always was (the per-layer tests against the oracle run the keep_acts form).
Nothing here reads /root/reference.
"""
import numpy as np
import pytest

import ctpn_amd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def weights(arena):
    return ctpn_amd.arena_views(arena)


def _run(arena, prec, imgs, options, tensors=("pool1", "heads")):
    n, h, w = imgs.shape[:3]
    with ctpn_amd.Context(0, n, h, w, prec, options=options) as ctx:
        ctx.load_weights(arena)
        ctx.forward(imgs)
        first = {t: ctx.get_tensor(t).copy() for t in tensors}
        ctx.forward(imgs)                                     # the second forward of the ctx: same bytes (no state left behind)
        second = {t: ctx.get_tensor(t).copy() for t in tensors}
    return first, second


# geometries: the benchmark's (W = 900 = 28 tiles + 4 ragged columns through the edge kernel), one tile column only, ragged in both
# directions with odd sizes, fewer tiles than workgroups, a wide one (config 5's) -- and a batch that spans several tile ranges
GEOMS = [(2, 600, 900), (1, 64, 32), (3, 101, 203), (1, 16, 16), (2, 130, 1000), (5, 88, 96)]


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_fused_conv1_gives_the_stored_forms_bytes(arena, prec, geom):
    n, h, w = geom
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 7 + h)
    f1, f2 = _run(arena, prec, imgs, {"conv1_fuse": 1})
    u1, _ = _run(arena, prec, imgs, {"conv1_fuse": 0})
    for t in f1:
        assert np.array_equal(f1[t], f2[t]), f"{t}: the second fused forward differs from the first"
        d = np.flatnonzero(f1[t].ravel() != u1[t].ravel())
        assert d.size == 0, f"{t}: {d.size} of {f1[t].size} values differ between the fused and the stored form; first at {np.unravel_index(d[0], f1[t].shape)}"


def test_fused_conv1_is_the_default_and_does_not_store_conv1_1(arena):
    imgs = ctpn_amd.weights.synthetic_images(1, 96, 160, 3)
    with ctpn_amd.Context(0, 1, 96, 160, "bf16") as ctx:
        assert ctx.get_option("conv1_kernel") == 2 and ctx.get_option("conv1_fuse") == 1
        ctx.load_weights(arena)
        ctx.forward(imgs)
        with pytest.raises(Exception, match="window stage"):
            ctx.get_tensor("conv1_1")
        ctx.set_option("keep_acts", 1)
        ctx.forward(imgs)
        assert ctx.get_tensor("conv1_1").shape[-1] == 64


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_q_image_conv1_1_against_the_oracle_layer(arena, weights, prec):
    """conv1_1 from the q-image (what keep_acts stores, and what the fused producer computes) against the oracle's conv1_1 on the same
    bytes with the weights rounded to the mode's 16-bit type: within one rounding of the output type, interior and border pixels alike
    (at the border the mean correction rides on the pixels' P slots, and the parts of G the 16-bit rounding dropped are not taken out
    again for the missing taps: < 5 x 2^-9 |G|, far below the output rounding)."""
    from oracle import network as N
    imgs = ctpn_amd.weights.synthetic_images(2, 72, 104, 5)
    with ctpn_amd.Context(0, 2, 72, 104, prec, options={"keep_acts": 1}) as ctx:
        ctx.load_weights(arena)
        ctx.forward(imgs)
        got = ctx.get_tensor("conv1_1")
    wq = weights["conv1_1/weights"].astype(np.float32)
    if prec == "bf16":
        u = wq.view(np.uint32).astype(np.uint64)
        wq = (((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)
    else:
        wq = wq.astype(np.float16).astype(np.float32)
    want = N.conv3x3_relu(N.image_blob(imgs), wq, weights["conv1_1/biases"])
    assert got.shape == want.shape
    ulp = float(np.abs(want).max()) * (2.0 ** -8 if prec == "bf16" else 2.0 ** -11)
    err = np.abs(got - want)
    assert err.max() <= ulp, err.max() / ulp
    border = np.ones(err.shape[1:3], bool)
    border[1:-1, 1:-1] = False
    assert err[:, border].max() <= ulp


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_device_conv1_1_equals_its_arithmetic_specification(arena, weights, prec):
    """oracle/conv1_q.py is the numpy restatement of the q-image form (slot layout, G / V constants, zero frame); the device's conv1_1 must
    be that, value for value, up to the MFMA's fp32 summation order: equal after the output rounding except on rounding boundaries."""
    from oracle import conv1_q as Q
    imgs = ctpn_amd.weights.synthetic_images(2, 45, 70, 21)
    with ctpn_amd.Context(0, 2, 45, 70, prec, options={"keep_acts": 1}) as ctx:
        ctx.load_weights(arena)
        ctx.forward(imgs)
        got = ctx.get_tensor("conv1_1")
    want = Q.conv1_1_from_q(imgs, weights["conv1_1/weights"], weights["conv1_1/biases"], prec)
    assert got.shape == want.shape
    ulp = float(np.abs(want).max()) * (2.0 ** -8 if prec == "bf16" else 2.0 ** -11)
    assert (got != want).mean() < 2e-3, (got != want).mean()
    assert np.abs(got - want).max() <= ulp
