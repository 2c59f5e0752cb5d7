"""Weight import: frozen-graph and VGG-npy readers against files written with the same wire format."""
import numpy as np
import pytest

import ctpn_amd
from ctpn_amd import weights_import as WI


def test_frozen_graph_round_trip(tmp_path, arena):
    views = ctpn_amd.arena_views(arena)
    path = str(tmp_path / "ctpn.pb")
    extra = {"Placeholder_shape": np.zeros((4,), np.float32)}                     # unrelated Const nodes are ignored by name
    WI.write_frozen_graph(path, dict(list(views.items()) + list(extra.items())))
    got = WI.read_frozen_graph(path)
    assert set(views) <= set(got)
    for k, v in views.items():
        assert got[k].shape == v.shape and np.array_equal(got[k], v), k
    again = WI.load_any(path)
    assert np.array_equal(again, arena)


def test_frozen_graph_missing_variable_is_loud(tmp_path, arena):
    views = dict(ctpn_amd.arena_views(arena))
    views.pop("rpn_cls_score/biases")
    path = str(tmp_path / "broken.pb")
    WI.write_frozen_graph(path, views)
    with pytest.raises(KeyError):
        WI.load_any(path)
    bad = dict(ctpn_amd.arena_views(arena))
    bad["conv1_1/weights"] = np.zeros((3, 3, 3, 32), np.float32)
    WI.write_frozen_graph(path, bad)
    with pytest.raises(ValueError):
        WI.load_any(path)


def test_vgg_imagenet_npy_layout(tmp_path, arena):
    """Network.load's format (reference lib/networks/network.py:40-53): {layer: {'weights', 'biases'}}, conv layers only."""
    views = ctpn_amd.arena_views(arena)
    nested = {}
    for name in ["conv%d_%d" % (b, i) for b, n in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)) for i in range(1, n + 1)]:
        nested[name] = {"weights": views[name + "/weights"] * 2, "biases": views[name + "/biases"] + 1}
    nested["fc6"] = {"weights": np.zeros((8, 8), np.float32), "biases": np.zeros((8,), np.float32)}   # present in the real file, unused
    path = str(tmp_path / "VGG_imagenet.npy")
    np.save(path, nested, allow_pickle=True)
    out, missing = WI.arena_from_vgg_npy(path, base=arena)
    v2 = ctpn_amd.arena_views(out)
    assert np.array_equal(v2["conv3_2/weights"], views["conv3_2/weights"] * 2)
    assert np.array_equal(v2["conv5_3/biases"], views["conv5_3/biases"] + 1)
    assert np.array_equal(v2["rpn_conv/3x3/weights"], views["rpn_conv/3x3/weights"])           # untouched: not in the file
    assert "lstm_o/weights" in missing and "conv1_1/weights" not in missing


def test_varint_and_splat_encodings():
    assert WI._varint(bytes([0xAC, 0x02]), 0) == (300, 2)
    # TensorProto with a single float_val and shape [3]: constant splat
    t = WI._enc_varint(1 << 3) + WI._enc_varint(1) + WI._enc(2, WI._enc(2, WI._enc_varint(1 << 3) + WI._enc_varint(3))) + bytes([(5 << 3) | 5]) + np.float32(1.5).tobytes()
    assert WI._tensor(memoryview(t)).tolist() == [1.5, 1.5, 1.5]


def test_saver_v2_checkpoint_round_trip(tmp_path, arena):
    """tensor bundle = LevelDB-format table (.index) + raw shard (.data-00000-of-00001); optimizer slots and non-float
    entries that a real training checkpoint carries next to the model variables are ignored by name / dtype."""
    views = dict(ctpn_amd.arena_views(arena))
    prefix = str(tmp_path / "VGGnet_fast_rcnn_iter_50000.ckpt")
    extra = dict(views)
    extra["conv1_1/weights/Adam"] = np.zeros((3, 3, 3, 64), np.float32)
    extra["global_step_f"] = np.zeros((), np.float32)
    WI.write_checkpoint(prefix, extra)
    got = WI.read_checkpoint(prefix)
    assert set(extra) == set(got)
    for k, v in views.items():
        assert got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert np.array_equal(WI.load_any(prefix), arena)
    assert np.array_equal(WI.load_any(prefix + ".index"), arena)
    (tmp_path / "junk.index").write_bytes(b"\x00" * 64)
    with pytest.raises(ValueError):
        WI.read_checkpoint(str(tmp_path / "junk"))


def test_table_block_prefix_compression():
    """keys inside a block are prefix-compressed against the previous key; TF uses a restart interval of 16."""
    body = bytearray()
    for shared, key, val in ((0, b"conv1_1/biases", b"A"), (8, b"weights", b"B"), (4, b"2_1/weights", b"C")):
        body += WI._enc_varint(shared) + WI._enc_varint(len(key)) + WI._enc_varint(len(val)) + key + val
    import struct
    body += struct.pack("<II", 0, 1)
    got = [(k, bytes(v)) for k, v in WI._block_entries(memoryview(bytes(body)))]
    assert got == [(b"conv1_1/biases", b"A"), (b"conv1_1/weights", b"B"), (b"conv2_1/weights", b"C")]
