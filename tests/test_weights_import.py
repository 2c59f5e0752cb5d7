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


def _leveldb_table(pairs, block_size=4096, restart_interval=16):
    """An index file the way LevelDB's TableBuilder (tensorflow/core/lib/io/table_builder.cc) lays it out, written independently of
    weights_import's own writer: sorted keys, prefix compression against the previous key with a restart point every 16 entries,
    a new data block once the current one exceeds block_size, an index block of (separator key >= last key of the block, handle)
    entries with restart interval 1, an empty metaindex block, block trailers (type 0 + 4 crc bytes), 48-byte footer."""
    import struct

    def varint(v):
        out = bytearray()
        while True:
            b = v & 0x7F
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)

    def build_block(entries, interval):
        body, restarts, prev = bytearray(), [], b""
        for n, (k, v) in enumerate(entries):
            shared = 0
            if n % interval == 0:
                restarts.append(len(body))
            else:
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            body += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
            prev = k
        for r in restarts or [0]:
            body += struct.pack("<I", r)
        body += struct.pack("<I", max(len(restarts), 1))
        return bytes(body)

    blob, index_entries, cur = bytearray(), [], []
    pairs = sorted(pairs)

    def flush():
        if not cur:
            return
        body = build_block(cur, restart_interval)
        handle = varint(len(blob)) + varint(len(body))
        blob.extend(body + b"\x00" + b"\xde\xad\xbe\xef")
        index_entries.append((cur[-1][0] + b"\x00", handle))     # any key >= last key of the block and < first key of the next
        cur.clear()

    size = 0
    for k, v in pairs:
        cur.append((k, v))
        size += len(k) + len(v) + 3
        if size >= block_size:
            flush()
            size = 0
    flush()
    meta = build_block([], 1)
    meta_handle = varint(len(blob)) + varint(len(meta))
    blob.extend(meta + b"\x00" + b"\x00\x00\x00\x00")
    index = build_block(index_entries, 1)
    index_handle = varint(len(blob)) + varint(len(index))
    blob.extend(index + b"\x00" + b"\x00\x00\x00\x00")
    footer = meta_handle + index_handle
    blob.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57))
    return bytes(blob), len(index_entries)


def test_saver_v2_reader_on_independent_multi_block_table(tmp_path, arena):
    """the bundle reader was only tested on the one-block, restart-per-entry tables weights_import writes itself. Here the
    index comes from an independent LevelDB-style builder: several data blocks, 16-entry restart intervals with real prefix
    compression, separator index keys, plus non-float and sliced entries the reader has to skip."""
    views = ctpn_amd.arena_views(arena)
    enc, ev = WI._enc, WI._enc_varint
    data = bytearray()
    pairs = [(b"", ev(1 << 3) + ev(1))]                                                   # BundleHeaderProto{num_shards: 1}

    def entry(dtype, shape, off, size, sliced=False):
        sh = b"".join(enc(2, ev(1 << 3) + ev(int(d))) for d in shape)
        e = ev(1 << 3) + ev(dtype) + enc(2, sh) + ev(4 << 3) + ev(off) + ev(5 << 3) + ev(size)
        return e + (enc(7, b"\x08\x01") if sliced else b"")

    for name in sorted(views):
        a = np.ascontiguousarray(views[name], "<f4")
        pairs.append((name.encode(), entry(1, a.shape, len(data), a.nbytes)))
        data += a.tobytes()
    pairs.append((b"global_step", entry(9, (), len(data), 8)))                            # DT_INT64: skipped
    data += (50000).to_bytes(8, "little")
    # Adam slots / padding variables give the table more keys than fit one 4 KB block, sharing long prefixes with the real ones
    for i in range(300):
        k = ("lstm_o/bidirectional_rnn/fw/lstm_cell/kernel/Adam_%03d" % i).encode()
        pairs.append((k, entry(1, (2,), len(data), 8, sliced=(i % 7 == 0))))
        data += np.array([i, -i], "<f4").tobytes()
    table, nblocks = _leveldb_table(pairs)
    assert nblocks >= 3
    prefix = str(tmp_path / "VGGnet_fast_rcnn_iter_50000.ckpt")
    open(prefix + ".index", "wb").write(table)
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    got = WI.read_checkpoint(prefix)
    for k, v in views.items():
        assert got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert "global_step" not in got
    assert np.array_equal(got["lstm_o/bidirectional_rnn/fw/lstm_cell/kernel/Adam_001"], np.array([1, -1], np.float32))
    assert "lstm_o/bidirectional_rnn/fw/lstm_cell/kernel/Adam_007" not in got              # sliced entries are not assembled
    assert np.array_equal(WI.load_any(prefix), arena)
    # an entry that runs past its data shard is reported, not read
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data[: len(data) // 2]))
    with pytest.raises(ValueError, match="does not fit its data shard"):
        WI.read_checkpoint(prefix)


# ---- an independent WRITER for the protobuf layer: Google's protobuf runtime on a schema typed in from TensorFlow's public .proto files -----------
def _tf_messages():
    """GraphDef / NodeDef / AttrValue / TensorProto / TensorShapeProto / BundleHeaderProto / BundleEntryProto as dynamic messages. Field
    numbers and types from tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape,types}.proto and
    tensorflow/core/protobuf/tensor_bundle.proto (TF 1.3); only the fields a frozen CTPN graph / a Saver-V2 index uses."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    T = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="tf_subset.proto", package="tfsub", syntax="proto3")

    def msg(parent, name):
        m = parent.message_type.add() if hasattr(parent, "message_type") else parent.nested_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=T.LABEL_OPTIONAL, type_name=None):
        f = m.field.add(name=name, number=num, type=typ, label=label)
        if type_name:
            f.type_name = ".tfsub." + type_name
        return f
    shape = msg(fd, "TensorShapeProto")
    dim = msg(shape, "Dim")
    field(dim, "size", 1, T.TYPE_INT64)
    field(dim, "name", 2, T.TYPE_STRING)
    field(shape, "dim", 2, T.TYPE_MESSAGE, T.LABEL_REPEATED, "TensorShapeProto.Dim")
    field(shape, "unknown_rank", 3, T.TYPE_BOOL)
    tensor = msg(fd, "TensorProto")
    field(tensor, "dtype", 1, T.TYPE_INT32)
    field(tensor, "tensor_shape", 2, T.TYPE_MESSAGE, type_name="TensorShapeProto")
    field(tensor, "version_number", 3, T.TYPE_INT32)
    field(tensor, "tensor_content", 4, T.TYPE_BYTES)
    field(tensor, "float_val", 5, T.TYPE_FLOAT, T.LABEL_REPEATED)
    field(tensor, "int_val", 7, T.TYPE_INT32, T.LABEL_REPEATED)
    attr = msg(fd, "AttrValue")
    field(attr, "s", 2, T.TYPE_BYTES)
    field(attr, "i", 3, T.TYPE_INT64)
    field(attr, "f", 4, T.TYPE_FLOAT)
    field(attr, "b", 5, T.TYPE_BOOL)
    field(attr, "type", 6, T.TYPE_INT32)
    field(attr, "shape", 7, T.TYPE_MESSAGE, type_name="TensorShapeProto")
    field(attr, "tensor", 8, T.TYPE_MESSAGE, type_name="TensorProto")
    node = msg(fd, "NodeDef")
    field(node, "name", 1, T.TYPE_STRING)
    field(node, "op", 2, T.TYPE_STRING)
    field(node, "input", 3, T.TYPE_STRING, T.LABEL_REPEATED)
    field(node, "device", 4, T.TYPE_STRING)
    entry = msg(node, "AttrEntry")
    entry.options.map_entry = True
    field(entry, "key", 1, T.TYPE_STRING)
    field(entry, "value", 2, T.TYPE_MESSAGE, type_name="AttrValue")
    field(node, "attr", 5, T.TYPE_MESSAGE, T.LABEL_REPEATED, "NodeDef.AttrEntry")
    ver = msg(fd, "VersionDef")
    field(ver, "producer", 1, T.TYPE_INT32)
    field(ver, "min_consumer", 2, T.TYPE_INT32)
    graph = msg(fd, "GraphDef")
    field(graph, "node", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, "NodeDef")
    field(graph, "version", 3, T.TYPE_INT32)
    field(graph, "versions", 4, T.TYPE_MESSAGE, type_name="VersionDef")
    hdr = msg(fd, "BundleHeaderProto")
    field(hdr, "num_shards", 1, T.TYPE_INT32)
    field(hdr, "endianness", 2, T.TYPE_INT32)
    field(hdr, "version", 3, T.TYPE_MESSAGE, type_name="VersionDef")
    ent = msg(fd, "BundleEntryProto")
    field(ent, "dtype", 1, T.TYPE_INT32)
    field(ent, "shape", 2, T.TYPE_MESSAGE, type_name="TensorShapeProto")
    field(ent, "shard_id", 3, T.TYPE_INT32)
    field(ent, "offset", 4, T.TYPE_INT64)
    field(ent, "size", 5, T.TYPE_INT64)
    field(ent, "crc32c", 6, T.TYPE_FIXED32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return {n: get(pool.FindMessageTypeByName("tfsub." + n)) for n in ("GraphDef", "BundleHeaderProto", "BundleEntryProto")}


def test_frozen_graph_written_by_the_protobuf_runtime(tmp_path, arena):
    """The frozen-graph reader against bytes it has never produced: a GraphDef serialised by Google's protobuf runtime (its own field order,
    packed repeated floats, map entries, version fields, non-Const nodes with inputs and device strings, an int32 Const to be skipped)."""
    M = _tf_messages()
    views = ctpn_amd.arena_views(arena)
    g = M["GraphDef"]()
    g.versions.producer = 24
    ph = g.node.add(name="Placeholder", op="Placeholder")
    ph.attr["dtype"].type = 1
    ph.attr["shape"].shape.unknown_rank = True
    splat = "conv5_3/biases"                                                        # all zeros in the synthetic arena: the float_val splat form
    for name in sorted(views):
        a = np.ascontiguousarray(views[name], "<f4")
        n = g.node.add(name=name, op="Const", device="/device:GPU:0")
        n.attr["dtype"].type = 1
        t = n.attr["value"].tensor
        t.dtype = 1
        for d in a.shape:
            t.tensor_shape.dim.add(size=int(d))
        if name == splat:
            t.float_val.append(float(a.ravel()[0]))
        else:
            t.tensor_content = a.tobytes()
        rd = g.node.add(name=name + "/read", op="Identity", input=[name])
        rd.attr["T"].type = 1
        rd.attr["_class"].s = ("loc:@" + name).encode()
    k = g.node.add(name="Reshape/shape", op="Const")
    k.attr["dtype"].type = 3
    k.attr["value"].tensor.dtype = 3
    k.attr["value"].tensor.tensor_shape.dim.add(size=4)
    k.attr["value"].tensor.int_val.extend([1, -1, 2, 3])
    conv = g.node.add(name="conv1_1/Conv2D", op="Conv2D", input=["Placeholder", "conv1_1/weights/read"])
    conv.attr["padding"].s = b"SAME"
    conv.attr["use_cudnn_on_gpu"].b = True
    path = str(tmp_path / "runtime.pb")
    open(path, "wb").write(g.SerializeToString())
    got = WI.read_frozen_graph(path)
    for kname, v in views.items():
        assert got[kname].shape == v.shape and np.array_equal(got[kname], v), kname
    assert "Reshape/shape" not in got                                               # an int32 Const is not a weight
    assert np.array_equal(WI.load_any(path), arena)


def test_saver_v2_index_entries_written_by_the_protobuf_runtime(tmp_path, arena):
    """Saver-V2 index values (BundleHeaderProto / BundleEntryProto) serialised by the protobuf runtime -- shard_id and the fixed32 crc32c
    TensorFlow always writes included -- inside the independent multi-block table of the test above."""
    M = _tf_messages()
    views = ctpn_amd.arena_views(arena)
    hdr = M["BundleHeaderProto"](num_shards=1)
    hdr.version.producer = 1
    pairs, data = [(b"", hdr.SerializeToString())], bytearray()
    for name in sorted(views):
        a = np.ascontiguousarray(views[name], "<f4")
        e = M["BundleEntryProto"](dtype=1, shard_id=0, offset=len(data), size=a.nbytes, crc32c=0x9e3779b9)
        for d in a.shape:
            e.shape.dim.add(size=int(d))
        pairs.append((name.encode(), e.SerializeToString()))
        data += a.tobytes()
    step = M["BundleEntryProto"](dtype=9, offset=len(data), size=8, crc32c=1)
    pairs.append((b"global_step", step.SerializeToString()))
    data += (50000).to_bytes(8, "little")
    table, _ = _leveldb_table(sorted(pairs))
    prefix = str(tmp_path / "runtime.ckpt")
    open(prefix + ".index", "wb").write(table)
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    got = WI.read_checkpoint(prefix)
    for kname, v in views.items():
        assert got[kname].shape == v.shape and np.array_equal(got[kname], v), kname
    assert np.array_equal(WI.load_any(prefix), arena)


def test_damaged_files_raise_value_error_naming_the_file(tmp_path):
    """Whatever the protobuf / table walkers trip over in a damaged file (an index past the buffer, a wrong wire type, bytes that are not UTF-8),
    the caller sees ValueError with the file's name (or IOError for a missing shard) -- 170 k mutants in round 4's fuzz run, none hung or
    allocated more than the file's size."""
    from ctpn_amd import weights_import as W
    rng = np.random.default_rng(0)
    small = {"a/weights": rng.normal(size=(3, 3, 2, 4)).astype(np.float32), "a/biases": rng.normal(size=(4,)).astype(np.float32)}
    W.write_frozen_graph(str(tmp_path / "g.pb"), small)
    W.write_checkpoint(str(tmp_path / "ck"), small)
    pb, idx = (tmp_path / "g.pb").read_bytes(), (tmp_path / "ck.index").read_bytes()
    data = (tmp_path / "ck.data-00000-of-00001").read_bytes()
    seen = set()
    for k in range(400):
        for name, blob, reader, target in (("m.pb", pb, W.read_frozen_graph, "m.pb"), ("mk.index", idx, W.read_checkpoint, "mk")):
            b = bytearray(blob)
            if k % 3 == 0:
                b = b[: int(rng.integers(0, len(b)))]
            else:
                for pos in rng.integers(0, len(b), int(rng.integers(1, 5))):
                    b[pos] = int(rng.integers(0, 256))
            (tmp_path / name).write_bytes(bytes(b))
            (tmp_path / "mk.data-00000-of-00001").write_bytes(data)
            try:
                reader(str(tmp_path / target))
                seen.add("ok")
            except ValueError as e:
                seen.add("ValueError")
                assert target in str(e), e
            except IOError:
                seen.add("IOError")
    assert "ValueError" in seen
