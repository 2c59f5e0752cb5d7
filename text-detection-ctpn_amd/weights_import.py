"""Weight import from the reference's own formats into the flat arena.

The reference restores weights three ways, none of which needs TensorFlow to READ:
  * `Network.load` (lib/networks/network.py:40-53): a pickled dict {layer: {'weights': ..., 'biases': ...}} saved with
    np.save -- the ImageNet VGG16 initialisation `data/pretrain/VGG_imagenet.npy` (conv layers only);
  * a frozen graph `data/ctpn.pb` (ctpn/demo_pb.py:60-66; written by ctpn/generate_pb.py): a serialized GraphDef whose
    Const nodes carry every variable under its scope name (`conv1_1/weights`, ...);
  * a Saver-V2 checkpoint (ctpn/demo.py:84-92): `*.index` (an SSTable of BundleEntryProto) + `*.data-00000-of-00001`.
This module reads all three without TensorFlow: a ~60-line protobuf wire-format walker (GraphDef / NodeDef / AttrValue /
TensorProto / BundleEntryProto field numbers from tensorflow/core/{framework,protobuf}/*.proto, TF 1.3) and a reader for the
tensor-bundle index, which is a LevelDB-format table (tensorflow/core/lib/io/table*.cc: prefix-compressed key blocks with a
restart array, block handles as varint64 pairs, 48-byte footer ending in the magic 0xdb4775248b80fb57; TF writes it
uncompressed). There is no trained model in the reference tree and no TensorFlow in this image: the readers are tested on
files this package WRITES itself in the same formats (`write_frozen_graph`, `write_checkpoint`) and on a table built by an
independent LevelDB-style builder in tests/test_weights_import.py (4 KB data blocks, restart interval 16, prefix-compressed keys,
shortest-separator index keys -- the shape TF's TableBuilder gives a real bundle); parity with a real TF-written file stays
UNPINNED.
"""
import struct

import numpy as np

from . import weights as _w

_DT_FLOAT = 1


# ---- protobuf wire format -------------------------------------------------------------------------------------
def _varint(buf, i):
    v, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, i
        s += 7


def _fields(buf):
    """yield (field_number, wire_type, value) for one message; value is int (varint / fixed) or a memoryview (bytes)."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v = bytes(buf[i:i + 8]); i += 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]; i += ln
        elif wt == 5:
            v = bytes(buf[i:i + 4]); i += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _tensor(buf):
    """TensorProto -> ndarray (float only): dtype = 1, tensor_shape = 2, tensor_content = 4, float_val = 5."""
    dtype, shape, content, fvals = None, [], None, []
    for f, wt, v in _fields(buf):
        if f == 1:
            dtype = v
        elif f == 2:
            for f2, _, v2 in _fields(v):          # TensorShapeProto.dim = 2
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _fields(v2):  # Dim.size = 1
                        if f3 == 1:
                            size = v3
                    shape.append(int(size))
        elif f == 4:
            content = bytes(v)
        elif f == 5:
            if wt == 2:
                fvals.extend(np.frombuffer(bytes(v), "<f4").tolist())
            else:
                fvals.append(struct.unpack("<f", v)[0])
    if dtype != _DT_FLOAT:
        return None
    n = int(np.prod(shape)) if shape else 1
    if content is not None:
        a = np.frombuffer(content, "<f4")
    elif len(fvals) == 1 and n > 1:
        a = np.full((n,), fvals[0], np.float32)    # splat encoding of constant tensors
    else:
        a = np.asarray(fvals, np.float32)
    return a.reshape(shape).astype(np.float32)


def _corrupt_as_value_error(what):
    """A damaged file surfaces as ONE exception type, ValueError naming the file, whatever the walker tripped over (an index past the buffer,
    a field of the wrong wire type, bytes that are not UTF-8 ...): callers catch `ValueError` / `IOError`, not the parser's internals."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(path, *a, **kw):
            try:
                return fn(path, *a, **kw)
            except (IndexError, struct.error, UnicodeDecodeError, TypeError, KeyError, OverflowError, ValueError) as e:
                if isinstance(e, ValueError) and not isinstance(e, UnicodeDecodeError) and str(path) in str(e):
                    raise
                raise ValueError("%s is not a readable %s (%s: %s)" % (path, what, type(e).__name__, e)) from e
        return wrapped
    return deco


@_corrupt_as_value_error("frozen graph")
def read_frozen_graph(path):
    """{node name: ndarray} for every float Const node of a serialized GraphDef (GraphDef.node = 1; NodeDef.name = 1,
    op = 2, attr = 5 (map entry: key = 1, value = 2); AttrValue.tensor = 8)."""
    buf = memoryview(open(path, "rb").read())
    out = {}
    for f, _, node in _fields(buf):
        if f != 1:
            continue
        name, op, tensor = None, None, None
        for f2, _, v2 in _fields(node):
            if f2 == 1:
                name = bytes(v2).decode()
            elif f2 == 2:
                op = bytes(v2).decode()
            elif f2 == 5:
                key, val = None, None
                for f3, _, v3 in _fields(v2):
                    if f3 == 1:
                        key = bytes(v3).decode()
                    elif f3 == 2:
                        val = v3
                if key == "value" and val is not None:
                    for f4, _, v4 in _fields(val):
                        if f4 == 8:
                            tensor = v4
        if op == "Const" and tensor is not None:
            a = _tensor(tensor)
            if a is not None:
                out[name] = a
    return out


# ---- writers (test fixtures, and an export path for anyone who wants a .pb of the arena) ---------------------------
def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc(field, payload):
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def write_frozen_graph(path, tensors):
    """Serialize {name: float32 array} as a GraphDef of Const nodes (what tf.graph_util.convert_variables_to_constants
    leaves of the variables)."""
    blob = bytearray()
    for name, arr in tensors.items():
        a = np.ascontiguousarray(arr, "<f4")
        shape = b"".join(_enc(2, _enc_varint(1 << 3) + _enc_varint(int(d))) for d in a.shape)
        tensor = _enc_varint(1 << 3) + _enc_varint(_DT_FLOAT) + _enc(2, shape) + _enc(4, a.tobytes())
        attr_value = _enc(8, tensor)
        attr = _enc(5, _enc(1, b"value") + _enc(2, attr_value))
        dtype_attr = _enc(5, _enc(1, b"dtype") + _enc(2, _enc_varint(6 << 3) + _enc_varint(_DT_FLOAT)))
        node = _enc(1, name.encode()) + _enc(2, b"Const") + attr + dtype_attr
        blob += _enc(1, node)
    with open(path, "wb") as f:
        f.write(bytes(blob))


# ---- Saver-V2 checkpoints (tensor bundle: <prefix>.index + <prefix>.data-00000-of-00001) -----------------------------
_TABLE_MAGIC = 0xdb4775248b80fb57


def _block_entries(block):
    """(key, value) pairs of one LevelDB table block (without its 5-byte trailer)."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    i, key = 0, b""
    while i < end:
        shared, i = _varint(block, i)
        non_shared, i = _varint(block, i)
        vlen, i = _varint(block, i)
        key = key[:shared] + bytes(block[i:i + non_shared])
        i += non_shared
        yield key, block[i:i + vlen]
        i += vlen


def _read_block(buf, off, size):
    ctype = buf[off + size]
    if ctype != 0:
        raise ValueError("compressed table block (type %d): tensor bundles are written uncompressed" % ctype)
    return buf[off:off + size]


@_corrupt_as_value_error("Saver-V2 checkpoint")
def read_checkpoint(prefix):
    """{variable name: ndarray} of a Saver-V2 checkpoint given its prefix (e.g. checkpoints/VGGnet_fast_rcnn_iter_50000.ckpt).
    Only float32 tensors of single-slice entries are returned (that is all this model has)."""
    idx = memoryview(open(prefix + ".index", "rb").read())
    if len(idx) < 48 or struct.unpack_from("<Q", idx, len(idx) - 8)[0] != _TABLE_MAGIC:
        raise ValueError(prefix + ".index is not a tensor-bundle table")
    foot = idx[len(idx) - 48:]
    _, i = _varint(foot, 0)           # metaindex handle: offset, size
    _, i = _varint(foot, i)
    ioff, i = _varint(foot, i)        # index handle
    isize, i = _varint(foot, i)
    entries = {}
    num_shards = 1
    for _, handle in _block_entries(_read_block(idx, ioff, isize)):
        boff, j = _varint(handle, 0)
        bsize, j = _varint(handle, j)
        for key, val in _block_entries(_read_block(idx, boff, bsize)):
            if key == b"":
                for f, _, v in _fields(val):          # BundleHeaderProto.num_shards = 1
                    if f == 1:
                        num_shards = v
                continue
            e = {"dtype": 0, "shape": [], "shard": 0, "offset": 0, "size": 0, "sliced": False}
            for f, _, v in _fields(val):
                if f == 1:
                    e["dtype"] = v
                elif f == 2:
                    for f2, _, v2 in _fields(v):
                        if f2 == 2:
                            size = 0
                            for f3, _, v3 in _fields(v2):
                                if f3 == 1:
                                    size = v3
                            e["shape"].append(int(size))
                elif f == 3:
                    e["shard"] = v
                elif f == 4:
                    e["offset"] = v
                elif f == 5:
                    e["size"] = v
                elif f == 7:
                    e["sliced"] = True
            entries[key.decode()] = e
    shards = {}
    out = {}
    for name, e in entries.items():
        if e["dtype"] != _DT_FLOAT or e["sliced"]:
            continue
        if e["shard"] not in shards:
            shards[e["shard"]] = np.memmap("%s.data-%05d-of-%05d" % (prefix, e["shard"], num_shards), dtype=np.uint8, mode="r")
        n_el = int(np.prod(e["shape"])) if e["shape"] else 1
        if e["offset"] + e["size"] > shards[e["shard"]].shape[0] or e["size"] != 4 * n_el:
            raise ValueError("%s: entry %s (offset %d, size %d, shape %s) does not fit its data shard of %d bytes" % (
                prefix, name, e["offset"], e["size"], e["shape"], shards[e["shard"]].shape[0]))
        raw = shards[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        out[name] = np.frombuffer(bytes(raw), "<f4").reshape(e["shape"]).astype(np.float32)
    return out


def _table_block(pairs):
    """one table block with a restart point at every entry (shared = 0), plus its trailer (no compression, crc unchecked)."""
    body, restarts = bytearray(), []
    for k, v in pairs:
        restarts.append(len(body))
        body += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
    for r in restarts or [0]:
        body += struct.pack("<I", r)
    body += struct.pack("<I", max(len(restarts), 1))
    return bytes(body), bytes(body) + b"\x00" + b"\x00\x00\x00\x00"


def write_checkpoint(prefix, tensors):
    """Write {name: float32 array} as a single-shard tensor bundle (the layout tf.train.Saver(write_version=V2) produces)."""
    data = bytearray()
    pairs = [(b"", _enc_varint(1 << 3) + _enc_varint(1))]                    # BundleHeaderProto{num_shards: 1}
    for name in sorted(tensors):
        a = np.ascontiguousarray(tensors[name], "<f4")
        shape = b"".join(_enc(2, _enc_varint(1 << 3) + _enc_varint(int(d))) for d in a.shape)
        entry = (_enc_varint(1 << 3) + _enc_varint(_DT_FLOAT) + _enc(2, shape) + _enc_varint(4 << 3) + _enc_varint(len(data)) +
                 _enc_varint(5 << 3) + _enc_varint(a.nbytes))
        pairs.append((name.encode(), entry))
        data += a.tobytes()
    blob = bytearray()
    body, full = _table_block(pairs)
    data_handle = _enc_varint(0) + _enc_varint(len(body))
    blob += full
    meta_body, meta_full = _table_block([])
    meta_handle = _enc_varint(len(blob)) + _enc_varint(len(meta_body))
    blob += meta_full
    index_body, index_full = _table_block([(pairs[-1][0] + b"\xff", data_handle)])
    index_handle = _enc_varint(len(blob)) + _enc_varint(len(index_body))
    blob += index_full
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    blob += footer
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(blob))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))


# ---- name mapping ----------------------------------------------------------------------------------------------
def arena_from_named(tensors, strict=True, base=None):
    """{TF variable name: array} -> flat arena. Accepts the scope names of the manifest, optionally with a ':0' suffix or a
    '/read' alias (frozen graphs keep `name/read` Identity nodes next to the Const). Missing entries keep `base`'s values
    (default zeros) unless strict."""
    arena = np.zeros((_w.WEIGHT_FLOATS,), np.float32) if base is None else np.array(base, np.float32).reshape(-1).copy()
    views = _w.arena_views(arena)
    norm = {}
    for k, v in tensors.items():
        k = k[:-2] if k.endswith(":0") else k
        norm[k] = v
    missing = []
    for name, shape, _ in _w.MANIFEST:
        if name in norm:
            a = np.asarray(norm[name], np.float32)
            if a.size != int(np.prod(shape)):
                raise ValueError("%s: expected shape %s, file has %s" % (name, shape, a.shape))
            views[name][...] = a.reshape(shape)
        else:
            missing.append(name)
    if strict and missing:
        raise KeyError("weight file lacks %d variables, first: %s" % (len(missing), missing[0]))
    return arena, missing


def arena_from_vgg_npy(path, base=None):
    """`VGG_imagenet.npy` (Network.load, network.py:40-53: dict layer -> {'weights', 'biases'}): fills conv1_1 .. conv5_3,
    everything else keeps `base` (the reference trains those from their initialisers)."""
    d = np.load(path, allow_pickle=True, encoding="latin1")
    d = d.item() if isinstance(d, np.ndarray) else d
    named = {}
    for layer, params in d.items():
        if isinstance(params, dict):
            for k, v in params.items():
                named["%s/%s" % (layer, k)] = v
    return arena_from_named(named, strict=False, base=base)


def load_any(path, base=None):
    """.pb -> frozen graph, <prefix>(.index) -> Saver-V2 checkpoint, .npy / .npz -> flat arena, manifest-keyed dict or VGG-style
    nested dict."""
    import os
    if path.endswith(".pb"):
        return arena_from_named(read_frozen_graph(path), strict=True)[0]
    if path.endswith(".index") or os.path.exists(path + ".index"):
        prefix = path[:-6] if path.endswith(".index") else path
        return arena_from_named(read_checkpoint(prefix), strict=True)[0]
    obj = np.load(path, allow_pickle=True, encoding="latin1")
    if isinstance(obj, np.ndarray) and obj.dtype != object:
        a = np.asarray(obj, np.float32).reshape(-1)
        if a.size != _w.WEIGHT_FLOATS:
            raise ValueError("flat arena must hold %d floats" % _w.WEIGHT_FLOATS)
        return a
    d = obj.item() if isinstance(obj, np.ndarray) else dict(obj)
    if any(isinstance(v, dict) for v in d.values()):
        return arena_from_vgg_npy(path, base=base)[0]
    return arena_from_named(d, strict=True)[0]
