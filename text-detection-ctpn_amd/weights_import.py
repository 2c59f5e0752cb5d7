"""Weight import from the reference's own formats into the flat arena.

The reference restores weights three ways, none of which needs TensorFlow to READ:
  * `Network.load` (lib/networks/network.py:40-53): a pickled dict {layer: {'weights': ..., 'biases': ...}} saved with
    np.save -- the ImageNet VGG16 initialisation `data/pretrain/VGG_imagenet.npy` (conv layers only);
  * a frozen graph `data/ctpn.pb` (ctpn/demo_pb.py:60-66; written by ctpn/generate_pb.py): a serialized GraphDef whose
    Const nodes carry every variable under its scope name (`conv1_1/weights`, ...);
  * a Saver-V2 checkpoint (ctpn/demo.py:84-92): `*.index` (an SSTable of BundleEntryProto) + `*.data-00000-of-00001`.
This module reads the first two with a ~60-line protobuf wire-format walker (GraphDef / NodeDef / AttrValue / TensorProto
field numbers from tensorflow/core/framework/*.proto, TF 1.3) and, for checkpoints, the uncompressed single-shard bundle
layout. There is no trained model in the reference tree and no TensorFlow in this image: the readers are tested on
files this package WRITES itself with the same wire format (`write_frozen_graph`), i.e. parity with a real TF-written
file is UNPINNED.
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
    """.pb -> frozen graph, .npy / .npz -> flat arena, manifest-keyed dict or VGG-style nested dict."""
    if path.endswith(".pb"):
        return arena_from_named(read_frozen_graph(path), strict=True)[0]
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
