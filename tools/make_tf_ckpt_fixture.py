#!/usr/bin/env python3
"""Writes tests/golden/tf_v1_fixture.ckpt (+ .npz with the arrays it holds): a TensorFlow V1 checkpoint built from
TensorFlow's PUBLISHED on-disk format by a writer that shares NO code with multibox_amd/tf_checkpoint.py (the reader
under test).  No TF-written file exists in this image (the reference ships none, TF 0.11 cannot be
installed), so this is the strongest pin available: an independent implementation of the same specification.

  * protobuf layer: the messages of tensorflow/core/util/saved_tensor_slice.proto, tensor.proto, tensor_shape.proto,
    tensor_slice.proto and versions.proto are declared here field by field (numbers and types as published) and
    serialised by the google.protobuf RUNTIME -- not by hand-written wire code;
  * keys: saved_tensor_slice_util.cc EncodeTensorNameSlice = OrderedCode (lib/strings/ordered_code.cc):
    WriteNumIncreasing(0), WriteString(name), WriteNumIncreasing(rank), then per dimension
    WriteSignedNumIncreasing(start), WriteSignedNumIncreasing(length) with length -1 for a full extent;
  * table layer: lib/io/table_builder.cc / block_builder.cc / format.cc (leveldb's): prefix-compressed entries with a
    restart point every 16 entries, per-block trailer (compression type, masked crc32c of block + type), metaindex and
    index blocks, 48-byte footer with the magic 0xdb4775248b80fb57; index keys are SHORTENED separators as leveldb's
    FindShortestSeparator produces them;
  * compression: blocks are snappy-compressed (format_description.txt of snappy: varint length, literal and copy
    elements) when that saves at least 12.5 %, as table_builder.cc does -- with a small greedy matcher written here.

Contents: float / double / int32 / int64 tensors, a scalar, a variable saved in TWO partitions (extents with start and
length), slim-style names incl. EMA shadows, enough bytes for several data blocks, compressible and incompressible ones.
usage: python tools/make_tf_ckpt_fixture.py [outdir]"""
import os
import struct
import sys

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

MAGIC = 0xDB4775248B80FB57
DT = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9}


# ------------------------------------------------------------------------------------------------ protobuf messages
def _messages():
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="tf_v1_ckpt_fixture.proto", package="fx", syntax="proto3")

    def msg(name, fields, nested=()):
        m = descriptor_pb2.DescriptorProto(name=name)
        for fname, num, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
        for n in nested:
            m.nested_type.add().CopyFrom(n)
        return m
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    dim = msg("Dim", [("size", 1, F.TYPE_INT64, OPT, None), ("name", 2, F.TYPE_STRING, OPT, None)])
    fd.message_type.add().CopyFrom(msg("TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, ".fx.TensorShapeProto.Dim"),
                                                            ("unknown_rank", 3, F.TYPE_BOOL, OPT, None)], [dim]))
    ext = msg("Extent", [("start", 1, F.TYPE_INT64, OPT, None), ("length", 2, F.TYPE_INT64, OPT, None)])
    ext.oneof_decl.add(name="has_length")
    ext.field[1].oneof_index = 0                                      # oneof has_length { int64 length = 2; }
    fd.message_type.add().CopyFrom(msg("TensorSliceProto", [("extent", 1, F.TYPE_MESSAGE, REP, ".fx.TensorSliceProto.Extent")], [ext]))
    fd.message_type.add().CopyFrom(msg("TensorProto", [
        ("dtype", 1, F.TYPE_INT32, OPT, None), ("tensor_shape", 2, F.TYPE_MESSAGE, OPT, ".fx.TensorShapeProto"),
        ("version_number", 3, F.TYPE_INT32, OPT, None), ("tensor_content", 4, F.TYPE_BYTES, OPT, None),
        ("float_val", 5, F.TYPE_FLOAT, REP, None), ("double_val", 6, F.TYPE_DOUBLE, REP, None),
        ("int_val", 7, F.TYPE_INT32, REP, None), ("int64_val", 10, F.TYPE_INT64, REP, None)]))
    fd.message_type.add().CopyFrom(msg("VersionDef", [("producer", 1, F.TYPE_INT32, OPT, None), ("min_consumer", 2, F.TYPE_INT32, OPT, None)]))
    fd.message_type.add().CopyFrom(msg("SavedSliceMeta", [
        ("name", 1, F.TYPE_STRING, OPT, None), ("shape", 2, F.TYPE_MESSAGE, OPT, ".fx.TensorShapeProto"),
        ("type", 3, F.TYPE_INT32, OPT, None), ("slice", 4, F.TYPE_MESSAGE, REP, ".fx.TensorSliceProto")]))
    fd.message_type.add().CopyFrom(msg("SavedTensorSliceMeta", [("tensor", 1, F.TYPE_MESSAGE, REP, ".fx.SavedSliceMeta"),
                                                                ("versions", 2, F.TYPE_MESSAGE, OPT, ".fx.VersionDef")]))
    fd.message_type.add().CopyFrom(msg("SavedSlice", [("name", 1, F.TYPE_STRING, OPT, None),
                                                      ("slice", 2, F.TYPE_MESSAGE, OPT, ".fx.TensorSliceProto"),
                                                      ("data", 3, F.TYPE_MESSAGE, OPT, ".fx.TensorProto")]))
    fd.message_type.add().CopyFrom(msg("SavedTensorSlices", [("meta", 1, F.TYPE_MESSAGE, OPT, ".fx.SavedTensorSliceMeta"),
                                                             ("data", 2, F.TYPE_MESSAGE, OPT, ".fx.SavedSlice")]))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return {n: get(pool.FindMessageTypeByName("fx." + n)) for n in ("SavedTensorSlices", "TensorSliceProto", "TensorShapeProto")}


# -------------------------------------------------------------------------------------------------- OrderedCode keys
def oc_num_increasing(v):
    body = b"" if v == 0 else v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([len(body)]) + body


def oc_string(s):
    """WriteString: 0x00 -> 00 ff, 0xff -> ff 00, terminator 00 01."""
    return b"".join(b"\x00\xff" if c == 0 else b"\xff\x00" if c == 255 else bytes([c]) for c in s) + b"\x00\x01"


def oc_signed_increasing(v):
    x = ~v if v < 0 else v
    if x < 64:
        return bytes([(0x80 ^ v) & 0xFF])
    bits = x.bit_length() + 1                              # magnitude bits + sign
    n = (bits + 6) // 7                                    # 7 payload bits per byte: 2 bytes up to 13 bits + sign, ...
    buf = bytearray((v & ((1 << 80) - 1)).to_bytes(10, "big"))[10 - n:]
    header = ((1 << n) - 1) << (16 - n) if n <= 8 else (0xFF00 | (((1 << (n - 8)) - 1) << (16 - n)))
    buf[0] ^= (header >> 8) & 0xFF
    if n >= 2:
        buf[1] ^= header & 0xFF
    return bytes(buf)


def slice_key(name, extents):
    k = oc_num_increasing(0) + oc_string(name.encode()) + oc_num_increasing(len(extents))
    for start, length in extents:
        k += oc_signed_increasing(start) + oc_signed_increasing(-1 if length is None else length)
    return k


# ------------------------------------------------------------------------------------------------------- crc32c
_T = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (0x82F63B78 if _c & 1 else 0)
    _T.append(_c)


def crc32c(b):
    c = 0xFFFFFFFF
    for x in b:
        c = _T[(c ^ x) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(b):
    c = crc32c(b)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------------- snappy
def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def snappy_compress(data):
    """Greedy 4-byte-hash matcher; emits literals, 2-byte-offset copies (tag 10) and 1-byte-offset copies (tag 01)."""
    out = bytearray(_varint(len(data)))
    table = {}
    i = lit = 0
    n = len(data)

    def literal(a, b):
        while a < b:
            ln = min(b - a, 65536)
            if ln <= 60:
                out.append((ln - 1) << 2)
            elif ln <= 256:
                out.extend([60 << 2, ln - 1])
            else:
                out.extend([61 << 2, (ln - 1) & 0xFF, (ln - 1) >> 8])
            out.extend(data[a:a + ln])
            a += ln
    while i + 4 <= n:
        key = data[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and i - j < 65536:
            ln = 4
            while i + ln < n and ln < 64 and data[j + ln] == data[i + ln]:
                ln += 1
            literal(lit, i)
            off = i - j
            if 4 <= ln <= 11 and off < 2048:
                out.extend([((off >> 8) << 5) | ((ln - 4) << 2) | 1, off & 0xFF])
            else:
                out.extend([((ln - 1) << 2) | 2, off & 0xFF, off >> 8])
            i += ln
            lit = i
        else:
            i += 1
    literal(lit, n)
    return bytes(out)


# -------------------------------------------------------------------------------------------------------- table
class TableBuilder:
    def __init__(self, block_size=1024, restart_interval=16):
        self.buf = bytearray()
        self.block_size, self.restart_interval = block_size, restart_interval
        self.index = []                       # (separator key, offset, size)
        self._reset()
        self.pending = None                   # (last key of the finished block, handle)
        self.stats = {"snappy_blocks": 0, "raw_blocks": 0}

    def _reset(self):
        self.body, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def _add_entry(self, key, value):
        shared = 0
        if self.count % self.restart_interval == 0:
            if self.count:
                self.restarts.append(len(self.body))
        else:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.body += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def _finish_block(self):
        raw = bytes(self.body) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))
        self._reset()
        return raw

    def _write_raw_block(self, raw, compress=True):
        comp = snappy_compress(raw) if compress else raw
        if compress and len(comp) < len(raw) - len(raw) // 8:         # table_builder.cc: keep only if >= 12.5 % smaller
            payload, ctype = comp, 1
            self.stats["snappy_blocks"] += 1
        else:
            payload, ctype = raw, 0
            self.stats["raw_blocks"] += 1
        off = len(self.buf)
        self.buf += payload + bytes([ctype]) + struct.pack("<I", masked(payload + bytes([ctype])))
        return off, len(payload)

    @staticmethod
    def _separator(a, b):
        """leveldb BytewiseComparator::FindShortestSeparator: a <= sep < b, as short as possible."""
        m = min(len(a), len(b))
        d = 0
        while d < m and a[d] == b[d]:
            d += 1
        if d < m and a[d] < 0xFF and a[d] + 1 < b[d]:
            return a[:d] + bytes([a[d] + 1])
        return a

    def add(self, key, value):
        assert self.count == 0 or key > self.last
        if self.pending is not None:
            last_key, handle = self.pending
            self.index.append((self._separator(last_key, key), handle))
            self.pending = None
        self._add_entry(key, value)
        if len(self.body) >= self.block_size:
            last = self.last
            self.pending = (last, self._write_raw_block(self._finish_block()))

    def finish(self):
        if self.count:
            last = self.last
            self.pending = (last, self._write_raw_block(self._finish_block()))
        if self.pending is not None:
            last_key, handle = self.pending
            # FindShortSuccessor of the last key: first byte that can be incremented, incremented, rest dropped
            succ = last_key
            for i, c in enumerate(last_key):
                if c != 0xFF:
                    succ = last_key[:i] + bytes([c + 1])
                    break
            self.index.append((succ, handle))
        meta = self._write_raw_block(self._finish_block(), compress=False)          # empty metaindex block
        for k, (off, size) in self.index:
            self._add_entry(k, _varint(off) + _varint(size))
        idx = self._write_raw_block(self._finish_block(), compress=False)
        footer = _varint(meta[0]) + _varint(meta[1]) + _varint(idx[0]) + _varint(idx[1])
        self.buf += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
        return bytes(self.buf)


# ------------------------------------------------------------------------------------------------------ contents
def fixture_arrays():
    rng = np.random.RandomState(20161109)
    P = "InceptionResnetV2/"
    a = {
        "global_step": np.array(123456, np.int64),
        P + "Conv2d_1a_3x3/weights": rng.randn(3, 3, 3, 32).astype(np.float32),                  # HWIO, as slim stores it
        P + "Conv2d_1a_3x3/BatchNorm/beta": rng.randn(32).astype(np.float32),
        P + "Conv2d_1a_3x3/BatchNorm/moving_mean": rng.randn(32).astype(np.float32),
        P + "Conv2d_1a_3x3/BatchNorm/moving_variance": rng.rand(32).astype(np.float32) + 0.5,
        P + "Conv2d_1a_3x3/weights/ExponentialMovingAverage": rng.randn(3, 3, 3, 32).astype(np.float32),
        P + "Repeat/block35_1/Conv2d_1x1/weights": np.zeros((1, 1, 128, 320), np.float32),      # all zeros: compresses
        P + "Repeat/block35_1/Conv2d_1x1/biases": np.linspace(-1, 1, 320).astype(np.float32),
        "Multibox/8x8/Conv_2/weights": rng.randn(1, 1, 96, 20).astype(np.float32),
        "fixture/partitioned": np.arange(8 * 6, dtype=np.float32).reshape(8, 6) * 0.5,          # saved as rows 0:4 and 4:8
        "fixture/partitioned_cols": rng.randn(5, 400).astype(np.float32),                        # columns 0:192 | 192:400
        "fixture/doubles": rng.randn(7).astype(np.float64),
        "fixture/ints": np.array([-5, 0, 7, 2 ** 31 - 1, -2 ** 31], np.int32),
        "fixture/scalar_float": np.array(0.25, np.float32),
    }
    parts = {"fixture/partitioned": [[(0, 4), (0, None)], [(4, 4), (0, None)]],
             "fixture/partitioned_cols": [[(0, None), (0, 192)], [(0, None), (192, 208)]]}
    return a, parts


def build(outdir):
    M = _messages()
    arrays, parts = fixture_arrays()
    entries = []
    top = M["SavedTensorSlices"]()
    top.meta.versions.producer = 0
    for name in sorted(arrays):
        arr = arrays[name]
        t = top.meta.tensor.add()
        t.name, t.type = name, DT[arr.dtype]
        for d in arr.shape:
            t.shape.dim.add().size = d
        for extents in parts.get(name, [[(0, None)] * arr.ndim]):
            sl = t.slice.add()
            rec = M["SavedTensorSlices"]()
            rec.data.name = name
            idx = []
            for start, length in extents:
                for target in (sl, rec.data.slice):
                    e = target.extent.add()
                    if length is not None:                  # a full extent leaves start and length unset (tensor_slice.cc AsProto)
                        e.start, e.length = start, length
                idx.append(slice(start, None if length is None else start + length))
            piece = np.asarray(arr[tuple(idx)])
            rec.data.data.dtype = DT[arr.dtype]
            for d in piece.shape:
                rec.data.data.tensor_shape.dim.add().size = d
            field = {1: "float_val", 2: "double_val", 3: "int_val", 9: "int64_val"}[DT[arr.dtype]]
            getattr(rec.data.data, field).extend(piece.reshape(-1).tolist())
            entries.append((slice_key(name, extents), rec.SerializeToString()))
    entries.append((b"", top.SerializeToString()))
    tb = TableBuilder()
    for k, v in sorted(entries):
        tb.add(k, v)
    blob = tb.finish()
    assert tb.stats["snappy_blocks"] >= 2 and tb.stats["raw_blocks"] >= 4, tb.stats     # both kinds of data block
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "tf_v1_fixture.ckpt"), "wb") as f:
        f.write(blob)
    np.savez(os.path.join(outdir, "tf_v1_fixture.npz"), **{k.replace("/", "|"): v for k, v in arrays.items()})
    print("wrote %d bytes, %d entries, blocks %s" % (len(blob), len(entries), tb.stats))


if __name__ == "__main__":
    build(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
