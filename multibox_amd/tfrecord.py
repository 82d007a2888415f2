"""TFRecord files of tf.train.Example without TensorFlow.

The reference reads its data with tf.TFRecordReader + tf.parse_single_example
(inputs.py:225-247, detect.py:158-171).  TF is not available here, so this is an own
implementation of the two public formats involved:
  * TFRecord framing: uint64 length, uint32 masked-crc32c(length), payload, uint32 masked-crc32c(payload);
  * protobuf wire format of Example { Features { map<string, Feature{bytes_list|float_list|int64_list}> } }.
"""
from __future__ import annotations

import struct

_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    if len(data) >= 256:                       # libmbx's slice-by-8 (host code); the Python loop below gives the same value
        try:
            from . import _lib
            buf = bytes(data)
            return int(_lib.lib().mbx_crc32c(buf, len(buf), 0))
        except (OSError, RuntimeError, AttributeError):
            pass
    t, c = _crc_table(), 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def read_records(path, verify=False):
    """Yield the raw payload of every record of a TFRecord file."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError("truncated TFRecord header in %s" % path)
            (n,), (hcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            data = f.read(n)
            tail = f.read(4)
            if len(data) < n or len(tail) < 4:
                raise IOError("truncated TFRecord payload in %s" % path)
            if verify and (masked_crc(head[:8]) != hcrc or masked_crc(data) != struct.unpack("<I", tail)[0]):
                raise IOError("TFRecord crc mismatch in %s" % path)
            yield data


def write_records(path, payloads):
    with open(path, "wb") as f:
        for d in payloads:
            n = struct.pack("<Q", len(d))
            f.write(n + struct.pack("<I", masked_crc(n)) + d + struct.pack("<I", masked_crc(d)))


# ------------------------------------------------------------------------- protobuf wire format
def _varint(buf, i):
    r, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, i
        s += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) of one message."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            l, i = _varint(buf, i)
            v, i = buf[i:i + l], i + l
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _parse_feature(buf):
    for fn, wt, v in _fields(buf):
        if fn == 1:      # BytesList
            return [bytes(x) for f2, _, x in _fields(v) if f2 == 1]
        if fn == 2:      # FloatList (packed or not)
            out = []
            for f2, w2, x in _fields(v):
                if f2 == 1:
                    out += list(struct.unpack("<%df" % (len(x) // 4), x)) if w2 == 2 else [struct.unpack("<f", x)[0]]
            return out
        if fn == 3:      # Int64List (packed or not)
            out = []
            for f2, w2, x in _fields(v):
                if f2 != 1:
                    continue
                if w2 == 2:
                    j = 0
                    while j < len(x):
                        val, j = _varint(x, j)
                        out.append(val - (1 << 64) if val >> 63 else val)
                else:
                    out.append(x - (1 << 64) if x >> 63 else x)
            return out
    return []


def parse_example(payload: bytes) -> dict:
    """tf.train.Example -> {feature name: list of bytes / float / int}."""
    out = {}
    for fn, _, features in _fields(payload):
        if fn != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            key, val = None, []
            for f3, _, x in _fields(entry):
                if f3 == 1:
                    key = bytes(x).decode("utf-8")
                elif f3 == 2:
                    val = _parse_feature(x)
            if key is not None:
                out[key] = val
    return out


# ------------------------------------------------------------------------------ writer (tools / tests)
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(features: dict) -> bytes:
    """features: name -> bytes | str | list of int | list of float."""
    entries = b""
    for k, v in features.items():
        if isinstance(v, (bytes, str)):
            v = [v]
        v = list(v)
        if len(v) and isinstance(v[0], (bytes, str)):
            feat = _ld(1, b"".join(_ld(1, x if isinstance(x, bytes) else x.encode()) for x in v))
        elif len(v) and isinstance(v[0], float):
            feat = _ld(2, _ld(1, struct.pack("<%df" % len(v), *v)))
        else:
            feat = _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in v)))
        entries += _ld(1, _ld(1, k.encode()) + _ld(2, feat))
    return _ld(1, entries)
