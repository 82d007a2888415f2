"""TensorFlow V1 checkpoints ("model.ckpt-N" written by tf.train.Saver of TF <= 0.11) without TensorFlow -- SURVEY 8(f) F2.

The reference restores `inception_resnet_v2_2016_08_30.ckpt` or one of its own checkpoints through
`slim.assign_from_checkpoint_fn` (train.py:15-90) and, for inference, the EMA shadows (detect.py:336-346).  TF is not
installable here, so this module restates the public on-disk format:

  * the file is a leveldb-style sorted table (tensorflow/core/lib/io/table*.cc): data blocks of prefix-compressed
    (key, value) entries with a restart array, each block followed by a 1-byte compression type (0 none, 1 snappy) and
    a masked crc32c; an index block mapping last keys to block handles; a 48-byte footer ending in the magic
    0xdb4775248b80fb57;
  * every value is a `SavedTensorSlices` protobuf (tensorflow/core/util/saved_tensor_slice.proto): the entry with the
    empty key carries `meta` (name, shape, dtype, slices of every tensor), every other entry carries `data` (one slice
    of one tensor as a TensorProto: float_val / double_val / int_val / int64_val packed, or tensor_content).

PARITY UNPINNED: no TF-written file is available in this environment (the reference ships none), so the reader is checked
against this module's own writer, against hand-built snappy / table / protobuf byte strings, and nothing else.  The
variable names are slim's scopes as they appear in model.py ("inferred"); a real checkpoint's key list may
still differ, in which case `restore` reports every missing name at once.
"""
from __future__ import annotations

import struct

import numpy as np

from .tfrecord import masked_crc

MAGIC = 0xDB4775248B80FB57
EMA_SUFFIX = "/ExponentialMovingAverage"          # tf.train.ExponentialMovingAverage.average_name (train.py:56-58)
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_NP = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64}
_DT = {np.dtype(v): k for k, v in _NP.items()}


# ------------------------------------------------------------------------------------------ varints / protobuf wire
def _varint(buf, i):
    v = s = 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        if b < 0x80:
            return v, i
        s += 7


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _fields(buf):
    """(field number, wire type, value) of one protobuf message; value is an int or a memoryview."""
    buf = memoryview(buf)
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _ld(field, payload):
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + bytes(payload)


def _vi(field, value):
    return _enc_varint(field << 3) + _enc_varint(value)


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ------------------------------------------------------------------------------------------------------- snappy
def snappy_decompress(data):
    """Raw snappy block format (the table's compression type 1)."""
    data = memoryview(data)
    n, i = _varint(data, 0)
    out = bytearray()
    while i < len(data):
        tag = data[i]
        i += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[i:i + nb], "little")
                i += nb
            ln += 1
            out += data[i:i + ln]
            i += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | data[i]
            i += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = data[i] | (data[i + 1] << 8)
            i += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[i:i + 4], "little")
            i += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy stream")
        start = len(out) - off
        if off >= ln:
            out += out[start:start + ln]
        else:                                           # overlapping copy: byte by byte semantics
            for k in range(ln):
                out.append(out[start + k])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# -------------------------------------------------------------------------------------------------- sorted table
def _read_block(buf, off, size, verify):
    raw = buf[off:off + size]
    ctype = buf[off + size]
    if verify:
        (crc,) = struct.unpack("<I", buf[off + size + 1:off + size + 5])
        if masked_crc(bytes(buf[off:off + size + 1])) != crc:
            raise IOError("table block checksum mismatch at offset %d" % off)
    if ctype == 1:
        raw = memoryview(snappy_decompress(raw))
    elif ctype != 0:
        raise IOError("unknown block compression type %d" % ctype)
    return raw


def _block_entries(block):
    n = len(block)
    (num_restarts,) = struct.unpack("<I", block[n - 4:n])
    end = n - 4 - 4 * num_restarts
    i, key = 0, b""
    while i < end:
        shared, i = _varint(block, i)
        non_shared, i = _varint(block, i)
        vlen, i = _varint(block, i)
        key = key[:shared] + bytes(block[i:i + non_shared])
        i += non_shared
        yield key, block[i:i + vlen]
        i += vlen


def read_table(path, verify=False):
    """Yield (key, value) of every entry of a leveldb-format table file, in key order."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != MAGIC:
        raise IOError("%s is not a TensorFlow V1 checkpoint table (bad magic)" % path)
    footer = buf[-48:]
    _, i = _varint(footer, 0)          # metaindex handle: offset, size (unused)
    _, i = _varint(footer, i)
    ioff, i = _varint(footer, i)
    isize, i = _varint(footer, i)
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, j = _varint(handle, 0)
        bsize, j = _varint(handle, j)
        for kv in _block_entries(_read_block(buf, boff, bsize, verify)):
            yield kv


class _TableWriter:
    """Uncompressed blocks, one restart point per entry (no prefix sharing): the simplest valid table."""

    def __init__(self, f, block_size=4096):
        self.f, self.block_size = f, block_size
        self.off = 0
        self.entries, self.bytes = [], 0
        self.index = []                   # (last key of block, offset, size)
        self.last_key = None

    def _emit(self, entries):
        body, restarts = bytearray(), []
        for k, v in entries:
            restarts.append(len(body))
            body += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
        for r in restarts:
            body += struct.pack("<I", r)
        body += struct.pack("<I", len(restarts))
        trailer = b"\x00"
        blob = bytes(body) + trailer
        self.f.write(blob + struct.pack("<I", masked_crc(blob)))
        handle = (self.off, len(body))
        self.off += len(body) + 5
        return handle

    def add(self, key, value):
        assert self.last_key is None or key > self.last_key, "table keys must be added in increasing order"
        self.last_key = key
        self.entries.append((key, value))
        self.bytes += len(key) + len(value)
        if self.bytes >= self.block_size:
            self.flush()

    def flush(self):
        if self.entries:
            off, size = self._emit(self.entries)
            self.index.append((self.entries[-1][0], off, size))
            self.entries, self.bytes = [], 0

    def finish(self):
        self.flush()
        moff, msize = self._emit([])                                  # empty metaindex block
        ioff, isize = self._emit([(k, _enc_varint(o) + _enc_varint(s)) for k, o, s in self.index])
        footer = _enc_varint(moff) + _enc_varint(msize) + _enc_varint(ioff) + _enc_varint(isize)
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
        self.f.write(footer)


# --------------------------------------------------------------------------------- SavedTensorSlices <-> arrays
def _parse_shape(buf):
    dims = []
    for f, _, v in _fields(buf):
        if f == 2:                                     # TensorShapeProto.Dim
            size = 0
            for g, _, w in _fields(v):
                if g == 1:
                    size = _signed(w)
            dims.append(size)
    return tuple(dims)


def _parse_slice(buf):
    """TensorSliceProto -> [(start, length or None)] per dimension."""
    ext = []
    for f, _, v in _fields(buf):
        if f == 1:
            start, length = 0, None
            for g, _, w in _fields(v):
                if g == 1:
                    start = _signed(w)
                elif g == 2:
                    length = _signed(w)
            ext.append((start, length))
    return ext


def _parse_tensor(buf):
    dtype, content, vals = DT_FLOAT, None, None
    for f, wt, v in _fields(buf):
        if f == 1:
            dtype = v
        elif f == 4:
            content = bytes(v)
        elif f in (5, 6) and wt == 2:                   # packed float_val / double_val
            vals = np.frombuffer(bytes(v), dtype="<f4" if f == 5 else "<f8")
        elif f == 5 and wt == 5:
            vals = np.append(vals if vals is not None else np.zeros(0, "<f4"), np.frombuffer(bytes(v), "<f4"))
        elif f in (7, 10):                              # int_val / int64_val: varints, packed or not
            if wt == 2:
                out, i = [], 0
                while i < len(v):
                    x, i = _varint(v, i)
                    out.append(_signed(x))
            else:
                out = list(vals) if vals is not None else []
                out.append(_signed(v))
            vals = np.array(out, dtype=np.int64)
    if dtype not in _NP:
        raise ValueError("unsupported tensor dtype %d in checkpoint" % dtype)
    if content is not None:
        return np.frombuffer(content, dtype=np.dtype(_NP[dtype]).newbyteorder("<"))
    return (vals if vals is not None else np.zeros(0)).astype(_NP[dtype])


def load(path, verify=False, names=None):
    """{variable name: ndarray} of a V1 checkpoint (every dtype the reference saves: float32 variables, int64 step).
    `names`: optional predicate / container to skip tensors that are not wanted."""
    meta, parts = {}, {}
    want = (lambda n: True) if names is None else (names if callable(names) else (lambda n: n in names))
    for key, value in read_table(path, verify):
        for f, _, v in _fields(value):
            if f == 1:                                  # SavedTensorSliceMeta
                for g, _, w in _fields(v):
                    if g == 1:                          # SavedSliceMeta
                        name, shape = None, ()
                        for h, _, x in _fields(w):
                            if h == 1:
                                name = bytes(x).decode()
                            elif h == 2:
                                shape = _parse_shape(x)
                        meta[name] = shape
            elif f == 2:                                # SavedSlice
                name, ext, arr = None, [], None
                for g, _, w in _fields(v):
                    if g == 1:
                        name = bytes(w).decode()
                    elif g == 2:
                        ext = _parse_slice(w)
                    elif g == 3 and want(name):
                        arr = _parse_tensor(w)
                if arr is not None:
                    parts.setdefault(name, []).append((ext, arr))
    out = {}
    for name, pieces in parts.items():
        shape = meta.get(name)
        if shape is None:
            raise IOError("checkpoint has data for %r but no metadata entry" % name)
        full = None
        for ext, arr in pieces:
            if all(length is None for _, length in ext):           # the usual case: one full slice
                full = arr.reshape(shape)
                continue
            if full is None:
                full = np.zeros(shape, arr.dtype)
            idx = tuple(slice(s, None if ln is None else s + ln) for s, ln in ext)
            full[idx] = arr.reshape(full[idx].shape)
        out[name] = full
    return out


def _ordered_string(s):
    """OrderedCode::WriteString: 0x00 -> 00 ff, 0xff -> ff 00, terminator 00 01."""
    return b"".join(b"\x00\xff" if c == 0 else b"\xff\x00" if c == 255 else bytes([c]) for c in s) + b"\x00\x01"


def _slice_key(name, ndim):
    """EncodeTensorNameSlice: ordered code of (0, name, ndim, (start=0, length=-1) per dim) -- sorts after ""."""
    def num(v):
        if v == 0:
            return b"\x00"
        b = v.to_bytes((v.bit_length() + 7) // 8, "big")
        return bytes([len(b)]) + b
    return num(0) + _ordered_string(name.encode()) + num(ndim) + (b"\x80\x7f" * ndim)


def save(path, tensors):
    """Write {name: ndarray} as a V1 checkpoint table (float32 / float64 / int32 / int64), one full slice per tensor."""
    items = sorted(tensors.items(), key=lambda kv: _slice_key(kv[0], np.ndim(kv[1])))
    meta = b""
    for name, a in items:
        a = np.asarray(a)
        shape = b"".join(_ld(2, _vi(1, d)) for d in a.shape)
        full = b"".join(_ld(1, b"") for _ in a.shape)
        meta += _ld(1, _ld(1, name.encode()) + _ld(2, shape) + _vi(3, _DT[a.dtype]) + _ld(4, full))
    with open(path, "wb") as f:
        w = _TableWriter(f)
        w.add(b"", _ld(1, meta + _ld(2, _ld(1, _vi(1, 11)))))       # meta + VersionDef{producer}
        for name, a in items:
            a = np.ascontiguousarray(a)
            dt = _DT[a.dtype]
            tshape = b"".join(_ld(2, _vi(1, d)) for d in a.shape)
            if dt in (DT_FLOAT, DT_DOUBLE):
                payload = _ld(5 if dt == DT_FLOAT else 6, a.astype(a.dtype.newbyteorder("<")).tobytes())
            else:
                payload = _ld(7 if dt == DT_INT32 else 10, b"".join(_enc_varint(int(x)) for x in a.reshape(-1)))
            tensor = _vi(1, dt) + _ld(2, tshape) + payload
            full = b"".join(_ld(1, b"") for _ in a.shape)
            w.add(_slice_key(name, a.ndim), _ld(2, _ld(1, name.encode()) + _ld(2, full) + _ld(3, tensor)))
        w.finish()


# ------------------------------------------------------------------------------ variables <-> the engine's buffers
def _to_engine(name, value, shape):
    """TF layout -> the engine's: conv filters are HWIO [R,S,C_in,C_out] in slim, KRSC [C_out,R,S,C_in] here."""
    v = np.asarray(value, np.float32)
    if name.endswith("/weights"):
        if v.ndim != 4:
            raise ValueError("%s: expected a 4-D filter, checkpoint has shape %s" % (name, v.shape))
        v = v.transpose(3, 0, 1, 2)
    if tuple(v.shape) != tuple(shape):
        raise ValueError("%s: checkpoint shape %s does not match the network's %s" % (name, v.shape, tuple(shape)))
    return v


def _from_engine(name, t):
    v = t.detach().cpu().numpy().astype(np.float32)
    return v.transpose(1, 2, 3, 0) if name.endswith("/weights") else v


def model_variable_names(net, backbone_only=False):
    """slim.get_model_variables() of the built graph (train.py:48-52); backbone_only = the dict model.build returns
    for --fine_tune (model.py:326-337: everything under InceptionResnetV2/)."""
    return [n for n in net.param_index if not backbone_only or n.startswith("InceptionResnetV2/")]


def restore(path, net, fine_tune=False, use_moving_averages=False, restore_moving_averages=False, ema=None,
            tensors=None):
    """train.py:15-90 (get_init_function) on a TF V1 checkpoint.

    fine_tune: only the backbone variables are restored (the heads keep their initial values);
    use_moving_averages: every variable is read from its `<name>/ExponentialMovingAverage` entry;
    restore_moving_averages: the EMA shadows (`ema`: an object with Wema / Btema / MMema / MVema flat buffers laid
    out like net.W / Bt / MM / MV, e.g. the Trainer) are restored as well, from the shadow entries;
    missing variables raise (ignore_missing_vars=False, train.py:87-90).  Returns the restored names."""
    import torch
    names = model_variable_names(net, backbone_only=fine_tune)
    wanted = set()
    for n in names:
        if use_moving_averages or restore_moving_averages:
            wanted.add(n + EMA_SUFFIX)
        if not use_moving_averages:
            wanted.add(n)
    ck = tensors if tensors is not None else load(path, names=wanted)
    missing = sorted(w for w in wanted if w not in ck)
    if missing:
        raise KeyError("%d variables are not in the checkpoint %s, e.g. %s" % (len(missing), path, ", ".join(missing[:5])))
    if restore_moving_averages and ema is None:
        raise ValueError("restore_moving_averages needs the EMA buffers (pass the Trainer as `ema`)")
    shadow = {"W": "Wema", "Bt": "Btema", "MM": "MMema", "MV": "MVema"}
    for n in names:
        buf, off, shape, cpad = net.param_index[n]
        src = ck[n + EMA_SUFFIX] if use_moving_averages else ck[n]
        net.set_param(n, _to_engine(n, src, shape))
        if restore_moving_averages:
            v = torch.as_tensor(_to_engine(n, ck[n + EMA_SUFFIX], shape))
            flat = getattr(ema, shadow[buf])
            if cpad is not None and cpad != shape[-1]:
                K, R, S_, Cc = shape
                flat[off:off + K * R * S_ * cpad].reshape(K, R, S_, cpad)[..., :Cc].copy_(v)
            else:
                flat[off:off + v.numel()].copy_(v.reshape(-1))
    if hasattr(net, "Wb"):
        net.Wb.copy_(net.W.to(net.Wb.dtype))
    return names


def restore_for_inference(path, net, use_moving_averages=True):
    """detect.py:336-346: the live variables take the EMA shadows' values.  Returns the global step (0 if absent)."""
    restore(path, net, fine_tune=False, use_moving_averages=use_moving_averages)
    step = load(path, names=("global_step",)).get("global_step")
    return int(np.asarray(step).reshape(-1)[0]) if step is not None and np.size(step) else 0


def export(path, net, ema=None, global_step=None):
    """Write the network (and, with `ema`, the shadows under slim's EMA names) as a TF V1 checkpoint."""
    out = {}
    shadow = {"W": "Wema", "Bt": "Btema", "MM": "MMema", "MV": "MVema"}
    for n, (buf, off, shape, cpad) in net.param_index.items():
        out[n] = _from_engine(n, net.get_param(n))
        if ema is not None:
            flat = getattr(ema, shadow[buf])
            if cpad is not None and cpad != shape[-1]:
                K, R, S_, Cc = shape
                t = flat[off:off + K * R * S_ * cpad].reshape(K, R, S_, cpad)[..., :Cc]
            else:
                t = flat[off:off + int(np.prod(shape))].reshape(shape)
            out[n + EMA_SUFFIX] = _from_engine(n, t)
    if global_step is not None:
        out["global_step"] = np.array(global_step, np.int64)
    save(path, out)
    return path
