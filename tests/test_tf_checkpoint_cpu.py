"""TF V1 checkpoint reader / writer and the restore modes of train.py:15-90.
No TF-written file exists in this environment (parity unpinned, see multibox_amd/tf_checkpoint.py): the reader is
checked against hand-built byte strings of the public formats and against the module's own writer."""
import os
import struct
import types

import numpy as np
import pytest

from multibox_amd import tf_checkpoint as T


def test_snappy_known_streams():
    # literal "abcd", then a 1-byte-offset copy (offset 4, length 8: overlapping -> repeats the pattern)
    assert T.snappy_decompress(bytes([12, 0x0C]) + b"abcd" + bytes([0x11, 4])) == b"abcdabcdabcd"
    # literal "xyz12", 2-byte-offset copy (offset 5, length 5), literal "!"
    s = bytes([11, 4 << 2]) + b"xyz12" + bytes([(4 << 2) | 2, 5, 0]) + bytes([0]) + b"!"
    assert T.snappy_decompress(s) == b"xyz12xyz12!"
    # long literal with a 2-byte length field (tag 61)
    body = bytes(range(200)) * 2
    s = T._enc_varint(400) + bytes([61 << 2]) + struct.pack("<H", 399) + body
    assert T.snappy_decompress(s) == body
    with pytest.raises(ValueError):
        T.snappy_decompress(bytes([4, 0x11, 9]))                    # copy before any output


def test_block_with_shared_key_prefixes_and_snappy_block():
    # entries: "apple"->"1", "apply"->"22" (shares "appl"), "b"->"" ; one restart point
    body = bytes([0, 5, 1]) + b"apple" + b"1" + bytes([4, 1, 2]) + b"y" + b"22" + bytes([0, 1, 0]) + b"b"
    body += struct.pack("<II", 0, 1)
    got = [(k, bytes(v)) for k, v in T._block_entries(memoryview(body))]
    assert got == [(b"apple", b"1"), (b"apply", b"22"), (b"b", b"")]
    # the same block stored snappy-compressed (all literal) with its type byte and crc
    comp = T._enc_varint(len(body)) + bytes([(len(body) - 1) << 2]) + body
    blob = comp + b"\x01"
    buf = memoryview(blob + struct.pack("<I", T.masked_crc(blob)))
    blk = T._read_block(buf, 0, len(comp), verify=True)
    assert [(k, bytes(v)) for k, v in T._block_entries(blk)] == got
    bad = bytearray(buf); bad[3] ^= 1
    with pytest.raises(IOError):
        T._read_block(memoryview(bytes(bad)), 0, len(comp), verify=True)


def test_save_load_roundtrip_all_dtypes(tmp_path):
    rng = np.random.RandomState(0)
    tensors = {"a/weights": rng.randn(3, 3, 8, 16).astype(np.float32), "a/biases": rng.randn(16).astype(np.float32),
               "z/double": rng.randn(5, 2), "counts": np.arange(-3, 9, dtype=np.int32).reshape(3, 4),
               "global_step": np.array(123456789012, np.int64), "empty_name_\xff": np.zeros((2, 0), np.float32)}
    for i in range(40):                                              # many blocks
        tensors["layer_%02d/BatchNorm/beta" % i] = rng.randn(700).astype(np.float32)
    p = str(tmp_path / "model.ckpt-7")
    T.save(p, tensors)
    got = T.load(p, verify=True)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == np.asarray(v).dtype and got[k].shape == np.asarray(v).shape, k
        assert np.array_equal(got[k], v), k
    only = T.load(p, names=("a/biases", "global_step"))
    assert set(only) == {"a/biases", "global_step"}
    keys = [k for k, _ in T.read_table(p)]
    assert keys[0] == b"" and keys == sorted(keys)
    with open(p, "rb") as f:
        raw = f.read()
    assert struct.unpack("<Q", raw[-8:])[0] == T.MAGIC
    (tmp_path / "junk").write_bytes(b"x" * 100)
    with pytest.raises(IOError):
        T.load(str(tmp_path / "junk"))


def test_partitioned_slices_are_assembled(tmp_path):
    """A variable saved as two slices (tf partitioned variables): extents carry start / length."""
    a = np.arange(24, dtype=np.float32).reshape(4, 6)

    def ext(start=None, length=None):
        e = b""
        if start:
            e += T._vi(1, start)
        if length is not None:
            e += T._vi(2, length)
        return T._ld(1, e)

    def entry(sl_bytes, arr):
        tensor = T._vi(1, T.DT_FLOAT) + T._ld(5, arr.astype("<f4").tobytes())
        return T._ld(2, T._ld(1, b"v") + T._ld(2, sl_bytes) + T._ld(3, tensor))
    shape = T._ld(2, T._vi(1, 4)) + T._ld(2, T._vi(1, 6))
    meta = T._ld(1, T._ld(1, T._ld(1, b"v") + T._ld(2, shape) + T._vi(3, T.DT_FLOAT)))
    p = str(tmp_path / "part.ckpt")
    with open(p, "wb") as f:
        w = T._TableWriter(f)
        w.add(b"", meta)
        w.add(b"\x00v\x00\x01a", entry(ext(0, 1) + ext(), a[:1]))
        w.add(b"\x00v\x00\x01b", entry(ext(1, 3) + ext(), a[1:]))
        w.finish()
    assert np.array_equal(T.load(p)["v"], a)


@pytest.fixture(scope="module")
def nets():
    from multibox_amd.engine import Net
    a = Net(batch=1, mode="train", device="cpu", seed=3, repeats=(1, 1, 1))
    b = Net(batch=1, mode="train", device="cpu", seed=4, repeats=(1, 1, 1))
    return a, b


def _ema_like(net, fill):
    import torch
    return types.SimpleNamespace(Wema=torch.full_like(net.W, fill), Btema=torch.full_like(net.Bt, fill),
                                 MMema=torch.full_like(net.MM, fill), MVema=torch.full_like(net.MV, fill))


def test_export_restore_modes(tmp_path, nets):
    """train.py:15-90: all variables / backbone only (--fine_tune) / from the shadows (--use_moving_averages) /
    shadows restored too (--restore_moving_averages); missing variables raise."""
    import torch
    src, dst = nets
    gen = torch.Generator().manual_seed(0)
    src.Bt.copy_(torch.randn(src.Bt.shape, generator=gen)); src.MM.copy_(torch.randn(src.MM.shape, generator=gen))
    src.MV.copy_(torch.rand(src.MV.shape, generator=gen) + 0.5)
    ema = types.SimpleNamespace(Wema=src.W * 0.5, Btema=src.Bt * 0.5, MMema=src.MM * 0.5, MVema=src.MV * 0.5)
    p = str(tmp_path / "model.ckpt-11")
    T.export(p, src, ema=ema, global_step=11)
    ck = T.load(p)
    assert ck["InceptionResnetV2/Conv2d_1a_3x3/weights"].shape == (3, 3, 3, 32)          # HWIO, C_in un-padded
    assert ck["Multibox/8x8/Conv/weights" + T.EMA_SUFFIX].shape == (1, 1, 1536, 96)
    assert int(ck["global_step"]) == 11
    w0 = src.get_param("InceptionResnetV2/Conv2d_1a_3x3/weights")                        # KRSC in the engine
    assert np.array_equal(ck["InceptionResnetV2/Conv2d_1a_3x3/weights"][1, 2, 0, 5], w0[5, 1, 2, 0].numpy())

    def fresh():
        dst.init_weights(4)
        dst.Bt.zero_(); dst.MM.zero_(); dst.MV.fill_(1.0)
        return dst.W.clone()
    # 1. everything
    fresh()
    names = T.restore(p, dst)
    assert len(names) == len(dst.param_index)
    for t in ("W", "Bt", "MM", "MV"):
        assert torch.equal(getattr(dst, t), getattr(src, t)), t
    assert torch.equal(dst.Wb, dst.W.to(torch.bfloat16))
    # 2. --fine_tune: heads keep their initial values
    w_init = fresh()
    T.restore(p, dst, fine_tune=True)
    for n, (buf, off, shape, cpad) in dst.param_index.items():
        same_as_src = torch.equal(dst.get_param(n), src.get_param(n))
        if n.startswith("InceptionResnetV2/"):
            assert same_as_src, n
        elif buf == "W":
            ref = w_init[off:off + dst.get_param(n).numel()].reshape(dst.get_param(n).shape) if cpad in (None, shape[-1]) else None
            assert ref is None or torch.equal(dst.get_param(n), ref), n
    # 3. --use_moving_averages: live variables <- shadows
    fresh()
    T.restore(p, dst, use_moving_averages=True)
    assert torch.equal(dst.W, src.W * 0.5) and torch.equal(dst.MV, src.MV * 0.5)
    # 4. --restore_moving_averages: live <- live, shadows <- shadows
    fresh()
    e2 = _ema_like(dst, 7.0)
    T.restore(p, dst, restore_moving_averages=True, ema=e2)
    assert torch.equal(dst.W, src.W) and torch.equal(e2.Btema, src.Bt * 0.5) and torch.equal(e2.MVema, src.MV * 0.5)
    for n, (buf, off, shape, cpad) in dst.param_index.items():      # weights: per variable (padding lanes stay untouched)
        if buf == "W":
            k = int(np.prod(shape[:-1])) * (cpad or shape[-1]) if len(shape) == 4 else int(np.prod(shape))
            got = e2.Wema[off:off + k]
            got = got.reshape(shape[0], shape[1], shape[2], cpad)[..., :shape[3]] if len(shape) == 4 else got
            assert torch.equal(got, src.get_param(n) * 0.5), n
    with pytest.raises(ValueError):
        T.restore(p, dst, restore_moving_averages=True)
    # 5. inference restore (detect.py:336-346)
    fresh()
    assert T.restore_for_inference(p, dst) == 11
    assert torch.equal(dst.W, src.W * 0.5)
    # 6. a checkpoint without the heads: fine with --fine_tune, KeyError otherwise (ignore_missing_vars=False)
    p2 = str(tmp_path / "backbone.ckpt")
    T.save(p2, {k: v for k, v in ck.items() if k.startswith("InceptionResnetV2/")})
    fresh()
    T.restore(p2, dst, fine_tune=True)
    with pytest.raises(KeyError) as ei:
        T.restore(p2, dst)
    assert "Multibox/" in str(ei.value)
    # 7. wrong shape is reported by name
    bad = dict(ck); bad["InceptionResnetV2/Conv2d_1a_3x3/weights"] = np.zeros((3, 3, 3, 16), np.float32)
    with pytest.raises(ValueError) as ei:
        T.restore(None, dst, tensors=bad)
    assert "Conv2d_1a_3x3" in str(ei.value)


def test_reader_against_an_independently_written_checkpoint(tmp_path):
    """Row F2's pin: tests/golden/tf_v1_fixture.ckpt is a V1 checkpoint written from TensorFlow's
    PUBLISHED format by tools/make_tf_ckpt_fixture.py, which shares no code with the reader -- protobufs serialised by
    the google.protobuf runtime from the published field numbers, OrderedCode slice keys, leveldb blocks WITH prefix
    compression, shortened index separators, snappy-compressed and raw blocks, masked crc32c.  It holds float / double /
    int32 / int64 tensors, scalars and two PARTITIONED variables (rows 0:4 | 4:8 and columns 0:192 | 192:400).
    NO TF-WRITTEN FILE EXISTS IN THIS IMAGE (the reference ships none; TF 0.11 cannot be installed): this is an
    independent implementation of the same specification, not TensorFlow's own bytes."""
    import subprocess
    import sys
    from multibox_amd import tf_checkpoint as TF
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    got = TF.load(os.path.join(gold, "tf_v1_fixture.ckpt"), verify=True)          # every block checksum verified
    want = {k.replace("|", "/"): v for k, v in np.load(os.path.join(gold, "tf_v1_fixture.npz")).items()}
    assert sorted(got) == sorted(want) and len(want) == 14
    for k, w in want.items():
        assert got[k].shape == w.shape and got[k].dtype == w.dtype and np.array_equal(got[k], w), k
    assert got["global_step"].shape == () and int(got["global_step"]) == 123456
    assert np.array_equal(got["fixture/partitioned"], np.arange(48, dtype=np.float32).reshape(8, 6) * 0.5)
    # the table really exercises what the reader's own writer never produces: shared key prefixes and snappy blocks
    raw = open(os.path.join(gold, "tf_v1_fixture.ckpt"), "rb").read()
    kinds, shared = set(), 0
    buf = memoryview(raw)
    footer = buf[-48:]
    _, i = TF._varint(footer, 0); _, i = TF._varint(footer, i)
    ioff, i = TF._varint(footer, i); isize, i = TF._varint(footer, i)
    for _, handle in TF._block_entries(TF._read_block(buf, ioff, isize, True)):
        boff, j = TF._varint(handle, 0); bsize, j = TF._varint(handle, j)
        kinds.add(raw[boff + bsize])
        blk = TF._read_block(buf, boff, bsize, True)
        k = 0
        s, k = TF._varint(blk, 0)
        end = len(blk) - 4 - 4 * struct.unpack("<I", blk[-4:])[0]
        pos = 0
        while pos < end:
            sh, pos = TF._varint(blk, pos); ns, pos = TF._varint(blk, pos); vl, pos = TF._varint(blk, pos)
            shared += sh > 0
            pos += ns + vl
    assert kinds == {0, 1} and shared >= 5
    # the committed fixture is exactly what the committed script writes
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(gold), "..", "tools", "make_tf_ckpt_fixture.py"), str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    assert open(str(tmp_path / "tf_v1_fixture.ckpt"), "rb").read() == raw
