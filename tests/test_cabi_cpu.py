"""CPU-side checks of the C-ABI: the library loads, exports every symbol include/mbx.h
declares, and the HOST entry point (priors) is bit-exact.  No device compute here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from multibox_amd import _lib
    return _lib.lib()


def test_exports_every_declared_symbol(lib):
    from multibox_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mbx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mbx_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libmbx.so does not export %s" % name
    assert sorted(_lib.declared_symbols()) == declared, "ctypes table and header disagree"


def test_status_strings(lib):
    assert lib.mbx_status_string(0) == b"ok"
    assert lib.mbx_status_string(-4) != lib.mbx_status_string(-1)
    assert lib.mbx_version() >= 100


def test_priors_cabi_bit_exact(lib, golden):
    from multibox_amd import priors as P
    g = golden.priors
    for name, kw in [("k5_restrict", {}), ("k5_clip", dict(restrict_to_image_bounds=False)),
                     ("k7_restrict", {}), ("k8_restrict", {}), ("k5_scales", dict(min_scale=0.2, max_scale=0.8))]:
        out = P.generate_priors_array(g[name + "_ars"], **kw)
        assert out.tobytes() == g[name].tobytes(), name
    lst = P.generate_priors([1, 2, 3, 1 / 2., 1 / 3.])
    assert isinstance(lst, list) and len(lst) == 646 and isinstance(lst[0][0], float)


def test_priors_pickle_roundtrip(lib, tmp_path):
    from multibox_amd import priors as P
    p = P.generate_priors([1, 2, 3, 1 / 2., 1 / 3.])
    f = str(tmp_path / "priors.pkl")
    P.save_priors(f, p)
    q = P.load_priors(f)
    assert q.dtype == np.float32 and q.shape == (646, 4)
    assert np.array_equal(q, np.array(p).astype(np.float32))


def test_priors_bad_args(lib):
    from multibox_amd import priors as P, _lib
    with pytest.raises(_lib.MbxError):
        P.generate_priors_array([1.0], grids=[0])


def test_head_grids():
    from multibox_amd.priors import head_grids
    assert head_grids(299) == ([8, 6, 4, 3, 2], 1)
    assert head_grids(512) == ([14, 12, 7, 6, 5], 49)


def test_conv_stats_rows_follow_the_tile(lib):
    """mbx_conv_stats_rows (host): the number of BN-statistics partial rows a forward launch writes is ceil(M / BM) of the
    tile the launch will use -- igemm3 tiles 1..14 and the persistent igemm5 tiles 33..37 (csrc/conv5.hip); unknown
    igemm5 indices are rejected."""
    import ctypes as C
    from multibox_amd import ops, _lib
    d = ops.ConvDesc()
    d.N, d.H_in, d.W_in, d.C_in, d.C_out, d.R, d.S, d.stride = 64, 17, 17, 128, 160, 1, 7, 1
    d.H_out, d.W_out = 17, 17
    M = 64 * 17 * 17
    l = _lib.lib()
    bm3 = {1: 128, 2: 128, 3: 64, 4: 128, 5: 64, 6: 256, 7: 128, 8: 256, 9: 128, 10: 128, 11: 64, 12: 128, 13: 256, 14: 128}
    for cfg, bm in bm3.items():
        d.tile_config = cfg
        assert l.mbx_conv_stats_rows(C.byref(d)) == -(-M // bm), cfg
    for cfg, bm in zip(ops.I5_TILE_CONFIGS, (128, 128, 192, 256, 256, 128, 128)):
        d.tile_config = cfg
        assert l.mbx_conv_stats_rows(C.byref(d)) == -(-M // bm), cfg
    d.tile_config = ops.I5_FLAG + 9
    assert l.mbx_conv_stats_rows(C.byref(d)) < 0


def test_relu_bits_descriptor_rules(lib):
    """mbx_conv_desc.relu_bits through mbx_conv_supported (every check of mbx_conv, no launch, no GPU): written by
    RESIDUAL + relu, read by a plain STORE in place of `skip`; ld_bits a multiple of 4 and >= 4 ceil(C_out / 32), the table
    4-byte aligned; never with statistics, a `skip` mask, the stride-2 data gradient, the direct or split-K launches; the
    persistent tiles take the bits form except 128x192 (no accumulate + mask instantiation at all)."""
    import ctypes as C
    from multibox_amd import ops, _lib
    l = _lib.lib()

    def desc(epilogue, **kw):
        d = ops.ConvDesc()
        d.x, d.x_img_stride, d.ldx = 0x10000, 17 * 17 * 384, 384
        d.N, d.H_in, d.W_in, d.C_in = 4, 17, 17, 384
        d.w, d.C_out, d.R, d.S = 0x20000, 1088, 1, 1
        d.stride, d.H_out, d.W_out = 1, 17, 17
        d.y, d.y_img_stride, d.ldy = 0x30000, 17 * 17 * 1088, 1088
        d.epilogue = epilogue
        d.relu_bits, d.ld_bits = 0x50000, 136
        for k_, v in kw.items():
            setattr(d, k_, v)
        return d
    skip = dict(skip=0x40000, skip_img_stride=17 * 17 * 1088, ld_skip=1088)
    ok = lambda d: l.mbx_conv_supported(C.byref(d))
    assert ok(desc(ops.EPI_RESIDUAL, relu=1, rscale=0.1, **skip)) == 0                     # write
    assert ok(desc(ops.EPI_STORE, accumulate=1)) == 0 and ok(desc(ops.EPI_STORE)) == 0   # read
    assert ok(desc(ops.EPI_RESIDUAL, relu=0, rscale=0.1, **skip)) == -1                    # no relu: nothing to record
    assert ok(desc(ops.EPI_STORE, **skip)) == -1                                           # two masks
    assert ok(desc(ops.EPI_AFFINE, relu=1)) == -1
    assert ok(desc(ops.EPI_STORE, ld_bits=135)) == -1 and ok(desc(ops.EPI_STORE, ld_bits=132)) == -1
    assert ok(desc(ops.EPI_STORE, ld_bits=140)) == 0
    assert ok(desc(ops.EPI_STORE, relu_bits=0x50002)) == -1
    assert ok(desc(ops.EPI_STORE, stats_partial=0x60000)) != 0
    # stride-2 data gradient (3x3, transposed): the bits are indexed by the raster pixel index
    d = desc(ops.EPI_STORE, R=3, S=3, stride=2, transposed=1, H_in=8, W_in=8, x_img_stride=8 * 8 * 384, pad_t=2, pad_l=2)
    assert ok(d) == -2
    d.relu_bits = None
    assert ok(d) == 0
    for cfg in ops.I5_TILE_CONFIGS:
        want = -2 if cfg == 38 else 0
        assert ok(desc(ops.EPI_STORE, accumulate=1, tile_config=cfg)) == want, cfg
        assert ok(desc(ops.EPI_RESIDUAL, relu=1, rscale=0.1, tile_config=cfg, **skip)) == 0, cfg
    for cfg in (ops.DIRECT3_TILE_CONFIG, ops.DIRECTW_TILE_CONFIG, ops.RESIDENT_TILE_CONFIG, ops.SPLITK_FLAG + 4):
        assert ok(desc(ops.EPI_STORE, tile_config=cfg)) != 0, cfg
    # the pixel-resident pointwise launch (99): writes the bits beside a residual output, reads them as the mask of a store
    # (with or without an accumulate source); nothing else -- no tensor mask, no plain store
    assert ok(desc(ops.EPI_RESIDUAL, relu=1, rscale=0.1, tile_config=ops.PWRES_TILE_CONFIG, **skip)) == 0
    assert ok(desc(ops.EPI_STORE, accumulate=1, tile_config=ops.PWRES_TILE_CONFIG)) == 0
    assert ok(desc(ops.EPI_STORE, tile_config=ops.PWRES_TILE_CONFIG)) == 0
    d = desc(ops.EPI_STORE, tile_config=ops.PWRES_TILE_CONFIG, **skip)
    d.relu_bits = None
    assert ok(d) == -2
    d = desc(ops.EPI_STORE, tile_config=ops.PWRES_TILE_CONFIG)
    d.relu_bits = None
    assert ok(d) == -2


def test_ctypes_structs_match_the_header(tmp_path):
    """The ctypes mirrors of the C-ABI structs (ops.ConvDesc, ops.WgradJob, _lib.BnBwdStats) against include/mbx.h as a C
    compiler lays it out: size and the offset of every field of mbx_conv_desc (gcc on a ten-line program; no GPU).  A field
    added to the header and not to the mirror (or the other way round) shifts everything behind it -- mbx_wgrad_job embeds
    the descriptor, so the library would read the job array with the wrong stride."""
    import ctypes as C
    import shutil
    import subprocess
    from multibox_amd import ops, _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f[0] for f in ops.ConvDesc._fields_]
    hdr = {"shift": "shift", "stats_partial": "stats_partial"}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "mbx.h"', 'int main(void) {',
           '  printf("sizeof_desc %zu\\n", sizeof(mbx_conv_desc));', '  printf("sizeof_job %zu\\n", sizeof(mbx_wgrad_job));',
           '  printf("sizeof_bw %zu\\n", sizeof(mbx_bn_bwd_stats));',
           '  printf("sizeof_ba %zu\\n", sizeof(mbx_bn_apply_desc));', '  printf("sizeof_fb %zu\\n", sizeof(mbx_bn_bwd_fused));',
           '  printf("barrier_bytes %d\\n", MBX_GRID_BARRIER_BYTES);', '  printf("bwd_slots %d\\n", MBX_BN_BWD_SLOTS);']
    for f in fields:
        src.append('  printf("%s %%zu\\n", offsetof(mbx_conv_desc, %s));' % (f, hdr.get(f, f)))
    # (round 6) the tables of the fused launches: every field of both
    ba_names = {"moving_mean": "moving_mean", "moving_var": "moving_var"}
    for f, _ in ops.BnApplyDesc._fields_:
        src.append('  printf("ba.%s %%zu\\n", offsetof(mbx_bn_apply_desc, %s));' % (f, ba_names.get(f, f)))
    for f, _ in ops.BnBwdFused._fields_:
        src.append('  printf("fb.%s %%zu\\n", offsetof(mbx_bn_bwd_fused, %s));' % (f, f))
    src += ['  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(c), "-o", exe], check=True)
    out = dict(line.split() for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof_desc"]) == C.sizeof(ops.ConvDesc)
    assert int(out["sizeof_job"]) == C.sizeof(ops.WgradJob)
    assert int(out["sizeof_bw"]) == C.sizeof(_lib.BnBwdStats)
    assert int(out["sizeof_ba"]) == C.sizeof(ops.BnApplyDesc) and int(out["sizeof_fb"]) == C.sizeof(ops.BnBwdFused)
    assert int(out["barrier_bytes"]) == ops.GRID_BARRIER_BYTES and int(out["bwd_slots"]) == ops.BN_BWD_SLOTS
    for f in fields:
        assert int(out[f]) == getattr(ops.ConvDesc, f).offset, f
    for f, _ in ops.BnApplyDesc._fields_:
        assert int(out["ba." + f]) == getattr(ops.BnApplyDesc, f).offset, f
    for f, _ in ops.BnBwdFused._fields_:
        assert int(out["fb." + f]) == getattr(ops.BnBwdFused, f).offset, f


def test_no_kernel_uses_scratch_memory(tmp_path):
    """Every kernel of libmbx must fit its registers: a launch that needs scratch (private segment) makes the queue set
    scratch up, which showed as a ~0.1 ms stall per training step while the 192x192 weight-gradient tile spilled five
    registers (round 2).  Reads the code-object metadata out of the built objects (no GPU needed)."""
    import re
    import shutil
    import subprocess
    import __graft_entry__ as g
    g.build()
    llvm = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("ROCm LLVM tools not found")
    from multibox_amd import build as B
    seen = 0
    for obj in sorted(os.listdir(B.OBJ)):
        if not obj.endswith(".o"):
            continue
        path = os.path.join(B.OBJ, obj)
        sections = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-S", path], capture_output=True, text=True).stdout
        if ".hip_fatbin" not in sections:
            continue                                           # host-only source (priors.cpp)
        fat, co = str(tmp_path / (obj + ".fatbin")), str(tmp_path / (obj + ".co"))
        shutil.copy(path, str(tmp_path / obj))
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, str(tmp_path / obj)], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co, "--unbundle"], check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        names = re.findall(r"\.name:\s+(\S+)", notes)
        sizes = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
        spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)]
        kernels = [n for n in names if n.startswith("_Z")]
        assert len(sizes) == len(spills) > 0
        bad = [(n, s, v) for n, s, v in zip(kernels, sizes, spills) if s or v]
        assert not bad, "%s: kernels with scratch / spills: %s" % (obj, bad[:5])
        seen += len(sizes)
    assert seen > 150                                          # the igemm instantiations alone are > 100
