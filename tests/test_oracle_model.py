"""Structural pins of the torch-CPU oracle model and of the engine's graph (no GPU needed).

model_tests.py:53-56 pins [1,646,4]/[1,646,1] for k=5 at 299x299; SURVEY 8(c) pins the
parameter counts and layer counts derived from model.py.
"""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def net_cpu():
    import __graft_entry__ as g
    g.build()
    from multibox_amd.engine import Net
    return Net(batch=1, input_size=299, k=5, mode="train", device="cpu")


def params_from(net):
    return {name: net.get_param(name).clone() for name in net.param_index}


def test_engine_graph_structure(net_cpu):
    n = net_cpu
    assert n.P == 646 and n.grid_sizes == [8, 6, 4, 3, 2, 1]
    # 244 backbone + 22 head convs of the reference; fused siblings make it 205 launches
    ref_convs = sum(len(op.members) for op in n.convs)
    assert ref_convs == 244 + 22
    assert len(n.convs) == 205
    weights = sum(int(np.prod(s)) for k, (b, o, s, _) in n.param_index.items() if k.endswith("/weights"))
    biases = sum(int(np.prod(s)) for k, (b, o, s, _) in n.param_index.items() if k.endswith("/biases"))
    betas = sum(int(np.prod(s)) for k, (b, o, s, _) in n.param_index.items() if k.endswith("/beta"))
    assert abs((weights + biases + betas) / 1e6 - 60.00) < 0.01          # trainable (w, b, beta)
    assert abs((weights + biases + 3 * betas) / 1e6 - 60.06) < 0.01     # + moving mean/variance
    feat = n.features
    assert (feat.H, feat.W, feat.C) == (8, 8, 1536)
    macs = sum(op.M * op.K * op.R * op.S * (3 if op.Cin == 8 and op.R == 3 and op.x.H == 299 else op.Cin) for op in n.convs)
    assert abs(macs / 1e9 - 13.32) < 0.01                                # 13.154 + 0.166 GMAC / image


def test_oracle_model_shapes_and_names(net_cpu):
    from oracle.torch_model import Model
    P = params_from(net_cpu)
    m = Model(P, k=5)
    with torch.no_grad():
        locs, logits = m.build(torch.rand(1, 299, 299, 3) * 2 - 1)
    assert locs.shape == (1, 646, 4) and logits.shape == (1, 646)        # model_tests.py:53-56
    assert m.endpoints["Mixed_5b"].shape == (1, 320, 35, 35)
    assert m.endpoints["Mixed_6a"].shape == (1, 1088, 17, 17)
    assert m.endpoints["Mixed_7a"].shape == (1, 2080, 8, 8)
    assert m.endpoints["Conv2d_7b_1x1"].shape == (1, 1536, 8, 8)
    assert torch.isfinite(locs).all() and torch.isfinite(logits).all()
    # every parameter the engine owns was consumed by name (KeyError otherwise) and BN stats were produced
    assert len(m.new_moving) == sum(1 for k in P if k.endswith("/BatchNorm/beta"))


def test_k7_is_904():
    from multibox_amd.engine import Net
    n = Net(batch=1, input_size=299, k=7, mode="infer", device="cpu")
    assert n.P == 904


def test_512_geometry_d4():
    """at 512x512 the heads see 14x14 features -> grids 14,12,7,6,5 and a 7x7 single-prior head."""
    from multibox_amd.engine import Net
    from multibox_amd.priors import head_grids
    n = Net(batch=1, input_size=512, k=7, mode="infer", device="cpu")
    grids, last = head_grids(512)
    assert n.grid_sizes == grids + [7] and last == 49
    assert n.P == 7 * sum(g * g for g in grids) + 49 == 3199
