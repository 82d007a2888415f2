"""The reference's network graph (model.py:6-337), layer by layer, pins BOTH the torch oracle and the engine.

tests/golden/model_graph.json is generated from the reference's own source by tools/gen_model_graph.py (an AST walk
of model.py + the arg_scope of train.py:101-105; re-run it in a container that has /root/reference).  Every
convolution and pool of the reference must appear in oracle/torch_model.py's execution trace and in
multibox_amd.engine.Net.layer_table() with the same scope, channels, kernel, stride, padding, batch-norm / bias /
activation flags, residual scale and input tensors; the head flatten order must match model.py:295-322.
No GPU needed (the engine's graph is built on CPU; nothing is launched).
"""
import json
import os

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "model_graph.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def net_cpu():
    import __graft_entry__ as g
    g.build()
    from multibox_amd.engine import Net
    return Net(batch=1, input_size=299, k=5, mode="train", device="cpu")


def tf_pad(n, k, s, padding):
    """TF SAME / VALID: (output size, padding before)."""
    if padding == "VALID":
        return (n - k) // s + 1, 0
    out = -(-n // s)
    return out, max((out - 1) * s + k - n, 0) // 2


def test_golden_is_the_reference_network(golden):
    assert golden["n_conv"] == 266 and golden["n_backbone_conv"] == 244
    assert len([e for e in golden["layers"] if e["op"] != "conv2d"]) == 6
    assert golden["k"] == 5 and golden["input_size"] == 299


def test_oracle_model_matches_reference_graph(golden, net_cpu):
    from oracle.torch_model import Model
    P = {name: net_cpu.get_param(name).clone() for name in net_cpu.param_index}
    m = Model(P, k=5)
    with torch.no_grad():
        m.build(torch.rand(1, 299, 299, 3) * 2 - 1)
    # the oracle executes branch by branch, the reference lists layers in source order: compare by scope AND as a set
    got = {e["scope"]: e for e in m.trace}
    assert len(got) == len(m.trace) == len(golden["layers"])
    for ref in golden["layers"]:
        e = got[ref["scope"]]
        for key in ref:
            assert e[key] == ref[key], (ref["scope"], key, e[key], ref[key])
    # topological sanity of the oracle's own order: every input was produced earlier
    seen = {"inputs"}
    for e in m.trace:
        assert all(i in seen for i in e["inputs"]), e["scope"]
        seen.add(e["scope"])


def test_engine_matches_reference_graph(golden, net_cpu):
    table = net_cpu.layer_table()
    got = {e["scope"]: e for e in table}
    assert len(got) == len(table) == len(golden["layers"])
    hw = {"inputs": (299, 299)}
    for ref in golden["layers"]:
        e = got[ref["scope"]]
        hin = hw[ref["inputs"][0]]
        for key in ("op", "in_channels", "out_channels", "kernel", "stride", "inputs", "out_hw"):
            assert e[key] == ref[key], (ref["scope"], key, e[key], ref[key])
        # padding: the engine stores explicit top/left padding -- compare with TF's arithmetic for the golden's mode
        pads = [tf_pad(hin[0], ref["kernel"][0], ref["stride"], ref["padding"]),
                tf_pad(hin[1], ref["kernel"][1], ref["stride"], ref["padding"])]
        assert [pads[0][0], pads[1][0]] == ref["out_hw"] == e["out_hw"]
        assert e["pad"] == [pads[0][1], pads[1][1]], (ref["scope"], e["pad"], pads)
        if ref["op"] == "conv2d":
            for key in ("bn", "bias", "activation"):
                assert e[key] == ref[key], (ref["scope"], key, e[key], ref[key])
            assert e.get("residual") == ref.get("residual"), (ref["scope"], e.get("residual"), ref.get("residual"))
        hw[ref["scope"]] = tuple(ref["out_hw"])


def test_head_flatten_order(golden, net_cpu):
    """model.py:295-322: NHWC flatten per head, concatenated 8,6,4,3,2,1; prediction index off_g + cell*k + a."""
    n = net_cpu
    assert [op.members[0].scope for op in n.heads] == golden["locations"]["order"]
    assert [op.members[1].scope for op in n.heads] == golden["confidences"]["order"]
    off = 0
    by_scope = {e["scope"]: e for e in golden["layers"]}
    for op in n.heads:
        cells, kk, o = op.head
        ref = by_scope[op.members[0].scope]
        assert o == off and cells == ref["out_hw"][0] * ref["out_hw"][1] and 4 * kk == ref["out_channels"]
        assert by_scope[op.members[1].scope]["out_channels"] == kk
        off += cells * kk
    assert off == n.P == 646


def test_parameter_names_and_shapes(golden, net_cpu):
    """Every reference variable exists in the engine's flat buffers with the reference's shape."""
    idx = net_cpu.param_index
    n_w = 0
    for e in golden["layers"]:
        if e["op"] != "conv2d":
            continue
        buf, off, shape, _ = idx[e["scope"] + "/weights"]
        assert tuple(shape) == (e["out_channels"], e["kernel"][0], e["kernel"][1], e["in_channels"]), e["scope"]
        n_w += 1
        assert ((e["scope"] + "/biases") in idx) == e["bias"]
        assert ((e["scope"] + "/BatchNorm/beta") in idx) == e["bn"]
    assert n_w == 266
    assert sum(k.endswith("/weights") for k in idx) == 266
