#!/usr/bin/env python3
"""Generate tests/golden/model_graph.json from the reference's model.py -- runs HERE only (reads /root/reference).

The reference network (model.py:6-337) is TF-0.11 graph code and cannot be imported (no tensorflow), but it is
plain `slim.conv2d / max_pool2d / avg_pool2d` calls with literal arguments, `tf.variable_scope` / `slim.arg_scope`
`with` blocks, `slim.repeat` and `tf.concat`.  This script walks its AST with a small symbolic interpreter (no
text of the reference is copied; nothing is exec'd) and emits the ordered layer table

    {scope, op, in_channels, out_channels, kernel, stride, padding, bn, bias, activation, inputs[, residual]}

for every convolution and pool, plus the head flatten/concat order (model.py:295-322).  The fixture pins BOTH
oracle/torch_model.py and multibox_amd.engine.Net to the reference's graph entry for entry
(tests/test_model_graph.py).  What the interpreter has to know that is NOT in /root/reference (slim/TF conventions,
un-vendored; stated here so they can be checked):
  * tf.variable_scope(None, default_name) and layers called without scope= get `default_name`, uniquified inside
    the enclosing scope as name, name_1, name_2, ... (slim.conv2d's default name is 'Conv', slim.repeat's 'Repeat');
  * slim.repeat(net, n, fn, **kw) calls fn n times with scope = fn.__name__ + '_' + str(i + 1);
  * slim.conv2d defaults: stride 1, padding 'SAME', activation_fn relu, normalizer_fn None, biases present unless a
    normalizer_fn is set or biases_initializer=None; arg_scope keyword defaults apply unless overridden at the call;
  * the outermost conv2d arg_scope (activation_fn, normalizer_fn=slim.batch_norm) is the one train.py:101-105 wraps
    around model.build -- read from train.py's AST below, not typed in.
Spatial sizes are inferred for a given input size with TF's SAME/VALID arithmetic.

Usage: python tools/gen_model_graph.py [--out tests/golden/model_graph.json] [--k 5] [--input-size 299]
"""
import argparse
import ast
import json
import os

REF = "/root/reference"


class Tensor:
    def __init__(self, producers, C, H, W):
        self.producers, self.C, self.H, self.W = list(producers), C, H, W


class Scope:
    def __init__(self):
        self.stack = []            # names
        self.used = [{}]           # per-level {base name: count}

    def unique(self, base):
        u = self.used[-1]
        n = u.get(base, 0)
        u[base] = n + 1
        return base if n == 0 else "%s_%d" % (base, n)

    def push(self, name):
        self.stack.append(name)
        self.used.append({})

    def pop(self):
        self.stack.pop()
        self.used.pop()

    def full(self, leaf):
        return "/".join([s for s in self.stack if s] + [leaf])


class Interp:
    def __init__(self, tree, k, input_size, conv_defaults):
        self.funcs = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
        self.k, self.S = k, input_size
        self.scope = Scope()
        self.arg_scopes = [{"conv2d": dict(conv_defaults)}]
        self.entries = []
        self.flatten = {}

    # ------------------------------------------------------------------ expression evaluation
    def ev(self, node, env):
        if isinstance(node, ast.Constant):
            return node.value
        if isinstance(node, ast.Name):
            if node.id in env:
                return env[node.id]
            if node.id in self.funcs:
                return ("func", node.id)
            raise KeyError(node.id)
        if isinstance(node, ast.List) or isinstance(node, ast.Tuple):
            return [self.ev(e, env) for e in node.elts]
        if isinstance(node, ast.Dict):
            return {}
        if isinstance(node, ast.Attribute):
            path = self.attr_path(node)
            if path == "tf.nn.relu":
                return "relu"
            if path == "slim.batch_norm":
                return "batch_norm"
            return ("attr", path)
        if isinstance(node, ast.BinOp):
            a, b = self.ev(node.left, env), self.ev(node.right, env)
            if isinstance(node.op, ast.Mult):
                if isinstance(a, Tensor) or isinstance(b, Tensor):     # scale * up
                    t, s = (a, b) if isinstance(a, Tensor) else (b, a)
                    return ("scaled", s, t)
                return a * b
            raise NotImplementedError(ast.dump(node))
        if isinstance(node, ast.Subscript):
            base = node.value
            idx = self.ev(node.slice, env)
            # net.get_shape()[3]  -> channels;  tf.shape(inputs)[0] -> batch
            if isinstance(base, ast.Call) and isinstance(base.func, ast.Attribute) and base.func.attr == "get_shape":
                t = self.ev(base.func.value, env)
                assert idx == 3
                return t.C
            v = self.ev(base, env)
            if isinstance(v, dict):
                return v[idx]
            if v == "batch":
                return "batch"
            raise NotImplementedError(ast.dump(node))
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
            return -self.ev(node.operand, env)
        if isinstance(node, ast.Call):
            return self.call(node, env)
        raise NotImplementedError(ast.dump(node))

    @staticmethod
    def attr_path(node):
        parts = []
        while isinstance(node, ast.Attribute):
            parts.append(node.attr)
            node = node.value
        if isinstance(node, ast.Name):
            parts.append(node.id)
        return ".".join(reversed(parts))

    # ------------------------------------------------------------------------------ layers
    def layer_kwargs(self, op, kw):
        out = {}
        for sc in self.arg_scopes:
            out.update(sc.get(op, {}))
        out.update(kw)
        return out

    @staticmethod
    def out_size(n, k, s, padding):
        return -(-n // s) if padding == "SAME" else (n - k) // s + 1

    def add_layer(self, op, x, kw, nout=None, ksize=None):
        a = self.layer_kwargs(op, kw)
        kh, kw_ = (ksize, ksize) if isinstance(ksize, int) else ksize
        stride, padding = a.get("stride", 1), a.get("padding", "SAME")
        default_name = {"conv2d": "Conv", "max_pool2d": "MaxPool2D", "avg_pool2d": "AvgPool2D"}[op]
        name = a.get("scope") or self.scope.unique(default_name)
        if a.get("scope"):
            self.scope.unique(name)
        e = {"scope": self.scope.full(name), "op": op, "in_channels": x.C, "kernel": [kh, kw_], "stride": stride,
             "padding": padding, "inputs": list(x.producers)}
        H, W = self.out_size(x.H, kh, stride, padding), self.out_size(x.W, kw_, stride, padding)
        if op == "conv2d":
            bn = a.get("normalizer_fn") is not None
            e.update(out_channels=nout, bn=bn,
                     bias=(not bn) and ("biases_initializer" not in a or a["biases_initializer"] is not None),
                     activation=a.get("activation_fn"))
            C = nout
        else:
            e.update(out_channels=x.C)
            C = x.C
        e["out_hw"] = [H, W]
        self.entries.append(e)
        return Tensor([e["scope"]], C, H, W)

    # -------------------------------------------------------------------------------- calls
    def call(self, node, env):
        path = self.attr_path(node.func) if isinstance(node.func, ast.Attribute) else getattr(node.func, "id", None)
        args = [self.ev(a, env) for a in node.args]
        kw = {k.arg: self.ev(k.value, env) for k in node.keywords}
        if path == "slim.conv2d":
            return self.add_layer("conv2d", args[0], kw, nout=args[1], ksize=args[2])
        if path in ("slim.max_pool2d", "slim.avg_pool2d"):
            return self.add_layer(path.split(".")[1], args[0], kw, ksize=args[1])
        if path == "tf.concat":
            dim, ts = args
            if dim == 3:
                assert all((t.H, t.W) == (ts[0].H, ts[0].W) for t in ts)
                return Tensor(sum((t.producers for t in ts), []), sum(t.C for t in ts), ts[0].H, ts[0].W)
            assert dim == 1
            return ("flat_concat", [t[1] for t in ts])
        if path == "tf.reshape":
            t, shape = args
            if isinstance(t, Tensor):                       # [batch, -1]: NHWC flatten of one head output
                assert shape == ["batch", -1]
                return ("flat", t.producers[0])
            if t[0] == "flat_concat":                       # [batch, -1, 4] / [batch, -1, 1]
                return ("pred", t[1], shape[2])
        if path == "tf.sigmoid":
            return ("pred", args[0][1], args[0][2], "sigmoid")
        if path == "tf.shape":
            return "batch"
        if path == "slim.repeat":
            net, n, fn = args[0], args[1], args[2]
            rep = self.scope.unique("Repeat")
            self.scope.push(rep)
            for i in range(n):
                net = self.run_function(fn[1], [net], dict(kw, scope="%s_%d" % (fn[1], i + 1)))
            self.scope.pop()
            return net
        if path == "slim.get_model_variables":
            return ("vars",)
        if path in self.funcs:
            return self.run_function(path, args, kw)
        raise NotImplementedError(path)

    def run_function(self, name, args, kw):
        fn = self.funcs[name]
        params = [a.arg for a in fn.args.args]
        defaults = fn.args.defaults
        env = {}
        for p, d in zip(params[len(params) - len(defaults):], defaults):
            env[p] = self.ev(d, {})
        for p, a in zip(params, args):
            env[p] = a
        env.update(kw)
        return self.run_body(fn.body, env)

    # --------------------------------------------------------------------------- statements
    def run_body(self, body, env):
        for st in body:
            r = self.stmt(st, env)
            if r is not None:
                return r[0]
        return None

    def stmt(self, st, env):
        if isinstance(st, ast.Expr):
            return None
        if isinstance(st, ast.Return):
            v = self.ev(st.value, env) if st.value is not None else None
            return (v,)
        if isinstance(st, ast.Assign):
            # comprehension over model variables (model.py:333): irrelevant to the graph
            if isinstance(st.value, ast.DictComp):
                env[st.targets[0].id] = ("vars",)
                return None
            v = self.ev(st.value, env)
            for t in st.targets:
                if isinstance(t, ast.Name):
                    env[t.id] = v
                elif isinstance(t, ast.Tuple):
                    vals = v if isinstance(v, (list, tuple)) else [v]
                    for el, vv in zip(t.elts, vals):
                        env[el.id] = vv
                elif isinstance(t, ast.Subscript):
                    self.ev(t.value, env)[self.ev(t.slice, env)] = v
            return None
        if isinstance(st, ast.AugAssign):
            assert isinstance(st.op, ast.Add)
            net = env[st.target.id]
            tag, scale, up = self.ev(st.value, env)
            assert tag == "scaled" and len(up.producers) == 1
            e = next(x for x in self.entries if x["scope"] == up.producers[0])
            e["residual"] = {"scale": scale, "skip": list(net.producers), "activation": None}
            env[st.target.id] = Tensor(up.producers, net.C, net.H, net.W)
            env["__residual__"] = e
            return None
        if isinstance(st, ast.If):
            if self.ev(st.test, env):
                # `net = activation_fn(net)` after a residual add (model.py:22-23)
                for s in st.body:
                    if (isinstance(s, ast.Assign) and isinstance(s.value, ast.Call) and isinstance(s.value.func, ast.Name)
                            and s.value.func.id == "activation_fn"):
                        env["__residual__"]["residual"]["activation"] = env["activation_fn"]
                    else:
                        r = self.stmt(s, env)
                        if r is not None:
                            return r
            return None
        if isinstance(st, ast.With):
            pushed_scope = pushed_args = 0
            for item in st.items:
                c = item.context_expr
                path = self.attr_path(c.func)
                if path == "tf.variable_scope":
                    a = [self.ev(x, env) for x in c.args[:2]]
                    name = a[0] if a[0] is not None else self.scope.unique(a[1])
                    if a[0] is not None:
                        self.scope.unique(name)
                    self.scope.push(name)
                    pushed_scope += 1
                elif path == "slim.arg_scope":
                    ops = [self.attr_path(x).split(".")[1] for x in c.args[0].elts]
                    kw = {k.arg: self.ev(k.value, env) for k in c.keywords}
                    self.arg_scopes.append({o: dict(kw) for o in ops})
                    pushed_args += 1
                else:
                    raise NotImplementedError(path)
            r = None
            for s in st.body:
                r = self.stmt(s, env)
                if r is not None:
                    break
            for _ in range(pushed_scope):
                self.scope.pop()
            for _ in range(pushed_args):
                self.arg_scopes.pop()
            return r
        raise NotImplementedError(ast.dump(st))


def conv_defaults_from_train():
    """activation_fn / normalizer_fn of the arg_scope train.py:101-105 wraps around model.build."""
    src = open(os.path.join(REF, "train.py")).read()
    # train.py is Python 2 (print statements): parse only the function we need
    start = src.index("def build_fully_trainable_model")
    end = src.index("def build_finetunable_model")
    fn = ast.parse(src[start:end]).body[0]
    it = Interp(ast.parse(""), 0, 0, {})
    for node in ast.walk(fn):
        if isinstance(node, ast.With):
            c = node.items[0].context_expr
            if it.attr_path(c.func) == "slim.arg_scope":
                kw = {k.arg: k.value for k in c.keywords}
                return {"activation_fn": it.ev(kw["activation_fn"], {}), "normalizer_fn": it.ev(kw["normalizer_fn"], {})}
    raise RuntimeError("arg_scope not found in train.py")


def generate(k, input_size):
    tree = ast.parse(open(os.path.join(REF, "model.py")).read())
    it = Interp(tree, k, input_size, conv_defaults_from_train())
    inputs = Tensor(["inputs"], 3, input_size, input_size)
    locs, confs, _ = it.run_function("build", [], {"inputs": inputs, "num_bboxes_per_cell": k})
    assert locs[0] == "pred" and confs[0] == "pred" and confs[3] == "sigmoid"
    convs = [e for e in it.entries if e["op"] == "conv2d"]
    return {
        "source": "gvanhorn38/multibox model.py:6-337 + train.py:101-105, AST-walked by tools/gen_model_graph.py",
        "k": k, "input_size": input_size,
        "n_conv": len(convs), "n_backbone_conv": sum(e["scope"].startswith("InceptionResnetV2/") for e in convs),
        "layers": it.entries,
        "locations": {"flatten": "NHWC", "last_dim": locs[2], "order": locs[1]},
        "confidences": {"flatten": "NHWC", "last_dim": confs[2], "order": confs[1], "activation": "sigmoid"},
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "tests", "golden", "model_graph.json"))
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--input-size", type=int, default=299)
    a = ap.parse_args()
    g = generate(a.k, a.input_size)
    with open(a.out, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote %s: %d layers (%d convs, %d backbone)" % (a.out, len(g["layers"]), g["n_conv"], g["n_backbone_conv"]))
