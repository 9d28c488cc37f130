"""The Julia `ccall` stub in INTEGRATION.md cannot be executed here (no Julia in the image), so its array comprehensions are
checked mechanically: each `Float32[... for ...]` line of `set_params!` is parsed, its iteration order is emulated with
itertools following Julia's rules (a flattened generator `for a in A for b in B` nests left to right -- the rightmost `for`
runs fastest; a product `for a in A, b in B` fills column-major -- the LEFTMOST variable runs fastest), and the resulting
memory order is compared with the layouts include/dpmm_hip.h prescribes and the ctypes binding sends:
    mu [3K][D], inv_sigma [3K][D][D], logdet [3K], lr_weights [K][2], logp [3K][D].
(Round 1 shipped a stub whose mu / inv / logp comprehensions put the distribution index fastest.)"""
import itertools
import os
import re

import numpy as np

from __graft_entry__ import ROOT


def _stub_lines():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = {}
    for m in re.finditer(r"^\s*(\w+)\s*=\s*Float32\[(.+?)\]\s*(?:#.*)?$", txt, flags=re.M):
        out.setdefault(m.group(1), []).append(m.group(2))
    return out


def _loop_nest(comp):
    """-> (expression, [variable names, outermost first])."""
    parts = re.split(r"\s+for\s+", comp.strip())
    expr, clauses = parts[0], parts[1:]
    order = []
    for cl in clauses:
        # split a product clause on top-level commas only: `d in (p.cluster_dist, p.l_dist, p.r_dist)` holds commas too
        pieces, depth, cur = [], 0, ""
        for ch in cl:
            depth += ch == "("
            depth -= ch == ")"
            if ch == "," and depth == 0:
                pieces.append(cur); cur = ""
            else:
                cur += ch
        pieces.append(cur)
        vars_ = [re.match(r"\s*(\w+)\s+in\s", v).group(1) for v in pieces]
        order.extend(reversed(vars_))          # product: leftmost fastest => it is the INNERMOST loop
    return expr, order


def _emulate(order, sizes):
    """Flat list of index tuples (as dicts) in memory order."""
    ranges = [range(sizes[v]) for v in order]
    return [dict(zip(order, idx)) for idx in itertools.product(*ranges)]   # itertools.product: last varies fastest


K, D = 3, 4
SIZES = dict(p=K, d=3, i=D, j=D, s=2)


def _check(comp, want_shape, index_of):
    expr, order = _loop_nest(comp)
    flat = _emulate(order, SIZES)
    want = np.arange(int(np.prod(want_shape))).reshape(want_shape)
    got = np.array([want[index_of(ix)] for ix in flat])
    assert np.array_equal(got, np.arange(want.size)), (expr, order)


def test_stub_parameter_layouts_match_the_abi():
    lines = _stub_lines()
    assert {"mu", "inv", "ld", "lr", "logp"} <= set(lines), sorted(lines)
    for comp in lines["mu"]:
        _check(comp, (3 * K, D), lambda ix: (3 * ix["p"] + ix["d"], ix["i"]))
    for comp in lines["inv"]:
        # row-major [3K][D][D] with (i, j) -> [i][j]; the matrix is symmetric, so [j][i] is accepted as well
        try:
            _check(comp, (3 * K, D, D), lambda ix: (3 * ix["p"] + ix["d"], ix["i"], ix["j"]))
        except AssertionError:
            _check(comp, (3 * K, D, D), lambda ix: (3 * ix["p"] + ix["d"], ix["j"], ix["i"]))
    for comp in lines["ld"]:
        _check(comp, (3 * K,), lambda ix: (3 * ix["p"] + ix["d"],))
    for comp in lines["lr"]:
        _check(comp, (K, 2), lambda ix: (ix["p"], ix["s"]))
    for comp in lines["logp"]:
        _check(comp, (3 * K, D), lambda ix: (3 * ix["p"] + ix["d"], ix["i"]))


def test_the_checker_rejects_the_round1_stub():
    """The comprehension round 1 shipped (`for i in 1:D, p in params for d in dists(p)`) must fail this check."""
    bad = "d.μ[i] for i in 1:D, p in params for d in dists(p)"
    try:
        _check(bad, (3 * K, D), lambda ix: (3 * ix["p"] + ix["d"], ix["i"]))
    except AssertionError:
        return
    raise AssertionError("the layout checker accepted a known-bad comprehension")


def test_binding_sends_row_major_3k_by_d():
    """binding.Worker.set_params_* pass C-contiguous (3K, D) / (3K, D*D) arrays: row 3k+w, feature fastest."""
    src = open(os.path.join(ROOT, "dpmmsubclusters.jl_amd", "binding.py")).read()
    assert "mu.shape == (3 * K, self.D)" in src and "logp.shape == (3 * K, self.D)" in src and "np.ascontiguousarray" in src
