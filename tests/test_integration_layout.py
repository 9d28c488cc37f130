"""The Julia bindings under julia/ (gpu_engine.jl: master + worker native; gpu_worker.jl: worker side only) cannot be executed here (no
Julia in the image), so they are checked mechanically.  (1) Array comprehensions: each `Float32[... for ...]` line of `set_params!` is parsed, its iteration order is emulated with
itertools following Julia's rules (a flattened generator `for a in A for b in B` nests left to right -- the rightmost `for`
runs fastest; a product `for a in A, b in B` fills column-major -- the LEFTMOST variable runs fastest), and the resulting
memory order is compared with the layouts include/dpmm_hip.h prescribes and the ctypes binding sends:
    mu [3K][D], inv_sigma [3K][D][D], logdet [3K], lr_weights [K][2], logp [3K][D].
(Round 1 shipped a stub whose mu / inv / logp comprehensions put the distribution index fastest.)
(2) Every symbol a file names in a `ccall` / `dlsym` is declared in the header it binds; (3) every ccall passes as many argument TYPES
and as many VALUES as the C prototype has parameters; (4) `WorkerTable` has the members of `struct dpmmh_worker` in header order and
`native_table` fills them with the entry points the header names next to each member."""
import itertools
import os
import re

import numpy as np

from __graft_entry__ import ROOT


JL_FILES = ("gpu_engine.jl", "gpu_worker.jl", "host_transport_mpi.jl")


def _jl(name):
    return open(os.path.join(ROOT, "julia", name)).read()


def _stub_lines():
    txt = _jl("gpu_worker.jl")
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


# ---------------------------------------------------------------------------------------------- symbols and prototypes
def _c_prototypes():
    """name -> number of parameters, from the four public headers."""
    protos = {}
    for h in ("dpmm_hip.h", "dpmm_hip_master.h", "dpmm_hip_debug.h", "dpmm_host.h"):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
        src = src[:src.index("typedef struct dpmmh_worker {")] + src[src.index("} dpmmh_worker;"):] if "typedef struct dpmmh_worker {" in src else src
        for m in re.finditer(r"\b(dpmmh?_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
            args = m.group(2).strip()
            # split on top-level commas (function-pointer parameters hold commas of their own)
            depth, n = 0, 1 if args and args != "void" else 0
            for ch in args:
                depth += ch == "("
                depth -= ch == ")"
                if ch == "," and depth == 0:
                    n += 1
            protos[m.group(1)] = n
    return protos


def _split_top(sx):
    out, depth, cur = [], 0, ""
    for ch in sx:
        depth += ch in "([{"
        depth -= ch in ")]}"
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _ccalls(txt):
    """[(symbol, n_types, n_values)] of every ccall( ... ) in a Julia source."""
    res = []
    for m in re.finditer(r"ccall\(", txt):
        i, depth = m.end(), 1
        while depth:
            depth += txt[i] in "([{"
            depth -= txt[i] in ")]}"
            i += 1
        parts = _split_top(txt[m.end():i - 1])
        sym = re.search(r":(dpmmh?_[a-z0-9_]+)", parts[0])
        if sym is None:                     # ccall(step, ...) with step = dlsym(libhost, :dpmmh_group_step) just above
            var = parts[0].strip()
            sym = re.search(rf"\b{re.escape(var)}\s*=\s*[^\n]*:(dpmmh?_[a-z0-9_]+)", txt)
        assert sym is not None, parts[0]
        types = parts[2].strip()
        assert types.startswith("(") and types.endswith(")"), types
        ntypes = len([t for t in _split_top(types[1:-1]) if t])
        res.append((sym.group(1), ntypes, len(parts) - 3))
    return res


def test_julia_files_name_only_declared_symbols_with_the_declared_arity():
    protos = _c_prototypes()
    seen = set()
    for f in JL_FILES:
        txt = _jl(f)
        for name in re.findall(r":(dpmmh?_[a-z0-9_]+)", txt):
            assert name in protos, (f, name)
            seen.add(name)
        for sym, ntypes, nvals in _ccalls(txt):
            assert ntypes == protos[sym] and nvals == protos[sym], (f, sym, ntypes, nvals, protos[sym])
    # the worker-side file stays on the drop-in surface: nothing of the device master, nothing of the diagnostics
    hip_only = set(re.findall(r":(dpmm_[a-z0-9_]+)", _jl("gpu_worker.jl")))
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dpmm_hip.h")).read(), flags=re.S)
    assert hip_only <= set(re.findall(r"\b(dpmm_[a-z0-9_]+)\s*\(", src)), hip_only
    assert {"dpmm_create", "dpmm_sweep", "dpmm_step_stats", "dpmm_comm_init", "dpmmh_group_step", "dpmmh_model_bind_worker"} <= seen


def test_worker_table_mirrors_struct_dpmmh_worker():
    txt = _jl("gpu_engine.jl")
    body = txt[txt.index("struct WorkerTable"):]
    body = body[:body.index("\nend")]
    fields = re.findall(r"(\w+)::", body)
    hsrc = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dpmm_host.h")).read(), flags=re.S)
    struct = hsrc[hsrc.index("typedef struct dpmmh_worker {"):hsrc.index("} dpmmh_worker;")]
    members = re.findall(r"\(\*([a-z_]+)\)\s*\(", struct)
    assert fields == ["ctx", "rank", "world"] + members
    # native_table fills the members, in order, with the entry point the header comment names for each
    import importlib
    from __graft_entry__ import load_package
    load_package()
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    call = txt[txt.index("native_table(ctx, rank, world) = WorkerTable("):]
    call = call[:call.index("\n\n")]
    syms = re.findall(r"hip\(:(dpmm_[a-z0-9_]+)\)", call)
    assert syms == [sym for _, sym, _ in engine._NATIVE_MAP]


# ---------------------------------------------------------------------------------------------- fit / dp_parallel surface
# keyword arguments of the reference's fit (src/dp-parallel-sampling.jl:215-216) = positional tail of dp_parallel (:121-134), in order
REF_FIT_KEYWORDS = ["iters", "init_clusters", "seed", "verbose", "save_model", "burnout", "gt", "max_clusters", "outlier_weight",
                    "outlier_params", "smart_splits"]


def _jl_function(txt, name):
    """(signature text, body text) of `function name(...) ... end` (first method)."""
    i = txt.index(f"function {name}(")
    j, depth = i + len(f"function {name}("), 1
    while depth:
        depth += txt[j] == "("
        depth -= txt[j] == ")"
        j += 1
    end = txt.index("\nend\n", j)
    return txt[i:j], txt[j:end]


def test_julia_fit_consumes_every_argument_of_the_reference_signature():
    """gpu_fit / gpu_dp_parallel (julia/gpu_engine.jl) take the reference's arguments -- same names, same order, same defaults where the
    reference states one -- and USE each of them: a name that only appears in the signature (round 4: gt, verbose, save_model,
    max_clusters were accepted and dropped) fails here.  The histories run_model returns must be filled, not literal empty arrays."""
    txt = _jl("gpu_engine.jl")
    sig_fit, body_fit = _jl_function(txt, "gpu_fit")
    sig_dp, body_dp = _jl_function(txt, "gpu_dp_parallel")
    pos = [m.group(1) for m in re.finditer(r"(\w+)(?:::[\w{},. ]+)?\s*=", sig_dp.split(";")[0])]
    assert pos == REF_FIT_KEYWORDS, pos                                    # dp_parallel: positional, in the reference's order
    for kw in REF_FIT_KEYWORDS:
        assert re.search(rf"\b{kw}\b(?:::[\w{{}}]+)?\s*=", sig_fit), ("gpu_fit lacks keyword", kw)
        assert re.search(rf"\b{kw}\b", body_fit), ("gpu_fit drops", kw)
        assert len(re.findall(rf"\b{kw}\b", body_dp)) >= 1, ("gpu_dp_parallel drops", kw)
    for default in ("iters::Int64 = 100", "init_clusters::Int64 = 1", "seed = nothing", "verbose = true", "save_model = false", "gt = nothing",
                    "max_clusters = Inf", "outlier_weight = 0", "outlier_params = nothing", "smart_splits = false"):
        assert default in sig_fit and default in sig_dp, default
    assert "burnout = 20" in sig_fit and "burnout = 15" in sig_dp          # (the reference's two defaults: :216 and :129)
    # both methods dispatch on the abstract prior type, and both priors have a set_prior! method
    assert "local_hyper_params::distribution_hyper_params" in sig_fit and "local_hyper_params::distribution_hyper_params" in sig_dp
    assert "set_prior!(model, which, h::niw_hyperparams)" in txt and "set_prior!(model, which, h::multinomial_hyper)" in txt
    # run_model!: every flag acts, every history is pushed per iteration
    sig_rm, body_rm = _jl_function(txt, "run_model!")
    for name in ("verbose", "gt", "max_clusters", "save_model"):
        assert re.search(rf"\b{name}\b", sig_rm) and re.search(rf"\b{name}\b", body_rm), name
    for hist in ("iter_count", "nmi_score_history", "liklihood_history", "cluster_count_history"):
        assert f"push!({hist}," in body_rm, hist
    assert "K >= max_clusters" in body_rm and ":dpmm_contingency" in body_rm and ":dpmmh_log_posterior" in body_rm
    assert "Float64[], Float64[], Int[]" not in txt
