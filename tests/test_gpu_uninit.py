"""Nothing on the path may depend on what earlier kernels left in LDS or in the vector registers.

tests/tools/poison.py fills the LDS and the register files of every CU with a bit pattern; the draws and whole chains must come out
bit-identical with and without it.  (Round 3: the posterior kernel's LDS factorisation read never-written words right of the diagonal
and multiplied them by zero -- with a NaN pattern there, L[15][15] of a block became NaN and a chain differed on a fresh GPU box.)"""
import importlib

import numpy as np
import pytest

from tools import poison

pytestmark = pytest.mark.gpu

PATTERNS = [0xffffffff, 0x7fc00000]     # NaN as Float32 and Float64 / NaN as Float32, finite as Float64


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    pkg = load_package()
    try:
        poison.build()
    except Exception as e:  # noqa: BLE001 -- the tool is test infrastructure: without hipcc on the box there is nothing to run
        pytest.skip(f"tests/tools/libpoison.so cannot be built here: {e}")
    return pkg


@pytest.fixture(scope="module")
def mods(pkg):
    return (importlib.import_module(pkg.__name__ + ".binding"), importlib.import_module(pkg.__name__ + ".host"),
            importlib.import_module(pkg.__name__ + ".host.engine"))


@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("D", [2, 5, 64, 200])
def test_posteriors_and_draws_ignore_lds_and_register_contents(pkg, mods, D, pattern):
    """Posterior scalars, the factor behind them and the draws (mu, R, log det) of populated, one-sided and empty distributions: clean,
    with the pattern refilled before every library call, and with it refilled before every kernel launch inside the library."""
    import contextlib
    import test_gpu_master as tm
    n, K = 3000, 4
    lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
    out = []
    for dirty in (False, True, "kernels"):
        with (poison.poisoned_kernel_launches(mods[0], pattern) if dirty == "kernels" else contextlib.nullcontext()):
            wk, X, lab, sub, prior = tm._setup(pkg, D, n, K, seed=900 + D)
            lab = lab.copy(); sub = np.ones_like(sub); lab[lab == K] = 1
            wk.set_labels(lab, sub)
            wk.master_setup(*prior)
            slots = np.arange(K, dtype=np.int32)
            if dirty: poison.poison(pattern)
            wk.suffstats_device(None)
            if dirty: poison.poison(pattern)
            scal = np.array(wk.master_posterior(None, slots)).copy()
            draws = []
            for epoch in (1, 2):
                if dirty: poison.poison(pattern)
                wk.master_draw(epoch, slots, lr, w)
                draws.append([np.array(a).copy() for a in wk.master_draws(K)])
            wk.close()
            out.append((scal, draws))
    s0, d0 = out[0]
    for s1, d1 in out[1:]:
        assert np.array_equal(s0, s1, equal_nan=True)
        for a, b in zip(d0, d1):
            for u, v in zip(a, b):
                assert np.all(np.isfinite(u)) and np.array_equal(u, v)


def _chain(pkg, mods, kind, iters):
    binding, host, engine = mods
    if kind == "mult200":
        N, D, K = 20000, 200, 5
        x = host.generate_mnmm_data(N, D, K, 80, seed=4242)[0]
        hyper = host.multinomial_hyper(np.ones(D, np.float32))
        dev = 1
    else:
        D = int(kind[3:]); N, K = 30000, 5
        x = host.generate_gaussian_data(N, D, K, 100.0, seed=4242)[0]
        hyper = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
        dev = 1 if D >= 64 else 0
    wk = pkg.Worker(hyper.kind, D, N, device=0, seed=11)
    wk.upload_points(np.ascontiguousarray(np.asarray(x, np.float32).T))
    s = host.DPMMSampler(wk, hyper, 10.0, N, 11, burnout=5)
    s.model.set_option(engine.OPT_DEVICE_MASTER, dev)
    s.init_first_clusters(1)
    _, _, _, kh = s.run_model(iters)
    lab, sub = wk.get_labels()
    lab, sub = np.array(lab).copy(), np.array(sub).copy()
    wk.close()
    return kh, lab, sub


@pytest.mark.parametrize("kind", ["niw8", "niw64", "niw128", "mult200"])
def test_chain_unchanged_with_poison_before_every_kernel_launch(pkg, mods, kind):
    """The same with the pattern refilled in front of EVERY kernel launch inside the library (its test hook): no kernel sees the leftovers
    of the library's own previous kernel either (Int32 -1 sentinels and masks left in LDS are NaNs when a later kernel reads them as
    Float64)."""
    binding, host, engine = mods
    iters = 16
    clean = _chain(pkg, mods, kind, iters)
    with poison.poisoned_kernel_launches(binding, 0xffffffff) as launches:
        dirty = _chain(pkg, mods, kind, iters)
    assert launches[0] > 8 * iters, launches
    assert clean[0] == dirty[0], (clean[0], dirty[0])
    assert np.array_equal(clean[1], dirty[1]) and np.array_equal(clean[2], dirty[2])


@pytest.mark.parametrize("kind", ["niw8", "niw64", "niw128", "mult200"])
def test_chain_unchanged_with_poison_before_every_worker_call(pkg, mods, kind):
    """Whole chains (splits included) with LDS + registers refilled with 0xffffffff before EVERY worker call of the engine: same K
    history, labels and sub-labels as the clean chain.  niw8: host master + small-D sweep; niw64: device master + direct sweep kernel;
    niw128: device master + LDS-staged sweep kernel; mult200: Multinomial device master + byte sweep."""
    binding, host, engine = mods
    iters = 25
    clean = _chain(pkg, mods, kind, iters)
    with poison.poisoned_worker_calls(binding, engine, 0xffffffff) as calls:
        dirty = _chain(pkg, mods, kind, iters)
    assert calls[0] > 3 * iters, calls
    assert clean[0] == dirty[0], (clean[0], dirty[0])
    assert np.array_equal(clean[1], dirty[1]) and np.array_equal(clean[2], dirty[2])
    assert max(clean[0]) > 1, clean[0]          # the chain did split: the master's kernels ran on real work
