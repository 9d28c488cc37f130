"""The per-step statistics pass computes only the SMALLER sub-cluster of every cluster that no point entered or left since its
cluster-level statistics were cached, and derives the other one as cache - computed (DPMM_OPT_STATS_DERIVE; the reference recomputes
all three statistics of every cluster from scratch each sweep, src/local_clusters_actions.jl:149-169, src/priors/niw.jl:42-51).

Bar: the rows a per-step pass hands over equal a from-scratch Float64 pass over the same labels (the oracle, and the library's own
full pass) to rtol 1e-12 -- through label changes by sweeps, relabel operations (split / merge / remove-empty / set_labels), changes of K,
a new upload, a second context state; N counts exact."""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_niw import make_problem, gpu_worker
import test_gpu_mult as tm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


def _check(wk, X, D, K, tag, niw=True):
    """step_stats (derive on) against the oracle's statistics of the labels it leaves behind, and against the library's full pass."""
    packed, bad = wk.step_stats(reset_epoch=1000 + len(tag))
    lab, sub = wk.get_labels()
    if niw:
        N, s, S = wk.unpack(packed, K)
        oN, os_, oS = orc.suffstats_niw(X, D, lab, sub, K)
        assert np.array_equal(N, oN), tag
        np.testing.assert_allclose(s, os_, rtol=1e-12, atol=1e-9, err_msg=tag)
        np.testing.assert_allclose(S, oS, rtol=1e-12, atol=1e-9, err_msg=tag)
    else:
        N, s = wk.unpack(packed, K)
        oN, os_ = orc.suffstats_mult(X, D, lab, sub, K)
        assert np.array_equal(N, oN) and np.array_equal(s, os_), tag          # integer-valued sums: exact either way
    full = wk.suffstats_packed(None)
    np.testing.assert_allclose(packed, full, rtol=1e-12, atol=1e-9, err_msg=tag)
    return lab, sub, bad


@pytest.mark.parametrize("D,n,K", [(64, 30000, 6), (16, 9000, 4), (130, 6000, 3)])
def test_derived_rows_follow_every_label_change(pkg, D, n, K):
    from dpmmsubclusters_jl_amd import binding
    P = make_problem(D, n, K, seed=300 + D, sep=1.5)
    rng = np.random.default_rng(D)
    wk = gpu_worker(pkg, P, seed=17)
    X = P["X"]
    lab0 = rng.integers(1, K + 1, n); sub0 = 1 + (rng.random(n) < 0.2).astype(np.int64)      # unbalanced sub-clusters
    wk.set_labels(lab0, sub0)
    _check(wk, X, D, K, "first pass (every cluster computed in full)")
    _check(wk, X, D, K, "second pass, nothing changed (every larger sub-cluster derived)")
    # sub-labels change, labels do not: still derived
    wk.set_labels(lab0, 1 + (rng.random(n) < 0.7).astype(np.int64))
    _check(wk, X, D, K, "sub-labels changed")
    # a real sweep moves labels
    wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
    wk.sweep(3)
    lab, sub, _ = _check(wk, X, D, K, "after a sweep")
    wk.sweep(4)
    _check(wk, X, D, K, "after a second sweep (few labels move)")
    # a handful of points swap clusters: counts per cluster unchanged, membership not
    lab, sub = wk.get_labels()
    i, j = np.flatnonzero(lab == 1)[:5], np.flatnonzero(lab == 2)[:5]
    lab2 = lab.copy(); lab2[i] = 2; lab2[j] = 1
    wk.set_labels(lab2, sub)
    _check(wk, X, D, K, "five points swapped between clusters 1 and 2")
    # relabel operations
    wk.set_num_clusters(K + 1)
    wk.split(np.array([2]), np.array([K + 1]), epoch=9)
    _check(wk, X, D, K + 1, "after a split")
    _check(wk, X, D, K + 1, "pass after the split pass")
    wk.merge(np.array([1]), np.array([3]))
    _check(wk, X, D, K + 1, "after a merge (cluster 3 empty)")
    lab, _ = wk.get_labels()
    cnt = np.bincount(lab, minlength=K + 2)[1:K + 2]
    wk.remove_empty(cnt)
    wk.set_num_clusters(K)
    _check(wk, X, D, K, "after remove_empty (labels renumbered, K shrank)")
    _check(wk, X, D, K, "pass after remove_empty")
    # the option off gives the same rows; back on re-caches
    wk.set_option(binding.OPT_STATS_DERIVE, 0)
    _check(wk, X, D, K, "option off")
    wk.set_option(binding.OPT_STATS_DERIVE, 1)
    _check(wk, X, D, K, "option on again")
    _check(wk, X, D, K, "and derived")
    # new points under the same labels: the cache must not survive
    X2 = (X + np.float32(0.5)).astype(np.float32)
    wk.upload_points(X2)
    _check(wk, X2, D, K, "after a new upload")
    wk.close()


def test_derived_rows_multinomial(pkg):
    D, n, K = 200, 8000, 5
    P = tm.make_problem(D, n, K, 40, seed=9)
    wk = tm.worker(pkg, P, seed=3)
    rng = np.random.default_rng(1)
    wk.set_labels(rng.integers(1, K + 1, n), 1 + (rng.random(n) < 0.15).astype(np.int64))
    _check(wk, P["X"], D, K, "first", niw=False)
    _check(wk, P["X"], D, K, "second (derived)", niw=False)
    wk.sweep(2)
    _check(wk, P["X"], D, K, "after a sweep", niw=False)
    _check(wk, P["X"], D, K, "again", niw=False)
    wk.close()


def test_bad_cluster_reset_and_derivation_together(pkg):
    """A cluster whose right sub-cluster is empty is flagged, its sub-labels are re-drawn and -- its labels being unchanged -- its larger
    half is derived from the cache in the same pass."""
    D, n, K = 32, 12000, 4
    P = make_problem(D, n, K, seed=77, sep=2.0)
    rng = np.random.default_rng(5)
    wk = gpu_worker(pkg, P, seed=23)
    lab = rng.integers(1, K + 1, n); sub = rng.integers(1, 3, n)
    wk.set_labels(lab, sub)
    _check(wk, P["X"], D, K, "prime")
    sub2 = sub.copy(); sub2[lab == 2] = 1
    wk.set_labels(lab, sub2)
    l2, s2, bad = _check(wk, P["X"], D, K, "cluster 2 one-sided")
    assert bad[1] == 1 and bad.sum() == 1 and set(np.unique(s2[l2 == 2])) == {1, 2}
    wk.close()


@pytest.mark.parametrize("D,N,Kt", [(16, 40000, 6), (64, 60000, 5)])
def test_engine_chain_with_and_without_derivation(pkg, D, N, Kt):
    """The whole sampler from ONE initial cluster (labels move, clusters split, merge and are removed all the time while K grows), with the
    derivation on and off: same K history, same labels, statistics equal to rounding.  D = 64 runs the device master on the derived rows."""
    import importlib
    from dpmmsubclusters_jl_amd import binding
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    X, y = host.gaussian_mixture_shard(N, D, Kt, 100.0, 4321, 0, N)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    out = []
    for derive in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=11)
        wk.upload_points(X)
        wk.set_option(binding.OPT_STATS_DERIVE, derive)
        s = host.DPMMSampler(wk, prior, 10.0, N, 11, burnout=5)
        s.init_first_clusters(1)
        kh = []
        for it in range(90):
            s.group_step(it >= 80, False)
            kh.append(s.K)
        lab, sub = wk.get_labels()
        out.append((kh, lab, sub, s.model.get("packed"), s.log_posterior()))
        wk.close()
    (k1, l1, s1, p1, lp1), (k0, l0, s0, p0, lp0) = out
    assert k1 == k0 and k1[-1] >= Kt - 1, (k1, k0)
    assert int((l1 != l0).sum()) <= 2 and int((s1 != s0).sum()) <= 4
    if np.array_equal(l1, l0) and np.array_equal(s1, s0):
        np.testing.assert_allclose(p1, p0, rtol=1e-11, atol=1e-8)
        assert abs(lp1 - lp0) <= 1e-9 * abs(lp0)


@pytest.mark.parametrize("D,n,K,order", [(32, 70001, 5, "random"), (64, 300017, 9, "sorted"), (16, 2500000, 6, "random"), (8, 40000, 200, "random"),
                                         (8, 50000, 256, "sorted"), (8, 50000, 300, "random"), (4, 700, 3, "random")])
def test_reset_counted_ahead_is_the_reset_launch(pkg, D, n, K, order):
    """Round 6 (DPMM_OPT_CHAIN_FUSION bit 8): the per-step pass without a reset launch -- the histogram counts the outcome of reset_bad_clusters!
    (local_clusters_actions.jl:501-516) ahead for the clusters that are one-sided in a tile, the scan derives the flags and picks those counts
    for the flagged clusters, the scatter applies the re-draw while it places -- against the pass with the launch (bit 8 off): the same flags,
    the same labels and sub-labels for EVERY point, the same visiting order (the next sweep's tiles) and bitwise the same rows.  Cases: several
    bad clusters at once (left-empty, right-empty), a cluster that is one-sided in some tiles only (not flagged: its speculative counts must
    not be used), empty clusters, points in storage order and shuffled, tiles of 512 and 2048 points with a ragged last tile, 2K = 512 bins
    (the limit of the folded form), 2K = 600 (beyond it: the launch stays) and a shard smaller than one tile."""
    from dpmmsubclusters_jl_amd import binding
    rng = np.random.default_rng(1000 + K + D)
    X = rng.normal(size=(n, D)).astype(np.float32)
    lab = rng.integers(1, K + 1, n)
    if order == "sorted":
        lab.sort()
    sub = rng.integers(1, 3, n)
    bad_l, bad_r, partial, empty = 1, min(3, K), 2, (K if K > 4 else None)
    sub[lab == bad_l] = 2                         # left sub-cluster empty
    sub[lab == bad_r] = 1                         # right sub-cluster empty
    idx = np.flatnonzero(lab == partial)
    sub[idx[: len(idx) // 2]] = 1                 # one-sided in the first half of its points (whole tiles of them when sorted), mixed in the rest
    if empty is not None:
        lab[lab == empty] = partial if partial != empty else 1
    out = {}
    for fold in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, n, first_index=12345, device=0, seed=77)
        wk.upload_points(X)
        wk.set_option(binding.OPT_CHAIN_FUSION, 0x7fffffff if fold else 0x7fffffff & ~(8 | 32))        # (bit 32: the folded form below 4e6 points too)
        wk.set_num_clusters(K)
        wk.set_labels(lab, sub)
        passes = []
        for ep in (5, 6):                         # the second pass: derived halves + whatever the first reset left one-sided by chance
            packed, bad = wk.step_stats(reset_epoch=ep)
            passes.append((packed.copy(), bad.copy(), wk.get_labels(), wk.debug_perm() if hasattr(wk, "debug_perm") else None))
        out[fold] = passes
        wk.close()
    for (p1, b1, (l1, s1), o1), (p0, b0, (l0, s0), o0) in zip(out[1], out[0]):
        assert np.array_equal(b1, b0)
        assert np.array_equal(l1, l0) and np.array_equal(s1, s0)
        assert np.array_equal(p1, p0, equal_nan=True)
        if o1 is not None:
            assert np.array_equal(o1, o0)
    b = out[1][0][1]
    assert b[bad_l - 1] == 1 and b[bad_r - 1] == 1 and b[partial - 1] == 0
    l1, s1 = out[1][0][2]
    assert np.array_equal(l1, lab)                                                    # the reset moves sub-labels only
    for k in (bad_l, bad_r):
        if (lab == k).sum() > 40:
            assert set(np.unique(s1[lab == k])) == {1, 2}
    keep = ~np.isin(lab, [bad_l, bad_r] + [k + 1 for k in np.flatnonzero(b)])
    assert np.array_equal(s1[keep], sub[keep])                                        # nobody else's sub-label moved
