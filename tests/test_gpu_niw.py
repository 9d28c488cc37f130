"""GPU parity tests for the NIW worker path: HIP kernels (through the C ABI) vs the CPU oracle on
the same seeded inputs.  Tolerances (fp32 contraction on the matrix cores vs the oracle's f32/f64):
  per-point log-lik: atol 1e-3 + rtol 2e-5 against the Float64 evaluation of the same f32 parameters
  labels under shared uniforms: exact except counted near-boundary flips (< 1e-5 of points, at least 1; sub-labels <= 1e-4, at least 2;
      counts are printed; each label flip explained by a CDF margin below 1e-3 of the row mass)
  draw given the GPU's own table: bit-exact (same exp_det / scan arithmetic)
  N counts and all relabel bookkeeping: bit-exact; sum x, sum xx' vs Float64 oracle: rtol 1e-12
"""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


def make_problem(D, n, K, seed, sep=3.0, sorted_points=False):
    rng = np.random.default_rng(seed)
    mus = rng.normal(size=(3 * K, D)) * sep
    for k in range(K):
        d = rng.normal(size=D) * 0.4
        mus[3 * k + 1] = mus[3 * k] + d
        mus[3 * k + 2] = mus[3 * k] - d
    A = rng.normal(size=(3 * K, D, D)) * (0.3 / np.sqrt(D))
    Sig = A @ A.transpose(0, 2, 1) + np.eye(D) * (0.5 + rng.random((3 * K, 1, 1)))
    invS = np.linalg.inv(Sig)
    invS = 0.5 * (invS + invS.transpose(0, 2, 1))
    logdet = np.linalg.slogdet(Sig)[1]
    z = rng.integers(0, K, n)
    if sorted_points:
        z.sort()
    L = np.linalg.cholesky(Sig[3 * z])
    X = (mus[3 * z] + np.einsum("nij,nj->ni", L, rng.normal(size=(n, D)))).astype(np.float32)
    w = rng.dirichlet(np.ones(K) * 5).astype(np.float32)
    lr = rng.dirichlet(np.ones(2) * 5, size=K).astype(np.float32)
    return dict(D=D, n=n, K=K, X=X, mu=mus.astype(np.float32), invS=invS.reshape(3 * K, -1).astype(np.float32),
                logdet=logdet.astype(np.float32), w=w, lr=lr, z=z)


def gpu_worker(pkg, P, seed, first_index=0):
    wk = pkg.Worker(pkg.PRIOR_NIW, P["D"], P["n"], first_index=first_index, device=0, seed=seed)
    wk.upload_points(P["X"])
    wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
    return wk


def assert_sublabels_bit_exact(wk, lab, sub, u1):
    """The sub-label draw given the GPU's own Float32 left / right values must be bit-exact (same exp_det / scan arithmetic as
    the oracle): dpmm_debug_subloglik returns, for every point, b_l and b_r of every cluster; a point labelled k draws from
    rows 2(k-1), 2(k-1)+1 with its second uniform (create_subclusters_labels!, local_clusters_actions.jl:83-95)."""
    tab2 = wk.debug_subloglik()
    i = np.arange(len(lab))
    pair = np.stack([tab2[2 * (lab - 1), i], tab2[2 * (lab - 1) + 1, i]])
    assert np.array_equal(orc.sample_log_cat(pair, u1), sub)


def table_f64(P):
    K, n, D = P["K"], P["n"], P["D"]
    t = np.empty((K, n))
    for k in range(K):
        t[k] = orc.niw_loglik_f64(P["X"], D, P["mu"][3 * k], P["invS"][3 * k], P["logdet"][3 * k]) + np.log(np.float64(P["w"][k]))
    return t


@pytest.mark.parametrize("D,n,K", [(2, 1000, 5), (3, 4097, 3), (16, 3000, 4), (20, 2500, 6), (32, 3000, 4),
                                   (64, 6000, 8), (100, 1500, 3), (128, 2000, 4), (256, 1200, 3)])
def test_loglik_table(pkg, D, n, K):
    P = make_problem(D, n, K, seed=D * 7 + K)
    wk = gpu_worker(pkg, P, seed=1)
    got = wk.debug_loglik().astype(np.float64)
    # the GPU table omits the reference's constant -D^2/2 log(2 pi) term (mv_gaussian.jl:24)
    want = table_f64(P) + 0.5 * D * D * np.log(2 * np.pi)
    scale = np.abs(want).max(axis=0, keepdims=True)
    err = np.abs(got - want)
    print(f"D={D}: max |loglik err| = {err.max():.2e} (max |value| {np.abs(want).max():.1f})")
    assert np.all(err <= 1e-3 + 2e-5 * np.abs(want)), (err.max(), scale.max())
    # the differences that matter for a draw: relative to the row maximum
    rel = (got - got.max(0)) - (want - want.max(0))
    near = (want - want.max(0)) > -30
    assert np.abs(rel[near]).max() < 5e-3
    wk.close()


@pytest.mark.parametrize("D,n,K,sorted_points", [(2, 5000, 6, False), (64, 20000, 8, False), (64, 20000, 8, True),
                                                  (32, 9000, 5, False), (128, 4000, 4, False), (256, 2500, 3, True)])
def test_sweep_labels_vs_oracle(pkg, D, n, K, sorted_points):
    P = make_problem(D, n, K, seed=11 + D, sep=1.2, sorted_points=sorted_points)
    seed, epoch, first = 123456789, 5, 1000003
    wk = gpu_worker(pkg, P, seed=seed, first_index=first)
    wk.sweep(epoch)
    lab, sub = wk.get_labels()
    assert lab.min() >= 1 and lab.max() <= K and set(np.unique(sub)) <= {1, 2}
    # (1) draw given the GPU's own Float32 table must be bit-exact
    tab = wk.debug_loglik()
    u0, u1 = orc.uniforms(seed, epoch, 0, first, n)
    assert np.array_equal(orc.sample_log_cat(tab, u0), lab)
    assert_sublabels_bit_exact(wk, lab, sub, u1)
    # (2) against the oracle's independent Float32 evaluation: counted near-boundary flips only
    olab, osub, otab = orc.sweep_niw(P["X"], D, P["mu"], P["invS"], P["logdet"], np.log(P["w"]), np.log(P["lr"]),
                                     seed=seed, epoch=epoch, first_idx=first, want_parr=True)
    flips = np.flatnonzero(lab != olab)
    assert len(flips) <= max(1, int(1e-5 * n)), len(flips)
    t64 = table_f64(P)
    p = np.exp(t64 - t64.max(0))
    cdf = np.cumsum(p, 0) / p.sum(0)
    for i in flips:  # every flip must sit on a CDF edge
        assert np.min(np.abs(cdf[:, i] - u0[i])) < 1e-3, (i, u0[i], cdf[:, i])
    same = lab == olab
    sflips = int((sub[same] != osub[same]).sum())
    print(f"D={D} n={n}: label flips vs oracle {len(flips)}, sub-label flips {sflips}")
    assert sflips <= max(2, int(1e-4 * n)), sflips
    # the labels must be informative (not a degenerate draw)
    assert (lab == P["z"] + 1).mean() > 1.5 / K
    wk.close()


def test_loglik_reference_normaliser_switch(pkg):
    """DPMM_OPT_LOGLIK_REF_CONST adds back the reference's -D^2/2 log(2 pi) (mv_gaussian.jl:24 uses length(Sigma) = D^2): the table
    then equals the oracle's literal restatement of log_likelihood! (plus log w)."""
    from dpmmsubclusters_jl_amd import binding
    D, n, K = 6, 800, 3
    P = make_problem(D, n, K, seed=4)
    wk = gpu_worker(pkg, P, seed=1)
    plain = wk.debug_loglik()
    wk.set_option(binding.OPT_LOGLIK_REF_CONST, 1)
    ref = wk.debug_loglik()
    np.testing.assert_allclose(ref - plain, -0.5 * D * D * np.log(2 * np.pi), rtol=0, atol=2e-4)
    for k in range(K):
        want = orc.niw_loglik_ref(P["X"], D, P["mu"][3 * k], P["invS"][3 * k], P["logdet"][3 * k]) + np.log(P["w"][k])
        np.testing.assert_allclose(ref[k], want, rtol=2e-5, atol=1e-3)
    wk.close()


def test_final_argmax(pkg):
    P = make_problem(64, 7000, 7, seed=3, sep=0.8)
    wk = gpu_worker(pkg, P, seed=5)
    wk.sweep(9, final=True)
    lab, sub = wk.get_labels()
    tab = wk.debug_loglik()
    assert np.array_equal(lab, orc.argmax_rows(tab))
    assert np.array_equal(lab, tab.argmax(0) + 1)
    assert set(np.unique(sub)) == {1, 2}  # sub-labels are still sampled (local_clusters_actions.jl:83-95)
    wk.close()


def test_shard_invariance(pkg):
    """RNG counters use the global index: sharding the points over contexts changes nothing."""
    P = make_problem(64, 9000, 5, seed=21, sep=1.0)
    wk = gpu_worker(pkg, P, seed=77)
    wk.sweep(3)
    lab, sub = wk.get_labels()
    cut = 4321
    parts = []
    for lo, hi in ((0, cut), (cut, P["n"])):
        Q = dict(P); Q["X"] = np.ascontiguousarray(P["X"][lo:hi]); Q["n"] = hi - lo
        w2 = gpu_worker(pkg, Q, seed=77, first_index=lo)
        w2.sweep(3)
        parts.append(w2.get_labels())
        w2.close()
    assert np.array_equal(np.concatenate([p[0] for p in parts]), lab)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), sub)
    wk.close()


@pytest.mark.parametrize("D,n,K", [(2, 1000, 5), (3, 5000, 4), (16, 4000, 3), (40, 3000, 5), (64, 30000, 8),
                                   (128, 5000, 4), (200, 3000, 3), (256, 4000, 2), (2, 30000, 600)])     # K = 600: more than 1024 bins (one sort tile per workgroup)
def test_suffstats_vs_oracle(pkg, D, n, K):
    rng = np.random.default_rng(D + n)
    X = (rng.normal(size=(n, D)) * 2 + rng.normal(size=D) * 5).astype(np.float32)
    lab = rng.integers(1, K + 1, n).astype(np.int64)
    sub = rng.integers(1, 3, n).astype(np.int64)
    lab[lab == 2] = 1 if K > 2 else 2  # leave one cluster empty when K > 2
    wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1)
    wk.upload_points(X)
    wk.set_labels(lab, sub)
    g0, g1 = wk.get_labels()
    assert np.array_equal(g0, lab) and np.array_equal(g1, sub)
    wk.K = K
    # K is taken from the last parameter upload: give it dummy parameters
    wk.set_params_niw_chol(np.zeros((3 * K, D)), np.tile(np.eye(D).ravel(), (3 * K, 1)), np.zeros(3 * K),
                           np.full((K, 2), 0.5), np.full(K, 1.0 / K))
    N, s, S = wk.suffstats()
    oN, os_, oS = orc.suffstats_niw(X, D, lab, sub, K)
    assert np.array_equal(N, oN)  # integer counts: exact
    np.testing.assert_allclose(s, os_, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(S, oS, rtol=1e-12, atol=1e-9)
    # subset pass (update_suff_stats_posterior!(group, indices), local_clusters_actions.jl:668)
    idx = [K, 1]
    pk = wk.suffstats_packed(idx)
    N2, s2, S2 = wk.unpack(pk)
    for k in range(K):
        if k + 1 in idx:
            assert np.array_equal(N2[k], oN[k]); np.testing.assert_allclose(S2[k], oS[k], rtol=1e-12, atol=1e-9)
        else:
            assert not N2[k].any() and not S2[k].any()
    # run-to-run bitwise reproducibility
    assert np.array_equal(wk.suffstats_packed(), wk.suffstats_packed())
    # the per-step pass (histogram with running totals, bad-cluster flags, derived rows) returns the same rows when nothing is reset: a
    # cluster is flagged when one of its sub-clusters is empty, and a flagged cluster WITH points has its sub-labels re-drawn
    flagged_with_points = [(oN[k, 1] == 0) != (oN[k, 2] == 0) for k in range(K)]
    if not any(flagged_with_points):
        full = wk.suffstats_packed()
        rows, bad = wk.step_stats(7)
        assert np.array_equal(bad != 0, (oN[:, 1] == 0) | (oN[:, 2] == 0))
        np.testing.assert_allclose(rows, full, rtol=1e-12, atol=1e-9)
    wk.close()


def test_suffstats_golden_niw(pkg, golden_dir):
    """The reference's own checkpoint (examples/save_load_model/checkpoint__50.jld2) as known answer."""
    g = np.load(f"{golden_dir}/niw_golden.npz")
    X = np.ascontiguousarray(g["X"], np.float32)
    wk = pkg.Worker(pkg.PRIOR_NIW, 2, 1000, device=0, seed=1)
    wk.upload_points(X)
    wk.set_labels(g["labels"], g["sub"])
    K = 5
    wk.set_params_niw_chol(np.zeros((3 * K, 2)), np.tile(np.eye(2).ravel(), (3 * K, 1)), np.zeros(3 * K),
                           np.full((K, 2), 0.5), np.full(K, 0.2))
    N, s, S = wk.suffstats()
    i = 0
    for k in range(5):
        for w in range(3):
            assert N[k, w] == g["counts"][i]
            np.testing.assert_allclose(s[k, w], g["points_sum"][i], rtol=0, atol=5e-7 * 10 * max(N[k, w], 1))
            np.testing.assert_allclose(S[k, w], g["S"][i], rtol=2e-6, atol=1e-4)
            i += 1
    wk.close()


def test_relabel_ops_bit_exact(pkg):
    rng = np.random.default_rng(9)
    n, D = 50000, 2
    X = rng.normal(size=(n, D)).astype(np.float32)
    seed, first = 42, 777
    wk = pkg.Worker(pkg.PRIOR_NIW, D, n, first_index=first, device=0, seed=seed)
    wk.upload_points(X)
    wk.init_labels(5, epoch=1)
    lab, sub = wk.get_labels()
    olab, osub = orc.init_labels(n, 5, seed, 1, first)
    assert np.array_equal(lab, olab) and np.array_equal(sub, osub)
    # split (local_clusters_actions.jl:265-278)
    wk.split([2, 4], [6, 7], epoch=2)
    orc.split_relabel(olab, osub, [2, 4], [6, 7], seed, 2, first)
    lab, sub = wk.get_labels()
    assert np.array_equal(lab, olab) and np.array_equal(sub, osub)
    # merge (:293-304)
    wk.merge([1, 3], [5, 6])
    orc.merge_relabel(olab, osub, [1, 3], [5, 6])
    lab, sub = wk.get_labels()
    assert np.array_equal(lab, olab) and np.array_equal(sub, osub)
    # remove empty (:446-455)
    cnt = np.bincount(olab, minlength=8)[1:8]
    assert (cnt == 0).sum() == 2
    wk.remove_empty(cnt)
    orc.remove_empty(olab, cnt)
    lab, sub = wk.get_labels()
    assert np.array_equal(lab, olab) and np.array_equal(sub, osub) and lab.max() == 5
    # reset bad clusters (:481-488) and reset all (:257-261)
    wk.reset_sublabels([2, 5], epoch=3)
    orc.reset_sub(olab, osub, [2, 5], seed, 3, first)
    lab, sub = wk.get_labels()
    assert np.array_equal(sub, osub)
    wk.reset_sublabels(None, epoch=4)
    orc.reset_sub(olab, osub, None, seed, 4, first)
    assert np.array_equal(wk.get_labels()[1], osub)
    wk.close()


def test_error_paths(pkg):
    with pytest.raises(pkg.DpmmError):
        pkg.Worker(pkg.PRIOR_NIW, 300, 10, device=0)  # D beyond DPMM_MAX_DIM_NIW
    with pytest.raises(pkg.DpmmError):
        pkg.Worker(pkg.PRIOR_NIW, 4, 10, device=99)
    wk = pkg.Worker(pkg.PRIOR_NIW, 4, 10, device=0)
    with pytest.raises(pkg.DpmmError):
        wk.sweep(1)  # no points / params yet
    wk.close()
    # empty shard is legal (a worker can own zero columns)
    wk = pkg.Worker(pkg.PRIOR_NIW, 4, 0, device=0)
    wk.upload_points(np.zeros((0, 4), np.float32))
    wk.init_labels(1, 0)
    wk.set_params_niw_chol(np.zeros((3, 4)), np.tile(np.eye(4).ravel(), (3, 1)), np.zeros(3), np.full((1, 2), 0.5), np.ones(1))
    wk.sweep(1)
    N, s, S = wk.suffstats()
    assert not N.any()
    wk.close()


def test_contingency_table(pkg):
    """On-device evaluation: K x n_gt contingency table == numpy's, and NMI from it == sklearn's."""
    import importlib
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    from sklearn.metrics import normalized_mutual_info_score
    rng = np.random.default_rng(4)
    n, K, G = 123457, 9, 7
    lab = rng.integers(1, K + 1, n); sub = rng.integers(1, 3, n); gt = rng.integers(0, G, n)
    gt[lab < 4] = lab[lab < 4]  # some structure
    wk = pkg.Worker(pkg.PRIOR_NIW, 2, n, device=0)
    wk.upload_points(np.zeros((n, 2), np.float32))
    wk.set_labels(lab, sub)
    wk.set_ground_truth_range(gt, G)
    C = wk.contingency(K)
    want = np.zeros((K, G), np.int64); np.add.at(want, (lab - 1, gt), 1)
    assert np.array_equal(C, want)
    nmi, vi = host.sampler.nmi_vi_from_contingency(C)
    assert abs(nmi - normalized_mutual_info_score(gt, lab)) < 1e-10 and vi >= 0
    wk.close()


@pytest.mark.parametrize("D,n,K", [(64, 9000, 70), (64, 6000, 130), (48, 5000, 40), (24, 6000, 12)])
def test_many_clusters_and_padded_dims_with_screening(pkg, D, n, K):
    """K > 64 exercises the chunked far-mask, K > ~150 the global-scratch table; D = 48 / 24 the zero-padded blocks.
    Runs two consecutive sweeps so that the second one uses previous labels (reference clusters), the bin-sorted
    processing order and the screen; labels must equal the oracle draw from the GPU's own full table."""
    P = make_problem(D, n, K, seed=5 + K, sep=2.5, sorted_points=False)
    seed, first = 31337, 17
    wk = gpu_worker(pkg, P, seed=seed, first_index=first)
    wk.sweep(1)
    wk.suffstats_packed()                       # builds the permutation the next sweep walks
    wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
    wk.sweep(2)
    lab, sub = wk.get_labels()
    tab = wk.debug_loglik()
    u0, u1 = orc.uniforms(seed, 2, 0, first, n)
    assert np.array_equal(orc.sample_log_cat(tab, u0), lab)
    assert_sublabels_bit_exact(wk, lab, sub, u1)        # phase 2 of the screened / ordered sweep (rb0 carry-over, prefetch chain)
    olab, osub = orc.sweep_niw(P["X"], D, P["mu"], P["invS"], P["logdet"], np.log(P["w"]), np.log(P["lr"]), seed=seed, epoch=2, first_idx=first)
    same = lab == olab
    print(f"D={D} K={K} n={n}: label flips vs oracle {(lab != olab).sum()}, sub-label flips {(sub[same] != osub[same]).sum()}")
    assert (lab != olab).sum() <= max(1, int(1e-5 * n))
    assert (sub[same] != osub[same]).sum() <= max(2, int(1e-4 * n))
    wk.sweep(3, final=True)
    assert np.array_equal(wk.get_labels()[0], orc.argmax_rows(tab))
    wk.close()


def test_large_k_global_table_path(pkg):
    """K = 300 > LDS table budget: a_k table in the global scratch, screening bits over 10 words."""
    P = make_problem(32, 4000, 300, seed=77, sep=3.0)
    wk = gpu_worker(pkg, P, seed=3)
    wk.sweep(1); wk.suffstats_packed()
    wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
    wk.sweep(2)
    lab, sub = wk.get_labels()
    tab = wk.debug_loglik()
    u0, u1 = orc.uniforms(3, 2, 0, 0, 4000)
    assert np.array_equal(orc.sample_log_cat(tab, u0), lab)
    assert_sublabels_bit_exact(wk, lab, sub, u1)
    wk.close()


def test_maximum_cluster_count(pkg):
    """K = DPMM_MAX_CLUSTERS (1024): far-mask chunks of 64, 32 words of screening bits, 2048 statistic bins; one more is refused."""
    K = 1024
    P = make_problem(32, 5000, K, seed=5, sep=2.0)
    wk = gpu_worker(pkg, P, seed=4)
    wk.set_labels(P["z"] + 1, 1 + (np.arange(5000) & 1))
    wk.sweep(3)
    lab, sub = wk.get_labels()
    assert lab.min() >= 1 and lab.max() <= K
    tab = wk.debug_loglik()
    u0, _ = orc.uniforms(4, 3, 0, 0, 5000)
    assert np.array_equal(orc.sample_log_cat(tab, u0), lab)
    N, sums, S = wk.suffstats()
    assert N[:, 0].sum() == 5000 and np.array_equal(N[:, 0], np.bincount(lab, minlength=K + 1)[1:])
    cnt = wk.bin_counts()
    assert np.array_equal(cnt[:, 0], N[:, 1]) and np.array_equal(cnt[:, 1], N[:, 2])
    with pytest.raises(pkg.DpmmError):
        wk.set_num_clusters(K + 1)
    wk.close()


@pytest.mark.parametrize("D,n,K", [(64, 40000, 6), (128, 30000, 5), (256, 20000, 4)])
def test_tile_schedule_does_not_change_results(pkg, D, n, K):
    """The random stream is keyed by the point, so neither the number of workgroups (DPMM_OPT_SWEEP_GRID) nor the order in which
    the tile queues (D >= 128: every tile; D <= 64: the last DPMM_OPT_SWEEP_QUEUE_ROUNDS rounds, eight queue heads with stealing) hand
    tiles out may change a label or a sub-label; the second sweep runs with the ordered visiting order and the boundary-tile references
    active."""
    from dpmmsubclusters_jl_amd import binding
    P = make_problem(D, n, K, seed=31, sep=1.2, sorted_points=True)
    ref = None
    for grid, qrounds in ((0, -1), (7, -1), (64, -1), (0, 0), (7, 3), (16, 1000)):
        wk = gpu_worker(pkg, P, seed=11)
        if grid:
            wk.set_option(binding.OPT_SWEEP_GRID, grid)
        wk.set_option(binding.OPT_SWEEP_QUEUE_ROUNDS, qrounds)       # D <= 64: static schedule / a few rounds / everything from the queues
        wk.sweep(1)
        wk.suffstats_packed(None)          # builds the bin-sorted order the next sweep visits the points in
        wk.sweep(2)
        got = wk.get_labels()
        wk.close()
        if ref is None:
            ref = got
        else:
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_numa_node_query(pkg):
    P = make_problem(4, 100, 2, seed=1)
    wk = gpu_worker(pkg, P, seed=1)
    assert wk.numa_node() >= -1
    wk.close()


def test_work_counters_accumulate_until_read(pkg):
    """dpmm_last_sweep_work returns the totals of the sweeps since the previous call together with their number, and clears them: one
    read after a loop of launches gives the loop's per-launch average (what bench.py reports), a read after every launch that launch's."""
    P = make_problem(64, 20000, 6, seed=3, sep=1.5, sorted_points=True)
    wk = gpu_worker(pkg, P, seed=5)
    wk.sweep(1)
    one = wk.last_sweep_work()
    assert one["launches"] == 1 and one["wave_tiles"] == (20000 + 63) // 64 and one["full_evals"] >= 3 * one["wave_tiles"] - 1e-9
    assert one["executed_flops"] > 0
    for ep in (2, 3, 4):
        wk.sweep(ep)
    three = wk.last_sweep_work()
    assert three["launches"] == 3 and three["wave_tiles"] == one["wave_tiles"]
    none = wk.last_sweep_work()
    assert none["launches"] == 0 and none["wave_tiles"] == 0 and none["executed_flops"] == 0
    wk.close()


@pytest.mark.parametrize("D,sep,K", [(64, 40.0, 7), (64, 0.6, 7), (52, 40.0, 7), (60, 0.8, 7), (64, 6.0, 7), (64, 40.0, 100), (64, 1.0, 100),
                                     (128, 40.0, 6), (128, 0.8, 5), (100, 40.0, 6), (256, 40.0, 5), (256, 1.0, 4), (200, 3.0, 5),
                                     (64, 40.0, 2), (64, 0.5, 2), (32, 40.0, 2), (256, 40.0, 2), (16, 3.0, 2)])
def test_reference_bracket_does_not_change_labels(pkg, D, sep, K):
    """DPMM_OPT_REF_BRACKET (D in 33 .. 64; D in 65 .. 256 as a launch of its own per 128-point tile): on a wave whose points all carried the same label the reference cluster's value is first
    bracketed with two bf16 matrix passes and a certified rounding bound; its Float32 evaluation runs only if another cluster survives the
    screens against the bracket's lower end.  Labels and sub-labels must be those of the always-evaluate kernel, bit for bit -- on separated
    clusters (the evaluation is skipped on most waves: fewer full evaluations are counted) and on overlapping ones (survivors: the exact
    value is computed after all), with D below 64 (zero-padded features), and they equal the oracle's draw on the kernel's own table."""
    from dpmmsubclusters_jl_amd import binding
    n = 30000 if D <= 64 else 12000                    # (K = 100: beyond the LDS table's rows -- the generic kernel with the compact table)
    P = make_problem(D, n, K, seed=5 + D, sep=sep, sorted_points=True)
    out = {}
    for br in (1, 0):
        wk = gpu_worker(pkg, P, seed=21)
        wk.set_option(binding.OPT_REF_BRACKET, br)
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)                      # the bin-sorted visiting order: waves of one label
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.last_sweep_work()
        labs = []
        for ep in (1, 2):
            wk.sweep(ep)
            labs.append(wk.get_labels())
            wk.suffstats_packed(None)
            wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        work = wk.last_sweep_work()
        if br:
            tab = wk.debug_loglik()
            u0, u1 = orc.uniforms(21, 2, 0, 0, n)
            assert np.array_equal(orc.sample_log_cat(tab, u0), labs[1][0])
            assert_sublabels_bit_exact(wk, labs[1][0], labs[1][1], u1)
        out[br] = (labs, work)
        wk.close()
    for a, b in zip(out[1][0], out[0][0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    f1, f0 = out[1][1]["full_evals"], out[0][1]["full_evals"]
    print(f"D={D} sep={sep} K={K}: full evaluations per wave tile {f1 / out[1][1]['wave_tiles']:.2f} with the bracket, {f0 / out[0][1]['wave_tiles']:.2f} without")
    # (per tile: with the lean launch the tiles are aligned to the sort's bins -- a few more, partly filled ones -- and it only runs with the bracket)
    assert f1 / out[1][1]['wave_tiles'] <= f0 / out[0][1]['wave_tiles'] + 1e-9
    if sep >= 40.0 and D >= 33:                       # (D <= 32 has no bracket: the option changes nothing there)
        assert f1 < 0.8 * f0


@pytest.mark.parametrize("D", [128, 256, 100])
def test_big_bracket_follows_the_visiting_order(pkg, D):
    """D = 65 .. 256: niw_bracket_big_kernel writes one flag per 128-point tile and one threshold per POSITION of the visiting order, and the
    sweep reads them by position -- both launches must walk the same order.  (Round 4 launched the bracket before the sweep's order was
    set: thresholds in storage order, read in bin-sorted order.)  The case that exposes it: storage order shuffled, EVERY point labelled
    cluster 1 (one dominant cluster over two components, as early in a chain), component B several sd out in cluster 1's tail and tightly
    fitted by cluster 2, the sweep run behind a statistics pass (perm != identity).  A tail point that receives a central point's threshold
    excludes cluster 2 and is forced to k0 without an evaluation.  Labels must be bit-equal with the bracket on and off and equal to the
    oracle's draw on the kernel's own table (sample_labels_worker!, local_clusters_actions.jl:112-134)."""
    from dpmmsubclusters_jl_amd import binding
    rng = np.random.default_rng(900 + D)
    n, K = 16384 + 77, 2
    mus = np.zeros((3 * K, D)); Sig = np.empty((3 * K, D, D))
    shift = np.zeros(D); shift[: D // 2] = 6.0 / np.sqrt(D // 2) * 3.0      # component B: 6 sd (x3 on the broad cluster's scale) away from cluster 1's mean
    mus[3:6] = shift
    for j in range(3 * K):
        d = rng.normal(size=D) * 0.05
        mus[j] += d if j % 3 else 0.0
        Sig[j] = np.eye(D) * (9.0 if j < 3 else 0.04)                         # cluster 1 broad (sd 3), cluster 2 tight (sd 0.2)
    invS = np.linalg.inv(Sig); logdet = np.linalg.slogdet(Sig)[1]
    comp = (rng.random(n) < 0.35).astype(np.int64)                            # shuffled storage order: components interleaved
    X = np.where(comp[:, None] == 0, rng.normal(size=(n, D)) * 3.0, shift + rng.normal(size=(n, D)) * 0.2).astype(np.float32)
    w = np.array([0.6, 0.4], np.float32); lr = np.full((K, 2), 0.5, np.float32)
    args = (mus.astype(np.float32), invS.reshape(3 * K, -1).astype(np.float32), logdet.astype(np.float32), lr, w)
    out = {}
    for br in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, n, first_index=0, device=0, seed=33)
        wk.upload_points(X)
        wk.set_option(binding.OPT_REF_BRACKET, br)
        wk.set_params_niw(*args)
        wk.set_labels(np.ones(n, np.int64), 1 + (rng.permutation(n) & 1) if br else out["sub0"])
        if br: out["sub0"] = wk.get_labels()[1]
        wk.suffstats_packed(None)                                             # perm: sub-label 1 first, then sub-label 2 -- not the identity
        wk.set_params_niw(*args)
        wk.sweep(3)
        lab, sub = wk.get_labels()
        if br:
            tab = wk.debug_loglik()
            u0, u1 = orc.uniforms(33, 3, 0, 0, n)
            want = orc.sample_log_cat(tab, u0)
            bad = np.flatnonzero(want != lab)
            assert len(bad) == 0, (len(bad), comp[bad][:10], lab[bad][:10])
            assert_sublabels_bit_exact(wk, lab, sub, u1)
            assert (lab[comp == 1] == 2).mean() > 0.99                         # the tail component moves to the cluster that fits it
        out[br] = (lab, sub)
        wk.close()
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])


def _below_bf16_midpoint(rng, shape, lo_exp, hi_exp, worst=0.5):
    """Float32 values just BELOW a bf16 rounding midpoint: (1 + m / 128 + 1 / 256) 2^e stepped one Float32 ulp towards zero, so that
    round-to-nearest-even drops 2^-8 / (1 + m / 128) of the value -- the whole unit round-off for m = 0 (a fraction `worst` of the entries)."""
    m = rng.integers(0, 128, shape)
    m[rng.random(shape) < worst] = 0
    e = rng.integers(lo_exp, hi_exp, shape)
    v = ((1.0 + m / 128.0 + 1.0 / 256.0) * np.exp2(e)).astype(np.float32)
    return np.nextafter(v, np.float32(0))


def _bracket_problem(kind, D, n, seed):
    """Adversarial operands for the reference bracket: cluster 1 has mu = 0 (z = x exactly) and the upper-triangular factor R[0]."""
    rng = np.random.default_rng(seed)
    K = 4
    R = np.zeros((3 * K, D, D), np.float32)
    mu = (rng.normal(size=(3 * K, D)) * 30).astype(np.float32)
    mu[0:3] = 0
    mu[1, :] = 0.25; mu[2, :] = -0.25
    if kind == "midpoints":
        # every entry of R and of x just below a bf16 midpoint, all positive: every product loses (1 + u)^2 - 1, nothing cancels
        for j in range(3 * K):
            R[j] = np.triu(_below_bf16_midpoint(rng, (D, D), -6, -3))
        X = _below_bf16_midpoint(rng, (n, D), -2, 2)
    elif kind == "trailing":
        # random factor; points displaced along ONE trailing feature: the last rows of R z have one or two terms, |y^| = e^ there
        for j in range(3 * K):
            A = rng.normal(size=(D, D)) * 0.2 + np.eye(D) * (1 + rng.random(D))
            R[j] = np.triu(A).astype(np.float32)
        X = (rng.normal(size=(n, D)) * 0.05).astype(np.float32)
        feat = D - 1 - rng.integers(0, 3, n)
        X[np.arange(n), feat] = np.sign(rng.normal(size=n)).astype(np.float32) * _below_bf16_midpoint(rng, n, 0, 5)
        R[0][np.arange(D - 3, D), np.arange(D - 3, D)] = _below_bf16_midpoint(rng, 3, -1, 2)
    else:  # "outlier": ordinary points plus far ones, q up to ~1e5 and beyond
        for j in range(3 * K):
            A = rng.normal(size=(D, D)) * 0.2 + np.eye(D) * (1 + rng.random(D))
            R[j] = np.triu(A).astype(np.float32)
        X = rng.normal(size=(n, D)).astype(np.float32)
        far = rng.random(n) < 0.05
        X[far] *= np.float32(40.0)
        X[::97] *= np.float32(1e4)
    logdet = np.array([-2.0 * np.log(np.abs(np.diag(R[j, :D, :D]).astype(np.float64))).sum() for j in range(3 * K)], np.float32)
    w = np.full(K, 1.0 / K, np.float32); lr = np.full((K, 2), 0.5, np.float32)
    return dict(D=D, n=n, K=K, X=X, mu=mu, R=R, logdet=logdet, w=w, lr=lr)


@pytest.mark.parametrize("kind,D", [("midpoints", 64), ("midpoints", 52), ("trailing", 64), ("trailing", 60), ("outlier", 64)])
def test_reference_bracket_is_an_upper_bound(pkg, kind, D):
    """The exactness argument of the default D <= 64 sweep (sample_labels_worker!, local_clusters_actions.jl:112-134): the bf16 bracket's upper
    end q_hi must not fall below the Float32 quadratic form q it stands in for -- for EVERY point, on operands chosen against the bound
    (both bf16 roundings at their full unit round-off 2^-8 and of the same sign; rows of R z with a single term; far outliers).
    dpmm_debug_ref_bracket runs the sweep's own device functions on the sweep's own operand images.  Round 3's constant (0.00395: one
    rounding of half the size) FAILS this test on the midpoint operands; the library's (2u + u^2 + accumulation = 0.00785) passes."""
    n = 6000
    P = _bracket_problem(kind, D, n, seed=11 + D)
    wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=3)
    wk.upload_points(P["X"])
    wk.set_params_niw_chol(P["mu"], P["R"], P["logdet"], P["lr"], P["w"])
    qhi, q = wk.debug_ref_bracket(1)
    z = P["X"].astype(np.float64) - P["mu"][0].astype(np.float64)
    q64 = ((z @ P["R"][0].astype(np.float64).T) ** 2).sum(axis=1)
    fin = np.isfinite(q)
    assert fin.sum() >= 0.9 * n
    np.testing.assert_allclose(q[fin], q64[fin], rtol=3e-5)                       # the Float32 evaluation itself
    bad = np.isfinite(qhi) & fin & ((qhi < q) | (qhi < q64))
    worst = float(np.min((qhi[fin] - q[fin]) / q[fin]))
    print(f"{kind} D={D}: min (q_hi - q) / q = {worst:.3e}, median {float(np.median((qhi[fin] - q[fin]) / q[fin])):.3e}, "
          f"q range {q[fin].min():.3g} .. {q[fin].max():.3g}")
    assert not bad.any(), f"{int(bad.sum())} points with q_hi < q (worst relative deficit {worst:.3e})"
    assert not np.isfinite(qhi[~fin]).any()                                       # overflow: not finite either -> the sweep takes the Float32 path
    assert float(np.median((qhi[fin] - q[fin]) / q[fin])) < 0.5                   # ... and still a useful bound (random signs: e^ is several |y^|)
    if kind == "midpoints":
        # the same operands against round 3's constant: the bound is violated (this is what "certified" missed)
        qhi_old, q_old = wk.debug_ref_bracket(1, c_override=0.00395)
        assert np.array_equal(q_old, q)
        assert (qhi_old < q).sum() > 0.5 * n
    wk.close()


@pytest.mark.parametrize("kind", ["midpoints", "trailing", "outlier"])
def test_reference_bracket_labels_on_adversarial_operands(pkg, kind):
    """Same operands through the sweep itself: labels and sub-labels with the bracket on equal those with it off, bit for bit."""
    from dpmmsubclusters_jl_amd import binding
    D, n = 64, 8000
    P = _bracket_problem(kind, D, n, seed=29)
    rng = np.random.default_rng(1)
    # half of the points sit around the other clusters' means so that every cluster has label-homogeneous waves
    own = rng.integers(0, P["K"], n); own[: n // 2] = 0
    X = P["X"].copy()
    X[own > 0] = (P["mu"][3 * own[own > 0]] + rng.normal(size=(int((own > 0).sum()), D)) * 0.3).astype(np.float32)
    order = np.argsort(own, kind="stable"); X = X[order]; own = own[order]
    out = {}
    for br in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=17)
        wk.upload_points(X)
        wk.set_option(binding.OPT_REF_BRACKET, br)
        wk.set_params_niw_chol(P["mu"], P["R"], P["logdet"], P["lr"], P["w"])
        wk.set_labels(own + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)                      # the bin-sorted visiting order: waves of one label
        wk.set_params_niw_chol(P["mu"], P["R"], P["logdet"], P["lr"], P["w"])
        wk.last_sweep_work()
        labs = []
        for ep in (1, 2):
            wk.sweep(ep)
            labs.append(wk.get_labels())
            wk.suffstats_packed(None)
            wk.set_params_niw_chol(P["mu"], P["R"], P["logdet"], P["lr"], P["w"])
        out[br] = (labs, wk.last_sweep_work())
        wk.close()
    for a, b in zip(out[1][0], out[0][0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    print(f"{kind}: full evaluations per wave tile {out[1][1]['full_evals'] / out[1][1]['wave_tiles']:.2f} with the bracket, "
          f"{out[0][1]['full_evals'] / out[0][1]['wave_tiles']:.2f} without")


@pytest.mark.parametrize("D,sep,K", [(64, 6.0, 7), (64, 2.0, 12), (64, 0.8, 7), (52, 2.0, 7), (36, 3.0, 9), (64, 1.5, 100), (64, 40.0, 7)])
def test_bf16_screens_do_not_change_labels(pkg, D, sep, K):
    """DPMM_OPT_BF16_SCREENS (D in 33 .. 64): a certified bf16 LOWER bound of the last / first block row's part of the quadratic form in front of
    the Float32 16-row screen / of a survivor's evaluation.  They only skip Float32 tests that would have excluded the cluster as well, so the
    set of evaluated clusters, the table and the labels are those of the kernel without them -- bit for bit, on overlapping clusters (where
    they do most of the screening), on separated ones, with zero-padded features, and beyond the LDS table's rows."""
    from dpmmsubclusters_jl_amd import binding
    n = 30000
    P = make_problem(D, n, K, seed=40 + D + K, sep=sep, sorted_points=True)
    out = {}
    for on in (1, 0):
        wk = gpu_worker(pkg, P, seed=23)
        wk.set_option(binding.OPT_BF16_SCREENS, on)
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)                      # the bin-sorted visiting order
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.last_sweep_work()
        labs = []
        for ep in (1, 2):
            wk.sweep(ep)
            labs.append(wk.get_labels())
            wk.suffstats_packed(None)
            wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        work = wk.last_sweep_work()
        if on:
            tab = wk.debug_loglik()
            u0, u1 = orc.uniforms(23, 2, 0, 0, n)
            assert np.array_equal(orc.sample_log_cat(tab, u0), labs[1][0])
        out[on] = (labs, work)
        wk.close()
    for a, b in zip(out[1][0], out[0][0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    w1, w0 = out[1][1], out[0][1]
    t = w1["wave_tiles"]
    print(f"D={D} sep={sep} K={K}: per wave tile -- Float32 screens {w0['screens16'] / t:.2f} -> {w1['screens16'] / t:.2f}, bf16 bottom {w1['bf16_bottom_screens'] / t:.2f}, "
          f"bf16 top {w1['bf16_top_screens'] / t:.2f}, full evaluations {w0['full_evals'] / t:.2f} -> {w1['full_evals'] / t:.2f}")
    # never MORE Float32 work: a cluster the bf16 screens exclude is one the Float32 tests would have excluded; fewer full evaluations where the
    # top screen removes the last survivor of a bracketed wave (the reference cluster's own Float32 evaluation is then skipped as well)
    assert w1["full_evals"] <= w0["full_evals"]
    assert w1["screens16"] <= w0["screens16"] and w0["bf16_bottom_screens"] == 0 and w0["bf16_top_screens"] == 0


@pytest.mark.parametrize("D,sep,K,n", [(64, 40.0, 7, 30000), (64, 40.0, 5, 200037), (60, 30.0, 9, 50001), (64, 2.0, 12, 30011), (64, 0.8, 7, 20000), (52, 2.0, 7, 30000),
                                       (36, 3.0, 9, 9999), (64, 1.5, 100, 30000), (64, 40.0, 300, 120000), (64, 6.0, 1, 5000), (64, 2.0, 12, 200000 + 33)])      # (the last: more handed-on spans than the list launch has waves)
def test_lean_tiles_do_not_change_labels(pkg, D, sep, K, n):
    """D in 33 .. 64: the sweep runs as niw_lean_kernel (finishes the tiles whose label candidates its screens settle -- tiles aligned to the bins
    of the sort --, hands the others on as a list of spans) + niw_sweep_direct_kernel<LSTORE, LIST> (labels and sub-labels of the listed spans);
    with DPMM_OPT_LEAN_TILES = 0 niw_sweep_direct_kernel<LSTORE> (labels) + niw_sub_kernel (sub-labels) do every tile.  Which launch finishes a tile must not show: labels AND sub-labels bit-equal over a
    chain of sweeps (every sub-cluster value is the bf16 three-plane one in both), on separated clusters (nearly every tile settled in the lean
    launch), overlapping ones (most handed on; the regime switch turns the lean launch off), padded D, a ragged last tile and K = 1.  With the
    lean launch on, the labels are the oracle's draw on the kernel's own table and the sub-labels its draw on the kernel's own sub-cluster
    values (sample_labels_worker! / create_subclusters_labels!, local_clusters_actions.jl:83-134)."""
    from dpmmsubclusters_jl_amd import binding
    P = make_problem(D, n, K, seed=140 + D + K, sep=sep, sorted_points=True)
    out = {}
    for on in (1, 0):
        wk = gpu_worker(pkg, P, seed=29)
        wk.set_option(binding.OPT_LEAN_TILES, on)
        wk.set_timing(15)
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)                      # the bin-sorted visiting order
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.last_sweep_work()
        labs, lean_ms = [], []
        for ep in (1, 2, 3, 4):
            wk.sweep(ep)
            labs.append(wk.get_labels())
            lean_ms.append(wk.last_sweep_parts_ms()[0])
            if ep == 4 and on:
                u0, u1 = orc.uniforms(29, 4, 0, 0, n)
                assert np.array_equal(orc.sample_log_cat(wk.debug_loglik(), u0), labs[-1][0])
                assert_sublabels_bit_exact(wk, labs[-1][0], labs[-1][1], u1)
            wk.suffstats_packed(None)
            wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        work = wk.last_sweep_work()
        assert work["b3_evals"] > 0 or K == 1          # the sub-cluster values came from the three-plane evaluation (K = 1 has no tail records: the Float32 chain)
        if not on or K == 1 or K > 64:                 # (set_params path beyond 64 clusters: the scalar pre-screen runs in the sweep kernel, no lean launch;
                                                       #  the device-master path has no pre-screen and does run it: test_lean_tiles_beyond_64_clusters_*)
            assert max(lean_ms) == 0.0
        else:
            assert max(lean_ms) > 0.0                  # (overlapping clusters: one lean launch, then the regime switch keeps it off)
        out[on] = labs
        wk.close()
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("D,sep,K,n,prescreen", [(64, 40.0, 7, 30000, 1), (64, 40.0, 32, 60000, 1), (60, 30.0, 9, 50001, 1), (64, 2.0, 12, 30011, 1), (52, 6.0, 6, 20000, 1),
                                                 (64, 30.0, 100, 40000, 0), (64, 25.0, 200, 60000, 0), (64, 25.0, 256, 70000, 0), (64, 4.0, 2, 9000, 1)])
def test_pair_ball_table_and_labels(pkg, D, sep, K, n, prescreen):
    """Round 6: the lean kernel's ball test in all D features on a K x K table tabulated per parameter set (DPMM_OPT_PAIR_BALL; niw_pair_ball_kernel):
    pd[k, j] must be a LOWER bound of |R_j (mu_k - mu_j)| = sqrt((mu_k - mu_j)' Sigma_j^-1 (mu_k - mu_j)) and sn[j] an UPPER bound of
    |R_j|_2 = sqrt(lambda_max(Sigma_j^-1)) -- both checked against Float64 values of the Float32 parameters, both tight enough to be of use --
    and a chain of sweeps with the test on must give the labels AND sub-labels of the chain without it, on separated clusters (where it clears
    every candidate: no tail-pair test left), overlapping ones, padded D, K = 2 and beyond 64 clusters (the path without the pre-screen)."""
    from dpmmsubclusters_jl_amd import binding
    P = make_problem(D, n, K, seed=900 + D + K, sep=sep, sorted_points=True)
    out, work = {}, {}
    for on in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, n, first_index=0, device=0, seed=37)
        wk.upload_points(P["X"])
        wk.set_option(binding.OPT_PRESCREEN, prescreen)
        wk.set_option(binding.OPT_PAIR_BALL, on)
        wk.set_timing(15)
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)                      # the bin-sorted visiting order
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        if on:
            pd, sn = wk.debug_pair_ball()
            mu = P["mu"][0::3].astype(np.float64)
            iS = P["invS"].reshape(3 * K, D, D)[0::3].astype(np.float64)
            dm = mu[:, None, :] - mu[None, :, :]                                  # [k, j] = mu_k - mu_j
            true_d = np.sqrt(np.maximum(np.einsum("kja,jab,kjb->kj", dm, iS, dm), 0.0))
            true_s = np.sqrt(np.linalg.eigvalsh(iS)[:, -1])
            assert np.all(pd <= true_d * (1 + 1e-5) + 1e-6) and np.all(np.diag(pd) == 0)
            off = ~np.eye(K, dtype=bool)
            assert np.all(pd[off] >= 0.995 * true_d[off] - 1e-3)                 # (a bound nobody could use would pass the first check as well)
            assert np.all(sn >= true_s * (1 - 1e-5)) and np.all(sn <= 2.0 * true_s)
        else:
            with pytest.raises(binding.DpmmError):
                wk.debug_pair_ball()
        wk.last_sweep_work()
        labs = []
        for ep in (1, 2, 3):
            wk.sweep(ep)
            labs.append(wk.get_labels())
            wk.suffstats_packed(None)
            wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        out[on] = labs
        work[on] = wk.last_sweep_work()
        wk.close()
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    t1, t0 = work[1]["tail_pairs"], work[0]["tail_pairs"]
    print(f"D={D} sep={sep} K={K}: tail-pair tests per sweep {t0 / 3:.0f} without the pair-ball test, {t1 / 3:.0f} with it")
    assert t1 <= t0                            # (these problems' covariances are close to isotropic: the 4-feature ball test leaves little to clear; the bench
                                               #  data's inverse-Wishart clusters are where it leaves 3.1 pairs per tile and this one 0.1: scripts/parts_trace.py)


@pytest.mark.parametrize("K,n,sep", [(70, 40000, 30.0), (100, 30011, 30.0), (127, 51000, 25.0), (128, 52000, 25.0), (128, 30000, 2.0), (129, 52001, 25.0), (200, 60000, 30.0),
                                     (300, 120000, 40.0)])
def test_lean_tiles_beyond_64_clusters_without_the_prescreen(pkg, K, n, sep):
    """ADVICE r5 (high + medium).  The device-master path never builds the K > 64 pre-screen (lam == nullptr), so THERE the lean launch runs at
    any K: the candidate loop's chunks beyond the first 64 clusters, ball records from global memory beyond 128 clusters, bin-aligned tiles up
    to 256 bins (K = 128: nbins == the workgroup size -- the table's sentinel used to be written by a thread that does not exist; the last
    bin's tiles then got garbage counts) and 64-position tiles beyond.  DPMM_OPT_PRESCREEN = 0 reaches exactly that kernel configuration
    through set_params: lean launch on / off must give bit-equal labels and sub-labels, the lean launch must have run, and the labels are
    the oracle's draw on the kernel's own table."""
    from dpmmsubclusters_jl_amd import binding
    P = make_problem(64, n, K, seed=700 + K, sep=sep, sorted_points=True)
    out = {}
    for on in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, 64, n, first_index=0, device=0, seed=31)
        wk.upload_points(P["X"])
        wk.set_option(binding.OPT_PRESCREEN, 0)
        wk.set_option(binding.OPT_LEAN_TILES, on)
        wk.set_timing(15)
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        labs, lean_ms = [], []
        for ep in (1, 2, 3):
            wk.sweep(ep)
            labs.append(wk.get_labels())
            lean_ms.append(wk.last_sweep_parts_ms()[0])
            if ep == 3 and on:
                u0, u1 = orc.uniforms(31, 3, 0, 0, n)
                assert np.array_equal(orc.sample_log_cat(wk.debug_loglik(), u0), labs[-1][0])
                assert_sublabels_bit_exact(wk, labs[-1][0], labs[-1][1], u1)
            N_, _, _ = wk.suffstats()
            assert int(N_[:, 0].sum()) == n and np.array_equal(N_[:, 0], N_[:, 1] + N_[:, 2])      # conservation: every point swept, every label in range
            wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        assert (max(lean_ms) > 0.0) == bool(on), lean_ms
        out[on] = labs
        wk.close()
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("K", [96, 128, 160])
def test_lean_tiles_beyond_64_clusters_device_master_chain(pkg, K):
    """The same through the engine's default path (dpmm_niw_master_draw: parameters drawn on the device, no pre-screen): 12 group_steps from
    the generator's labels with K_true = 96 / 128 / 160, lean launch on and off -- K history, labels and sub-labels identical."""
    import importlib
    from dpmmsubclusters_jl_amd import binding
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    N, D = 160000, 64
    X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 1000 + K, 0, N)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    out = {}
    for on in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=17)
        wk.upload_points(X)
        wk.set_option(binding.OPT_LEAN_TILES, on)
        wk.set_timing(15)
        s = host.DPMMSampler(wk, prior, 10.0, N, 17, burnout=4)
        s.start_from_labels(y, 1 + np.random.default_rng(2).integers(0, 2, N), K)
        ks, snaps, ran = [], [], 0.0
        for it in range(12):
            s.group_step(False, False)
            ks.append(s.K)
            ran = max(ran, wk.last_sweep_parts_ms()[0])
            if it % 4 == 3:
                snaps.append(wk.get_labels())
        assert (ran > 0.0) == bool(on)
        out[on] = (ks, snaps)
        wk.close()
    assert out[1][0] == out[0][0], (out[1][0], out[0][0])
    for a, b in zip(out[1][1], out[0][1]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    lab = out[1][1][-1][0]
    assert lab.min() >= 1 and lab.max() <= out[1][0][-1]


def test_a_predictive_table_in_between_does_not_change_the_chain(pkg):
    """the first parameter set behind a dpmm_set_predictive_* call used to be packed without the three-plane images (the flag was
    cleared one statement too late) -- that one sweep drew its sub-labels with the Float32 chain, so a run that calls predict in between was not the
    chain of a run that does not.  Also: the part events of a two-launch sweep are not reported behind a later one-launch sweep."""
    from dpmmsubclusters_jl_amd import binding
    D, n, K = 64, 30000, 6
    P = make_problem(D, n, K, seed=811, sep=20.0, sorted_points=True)
    R = np.stack([np.linalg.cholesky(np.linalg.inv(P["invS"][3 * k].reshape(D, D).astype(np.float64))).T for k in range(K)]).astype(np.float32)
    out = {}
    for with_predict in (0, 1):
        wk = gpu_worker(pkg, P, seed=37)
        wk.set_timing(15)
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)
        if with_predict:
            tab = wk.predict_table_niw(P["mu"][::3], R.reshape(K, -1), P["logdet"][::3], np.full(K, 30.0), P["w"])
            assert tab.shape == (K, n) and np.isfinite(tab).all()
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.last_sweep_work()
        wk.sweep(1)
        out[with_predict] = (wk.get_labels(), wk.last_sweep_work()["b3_evals"], wk.debug_subloglik())
        assert wk.last_sweep_parts_ms()[0] > 0.0
        # a one-launch sweep afterwards (three-plane evaluation off): no part times of the older sweep
        wk.set_option(binding.OPT_B3_SUBLABELS, 0)
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.sweep(2)
        assert tuple(wk.last_sweep_parts_ms()) == (0.0, 0.0, 0.0)
        wk.close()
    assert out[1][1] > 0 and out[1][1] == out[0][1]
    assert np.array_equal(out[0][0][0], out[1][0][0]) and np.array_equal(out[0][0][1], out[1][0][1])
    assert np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("D,N,K,var", [(64, 120000, 10, 100.0), (64, 90000, 6, 2.0), (48, 60000, 5, 50.0), (16, 50000, 4, 50.0), (128, 40000, 4, 100.0)])
def test_folded_launches_and_the_polling_wait_are_the_same_chain(pkg, D, N, K, var):
    """Round 6: the n-independent chain of a step lost launches -- the sort's starts inside the scatter launch, the three-plane images inside the
    hand-over launch (workgroups partitioned by role), the bad-cluster reset counted ahead by the histogram and applied by the scatter, the
    draws' normals generated inside the posteriors' launch, the pair list read from pinned memory, no event between the posteriors and the
    draws launched ahead (the host waits on the posteriors' own records in pinned memory).  Each is value-neutral: 25 group_steps from ONE cluster (splits, merges, bad-cluster resets,
    subset passes) with everything on, with DPMM_OPT_CHAIN_FUSION = 0 and with DPMM_OPT_MASTER_POLL = 0 -- K history, labels and sub-labels equal
    at every checkpoint, and the sub-cluster values the next sweep would draw from (dpmm_debug_subloglik: written by the folded launch in one
    run, by niw_b3_pack_kernel in the other) equal bit for bit."""
    import importlib
    from dpmmsubclusters_jl_amd import binding
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    X, y = host.gaussian_mixture_shard(N, D, K, var, 500 + D, 0, N)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    out = {}
    for name, opts in (("all", ((binding.OPT_CHAIN_FUSION, 0x7fffffff),)), ("default", ()), ("unfused", ((binding.OPT_CHAIN_FUSION, 0),)), ("event", ((binding.OPT_MASTER_POLL, 0),)),
                       ("round5", ((binding.OPT_CHAIN_FUSION, 0), (binding.OPT_MASTER_POLL, 0)))):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=3)
        wk.upload_points(X)
        for o, v in opts:
            wk.set_option(o, v)
        s = host.DPMMSampler(wk, prior, 10.0, N, 3, burnout=5)
        s.init_first_clusters(1)
        ks, snaps = [], []
        for it in range(25):
            s.group_step(False, False)
            ks.append(s.K)
            if it % 8 == 7:
                snaps.append(wk.get_labels())
        tab = None
        if D <= 64:
            s.sample_clusters()                  # a parameter set for the CURRENT clusters (the last step may have changed K)
            wk.K = s.K
            tab = wk.debug_subloglik()
        out[name] = (ks, snaps, tab, [int(v) for v in s.model.get("counters")[4:6]])
        wk.close()
    print("K history:", out["all"][0], "; bad-cluster resets (total, steps):", out["all"][3])
    assert out["all"][3] == out["round5"][3]
    if D == 64 and var == 100.0:
        assert out["all"][3][0] > 0          # (the folded reset did run: hist_kernel<.., SPEC> + scan_tiles_step_kernel + the scatter's re-draw)
    for name in ("default", "unfused", "event", "round5"):
        assert out[name][0] == out["all"][0], name
        for a, b in zip(out[name][1], out["all"][1]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), name
        if out["all"][2] is not None:
            assert np.array_equal(out[name][2], out["all"][2], equal_nan=True), name


def test_lean_tiles_behind_whole_bin_relabels(pkg):
    """niw_lean_kernel aligns its tiles to the bins of the LAST sort; between that sort and the sweep the master relabels whole bins: a split moves
    (k, right) to a new cluster K + 1 (local_clusters_actions.jl:265-278), a merge folds cluster b into a (:293-304), remove-empty renumbers
    (:446-455).  The sort's bin table is then stale -- more clusters than it has bins, bins whose points carry another label -- but every tile
    still holds one label, and the sweep must be the sweep without the lean launch, bit for bit: labels and sub-labels of two sweeps after each
    kind of relabel."""
    from dpmmsubclusters_jl_amd import binding
    D, n, K = 64, 40000 + 9, 6
    P = make_problem(D, n, K + 2, seed=303, sep=25.0, sorted_points=True)     # parameters for up to K + 2 clusters
    def params(wk, k):
        wk.set_params_niw(P["mu"][:3 * k], P["invS"][:3 * k], P["logdet"][:3 * k], P["lr"][:k], (P["w"][:k] / P["w"][:k].sum()).astype(np.float32))
    out = {}
    for on in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, n, first_index=0, device=0, seed=41)
        wk.upload_points(P["X"])
        wk.set_option(binding.OPT_LEAN_TILES, on)
        wk.set_timing(15)
        params(wk, K)
        z0 = np.minimum(P["z"], K - 1) + 1
        wk.set_labels(z0, 1 + (np.arange(n) // 7 & 1))
        labs, ran = [], []
        def two_sweeps(k, ep):
            for e in (ep, ep + 1):
                wk.suffstats_packed(None) if e == ep + 1 else None
                params(wk, k)
                wk.sweep(e)
                labs.append(wk.get_labels())
                ran.append(wk.last_sweep_parts_ms()[0] > 0.0)
        wk.suffstats_packed(None)                      # the sort: 2 K bins
        wk.split([2, 5], [K + 1, K + 2], epoch=3)      # (2, right) -> cluster 7, (5, right) -> cluster 8: the table knows 12 bins, the labels 8 clusters
        two_sweeps(K + 2, 10)
        wk.suffstats_packed(None)
        wk.merge([1], [7])                             # cluster 7 folded into 1
        two_sweeps(K + 2, 20)
        wk.suffstats_packed(None)
        lab = wk.get_labels()[0]
        cnt = np.bincount(lab, minlength=K + 3)[1:K + 3]
        if (cnt == 0).any():
            wk.remove_empty(cnt)
            two_sweeps(int((cnt > 0).sum()), 30)
        assert any(ran) == bool(on)
        out[on] = labs
        wk.close()
    assert len(out[1]) == len(out[0]) >= 4
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_lean_tile_lists_take_turns(pkg):
    """Two tile lists alternate between the sweeps that run the lean launch (a lean launch clears the OTHER list's counter for its successor, the
    sub-label launch reports the length to the host): a chain in which the lean launch is switched on and off between sweeps -- a list is reused
    after sweeps that did not touch it -- gives the labels and sub-labels of the chain that never ran it, and the reported lengths are those of
    the sweeps that did (none behind a sweep without the launch)."""
    from dpmmsubclusters_jl_amd import binding
    D, n, K = 64, 60000 + 21, 9
    P = make_problem(D, n, K, seed=171, sep=12.0, sorted_points=True)
    pattern = [1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 1]
    out = {}
    for mode in ("mixed", "off"):
        wk = gpu_worker(pkg, P, seed=31)
        wk.set_timing(15)
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        labs, ran = [], []
        for ep, on in enumerate(pattern, start=1):
            wk.set_option(binding.OPT_LEAN_TILES, on if mode == "mixed" else 0)
            wk.sweep(ep)
            labs.append(wk.get_labels())
            ran.append(wk.last_sweep_parts_ms()[0] > 0.0)
            wk.suffstats_packed(None)
            wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        if mode == "mixed":
            assert ran == [bool(v) for v in pattern], ran
        else:
            assert not any(ran)
        out[mode] = labs
        wk.close()
    for a, b in zip(out["mixed"], out["off"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("D,sep", [(64, 3.0), (64, 30.0), (40, 3.0)])
def test_three_plane_subcluster_values(pkg, D, sep):
    """DPMM_OPT_B3_SUBLABELS: the sub-cluster log-likelihoods come from bf16 matrix instructions on an exact three-plane split of
    z = x - mu_k and of the factor R_s (6 of the 9 plane products; the dropped ones are below 2^-24 of a product), with
    d_s = R_s (mu_k - mu_s) summed in Float64 and rounded once.  Against the Float64 value of the same Float32 parameters the error must stay
    in the Float32 evaluation's own range (the tolerance north_star gives the Float32 path: a few ulp of the quadratic form), near and far
    from the cluster (far: |z| large, where a split that lost bits would show first)."""
    from dpmmsubclusters_jl_amd import binding
    n, K = 8192 + 13, 6
    P = make_problem(D, n, K, seed=77 + D, sep=sep)
    tabs = {}
    for b3 in (1, 0):
        wk = gpu_worker(pkg, P, seed=3)
        wk.set_option(binding.OPT_B3_SUBLABELS, b3)
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.sweep(1)
        tabs[b3] = wk.debug_subloglik().astype(np.float64)
        wk.close()
    want = np.empty((2 * K, n))
    for k in range(K):
        for s_ in range(2):
            j = 3 * k + 1 + s_
            want[2 * k + s_] = orc.niw_loglik_f64(P["X"], D, P["mu"][j], P["invS"][j], P["logdet"][j]) + np.log(np.float64(P["lr"][k, s_]))
    want += 0.5 * D * D * np.log(2 * np.pi)            # (the GPU tables omit the reference's constant term, mv_gaussian.jl:24)
    scale = 1.0 + np.abs(want)
    e3, e1 = np.abs(tabs[1] - want) / scale, np.abs(tabs[0] - want) / scale
    print(f"D={D} sep={sep}: relative error of the sub-cluster values -- three-plane max {e3.max():.3g} mean {e3.mean():.3g}; Float32 max {e1.max():.3g} mean {e1.mean():.3g}")
    # measured: three-plane max 3.3e-7 .. 3.6e-7, mean 5.8e-8; the Float32 chain max 3.7e-7 .. 4.6e-7, mean 6.4e-8 .. 6.8e-8
    assert e3.max() < 2e-6 and e3.mean() < 2e-7 and e3.max() <= 2 * e1.max()


@pytest.mark.parametrize("D,sep,K", [(64, 2.0, 12), (64, 1.0, 20), (64, 0.8, 7), (52, 2.0, 7), (36, 3.0, 9), (64, 1.5, 60), (64, 40.0, 7), (64, 0.3, 5)])
def test_direction_screen_does_not_change_labels(pkg, D, sep, K):
    """DPMM_OPT_DIRECTION_SCREEN (D in 33 .. 64, K <= 64): q_k(x) >= (u' R_k (x - mu_k))^2 along the ONE direction u = R_k d / |R_k d|, d = mu_k0 - mu_k,
    that separates candidate k from the tile's reference cluster k0 -- all candidates of a tile at once, their K dot products w . (x - mu_k0) through
    one bf16 matrix product with a certified rounding bound (niw_sweep.hip, direction_far).  It only removes candidates the Float32 tests behind it
    would have excluded, so the table and the labels are those of the kernel without it, bit for bit: on overlapping clusters of every
    degree (where it does most of the screening), with zero-padded features, on separated clusters (where it never runs) and on clusters so
    close that it can exclude nothing."""
    from dpmmsubclusters_jl_amd import binding
    n = 30000
    P = make_problem(D, n, K, seed=140 + D + K, sep=sep, sorted_points=True)
    out = {}
    for on in (1, 0):
        wk = gpu_worker(pkg, P, seed=29)
        wk.set_option(binding.OPT_DIRECTION_SCREEN, on)
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)                      # the bin-sorted visiting order
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.last_sweep_work()
        labs = []
        for ep in (1, 2):
            wk.sweep(ep)
            labs.append(wk.get_labels())
            wk.suffstats_packed(None)
            wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        work = wk.last_sweep_work()
        if on:
            tab = wk.debug_loglik()                    # (margin 0, no screens: the full table)
            u0, u1 = orc.uniforms(29, 2, 0, 0, n)
            assert np.array_equal(orc.sample_log_cat(tab, u0), labs[1][0])
        out[on] = (labs, work)
        wk.close()
    for a, b in zip(out[1][0], out[0][0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    w1, w0 = out[1][1], out[0][1]
    t = w1["wave_tiles"]
    print(f"D={D} sep={sep} K={K}: per wave tile -- direction screens {w1['direction_screens'] / t:.2f}; bf16 bottom {w0['bf16_bottom_screens'] / t:.2f} -> "
          f"{w1['bf16_bottom_screens'] / t:.2f}, bf16 top {w0['bf16_top_screens'] / t:.2f} -> {w1['bf16_top_screens'] / t:.2f}, "
          f"Float32 screens {w0['screens16'] / t:.2f} -> {w1['screens16'] / t:.2f}, full evaluations {w0['full_evals'] / t:.2f} -> {w1['full_evals'] / t:.2f}")
    assert w0["direction_screens"] == 0
    # never more work behind it: every cluster it removes skips its 16-row screens
    assert w1["bf16_bottom_screens"] <= w0["bf16_bottom_screens"] and w1["screens16"] <= w0["screens16"] and w1["full_evals"] <= w0["full_evals"]


@pytest.mark.parametrize("D,sep,K,n", [(64, 2.0, 12, 60000), (64, 1.0, 20, 60000), (52, 2.0, 7, 30000), (64, 1.5, 60, 90000), (64, 0.3, 5, 20000), (64, 3.0, 3, 20000)])
def test_lean_kernel_with_the_direction_screen(pkg, D, sep, K, n):
    """Round 6 (DPMM_OPT_LEAN_DIRECTION): while the direction screen's tables exist, niw_lean_kernel runs the screen itself -- its operand is plane h
    of z0 = x - mu_k0, which the kernel holds, |z0| is accumulated by the conversion -- and settles the tiles the screen clears; rounds 4-5 ran
    no lean launch in that regime (labels + sub-labels in two launches over every tile).  The launch behind it walks the handed-on spans with a
    direction-screen instantiation of its own.  Which launch finishes a tile must not show: labels AND sub-labels of a chain of sweeps equal the
    chain with DPMM_OPT_LEAN_DIRECTION = 0 and the chain without any lean launch, bit for bit; the labels are the oracle's draw on the kernel's own
    unscreened table; and the lean launch did run direction screens (overlapping clusters) or none (clusters it cannot separate / too few)."""
    from dpmmsubclusters_jl_amd import binding
    P = make_problem(D, n, K, seed=640 + D + K, sep=sep, sorted_points=True)
    out = {}
    for name, opts in (("lean_dir", ((binding.OPT_DIRECTION_SCREEN, 1),)), ("no_lean_dir", ((binding.OPT_DIRECTION_SCREEN, 1), (binding.OPT_LEAN_DIRECTION, 0))),
                       ("no_lean", ((binding.OPT_DIRECTION_SCREEN, 1), (binding.OPT_LEAN_TILES, 0))), ("no_dir", ((binding.OPT_DIRECTION_SCREEN, 0),))):
        wk = gpu_worker(pkg, P, seed=53)
        for o, v in opts:
            wk.set_option(o, v)
        wk.set_timing(15)
        wk.set_labels(P["z"] + 1, 1 + (np.arange(n) & 1))
        wk.suffstats_packed(None)
        wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        wk.last_sweep_work()
        labs, lean_ms = [], []
        for ep in (1, 2, 3):
            wk.sweep(ep)
            labs.append(wk.get_labels())
            lean_ms.append(wk.last_sweep_parts_ms()[0])
            if ep == 3 and name == "lean_dir":
                u0, u1 = orc.uniforms(53, 3, 0, 0, n)
                assert np.array_equal(orc.sample_log_cat(wk.debug_loglik(), u0), labs[-1][0])
                assert_sublabels_bit_exact(wk, labs[-1][0], labs[-1][1], u1)
            wk.suffstats_packed(None)
            wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
        out[name] = (labs, lean_ms, wk.last_sweep_work())
        wk.close()
    for name in ("no_lean_dir", "no_lean", "no_dir"):
        for a, b in zip(out["lean_dir"][0], out[name][0]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), name
    w = out["lean_dir"][2]
    print(f"D={D} sep={sep} K={K}: lean launch ms {[round(v, 3) for v in out['lean_dir'][1]]} (without the screen in it: {[round(v, 3) for v in out['no_lean_dir'][1]]}); "
          f"direction screens per wave tile {w['direction_screens'] / max(1.0, w['wave_tiles']):.2f}")
    assert max(out["no_lean"][1]) == 0.0
    if K >= 3:
        assert out["lean_dir"][1][0] > 0.0                      # the lean launch runs in the screen's regime now
    if sep in (2.0, 1.0, 1.5) and K >= 7:
        assert w["direction_screens"] > 0


def test_direction_screen_switches_itself_on_and_off(pkg):
    """Automatic mode: the tables are built for the sweep AFTER one whose tiles kept eight or more candidates on average behind the 4-row
    tests (overlapping clusters), and no longer once they keep fewer than 4 (separated clusters)."""
    D, n, K = 64, 20000, 14
    Pov = make_problem(D, n, K, seed=7, sep=1.5, sorted_points=True)
    wk = gpu_worker(pkg, Pov, seed=3)
    wk.set_labels(Pov["z"] + 1, 1 + (np.arange(n) & 1))
    wk.suffstats_packed(None)
    counts = []
    for ep in (1, 2, 3):
        wk.set_params_niw(Pov["mu"], Pov["invS"], Pov["logdet"], Pov["lr"], Pov["w"])
        wk.last_sweep_work()
        wk.sweep(ep)
        wk.sync()
        counts.append(wk.last_sweep_work()["direction_screens"])
    assert counts[0] == 0 and counts[1] > 0 and counts[2] > 0, counts
    # the same worker on separated clusters: nothing left for it, one sweep later it is off
    Psep = make_problem(D, n, K, seed=8, sep=40.0, sorted_points=True)
    wk.upload_points(Psep["X"])
    wk.set_labels(Psep["z"] + 1, 1 + (np.arange(n) & 1))
    wk.suffstats_packed(None)
    counts = []
    for ep in (4, 5, 6):
        wk.set_params_niw(Psep["mu"], Psep["invS"], Psep["logdet"], Psep["lr"], Psep["w"])
        wk.last_sweep_work()
        wk.sweep(ep)
        wk.sync()
        counts.append(wk.last_sweep_work()["direction_screens"])
    assert counts[-1] == 0, counts
    wk.close()


def test_direction_screen_whole_chain_with_the_device_master(pkg):
    """The same chain with the direction screen forced on and switched off: 30 steps of the native engine (device master: the tables are
    built from the images the draw kernels write) on overlapping components of the reference generator (MixtureVar 2) -- labels,
    sub-labels and cluster counts identical, and the screen did run."""
    import importlib
    from dpmmsubclusters_jl_amd import binding
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    N, D, K = 200000, 64, 12
    X, y = host.gaussian_mixture_shard(N, D, K, 2.0, 77, 0, N)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    out = {}
    for mode in (1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=5)
        wk.upload_points(X)
        wk.set_option(binding.OPT_DIRECTION_SCREEN, mode)
        s = host.DPMMSampler(wk, prior, 10.0, N, 5, burnout=5)
        s.start_from_labels(y, 1 + np.random.default_rng(1).integers(0, 2, N), K)
        ks, snaps = [], []
        wk.last_sweep_work()
        for it in range(30):
            s.group_step(False, False)
            ks.append(s.K)
            if it % 10 == 9:
                snaps.append(wk.get_labels())
        work = wk.last_sweep_work()
        out[mode] = (ks, snaps, work)
        wk.close()
    assert out[1][0] == out[0][0]
    for a, b in zip(out[1][1], out[0][1]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    t = out[1][2]["wave_tiles"]
    print(f"direction screens per wave tile {out[1][2]['direction_screens'] / t:.2f}; bf16 bottom screens {out[0][2]['bf16_bottom_screens'] / t:.2f} -> "
          f"{out[1][2]['bf16_bottom_screens'] / t:.2f}")
    assert out[1][2]["direction_screens"] > 0 and out[0][2]["direction_screens"] == 0
    assert out[1][2]["bf16_bottom_screens"] < out[0][2]["bf16_bottom_screens"]


def test_direction_screen_growth_chain_is_the_same_chain(pkg):
    """From ONE cluster on overlapping components (MixtureVar 4): splits and merges change K, the screen switches itself on and off in between
    (automatic mode: candidate counts, yield, measuring sweeps) -- the chain must be the one with the screen off: same K history, same labels."""
    import importlib
    from dpmmsubclusters_jl_amd import binding
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    N, D, K = 300000, 64, 16
    X, y = host.gaussian_mixture_shard(N, D, K, 4.0, 99, 0, N)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    out = {}
    for mode in (-1, 0):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=11)
        wk.upload_points(X)
        wk.set_option(binding.OPT_DIRECTION_SCREEN, mode)
        s = host.DPMMSampler(wk, prior, 10.0, N, 11, burnout=8)
        s.init_first_clusters(1)
        ks, ran = [], 0.0
        wk.last_sweep_work()
        for it in range(120):
            s.group_step(False, False)
            ks.append(s.K)
            if it % 20 == 19:
                ran += wk.last_sweep_work()["direction_screens"]
        out[mode] = (ks, wk.get_labels(), ran)
        wk.close()
    print("K history (every 10th):", out[-1][0][::10], "; direction screens run in automatic mode:", out[-1][2] > 0)
    assert out[-1][0] == out[0][0]
    assert np.array_equal(out[-1][1][0], out[0][1][0]) and np.array_equal(out[-1][1][1], out[0][1][1])
    assert out[0][2] == 0


@pytest.mark.parametrize("kind,D", [("midpoints", 128), ("midpoints", 256), ("trailing", 256), ("outlier", 256), ("outlier", 100)])
def test_big_bracket_is_a_lower_bound_of_the_reference_value(pkg, kind, D):
    """D in 65 .. 256: the bracket launch in front of the sweep (niw_bracket_big_kernel) writes, for every point of a label-homogeneous
    128-point tile, a LOWER bound of a_k0 = cst - q / 2 from two bf16 matrix passes over 72 (D = 256) fragments and a certified rounding
    constant.  On operands chosen against the bound (every entry of R and x just below a bf16 midpoint and of one sign; points displaced
    along one trailing feature; far outliers) it must stay below the Float32 value of the kernel's own table for every point."""
    n = 4096
    P = _bracket_problem(kind, D, n, seed=3 + D)
    wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=2)
    wk.upload_points(P["X"])
    wk.set_labels(np.ones(n, np.int64), 1 + (np.arange(n) & 1))          # every point in cluster 1: every tile is homogeneous
    wk.set_params_niw_chol(P["mu"], P["R"], P["logdet"], P["lr"], P["w"])
    aref, flags = wk.debug_bracket_big()
    assert np.all(flags == 1)                                            # 1 + k0, k0 = 0
    tab = wk.debug_loglik()[0].astype(np.float64)                        # a_1 of every point, Float32 evaluation (includes log w)
    fin = np.isfinite(tab)
    gap = tab[fin] - aref[fin].astype(np.float64)
    print(f"{kind} D={D}: bracket width (a - a_lower) min {gap.min():.3g}, median {np.median(gap):.3g}, max {gap.max():.3g}; |a| up to {np.abs(tab[fin]).max():.3g}")
    assert np.all(gap >= -1e-6 * np.abs(tab[fin])), gap.min()
    assert np.median(gap / np.maximum(1.0, np.abs(tab[fin]))) < 0.2      # ... and it is a bracket, not a trivial bound
    wk.close()
