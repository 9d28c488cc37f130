"""GPU parity tests for the Multinomial worker path (histogram sufficient statistics + alpha'x log-likelihood).
Tolerances: log-lik table rtol 1e-5 / atol 1e-3 (f32 contraction over D terms, different summation order);
draw given the GPU's own table: bit-exact; labels vs the oracle's own f32 table: counted near-boundary flips;
N and sum x: exact for count data (integer-valued Float64 sums)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


def make_problem(D, n, K, trials, seed):
    rng = np.random.default_rng(seed)
    P = rng.dirichlet(np.ones(D) * 0.5, size=3 * K)
    for k in range(K):  # sub-clusters are perturbations of the cluster
        P[3 * k + 1] = 0.8 * P[3 * k] + 0.2 * rng.dirichlet(np.ones(D))
        P[3 * k + 2] = 0.8 * P[3 * k] + 0.2 * rng.dirichlet(np.ones(D))
    z = rng.integers(0, K, n)
    X = np.stack([rng.multinomial(trials, P[3 * k]) for k in z]).astype(np.float32)
    logp = np.log(np.maximum(P, 1e-30)).astype(np.float32)
    w = rng.dirichlet(np.ones(K) * 5).astype(np.float32)
    lr = rng.dirichlet(np.ones(2) * 5, size=K).astype(np.float32)
    return dict(D=D, n=n, K=K, X=X, logp=logp, w=w, lr=lr, z=z)


def worker(pkg, P, seed, first=0):
    wk = pkg.Worker(pkg.PRIOR_MULT, P["D"], P["n"], first_index=first, device=0, seed=seed)
    wk.upload_points(P["X"])
    wk.set_params_mult(P["logp"], P["lr"], P["w"])
    return wk


@pytest.mark.parametrize("D,n,K,trials", [(100, 1000, 2, 50), (10, 3000, 5, 30), (1000, 3000, 32, 100), (37, 2049, 7, 20), (257, 1500, 50, 60)])
def test_table_and_labels(pkg, D, n, K, trials):
    P = make_problem(D, n, K, trials, seed=D + K)
    seed, epoch, first = 99, 3, 12345
    wk = worker(pkg, P, seed, first)
    tab = wk.debug_loglik()
    want = np.stack([P["X"].astype(np.float64) @ P["logp"][3 * k].astype(np.float64) + np.log(np.float64(P["w"][k])) for k in range(K)])
    np.testing.assert_allclose(tab, want, rtol=1e-5, atol=1e-3)
    wk.sweep(epoch)
    lab, sub = wk.get_labels()
    u0, u1 = orc.uniforms(seed, epoch, 0, first, n)
    assert np.array_equal(orc.sample_log_cat(tab, u0), lab)           # draw arithmetic: bit-exact
    if 2 * K <= 1024:                                                 # sub-label draw from the GPU's own left / right values: bit-exact too
        tab2 = wk.debug_subloglik()
        i = np.arange(n)
        pair = np.stack([tab2[2 * (lab - 1), i], tab2[2 * (lab - 1) + 1, i]])
        assert np.array_equal(orc.sample_log_cat(pair, u1), sub)
    olab, osub = orc.sweep_mult(P["X"], D, P["logp"], np.log(P["w"]), np.log(P["lr"]), seed, epoch, first)
    same = lab == olab
    print(f"D={D} K={K} n={n}: label flips vs oracle {(lab != olab).sum()}, sub-label flips {(sub[same] != osub[same]).sum()}")
    assert (lab != olab).sum() <= max(1, int(1e-5 * n))
    assert (sub[same] != osub[same]).sum() <= max(2, int(1e-4 * n))
    assert (lab == P["z"] + 1).mean() > 0.5
    wk.sweep(epoch + 1, final=True)
    assert np.array_equal(wk.get_labels()[0], orc.argmax_rows(tab))
    wk.close()


def test_suffstats_golden_bit_exact(pkg, golden_dir):
    """The reference's own checkpoint (test/save_load_test/checkpoint_20.jld2): Float32 points_sum, bit-exact."""
    g = np.load(f"{golden_dir}/mnm_golden.npz")
    X = np.ascontiguousarray(g["X"], np.float32)
    wk = pkg.Worker(pkg.PRIOR_MULT, 100, 1000, device=0, seed=1)
    wk.upload_points(X)
    wk.set_labels(g["labels"], g["sub"])
    wk.set_num_clusters(2)
    N, s = wk.suffstats()
    i = 0
    for k in range(2):
        for w in range(3):
            assert np.array_equal(s[k, w].astype(np.float32), g["points_sum"][i])
            assert np.array_equal((g["prior_alpha"] + s[k, w].astype(np.float32)).astype(np.float32), g["post_alpha"][i])
            i += 1
    assert N[:, 0].tolist() == [463, 537]
    wk.close()


@pytest.mark.parametrize("D,n,K", [(1000, 20000, 32), (100, 5000, 3), (7, 3000, 4)])
def test_suffstats_vs_oracle(pkg, D, n, K):
    rng = np.random.default_rng(D)
    X = rng.poisson(0.3, size=(n, D)).astype(np.float32)
    lab = rng.integers(1, K + 1, n); sub = rng.integers(1, 3, n)
    wk = pkg.Worker(pkg.PRIOR_MULT, D, n, device=0, seed=1)
    wk.upload_points(X)
    wk.set_labels(lab, sub)
    wk.set_num_clusters(K)
    N, s = wk.suffstats()
    oN, os_ = orc.suffstats_mult(X, D, lab, sub, K)
    assert np.array_equal(N, oN.astype(np.float64)) and np.array_equal(s, os_.astype(np.float64))
    wk.close()


def test_fit_multinomial_module_test(pkg):
    """test/module_tests.jl:49-60 (without the save/load half): mnmm data N=10^3, D=100, 20 components, 50 trials;
    params of test/save_load_test/multinomial_params.jl (alpha=1e5, prior ones(100), 39 iterations): asserts K > 1."""
    import importlib
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    x, labels, _ = host.generate_mnmm_data(10 ** 3, 100, 20, 50, seed=12345)
    hyper = host.multinomial_hyper(np.ones(100, np.float32))
    dp = host.dp_parallel(x, hyper, np.float32(100000.0), 39, 1, None, False, False, 20)
    assert dp[0].num_clusters > 1
    # and a well-separated problem is actually recovered
    x, labels, _ = host.generate_mnmm_data(20000, 50, 5, 200, seed=3)
    res = host.fit(x, hyper.__class__(np.ones(50, np.float32)), 10.0, iters=60, burnout=5, gt=labels, seed=5, verbose=False)
    assert res[4][-1] > 0.9


def test_f32_fallback_for_non_count_data(pkg):
    """Data that is not exactly representable in bf16 must take the FP32-MFMA kernel and give the same quality."""
    P = make_problem(100, 3000, 4, 40, seed=77)
    P["X"] = (P["X"] + np.float32(0.3)).astype(np.float32)      # 0.3 is not bf16-exact
    wk = worker(pkg, P, seed=5)
    tab = wk.debug_loglik()
    want = np.stack([P["X"].astype(np.float64) @ P["logp"][3 * k].astype(np.float64) + np.log(np.float64(P["w"][k])) for k in range(4)])
    np.testing.assert_allclose(tab, want, rtol=1e-5, atol=1e-3)
    wk.sweep(1)
    lab, _ = wk.get_labels()
    u0, _ = orc.uniforms(5, 1, 0, 0, 3000)
    assert np.array_equal(orc.sample_log_cat(tab, u0), lab)
    wk.close()


@pytest.mark.parametrize("D,n,K", [(1000, 6000, 32), (129, 4000, 5), (64, 3000, 3)])
def test_byte_copy_path_equals_float_paths(pkg, D, n, K):
    """Integer data in [0, 255] is streamed from a lossless BYTE copy (1 byte per element, 128-feature k-steps).  Against the
    Float32-reading bf16 kernel (DPMM_OPT_MULT_NO_U8) on the same data: the table agrees to summation-order rounding, the label
    draws differ only where a CDF edge sits within that rounding, and the statistics -- integer sums -- are bit-identical."""
    from dpmmsubclusters_jl_amd import binding
    P = make_problem(D, n, K, 80, seed=3 * D + K)
    P["X"][::7, 0] = 255.0                                           # the largest byte value is in the data
    lab0 = (P["z"] + 1).astype(np.int64); sub0 = 1 + (np.arange(n) & 1)
    out = {}
    for name, no_u8 in (("u8", 0), ("bf16", 1)):
        wk = pkg.Worker(pkg.PRIOR_MULT, D, n, first_index=7, device=0, seed=11)
        wk.set_option(binding.OPT_MULT_NO_U8, no_u8)
        wk.upload_points(P["X"])
        wk.set_params_mult(P["logp"], P["lr"], P["w"])
        tab = wk.debug_loglik()
        wk.set_labels(lab0, sub0)
        st = wk.suffstats_packed()
        wk.sweep(4)
        lab, sub = wk.get_labels()
        u0, _ = orc.uniforms(11, 4, 0, 7, n)
        assert np.array_equal(orc.sample_log_cat(tab, u0), lab)       # each path draws bit-exactly from its own table
        out[name] = (tab, lab, sub, st, wk.suffstats_packed())
        wk.close()
    np.testing.assert_allclose(out["u8"][0], out["bf16"][0], rtol=2e-6, atol=2e-4)
    assert (out["u8"][1] != out["bf16"][1]).sum() <= max(1, int(1e-5 * n))
    assert np.array_equal(out["u8"][3], out["bf16"][3])              # statistics of the same labelling: bit-identical
    want = np.zeros((2 * K, 1 + D))
    np.add.at(want, 2 * (lab0 - 1) + (sub0 - 1), np.concatenate([np.ones((n, 1)), P["X"].astype(np.float64)], axis=1))
    assert np.array_equal(out["u8"][3], want)
    # non-integer or out-of-range data must NOT take the byte path (it falls back to the Float32-reading kernels)
    Xf = P["X"].copy(); Xf[5, 3] = 256.0
    wk = pkg.Worker(pkg.PRIOR_MULT, D, n, device=0, seed=11)
    wk.upload_points(Xf)
    wk.set_labels(lab0, sub0); wk.set_num_clusters(K)
    st = wk.suffstats_packed()
    assert st[2 * (lab0[5] - 1) + (sub0[5] - 1), 1 + 3] == want[2 * (lab0[5] - 1) + (sub0[5] - 1), 1 + 3] - P["X"][5, 3] + 256.0
    wk.close()


def test_device_dirichlet_draws_law_and_handover(pkg):
    """dpmm_mult_master_draw: log.(rand(Dirichlet(alpha'))) with alpha' = alpha + sum x (multinomial_prior.jl:16-25) for all 3K distributions
    on the device.  Law: E[p_d] = alpha'_d / sum alpha' and E[log p_d] = digamma(alpha'_d) - digamma(sum alpha') -- the latter is what a
    wrong small-alpha branch (alpha' < 1: prior components without counts) would miss -- to 5 standard errors over 500 draws, for populated,
    one-sided and empty distributions; rows sum to one; same epoch = same draw; the table the sweep kernels evaluate equals the one computed
    from the fetched log-probabilities."""
    from scipy.special import digamma, polygamma
    D, n, K, trials = 300, 6000, 4, 30
    P = make_problem(D, n, K, trials, seed=123)
    rng = np.random.default_rng(7)
    lab = rng.integers(1, K + 1, n); sub = rng.integers(1, 3, n)
    sub[lab == 2] = 1                      # cluster 2: right sub-cluster empty -> its draw comes from the prior
    lab[lab == 4] = 1                      # cluster 4: empty
    wk = worker(pkg, P, seed=31)
    wk.set_labels(lab, sub); wk.set_num_clusters(K)
    alpha = rng.uniform(0.15, 2.5, D).astype(np.float32)       # components below 1: the Gamma(a + 1) U^(1/a) branch
    wk.mult_master_setup(alpha)
    wk.suffstats_packed(None)
    N, sums = orc.suffstats_mult(P["X"], D, lab, sub, K)
    apost = alpha[None, None, :] + sums.astype(np.float32)      # (K, 3, D), Float32 like the reference
    apost = np.where(N[:, :, None] == 0, alpha[None, None, :], apost).reshape(3 * K, D).astype(np.float64)
    lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
    reps = 500
    m1 = np.zeros((3 * K, D)); l1 = np.zeros((3 * K, D)); l2 = np.zeros((3 * K, D))
    for ep in range(1, reps + 1):
        wk.mult_master_draw(ep, lr, w)
        lp = wk.mult_master_draws(K).astype(np.float64)
        p = np.exp(lp)
        assert np.allclose(p.sum(1), 1.0, atol=1e-5)
        m1 += p; l1 += lp; l2 += lp * lp
    a0 = apost.sum(1, keepdims=True)
    Ep = apost / a0
    se_p = np.sqrt(Ep * (1 - Ep) / (a0 + 1) / reps)
    assert np.all(np.abs(m1 / reps - Ep) < 5 * se_p + 1e-7), np.max(np.abs(m1 / reps - Ep) / se_p)
    El = digamma(apost) - digamma(a0)
    se_l = np.sqrt((polygamma(1, apost) - polygamma(1, a0)) / reps)
    assert np.all(np.abs(l1 / reps - El) < 5 * se_l + 1e-6), np.max(np.abs(l1 / reps - El) / se_l)
    # deterministic in (seed, epoch, position); rows of different distributions differ
    wk.mult_master_draw(3, lr, w); a = wk.mult_master_draws(K)
    wk.mult_master_draw(4, lr, w); wk.mult_master_draw(3, lr, w); b = wk.mult_master_draws(K)
    assert np.array_equal(a, b) and not np.array_equal(a[0], a[3])
    # hand-over: the sweep kernels' table from the packed planes == x . logp + log w from the fetched draws
    tab = wk.debug_loglik()
    want = np.stack([P["X"].astype(np.float64) @ b[3 * k].astype(np.float64) + np.log(np.float64(w[k])) for k in range(K)])
    np.testing.assert_allclose(tab, want, rtol=1e-5, atol=1e-3)
    # a subset pass leaves incomplete rows: the draw refuses them
    wk.suffstats_packed(np.array([1]))
    with pytest.raises(pkg.DpmmError):
        wk.mult_master_draw(5, lr, w)
    wk.close()


def _mult_chain(pkg, host, engine, x, D, N, dev, init=1, seed=5, burnout=5):
    hyper = host.multinomial_hyper(np.ones(D, np.float32))
    wk = pkg.Worker(hyper.kind, D, N, device=0, seed=seed)
    wk.upload_points(np.ascontiguousarray(x.T))
    s = host.DPMMSampler(wk, hyper, 10.0, N, seed, burnout=burnout)
    s.model.set_option(engine.OPT_DEVICE_MASTER, dev)
    s.init_first_clusters(init)
    return wk, s


def test_engine_uses_the_device_dirichlet_draws(pkg):
    """The engine's Multinomial master with the draws on the device (DPMMH_OPT_DEVICE_MASTER; default for D >= 128): a step that follows a
    full statistics pass draws on the device (the worker holds the rows of all K clusters), a step that follows an accepted split / merge /
    removal draws on the host (counters[6] bit 0 says so beforehand); both chains recover the components; the log-probabilities the engine
    reports are the ones the device drew (normalised rows)."""
    import importlib
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    D, N, Kt = 200, 20000, 5
    x, y, _ = host.generate_mnmm_data(N, D, Kt, 150, seed=3)[:3]
    x = np.ascontiguousarray(x, np.float32)
    res = {}
    for dev in (1, 0):
        wk, s = _mult_chain(pkg, host, engine, x, D, N, dev)
        on_dev = on_host = 0
        ks = [s.K]
        for it in range(60):
            host_next = int(s.model.get("counters")[6]) & 1
            s.group_step(False, False)
            ks.append(s.K)
            lp = s.model.get("logp")
            if ks[-1] == ks[-2]:          # (rows of clusters born or renumbered in this step have no draw yet)
                assert np.allclose(np.exp(lp.astype(np.float64)).sum(1), 1.0, atol=1e-4)
            if dev:
                try:
                    fetched = wk.mult_master_draws(lp.shape[0] // 3) if not host_next else None
                except pkg.DpmmError:
                    fetched = None
                if not host_next and it > 0:
                    on_dev += 1
                else:
                    on_host += 1
                if fetched is not None and s.K == lp.shape[0] // 3 and not host_next and it > 0 and ks[-1] == ks[-2]:
                    assert np.array_equal(fetched, lp)
        lab, _ = wk.get_labels()
        from dpmmsubclusters_jl_amd.host.sampler import nmi_vi_from_contingency
        C = np.zeros((int(lab.max()), int(y.max())))
        np.add.at(C, (lab - 1, np.asarray(y) - 1), 1)
        res[dev] = (ks, nmi_vi_from_contingency(C)[0], on_dev, on_host)
        wk.close()
    print("device draws:", res[1][2], "steps on the device,", res[1][3], "on the host; K", res[1][0][-1], "NMI", res[1][1], "| host K", res[0][0][-1], "NMI", res[0][1])
    assert res[1][2] >= 30 and res[1][3] >= 2            # both kinds of step happened (splits on the way to 5 components)
    for dev in (0, 1):
        assert res[dev][0][-1] >= Kt - 1 and res[dev][1] > 0.9


def test_resume_continues_the_chain_with_device_dirichlet_draws(pkg, tmp_path):
    """A checkpoint carries the rows and whether the NEXT draws of the running chain were due on the host (counters[6]): a resumed run puts
    the rows back on the device (dpmm_mult_master_put_rows) and continues onto the same draws -- labels equal after the remaining steps,
    from a checkpoint taken in a quiet step and from one taken right after a split."""
    import importlib
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    ckpt = importlib.import_module("dpmmsubclusters_jl_amd.host.checkpoint")
    D, N = 160, 8000
    x, y, _ = host.generate_mnmm_data(N, D, 4, 120, seed=9)[:3]
    x = np.ascontiguousarray(x, np.float32)
    wk, s = _mult_chain(pkg, host, engine, x, D, N, 1, burnout=3)
    saved, ks = {}, [s.K]
    total = 40
    for it in range(1, total + 1):
        s.group_step(False, False)
        ks.append(s.K)
        flag = int(s.model.get("counters")[6]) & 1
        kind = "split" if ks[-1] > ks[-2] else ("quiet" if (not flag and it > 8) else None)
        if kind and kind not in saved and it < total - 3:
            st = {f: s.model.get(f) for f in ckpt._STATE_FIELDS}
            st["K"], st["it"], st["flag"] = s.K, it, flag
            st["labels"], st["sub"] = wk.get_labels()
            saved[kind] = st
    want = wk.get_labels()
    wk.close()
    assert set(saved) == {"split", "quiet"}, ks
    assert saved["split"]["flag"] == 1 and saved["quiet"]["flag"] == 0
    for kind, st in saved.items():
        wk2, s2 = _mult_chain(pkg, host, engine, x, D, N, 1, burnout=3)
        wk2.set_labels(st["labels"], st["sub"])
        s2.model.set("K", st["K"])
        wk2.set_num_clusters(st["K"])
        for f in ckpt._STATE_FIELDS:
            s2.model.set(f, st[f])
        for _ in range(total - st["it"]):
            s2.group_step(False, False)
        got = wk2.get_labels()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), kind
        wk2.close()


@pytest.mark.parametrize("K,D,n", [(5, 300, 4000), (12, 300, 5000), (40, 260, 6000), (70, 130, 5000)])
@pytest.mark.parametrize("ordered", [True, False])
def test_selective_row_blocks_equal_all_rows(pkg, K, D, n, ordered):
    """The byte kernel evaluates, per tile, the cluster rows + the sub-cluster row blocks of the clusters its points WERE in, and takes a second
    pass for blocks a new label asks for (mult_sweep_u8_kernel).  With previous labels that are right for ~70 % of the points and random for the
    rest -- so that tiles take both routes -- in the bin-sorted visiting order and in storage order: labels and sub-labels equal the oracle's
    draw applied to the all-rows tables of the same kernel (bit-exact), for K below / across / beyond one block of 16 cluster rows."""
    P = make_problem(D, n, K, 60, seed=11 * K + D)
    rng = np.random.default_rng(K)
    prev = (P["z"] + 1).astype(np.int64)
    wrong = rng.random(n) < 0.3
    prev[wrong] = rng.integers(1, K + 1, wrong.sum())
    sub0 = rng.integers(1, 3, n)
    seed, first = 17, 999
    wk = worker(pkg, P, seed, first)
    wk.set_labels(prev, sub0); wk.set_num_clusters(K)
    if ordered:
        wk.suffstats_packed()                 # leaves the bin-sorted order of these labels
    wk.set_params_mult(P["logp"], P["lr"], P["w"])
    tab = wk.debug_loglik()
    tab2 = wk.debug_subloglik()
    for epoch in (1, 2):                       # the second sweep starts from the first one's labels (previous = current almost everywhere)
        wk.sweep(epoch)
        lab, sub = wk.get_labels()
        u0, u1 = orc.uniforms(seed, epoch, 0, first, n)
        assert np.array_equal(orc.sample_log_cat(tab, u0), lab), (K, ordered, epoch)
        i = np.arange(n)
        pair = np.stack([tab2[2 * (lab - 1), i], tab2[2 * (lab - 1) + 1, i]])
        assert np.array_equal(orc.sample_log_cat(pair, u1), sub), (K, ordered, epoch)
        if ordered:
            wk.suffstats_packed()
            wk.set_params_mult(P["logp"], P["lr"], P["w"])
    assert (lab != prev).mean() > 0.2          # (the sweep did move the points that started in a wrong cluster)
    wk.close()


def test_device_log_marginals_match_the_host(pkg):
    """mult_marginal_kernel behind the per-step statistics (DPMMH_OPT_DEVICE_MASTER for the Multinomial prior): N and the log-marginals of the 3K
    distributions (multinomial_prior.jl:34-39) and the Hastings ratios of every merge candidate (pooled statistics, LCA:385-413) equal the
    host's Float64 evaluation of the same rows to 1e-10 relative; an accepted split sends the step's pairs back to the host."""
    import importlib
    from scipy.special import gammaln
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    D, N = 200, 20000
    x, y, _ = host.generate_mnmm_data(N, D, 6, 150, seed=3)[:3]
    x = np.ascontiguousarray(x, np.float32)
    wk, s = _mult_chain(pkg, host, engine, x, D, N, 1, burnout=4)
    alpha = np.ones(D, np.float32)
    checked = 0
    for it in range(60):
        s.group_step(False, False)
        K = s.K
        hr = s.model.debug_merge_log_hr()
        if K < 3 or not np.isfinite(hr).any():
            continue
        rows = s.model.get("packed").reshape(K, 2, 1 + D)
        Nd, Ld = s.model.get("N"), s.model.get("log_marginal")
        l, r = rows[:, 0], rows[:, 1]
        for w, st in enumerate((l + r, l, r)):
            ap = (alpha[None, :] + st[:, 1:].astype(np.float32)).astype(np.float64)
            want = gammaln(alpha.astype(np.float64).sum()) - gammaln(ap.sum(1)) + (gammaln(ap) - gammaln(alpha.astype(np.float64))[None, :]).sum(1)
            want = np.where(st[:, 0] == 0, 0.0, want)
            assert np.array_equal(Nd[w::3], st[:, 0])
            np.testing.assert_allclose(Ld[w::3], want, rtol=1e-10, atol=1e-7)
        s.model.set_option(engine.OPT_DEVICE_MASTER, 0)          # the same candidates, pooled on the host
        hr_host = s.model.debug_merge_log_hr()
        s.model.set_option(engine.OPT_DEVICE_MASTER, 1)
        assert np.array_equal(np.isfinite(hr), np.isfinite(hr_host))
        m = np.isfinite(hr)
        np.testing.assert_allclose(hr[m], hr_host[m], rtol=1e-10, atol=1e-6)
        checked += int(m.sum())
    assert checked > 20 and s.K >= 5
    wk.close()


def test_draws_launched_ahead_are_the_draws_of_the_next_step(pkg):
    """DPMM_OPT_MULT_DRAWS_AHEAD (default 1): dpmm_step_stats launches the next Dirichlet draws behind the statistics into a second set of
    buffers and dpmm_mult_master_draw swaps the sets when its arguments are the ones guessed.  The chain -- K history, labels, sub-labels, the
    log-probabilities of every step -- is the chain with the option off; quiet steps take the draws made ahead, steps that follow a split,
    merge or removal do not."""
    import importlib
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    D, N, Kt = 200, 20000, 5
    x, y, _ = host.generate_mnmm_data(N, D, Kt, 150, seed=3)[:3]
    x = np.ascontiguousarray(x, np.float32)
    res = {}
    for ahead in (1, 0):
        wk, s = _mult_chain(pkg, host, engine, x, D, N, 1)
        wk.set_option(importlib.import_module("dpmmsubclusters_jl_amd.binding").OPT_MULT_DRAWS_AHEAD, ahead)
        ks, lps = [s.K], []
        for it in range(50):
            s.group_step(False, False)
            ks.append(s.K)
            lps.append(s.model.get("logp").copy())
        res[ahead] = (ks, wk.get_labels(), lps, wk.debug_mult_draws_ahead())
        wk.close()
    assert res[1][0] == res[0][0]
    assert np.array_equal(res[1][1][0], res[0][1][0]) and np.array_equal(res[1][1][1], res[0][1][1])
    for i, (a, b) in enumerate(zip(res[1][2], res[0][2])):
        if res[1][0][i + 1] == res[1][0][i]:            # (rows of clusters born or renumbered in this step have no draw yet: whatever the buffer held)
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), i
    changes = sum(1 for i in range(1, len(res[1][0])) if res[1][0][i] != res[1][0][i - 1])
    print("draws taken from the set made ahead:", res[1][3], "of 50 steps;", changes, "steps changed K")
    assert res[0][3] == 0 and 30 <= res[1][3] <= 50 - changes


def test_master_rows_follow_the_labels_through_merges_and_removals(pkg):
    """The Multinomial device master leaves a step's rows in the worker's pinned block and pulls them only when it must (accepted split / merge /
    removal, state access).  From 14 random clusters on 4-component data the chain merges and removes its way down: after EVERY step the rows the
    model reports ("packed": [N, sum x] per cluster and side) are the counts of the labels the worker holds, recomputed here with numpy --
    exactly (integer data)."""
    import importlib
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    D, N, Kt = 160, 6000, 4
    x, y, _ = host.generate_mnmm_data(N, D, Kt, 120, seed=11)[:3]
    x = np.ascontiguousarray(x, np.float32)            # (D, N)
    wk, s = _mult_chain(pkg, host, engine, x, D, N, 1, init=14, burnout=3)
    X = x.T.astype(np.float64)
    ks = [s.K]
    for it in range(45):
        s.group_step(False, False)
        ks.append(s.K)
        rows = np.asarray(s.model.get("packed"), np.float64).reshape(2 * s.K, -1)
        lab, sub = wk.get_labels()
        for k in range(s.K):
            for side in (1, 2):
                m = (lab == k + 1) & (sub == side)
                want = np.concatenate([[m.sum()], X[m].sum(0)])
                assert np.array_equal(rows[2 * k + side - 1][: D + 1], want), (it, k, side, ks)
    checked = wk.get_labels()
    wk.close()
    print("K history:", ks)
    assert ks[0] == 14 and min(ks) < 14 and ks[-1] <= 8        # merges / removals happened
    # the same chain when nobody asks for the rows in between (they are pulled only by the accepted merges / removals themselves)
    wk, s = _mult_chain(pkg, host, engine, x, D, N, 1, init=14, burnout=3)
    ks2 = [s.K]
    for it in range(45):
        s.group_step(False, False)
        ks2.append(s.K)
    quiet = wk.get_labels()
    wk.close()
    assert ks2 == ks and np.array_equal(quiet[0], checked[0]) and np.array_equal(quiet[1], checked[1])


def test_parameters_survive_a_new_upload_of_another_kind_of_data(pkg):
    """The parameter images are packed for the sweep kernel of the points in place (byte planes for small counts, bf16 planes for bf16-exact
    data, Float32 fragments otherwise).  A new upload that changes the kind re-packs them from the raw rows: the log-likelihood table of the
    new points with the OLD parameters is x . logp + log w in every order of the three kinds."""
    P = make_problem(130, 1500, 6, 40, seed=5)
    rng = np.random.default_rng(1)
    kinds = {"bytes": P["X"], "bf16": P["X"] * 256.0, "f32": (P["X"] + rng.random(P["X"].shape).astype(np.float32) * 0.37).astype(np.float32)}
    for order in (("bytes", "f32", "bf16", "bytes"), ("f32", "bytes"), ("bf16", "f32")):
        wk = pkg.Worker(pkg.PRIOR_MULT, P["D"], P["n"], device=0, seed=3)
        wk.upload_points(kinds[order[0]])
        wk.set_params_mult(P["logp"], P["lr"], P["w"])
        for name in order:
            if name != order[0] or name == order[-1]:
                wk.upload_points(kinds[name])
            X = kinds[name].astype(np.float64)
            want = np.stack([X @ P["logp"][3 * k].astype(np.float64) + np.log(np.float64(P["w"][k])) for k in range(P["K"])])
            np.testing.assert_allclose(wk.debug_loglik(), want, rtol=2e-5, atol=2e-3 * max(1.0, float(np.abs(want).max()) / 1e3), err_msg=f"{order} at {name}")
        wk.close()


def test_outlier_component_with_the_device_master(pkg):
    """fit(...; outlier_weight, outlier_params) on Multinomial data wide enough for the device master (D >= 128): cluster 1 is the fixed outlier
    component with its own prior (local_clusters_actions.jl:42-61, :424-437), drawn on the device with the others (outlier_first).  The chain is
    the same with the draws launched ahead and with them made inside dpmm_mult_master_draw, the outlier keeps its constant weight, and the
    components are recovered."""
    import importlib
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    binding = importlib.import_module("dpmmsubclusters_jl_amd.binding")
    D, N, Kt = 160, 6000, 3
    x, y, _ = host.generate_mnmm_data(N, D, Kt, 120, seed=13)[:3]
    x = np.ascontiguousarray(x, np.float32)
    rng = np.random.default_rng(2)
    out = rng.random(N) < 0.05
    x[:, out] = rng.multinomial(120, np.ones(D) / D, size=int(out.sum())).T          # uniform bags of words: nobody's component
    hyper = host.multinomial_hyper(np.ones(D, np.float32))
    ohyper = host.multinomial_hyper(np.full(D, 50.0, np.float32))
    res = {}
    for ahead in (1, 0):
        made = []

        def factory(*a, **kw):
            wk = binding.Worker(*a, **kw)
            wk.set_option(binding.OPT_MULT_DRAWS_AHEAD, ahead)
            made.append(wk)
            return wk
        r = host.fit(x, hyper, 10.0, iters=60, seed=21, burnout=5, verbose=False, outlier_weight=0.05, outlier_params=ohyper, worker_factory=factory)
        res[ahead] = (np.asarray(r[0]).copy(), np.asarray(r[2]).copy(), made[0].debug_mult_draws_ahead())
    assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1])
    labels, weights = res[1][0], res[1][1]
    assert weights[0] == np.float32(0.05)
    print("draws taken ahead:", res[1][2], "| clusters", len(weights), "| outliers labelled 1:", float((labels[out] == 1).mean()), "| others labelled 1:", float((labels[~out] == 1).mean()))
    assert res[0][2] == 0 and res[1][2] >= 20
    assert (labels[out] == 1).mean() > 0.8 and (labels[~out] == 1).mean() < 0.05
    from dpmmsubclusters_jl_amd.host.sampler import nmi_vi_from_contingency
    keep = ~out
    C = np.zeros((int(labels.max()), int(np.asarray(y).max())))
    np.add.at(C, (labels[keep] - 1, np.asarray(y)[keep] - 1), 1)
    assert nmi_vi_from_contingency(C)[0] > 0.9
