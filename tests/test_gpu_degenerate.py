"""The reference's degenerate rows on the HIP path:

  * sample_log_cat_array! (src/utils.jl:19-31): NaN -> -Inf (utils.jl:22); a row that is all -Inf becomes all NaN after the max
    subtraction and StatsBase.sample's scan (`cw < t && i < n` with t = NaN) returns index 1;
  * the argmax path (src/local_clusters_actions.jl:129-130) has no NaN guard: Julia's argmax returns the FIRST NaN, else the first
    maximum;
  * a zero mixture weight (log w = -Inf, which a Float32 Dirichlet draw of a tiny cluster can produce) or a zero sub-cluster weight.

Bar: labels and sub-labels equal the oracle's draw (orc.sample_log_cat / orc.argmax_rows) applied to the GPU's OWN table -- for the
screened D <= 64 kernel (first sweep, and a second sweep with previous labels / bin-sorted order / tail + 16-row screens), the
LDS-staged D > 64 kernel, and the Multinomial u8 / bf16 / f32 kernels."""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_niw import make_problem, gpu_worker, assert_sublabels_bit_exact
import test_gpu_mult as tm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


def _poison(P, rng):
    """Rows the reference's arithmetic turns into NaN / -Inf rows: a NaN feature, +Inf in the LAST feature (the one the tail screen
    reads), -Inf in the first, a point so far out that every quadratic form overflows Float32."""
    n, D = P["n"], P["D"]
    idx = rng.choice(n, 8, replace=False)
    X = P["X"]
    X[idx[0], min(3, D - 1)] = np.nan
    X[idx[1], D - 1] = np.inf
    X[idx[2], 0] = -np.inf
    X[idx[3], :] = 3e19
    X[idx[4], D // 2] = np.nan; X[idx[4], 0] = np.inf
    X[idx[5], D - 1] = np.nan
    X[idx[6], :] = -3e19
    X[idx[7], D - 2 if D > 1 else 0] = np.inf
    return idx


def _check_sweep(wk, seed, epoch, first, n, final=False):
    lab, sub = wk.get_labels()
    tab = wk.debug_loglik()
    u0, u1 = orc.uniforms(seed, epoch, 0, first, n)
    with np.errstate(all="ignore"):
        want = orc.argmax_rows(tab) if final else orc.sample_log_cat(tab, u0)
    bad = np.flatnonzero(want != lab)
    assert len(bad) == 0, (epoch, final, bad[:10], lab[bad[:10]], want[bad[:10]], tab[:, bad[:3]])
    assert_sublabels_bit_exact(wk, lab, sub, u1)
    return lab, sub, tab


@pytest.mark.parametrize("D,n,K", [(64, 8192, 6), (64, 5000, 40), (16, 3000, 5), (128, 2048, 4), (256, 1500, 4)])
def test_niw_degenerate_rows(pkg, D, n, K):
    rng = np.random.default_rng(1000 + D + K)
    P = make_problem(D, n, K, seed=60 + D + K, sep=2.0, sorted_points=True)
    idx = _poison(P, rng)
    P["w"][2] = 0.0                                   # cluster 3: log w = -Inf -> never drawn
    P["lr"][1] = (0.0, 1.0)                           # cluster 2: left sub-cluster weight 0 -> sub-label always 2
    seed, first = 2024, 77
    wk = gpu_worker(pkg, P, seed=seed, first_index=first)
    wk.sweep(1)
    lab, sub, tab = _check_sweep(wk, seed, 1, first, n)
    assert not np.any(lab == 3)
    assert np.all(sub[lab == 2] == 2)
    # rows without a finite entry draw label 1 (all -Inf -> NaN weights -> StatsBase's scan stops at index 1) and sub-label 1
    dead = ~np.isfinite(tab).any(0)
    assert dead[idx[[0, 3, 4, 5, 6]]].all() and np.all(lab[dead] == 1)
    # second sweep: previous labels as reference clusters, bin-sorted visiting order, tail / 16-row screens
    wk.suffstats_packed()
    wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
    wk.sweep(2)
    lab2, sub2, _ = _check_sweep(wk, seed, 2, first, n)
    assert not np.any(lab2 == 3) and np.all(lab2[dead] == 1)
    # argmax phase (`final`): first NaN wins, else first maximum
    wk.suffstats_packed()
    wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
    wk.sweep(3, final=True)
    _check_sweep(wk, seed, 3, first, n, final=True)
    # every weight zero: every row is all -Inf -> label 1 everywhere
    P["w"][:] = 0.0
    wk.set_params_niw(P["mu"], P["invS"], P["logdet"], P["lr"], P["w"])
    wk.sweep(4)
    lab4, _, tab4 = _check_sweep(wk, seed, 4, first, n)
    assert np.all(np.isneginf(tab4) | np.isnan(tab4)) and np.all(lab4 == 1)
    wk.close()


@pytest.mark.parametrize("path", ["u8", "bf16", "f32"])
def test_multinomial_degenerate_rows(pkg, path):
    """log-probabilities of -Inf (log of a Dirichlet component that underflowed to 0): 0 * -Inf = NaN for every point that does not
    use the word, -Inf for those that do (multinomial_dist.jl:13-15 in Float32); zero mixture / sub-cluster weights."""
    D, n, K, trials = 200, 4000, 6, 60
    P = tm.make_problem(D, n, K, trials, seed=303)
    if path == "bf16":
        P["X"][5, 7] = 300.0                          # > 255: bf16-exact, not a byte
    elif path == "f32":
        P["X"][5, 7] += np.float32(0.3)               # not bf16-exact
        P["X"][9, 3] = np.nan
    P["logp"][0, 11] = -np.inf                        # cluster 1, word 11
    P["logp"][3 * 2 + 1, :] = -np.inf                 # left sub-cluster of cluster 3: no support at all
    P["logp"][3 * 4, 0:D:2] = -np.inf                 # cluster 5: half the vocabulary
    P["w"][3] = 0.0
    P["lr"][0] = (1.0, 0.0)
    seed, first = 5, 1234
    wk = tm.worker(pkg, P, seed, first)
    tab = wk.debug_loglik()
    assert np.isnan(tab).any() or np.isneginf(tab).any()
    for epoch, final in ((1, False), (2, False), (3, True)):
        wk.sweep(epoch, final=final)
        lab, sub = wk.get_labels()
        u0, u1 = orc.uniforms(seed, epoch, 0, first, n)
        with np.errstate(all="ignore"):
            want = orc.argmax_rows(tab) if final else orc.sample_log_cat(tab, u0)
        assert np.array_equal(want, lab), (path, epoch, np.flatnonzero(want != lab)[:10])
        if not final:
            assert not np.any(lab == 4)
        tab2 = wk.debug_subloglik()
        i = np.arange(n)
        pair = np.stack([tab2[2 * (lab - 1), i], tab2[2 * (lab - 1) + 1, i]])
        with np.errstate(all="ignore"):
            assert np.array_equal(orc.sample_log_cat(pair, u1), sub), (path, epoch)
        wk.suffstats_packed()
        wk.set_params_mult(P["logp"], P["lr"], P["w"])
    wk.close()
