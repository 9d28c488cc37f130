#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference checkout.

Run ONCE in the build container (where /root/reference exists); the outputs are
committed.  Nothing here is needed at test time and nothing from the reference
tree travels to the GPU box.

What is extracted (data only -- inputs and expected outputs, no reference source):

* mnm_golden.npz  -- Multinomial known-answer vector
    X          : test/save_load_test/mnm_data.npy  (1000 x 100 f32, row = point)
    labels/sub : Int64 vectors stored raw inside test/save_load_test/checkpoint_20.jld2
    points_sum : the six stored `points_sum` vectors (cluster k=1,2 x {c,l,r})
    post_alpha : the six stored posterior `alpha` vectors
  pins `create_sufficient_statistics(::multinomial_hyper)` (priors/multinomial_prior.jl:27-32)
  and `calc_posterior` (priors/multinomial_prior.jl:16-21) bit-exactly.

* niw_golden.npz  -- NIW known-answer vector
    X          : examples/save_load_model/2d1ksample.npy (1000 x 2 f64)
    labels/sub : Int64 vectors inside examples/save_load_model/checkpoint__50.jld2
    per (cluster 1..5) x {c,l,r}: points_sum(2), S(2x2), posterior kappa, nu, m(2), psi(2x2)
  pins `create_sufficient_statistics(::niw_hyperparams)` (priors/niw.jl:42-51) and
  `calc_posterior` (priors/niw.jl:20-31); prior kappa=1, m=0, nu=5, psi=I
  (examples/save_load_model/params_2d.jl:24-27).

The .jld2 files are HDF5 containers written by Julia; the arrays of interest are
stored contiguously, so they are read by byte offset (offsets located by the
structural survey, SURVEY.md section 4, and re-verified by this script: every
extracted vector is cross-checked against a recomputation before it is saved).
"""
import os
import sys
import numpy as np

REF = os.environ.get("DPMM_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def mnm():
    X = np.load(f"{REF}/test/save_load_test/mnm_data.npy")
    assert X.shape == (1000, 100) and X.dtype == np.float32
    b = open(f"{REF}/test/save_load_test/checkpoint_20.jld2", "rb").read()
    labels = np.frombuffer(b, dtype="<i8", count=1000, offset=6150).copy()
    sub = np.frombuffer(b, dtype="<i8", count=1000, offset=14208).copy()
    ps_off = [24885, 27052, 29219, 31779, 33946, 36113]
    al_off = [25406, 27573, 29740, 32300, 34467, 36634]
    points_sum = np.stack([np.frombuffer(b, dtype="<f4", count=100, offset=o) for o in ps_off])
    post_alpha = np.stack([np.frombuffer(b, dtype="<f4", count=100, offset=o) for o in al_off])
    # order of rows: (k=1,c),(k=1,l),(k=1,r),(k=2,c),(k=2,l),(k=2,r)
    i = 0
    for k in (1, 2):
        for w in "clr":
            m = labels == k
            if w == "l":
                m &= sub == 1
            if w == "r":
                m &= sub == 2
            assert np.array_equal(X[m].sum(0, dtype=np.float32), points_sum[i]), (k, w)
            assert np.array_equal(points_sum[i] + np.float32(1), post_alpha[i])
            i += 1
    np.savez_compressed(f"{OUT}/mnm_golden.npz", X=X, labels=labels, sub=sub,
                        points_sum=points_sum, post_alpha=post_alpha,
                        prior_alpha=np.ones(100, np.float32))
    print("mnm_golden.npz ok", np.bincount(labels), np.bincount(sub))


def niw():
    X = np.load(f"{REF}/examples/save_load_model/2d1ksample.npy")
    assert X.shape == (1000, 2) and X.dtype == np.float64
    b = open(f"{REF}/examples/save_load_model/checkpoint__50.jld2", "rb").read()
    labels = np.frombuffer(b, dtype="<i8", count=1000, offset=5957).copy()
    sub = np.frombuffer(b, dtype="<i8", count=1000, offset=14015).copy()
    po = [24388, 25624, 26860, 28457, 29693, 30929, 32526, 33762, 34998,
          36595, 37831, 39067, 40664, 41900, 43136]
    f8 = lambda o, c: np.frombuffer(b, dtype="<f8", count=c, offset=o).copy()
    points_sum = np.stack([f8(o, 2) for o in po])
    S = np.stack([f8(o + 89, 4).reshape(2, 2) for o in po])
    kappa = np.array([f8(o + 168, 1)[0] for o in po])
    nu = np.array([f8(o + 184, 1)[0] for o in po])
    m = np.stack([f8(o + 265, 2) for o in po])
    psi = np.stack([f8(o + 354, 4).reshape(2, 2) for o in po])
    counts = []
    i = 0
    for k in range(1, 6):
        for w in "clr":
            msk = labels == k
            if w == "l":
                msk &= sub == 1
            if w == "r":
                msk &= sub == 2
            P = X[msk]
            assert np.allclose(P.sum(0), points_sum[i], rtol=1e-12, atol=1e-11)
            assert np.allclose(P.T @ P, S[i], rtol=1e-10, atol=1e-10)
            assert kappa[i] == 1 + msk.sum() and nu[i] == 5 + msk.sum()
            counts.append(msk.sum())
            i += 1
    np.savez_compressed(f"{OUT}/niw_golden.npz", X=X, labels=labels, sub=sub,
                        points_sum=points_sum, S=S, kappa=kappa, nu=nu, m=m, psi=psi,
                        counts=np.array(counts, np.int64),
                        prior_kappa=1.0, prior_nu=5.0, prior_m=np.zeros(2), prior_psi=np.eye(2))
    print("niw_golden.npz ok", np.bincount(labels), np.bincount(sub))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit(f"reference checkout not found at {REF}")
    mnm()
    niw()
