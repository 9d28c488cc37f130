"""bench.py's launch contract (CPU, no GPU call): `--gpus N` never degrades to a smaller run silently (VERDICT r2 / ADVICE r2)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)


def test_more_gpus_than_visible_is_an_error_not_a_smaller_run():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has the GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout) and "{" not in r.stdout


def test_world_size_must_equal_gpus():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4 but --gpus 1" in (r.stderr + r.stdout)


def test_counter_traffic_is_quoted_only_for_the_kernel_sources_it_was_collected_on(tmp_path, monkeypatch):
    """roofline.traffic comes from profiles/latest_*_pmc_summary.json and carries the hash of the kernel sources: a stale file yields
    None and says why."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    tag = bench.kernel_source_tag()
    assert len(tag) == 16
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_tag", lambda: tag)
    rec = {"dpmm::niw_sweep_direct_kernel<4, 4, 2, true>": {"FETCH_SIZE": {"median": 1000.0, "n": 3}, "WRITE_SIZE": {"median": 10.0, "n": 3}}}
    (prof / "latest_bench_pmc_summary.json").write_text(json.dumps(dict(rec, _meta={"kernel_source_tag": tag, "collected": "now"})))
    t, src = bench.pmc_traffic("bench", "niw_sweep_direct_kernel")
    assert t == (2 * 1000.0 + 10.0) * 1024.0 and "latest_bench_pmc_summary.json" in src
    (prof / "latest_bench_pmc_summary.json").write_text(json.dumps(dict(rec, _meta={"kernel_source_tag": "0" * 16})))
    t, src = bench.pmc_traffic("bench", "niw_sweep_direct_kernel")
    assert t is None and "stale" in src
    t, src = bench.pmc_traffic("mult", "mult_sweep")
    assert t is None and src is None
