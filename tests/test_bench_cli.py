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


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _fat_record():
    """A record at least as fat as round 5's 20 KB line: every block the default run adds, long texts, a 260-entry K history."""
    legs = {name: {"n": 10 ** 7, "ms_per_step": 2.5, "sweep_kernel_ms": 1.9, "workload": "w" * 300, "host_ms_per_step": {f"t{i}": 0.001 * i for i in range(12)},
                   "roofline": {f"k{i}": 1.2345678901 * i for i in range(25)}} for name in
            ("overlap_var4", "overlap_var1", "inseparable", "k256", "c2", "c3_shard", "shard8_projection", "c4", "c5_shard")}
    roof = {"kernel": "niw_lean_kernel+niw_sweep_direct_kernel+niw_sub_kernel", "bound": "hbm", "achieved": 2431.123456789, "peak": 8000.0, "unit": "GB/s",
            "frac": 0.30389, "hbm_frac": 0.30389, "traffic": 2.83e9, "traffic_source": "s" * 400, "traffic_is_current": True, "traffic_frac": 0.33,
            "algorithmic_bytes_per_launch": 2.6e9, "avg_launch_ms": 1.0721189320087432, "lean_kernel_ms": 1.0, "mfma_pipe_frac": 0.3403521610697151,
            "frac_definition": "d" * 1500, "launches_ms": {"a" * 90: 1.0, "note": "n" * 300}, "dense_f32_frac": 0.733, "dense_launch_ms": 14.58,
            "pruning_factor": 1087.6, "stats_kernels_ms": 0.456, "pmc_frac": 0.35, "mfma_busy": 0.37}
    return {"metric": "Gibbs iterations/sec, N=10M D=64 NIW", "value": 614.4183087120751, "unit": "iterations/s", "n_gpus": 1, "steps": 20, "warmup": 5, "settle": 100,
            "ms_per_step": 1.6275556665884021, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "NIW D=64 N=10000000 ... MixtureVar 100 ..." + "x" * 300, "points_per_gpu": 10 ** 7, "parallelism": "p" * 200, "master": "m" * 500,
                       "worker_options_overridden": None,
                       "also_measured": {"host_master_it_per_s": 551.5, "host_master_ratio_to_headline": 0.8976, "host_cpu_model": "AMD EPYC 9575F 64-Core Processor",
                                         "growth": {"it_per_s_whole_run": 537.19, "K_final": 32, "K_true": 32, "nmi": 1.0, "moving_labels_it_per_s": 602.9},
                                         "shard8": {"shard_ms_per_step": 0.3264, "shard_ms_per_step_one_collective_form": 0.3214, "assumed_allreduce_ms_per_step": 0.04,
                                                    "projected_speedup_1_to_8": 4.5, "speedup_without_collectives": 4.99},
                                         **{f"{n}_ms_per_step": 2.49220949990558 for n in ("overlap_var4", "overlap_var1", "k256", "c2", "c4", "c5_shard")},
                                         "inseparable_sweep_kernel_ms": 22.2, "inseparable_full_evals_per_tile": 32.0, "c4_traffic_frac": None}},
            "roofline": roof, "comm": {f"c{i}": i for i in range(12)}, "blocks": {"it_per_s": [600.0] * 5},
            "host_ms_per_step": {f"t{i}": 0.001 * i for i in range(14)},
            "growth": {"K_history": list(range(260)), "it_per_s_whole_run": 537.0}, "host_master": {"note": "h" * 400, "it_per_s": 551.5},
            "legs": legs,
            "cpu_baseline": {"value": 0.2846, "unit": "iterations/s", "cores": 16, "kind": "port", "parallelism": "multiprocess",
                             "cpu_model": "AMD EPYC 9575F 64-Core Processor", "julia_found": False, "sample": "s" * 420, "sample_fraction": 1.0}}


def test_the_contract_line_is_compact_and_parses(tmp_path, monkeypatch, capsys):
    """round 5's ONE stdout line had grown to 20 KB and the driver recorded `parsed: null`.  The line the driver reads -- the LAST
    line of stdout -- stays under 4 KB whatever the run measured, parses, and carries the contract fields + roofline + cpu_baseline; the
    rest goes to bench_details.json and stderr."""
    import json
    bench = _bench_module()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec = _fat_record()
    assert len(json.dumps(rec)) > 12000
    bench.emit(rec)
    cap = capsys.readouterr()
    last = cap.out.splitlines()[-1]
    assert len(cap.out.splitlines()) == 1 and len(last) < 4096 and len(last.encode()) < 8192
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == rec["value"] and line["ms_per_step"] == rec["ms_per_step"]
    assert "workload" in line["config"] and "MixtureVar" in line["config"]["workload"] and "model" not in line["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "hbm_frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"):
        assert k in line["roofline"], k
    assert line["roofline"]["bound"] in ("hbm", "mfma") and abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "legs" not in line and "growth" not in line and "K_history" not in last
    full = json.loads((tmp_path / "bench_details.json").read_text())
    assert full["growth"]["K_history"] == list(range(260)) and "legs" in full
    assert "bench.py details: " in cap.err


def test_traffic_is_reported_whenever_a_counter_summary_exists(tmp_path, monkeypatch):
    """`roofline.traffic` silently became null when the committed summary carried another source tag.  With
    allow_stale the figure is reported and the text says which sources it was collected on; a matching tag reports it as current."""
    import json
    bench = _bench_module()
    tag = bench.kernel_source_tag()
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_tag", lambda: tag)
    rec = {"dpmm::niw_lean_kernel": {"FETCH_SIZE": {"median": 1000.0}, "WRITE_SIZE": {"median": 10.0}},
           "dpmm::niw_sweep_direct_kernel<4, 4, 2, true, false, true, true>": {"FETCH_SIZE": {"median": 100.0}, "WRITE_SIZE": {"median": 1.0}}}
    f = tmp_path / "profiles" / "latest_bench_pmc_summary.json"
    f.write_text(json.dumps(dict(rec, _meta={"kernel_source_tag": tag, "collected": "now"})))
    t, src = bench.pmc_traffic("bench", bench.SWEEP_KERNELS_64, allow_stale=True)
    assert t == (2 * 1100.0 + 11.0) * 1024.0 and "stale" not in src and bench.pmc_is_current("bench")
    f.write_text(json.dumps(dict(rec, _meta={"kernel_source_tag": "f" * 16})))
    t, src = bench.pmc_traffic("bench", bench.SWEEP_KERNELS_64, allow_stale=True)
    assert t == (2 * 1100.0 + 11.0) * 1024.0 and "stale" in src and not bench.pmc_is_current("bench")


def test_the_committed_counter_summaries_parse_for_the_kernels_the_bench_quotes():
    """Whatever tag the committed summaries carry, the kernels bench.py looks for are in them (a renamed kernel would make traffic null for ever)."""
    bench = _bench_module()
    t, _ = bench.pmc_traffic("bench", bench.SWEEP_KERNELS_64, allow_stale=True)
    assert t is not None and t > 1e9
    t, _ = bench.pmc_traffic("mult", "mult_sweep", allow_stale=True)
    assert t is not None and t > 1e8
