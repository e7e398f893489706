"""bench.py's stdout contract: ONE compact JSON line (<= 4 KB, no prose) that carries the driver's keys, `roofline` and
`cpu_baseline`; the full record goes to a side file.  Round 5's line had grown to ~23 KB and the driver could not parse
it -- these tests hold the size and the keys on the CPU, at N = 1 and at N = 8."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fat_record(B, world):
    """A full record shaped like a real run's, with every string blown up to a paragraph and every block padded with the
    kind of nested extras round 5's line carried."""
    import types
    prose = "the quick brown fox explains its methodology at length; " * 40
    a = types.SimpleNamespace(steps=20, warmup=5, rounds=5)
    full = B.headline_record(a, world, [1.6261, 1.6259, 1.6302, 1.6257, 1.6413], 30, 12.345678901, True, True, True, None, False)
    roof = {"bound": "mfma", "achieved": 105.123456789, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.6683123456, "frac_min": 0.66,
            "frac_max": 0.67, "sets": 7, "traffic": 15900000.0, "avg_ms": 0.014543211, "avg_ms_sets": [0.0145] * 7,
            "frac_steady_state": 0.75, "kernel": "conv3x3_mfma_strip_kernel<48, 1> " + prose, "flop_per_layer": 1528823808,
            "flop_per_launch": 764411904, "launches_per_layer": 2, "timing": prose, "traffic_source": "profiles/r06_pmc_conv.csv",
            "traffic_is": prose, "rocprof": {"kernel_stats": "profiles/r06_bench_kernel_stats.csv", "lone_launch_us": 10.25,
                                             "frac_lone_launch": 0.474, "mfma_busy_frac": 0.474, "calls": 32959, "pmc": "x"},
            "in_step": {"what": prose, "forward_chain": {"us": 445.0}}}
    full.update({
        "roofline": roof, "roofline_single_chain": dict(roof), "roofline_c32": dict(roof), "roofline_c64": dict(roof),
        "roofline_wgrad": {"bound": "mfma", "frac": 0.8, "achieved": 125.8, "ms_all_weight_gradients": 0.4875, "traffic": 5.9e8,
                           "kernel": prose, "timing": prose, "in_step": {"what": prose}, "isolated_loop": dict(roof)},
        "step": {"flop_per_step": 183650000000, "achieved": 112.9, "unit": "TFLOP/s", "peak": 157.3, "frac_of_peak": 0.718,
                 "what": prose, "sustained_clock_ghz": 2.32},
        "infer": {"ms_per_batch": 0.525, "value": 1123.4, "unit": "HR Mpixels/s", "flop": 52080000000, "frac_of_peak": 0.63},
        "infer_full_image": {"lr_image": [3, 339, 510], "hr_pixels": 2766240,
                             "LarvaNet": {"ms_per_image": 2.12, "ms_min": 2.11, "ms_max": 2.14, "value": 1304.8, "unit": "HR Mpixels/s",
                                          "flop": 244232000000, "frac_of_peak": 0.73},
                             "LarvaNetV2": {"ms_per_image": 2.36}, "roofline": dict(roof, avg_us=60.4, by_epilogue={"p": {"relu": {}}})},
        "cpu_baseline": {"value": 8.7, "unit": "HR Mpixels/s", "cores": 16, "kind": "port", "ms_per_step": 68.0, "sample": prose,
                         "edsr_train_step": {"ms_per_step": 59.0, "sample": prose}, "forward_only": {"ms_per_batch": 25.0, "sample": prose}},
        "other_widths": {"num_filters_32": {"what": prose}}, "rccl_world1": {"what": prose, "note": prose},
        "dp_schedule_1gpu": {"what": prose}, "sections_s": {"x": 1.0}, "full_record": "gpurun_out/bench_full.json",
    })
    if world > 1:
        full.update({"rccl_ranks": world, "dist_backend": "nccl",
                     "allreduce_exposed_us": {"median": 41.3, "min": 38.0, "max": 95.5, "steps": 100, "overlap": False, "definition": prose},
                     "dp_schedule": {"choice": "flat", "allreduce_isolated_us": 61.2, "bucket_bytes": 3330816, "ranks": world, "rule": prose},
                     "ms_per_step_per_rank": {"min": 1.66, "max": 1.71, "ranks": [1.66 + 0.01 * r for r in range(world)]},
                     "wgrad_schedule": prose})
    return full


@pytest.mark.parametrize("world", [1, 8])
def test_compact_line_is_small_parseable_and_complete(world):
    B = _bench()
    full = _fat_record(B, world)
    assert len(json.dumps(full)) > 20000          # (the input really is round 5's kind of record)
    text = B.compact_line(full)
    assert "\n" not in text and len(text.encode()) <= 4096
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["vs_baseline"] is None and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert set(("workload", "global_batch", "parallelism")) <= set(line["config"]) and "model" not in line["config"]
    assert line["config"]["global_batch"] == 16 * world and line["config"]["parallelism"] == "dp%d" % world
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_min", "frac_max", "avg_ms", "kernel", "flop_per_layer"):
        assert k in line["roofline"], k
    assert line["roofline"]["rocprof"]["lone_launch_us"] == 10.25
    for k in ("value", "unit", "cores", "kind", "sample", "ms_per_step"):
        assert k in line["cpu_baseline"], k
    assert line["step"]["frac_of_peak"] == 0.718
    assert line["infer_full_image"]["ms_per_image"] == 2.12 and line["infer_full_image"]["roofline"]["avg_us"] == 60.4
    # no prose: every string of the line is short
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s) for s in strings(line)) <= B.STR_MAX
    for k in ("roofline_single_chain", "roofline_c32", "other_widths", "rccl_world1", "dp_schedule_1gpu", "sections_s"):
        assert k not in line
    if world > 1:
        assert line["rccl_ranks"] == world and line["dp_schedule"]["choice"] == "flat"
        assert line["allreduce_exposed_us"]["median"] == 41.3 and len(line["ms_per_step_per_rank"]["ranks"]) == world
        assert "wgrad_schedule" not in line


def test_compact_line_refuses_to_grow_past_the_limit():
    B = _bench()
    full = _fat_record(B, 1)
    old = B.COMPACT_MAX_BYTES
    B.COMPACT_MAX_BYTES = 512
    try:
        with pytest.raises(AssertionError):
            B.compact_line(full)
    finally:
        B.COMPACT_MAX_BYTES = old


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [1, 8])
def test_dry_run_emits_the_compact_line_and_the_full_record(n, tmp_path):
    """The whole emit path (launcher -> rank 0 -> stdout) at N = 1 and N = 8 with the N > 1 fields present."""
    env = dict(os.environ, LARVA_BENCH_DRY="1", OMP_NUM_THREADS="1", LARVA_BENCH_FULL=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "20", "--warmup", "5"], env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) <= 4096
    line = json.loads(lines[0])
    for k in CONTRACT:
        assert k in line, k
    assert line["dry_run"] is True and line["n_gpus"] == n and "frac" in line["roofline"]
    if n > 1:
        for k in ("rccl_ranks", "allreduce_exposed_us", "dp_schedule", "ms_per_step_per_rank"):
            assert k in line, k
        assert len(line["ms_per_step_per_rank"]["ranks"]) == n
    full = json.load(open(os.path.join(str(tmp_path), "bench_full.json")))
    assert full["n_gpus"] == n and "ms_per_step_rounds" in full
    assert any(l.startswith("bench_full: {") for l in r.stderr.splitlines())
