"""CPU: the committed bench lines (profiles/r02_bench_*.json, printed by bench.py on the MI355X) carry every field
the driver's contract names, with consistent arithmetic -- a regression guard for bench.py's JSON."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_one_gpu_line_has_the_contract_fields():
    d = _line("r02_bench_line.json")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "env-steps/s" and d["data"] == "synthetic" and d["dtype"] == "u64"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "9x9x5" in d["metric"] and "65536" in d["metric"]
    # value = envs x plies per launch x launches / time
    envs, chunk = d["config"]["envs_per_gpu"], d["config"]["chunk"]
    assert d["value"] == pytest.approx(envs * chunk / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.3 < r["frac"] < 1.0
    assert r["achieved"] == pytest.approx(r["alg_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9, rel=1e-6)
    # measured HBM traffic (PMC) within 1 % of the algorithmic bytes: nothing is re-read
    assert r["traffic"] == pytest.approx(r["alg_bytes_per_launch"], rel=0.01)
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "cpu_model", "host_logical_cpus", "usable_cpus"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e4
    assert d["value"] / c["value"] > 1e4  # the GPU path against the CPU restatement on the same box


def test_two_rank_rehearsal_line_describes_the_exchange():
    d = _line("r02_bench_gloo2_rehearsal_line.json")
    assert d["n_gpus"] == 2 and d["value_without_exchange"] > d["value"]
    x = d["exchange"]
    for key in ("what", "transport", "bytes_per_rank_per_chunk", "bytes_per_env_step", "allgather_ms",
                "recv_GBps_per_rank", "per_link_GBps", "compute_ms_per_chunk", "exposed_ms_per_chunk", "overlap_fraction"):
        assert key in x, key
    assert x["bytes_per_env_step"] == pytest.approx(1.140625)  # 36 B state per 256 plies + 1 B per action + meta
