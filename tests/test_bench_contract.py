"""CPU: the committed bench lines (profiles/r0N_bench_*.json, printed by bench.py on the MI355X) carry every field
the driver's contract names, with consistent arithmetic -- a regression guard for bench.py's JSON."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["r02_bench_line.json", "r03_bench_line.json"])
def test_one_gpu_line_has_the_contract_fields(name):
    d = _line(name)
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


def test_round3_line_names_the_device_the_variant_and_the_ceiling():
    """Round 3: GPU identity, the measured write ceiling next to the roofline, the one-launch
    API path with its launch floor, the kernel variant the multi-GPU runs use, the env side of config 3, and the
    exchange mode / kernel variant / transport at the top level of the line."""
    d = _line("r03_bench_line.json")
    assert d["gather"] == "none" and d["kernel_variant"].startswith("records only") and d["transport"] is None
    g = d["gpu"]
    assert g["compute_units"] == 256 and g["arch"].startswith("gfx950") and g["rocminfo"]["name"] == "gfx950"
    r = d["roofline"]
    assert 4000 < r["measured_write_ceiling_GBps"] < 8000
    assert r["frac_of_measured_write_ceiling"] == pytest.approx(r["achieved"] / r["measured_write_ceiling_GBps"])
    assert r["achieved"] <= 1.02 * r["measured_write_ceiling_GBps"]  # the kernel cannot beat a store-only kernel by much
    one = d["api_path_one_launch"]
    assert d["api_path_graphed_one_launch_env_steps_per_s"] == pytest.approx(65536 / (one["us_per_ply"] * 1e-6), rel=1e-6)
    assert one["launch_floor_us"] < one["us_per_ply"] and one["alg_bytes_per_env_step"] == 142
    assert d["api_path_graphed_one_launch_env_steps_per_s"] > d["api_path_graphed_fused_reset_env_steps_per_s"] > \
        d["api_path_graphed_env_steps_per_s"]
    w = d["with_action_log"]
    assert w["log_bytes_per_env_step"] == 0.875 and "7-bit" in w["kernel_variant"] and 0.85 * d["value"] < w["value"] <= 1.02 * d["value"]
    sp = d["selfplay"]
    assert sp["eager_sink_add_copied_bytes_per_agent_step"] < 30 and sp["eager_copying_add_copied_bytes_per_agent_step"] == 750.0
    assert sp["graphed_agent_steps_per_s"] > sp["eager_sink_agent_steps_per_s"]
    assert sp["alg_bytes_per_agent_step"] <= 1.1 * sp["survey_B_agent_bytes"]


def test_round3_rehearsal_lines_cover_every_exchange_mode():
    with open(os.path.join(ROOT, "profiles", "r03_bench_gloo2_rehearsal_lines.json")) as f:
        lines = [json.loads(x) for x in f.read().strip().splitlines()]
    assert len(lines) == 4
    per_step = [x["exchange"]["bytes_per_env_step"] for x in lines]
    # keyframe every 8th chunk, every chunk, never (state once up front), packed records
    assert per_step[0] == pytest.approx(0.875 + (36 + 4) / (256 * 8), abs=2e-3)   # 36 B state (+ padding) per keyframe
    assert per_step[1] == pytest.approx(0.875 + (36 + 4) / 256, abs=2e-2)
    assert per_step[2] == 0.875 and lines[2]["exchange"]["start_state_bytes_per_rank"] == 16384 * 36
    assert per_step[3] == 28.0
    for x in lines:
        assert x["n_gpus"] == 2 and x["gather"] == x["config"]["gather"] and x["transport"] == x["exchange"]["transport"]
        assert ("7-bit" in x["kernel_variant"]) == (x["gather"] == "actions")


def test_rehearsed_exchange_lines_time_both_forms():
    """Round 3: the multi-GPU code path rehearsed with a one-rank RCCL communicator (`bench.py --rehearse-exchange`): the C
    ABI's transport, the exchange algorithm in the line, both exchange forms timed alone."""
    with open(os.path.join(ROOT, "profiles", "r03_bench_rehearse_exchange_lines.json")) as f:
        lines = [json.loads(ln) for ln in f.read().splitlines() if ln.startswith("{")]
    assert len(lines) == 4
    for d in lines:
        assert d["n_gpus"] == 1 and "rehearsal" in d and d["cpu_baseline"] is None
        x = d["exchange"]
        assert "C ABI, RCCL" in x["transport"] and d["transport"] == x["transport"]
        assert ("direct" in x["transport"]) == x["algorithm"].startswith("one grouped ncclSend")
        assert x["alone"]["ncclAllGather_ms"] > 0 and x["alone"]["direct_sendrecv_ms"] > 0
    assert sorted(d["gather"] for d in lines) == ["actions", "actions", "actions", "records"]
    assert {round(d["exchange"]["bytes_per_env_step"], 3) for d in lines} == {0.893, 1.016, 28.0}


def test_round4_line_uses_one_clock_and_stays_under_the_peak():
    """Round 4: the roofline fraction is given on BOTH clocks of the line, named -- `frac` =
    frac_kernel_events (HIP events on the kernel's stream, the contract's definition), frac_wall from `ms_per_step`, the
    interval `value` is computed from -- the wall-clock one cannot exceed the kernel's, and neither exceeds the peak."""
    d = _line("r04_bench_line.json")
    r = d["roofline"]
    assert r["frac"] == r["frac_kernel_events"] == pytest.approx(r["achieved"] / r["peak"])
    wall = r["alg_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e9
    assert wall <= r["peak"] and r["achieved_wall"] == pytest.approx(wall, rel=1e-9)
    assert r["frac_wall"] == pytest.approx(wall / r["peak"]) and r["frac_wall"] <= r["frac_kernel_events"]
    assert d["value"] == pytest.approx(d["config"]["envs_per_gpu"] * d["config"]["chunk"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    sp = d["selfplay"]
    assert sp["launches_per_agent_step"] == 1
    tc = sp["train_cadence_384_envs"]
    assert tc["one_graph_inplace_opponent_swap_us"] < tc["eager_reference_loop_us"] < tc["recapture_per_rollout_us"] * 1.5
    assert tc["speedup_swap_vs_eager"] > 1.5 and tc["env_side_launches_per_agent_step"] == 2


def test_round4_profile_summary_splits_the_phases():
    """the tracked summary's timed-region dispatch time is at most the same run's ms_per_step"""
    import re

    text = open(os.path.join(ROOT, "profiles", "r04_rollout_9x9x5.md")).read()
    m = re.search(r"\| TIMED REGION \| (\d+) \| ([\d.]+) \|", text)
    assert m and int(m.group(1)) == 64
    line = json.loads(re.search(r"```json\n(.*?)\n```", text, flags=re.S).group(1))
    assert float(m.group(2)) <= line["ms_per_step"] * 1e3


def test_round4_rehearsal_lines_carry_the_exchange_group_size():
    """Round 4: `--exchange-every J` in the rehearsed multi-GPU code path (one RCCL rank) and with two gloo ranks: the line
    says how many chunks one all-gather carries and how many bytes that is; bytes per env-step do not change."""
    with open(os.path.join(ROOT, "profiles", "r04_bench_rehearse_exchange_lines.json")) as f:
        lines = [json.loads(ln) for ln in f.read().splitlines() if ln.startswith("{")]
    assert [d["exchange"]["chunks_per_exchange"] for d in lines] == [1, 2, 1, 2]
    for d in lines:
        x = d["exchange"]
        assert x["bytes_per_rank_per_exchange"] == pytest.approx(x["bytes_per_rank_per_chunk"] * x["chunks_per_exchange"])
        assert "C ABI, RCCL" in x["transport"] and x["alone"]["ncclAllGather_ms"] > 0 and "exchange_hung" not in d
        assert d["roofline"]["frac_wall"] <= 1.02 * d["roofline"]["frac_kernel_events"]
    assert {round(d["exchange"]["bytes_per_env_step"], 3) for d in lines} == {0.893, 1.016, 28.0}
    d = _line("r04_bench_gloo2_exchange_every_line.json")
    assert d["n_gpus"] == 2 and d["exchange"]["chunks_per_exchange"] == 2 and d["exchange"]["bytes_per_env_step"] == pytest.approx(0.893, abs=1e-3)
