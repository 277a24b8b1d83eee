"""Condense the rocprofv3 CSVs of tools/profile_bench.sh into the summary kept under profiles/."""
import collections
import csv
import glob
import json
import os
import sys


def counters(d, kernel_substr):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


def phases(root, kern, bench):
    """The dominant kernel's dispatches of the kernel-trace pass, split by the phase of bench.py they belong to (in
    dispatch order: --settle set-up launches at ramping clocks, --warmup launches, the --steps launches of the timed
    region, then the four repetitions): one average over all of them mixes the cold launches with the timed ones and
    reads higher than the run's own ms_per_step."""
    files = sorted(glob.glob(os.path.join(root, "trace", "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not files or not bench:
        return
    rows = [r for r in csv.DictReader(open(files[-1])) if kern in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    settle, warm, steps = bench.get("setup_launches", 0), bench["warmup"], bench["steps"]
    cuts = [("set-up (--settle: clocks ramping)", settle), ("warm-up", warm), ("TIMED REGION", steps), ("repetitions + extras", len(us))]
    print(f"## `{kern}` dispatches by bench phase (kernel-trace pass, in dispatch order)\n")
    print("| phase | dispatches | mean us | median us | min us | max us |\n|---|---|---|---|---|---|")
    lo = 0
    timed_mean = None
    for name, count in cuts:
        part = us[lo:lo + count]
        lo += count
        if not part:
            continue
        srt = sorted(part)
        mean = sum(part) / len(part)
        if name == "TIMED REGION":
            timed_mean = mean
        print(f"| {name} | {len(part)} | {mean:.2f} | {srt[len(srt) // 2]:.2f} | {srt[0]:.2f} | {srt[-1]:.2f} |")
    if timed_mean is not None:
        step_us = bench["ms_per_step"] * 1e3
        alg = bench["roofline"]["alg_bytes_per_launch"]
        print(f"\ntimed region under the profiler: {timed_mean:.2f} us per dispatch = {alg / timed_mean / 1e3:.0f} GB/s = "
              f"{alg / timed_mean / 1e3 / 8000:.3f} of 8 TB/s; the same run's ms_per_step (wall clock, barrier to barrier): "
              f"{step_us:.2f} us ({'>=' if step_us >= timed_mean else '<'} the kernel's own time); in-bench HIP events: "
              f"{bench['roofline']['avg_launch_us']:.2f} us\n")


def main():
    root, tag = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else "k_rollout_random"
    print(f"# rocprofv3 summary `{tag}` (MI355X, gfx950)\n")
    try:
        bench = json.loads(open(os.path.join(root, "trace.json")).read().strip().splitlines()[-1])
        print("bench line under `--kernel-trace --stats`:\n\n```json\n" + json.dumps(bench) + "\n```\n")
    except Exception as e:  # noqa: BLE001
        bench = None
        print(f"(no bench line: {e})\n")
    for f in glob.glob(os.path.join(root, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        print("## kernel stats (`rocprofv3 --kernel-trace --stats`)\n")
        print("| kernel | calls | total ns | average ns | % | min ns | max ns |")
        print("|---|---|---|---|---|---|---|")
        for r in list(csv.DictReader(open(f)))[:8]:
            name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:90]
            print(f"| `{name}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | "
                  f"{float(r['Percentage']):.2f} | {r['MinNs']} | {r['MaxNs']} |")
        print()
    phases(root, kern, bench)
    print(f"## counters of `{kern}` (mean per dispatch; separate `--pmc` passes, 4 timed launches each)\n")
    print("| counter | mean | dispatches |\n|---|---|---|")
    allc = {}
    for sub in ("fetch", "write", "sq"):
        c = counters(os.path.join(root, sub), kern)
        allc.update(c)
        for k, (v, n) in sorted(c.items()):
            print(f"| {k} | {v:.1f} | {n} |")
    print()
    if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly half of a wide coalesced
        # stream (MI355X_MICROARCH.md, HBM): double it.
        rd = allc["FETCH_SIZE"][0] * 1024 * 2
        wr = allc["WRITE_SIZE"][0] * 1024
        print(f"HBM traffic per dispatch: read {rd/1e6:.2f} MB (FETCH_SIZE x 1024 x 2, gfx950 correction) + "
              f"write {wr/1e6:.2f} MB (WRITE_SIZE x 1024) = {(rd+wr)/1e6:.2f} MB")
        if bench:
            cfg = bench["config"]
            n, chunk = cfg["envs_per_gpu"], cfg["chunk"]
            per = bench["roofline"]["alg_bytes_per_env_step"]
            print(f"\nalgorithmic bytes per dispatch: {n} envs x {chunk} plies x {per:.3f} B = {n*chunk*per/1e6:.2f} MB")
    if "SQ_INSTS_VALU" in allc and "SQ_WAVES" in allc:
        waves = allc["SQ_WAVES"][0]
        print(f"\nper wave: {allc['SQ_INSTS_VALU'][0]/waves:.0f} VALU + {allc['SQ_INSTS_SALU'][0]/waves:.0f} SALU instructions; "
              f"SQ_WAVE_CYCLES x4 / waves = {allc['SQ_WAVE_CYCLES'][0]*4/waves:.0f} cycles; "
              f"GRBM_GUI_ACTIVE / 8 XCDs = {allc['GRBM_GUI_ACTIVE'][0]/8:.0f} cycles")


if __name__ == "__main__":
    main()
