"""Condenses `rocprofv3 --kernel-trace --stats -- python3 tools/exp_kernels.py` into profiles/r01_api_kernels.md.
usage: python tools/summarize_api_kernels.py <dir with *_kernel_stats.csv>  > profiles/r01_api_kernels.md"""
import csv, glob, os, re, sys

# algorithmic bytes of one launch and what moves; 9x9x5 x 65 536 envs / 19x19x5 x 32 768 envs
N9, N19, C9, C19 = 65536, 32768, 81, 361
BYTES = {
    "k_unpack_records<3, 9, 5>": (N9 * 32 * (28 + 9 * C9 + 13), "32 plies of records -> RolloutBuffer layout"),
    "k_unpack_records<12, 19, 5>": (N19 * 32 * (100 + 9 * C19 + 13), "32 plies of records -> RolloutBuffer layout"),
    "k_observe<3, 9, 5>": (N9 * (32 + 9 * C9), "planes in, f32 obs + bool mask out"),
    "k_observe<12, 19, 5>": (N19 * (96 + 9 * C19), "planes in, f32 obs + bool mask out"),
    "k_selfplay_step_random<3, 9, 5>": (54.5e6, "state r/w, canonical obs + mask, action, reward, terminated, pending, side"),
    "k_selfplay_step_random<12, 19, 5>": (114.0e6, "state r/w, canonical obs + mask, action, reward, terminated, pending, side"),
    "k_selfplay_pre<3, 9, 5>": (53.0e6, "agent ply + opponent's obs + mask"),
    "k_selfplay_pre<12, 19, 5>": (112.2e6, "agent ply + opponent's obs + mask"),
    "k_selfplay_post<3, 9, 5>": (53.3e6, "opponent ply + agent's obs + mask"),
    "k_selfplay_post<12, 19, 5>": (112.4e6, "opponent ply + agent's obs + mask"),
    "k_sample_logits<3>": (27.1e6, "65 536 rows x 81 logits + mask in, action out"),
    "k_sample_logits<12>": (59.4e6, "32 768 rows x 361 logits + mask in, action out"),
}


def main(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(files[-1])))
    print("# rocprofv3 kernel durations of the API-level kernels (MI355X, gfx950)\n")
    print("`rocprofv3 --kernel-trace --stats -- python3 tools/exp_kernels.py` (9x9x5 with 65 536 envs, 19x19x5 with 32 768 envs; "
          "stationary random positions), condensed by `tools/summarize_api_kernels.py`.")
    print("Bytes = algorithmic bytes of that launch (DESIGN.md section 3); GB/s = bytes / average duration.\n")
    print("| kernel | calls | avg us | min us | max us | algorithmic MB | GB/s (avg) | what moves |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = re.sub(r"^void ", "", r["Name"])
        name = re.sub(r"\(.*$", "", name)
        if not name.startswith("k_"):
            continue
        avg, mn, mx = (float(r[k]) / 1e3 for k in ("AverageNs", "MinNs", "MaxNs"))
        b, what = BYTES.get(name, (None, ""))
        mb = f"{b / 1e6:.1f}" if b else ""
        gbs = f"{b / avg / 1e3:.0f}" if b else ""
        print(f"| `{name}` | {r['Calls']} | {avg:.2f} | {mn:.2f} | {mx:.2f} | {mb} | {gbs} | {what} |")


if __name__ == "__main__":
    main(sys.argv[1])
