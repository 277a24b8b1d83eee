"""Condenses `rocprofv3 --kernel-trace --stats -- python3 tools/exp_kernels.py` into profiles/rNN_api_kernels.md.

usage: python tools/summarize_api_kernels.py <dir with *_kernel_trace.csv>  > profiles/r02_api_kernels.md

Works from the per-dispatch trace (not the per-name stats) so that one kernel name measured in several
configurations gets one row per configuration: tools/exp_kernels.py times `k_step_full` with three output sets
(none / legal mask / legal mask + f32 observation), 55 consecutive dispatches each (5 warm-up + 50 timed).
Bytes = algorithmic bytes of one launch; GB/s = bytes / average duration."""
import collections
import csv
import glob
import os
import re
import sys

N9, N19, C9, C19 = 65536, 32768, 81, 361
S9, S19, W9, W19 = 36, 100, 2, 6


def b_step(s, w, c, mask, obs):
    """read action i64 + state; write mover plane + meta, reward f32, done u8 (+ mask bytes, + f32 observation)"""
    return 8 + s + 8 * w + 4 + 5 + (c if mask else 0) + (8 * c if obs else 0)


STEP_SETS = [("no outputs", False, False), ("+ legal mask  [B_step of SURVEY 8d]", True, False),
             ("+ legal mask + f32 observation", True, True)]
BYTES = {
    "k_unpack_records<3, 9, 5>": (N9 * 32 * (28 + 9 * C9 + 13), "32 plies of records -> RolloutBuffer layout"),
    "k_unpack_records<12, 19, 5>": (N19 * 32 * (100 + 9 * C19 + 13), "32 plies of records -> RolloutBuffer layout"),
    "k_observe<3, 9, 5>": (None, "planes in; f32 obs + bool mask out (mixed with mask-only calls)"),
    "k_observe<12, 19, 5>": (None, "planes in; f32 obs + bool mask out (mixed with mask-only calls)"),
    "k_selfplay_step_random<3, 9, 5>": (N9 * (2 * (S9 + 8 * W9 + 4) + 9 * C9 + 8 + 4 + 1 + 2 + 16),
                                        "state r/w, canonical obs + mask, action, reward, terminated, pending, side"),
    "k_selfplay_step_random<12, 19, 5>": (N19 * (2 * (S19 + 8 * W19 + 4) + 9 * C19 + 8 + 4 + 1 + 2 + 16),
                                          "state r/w, canonical obs + mask, action, reward, terminated, pending, side"),
    "k_selfplay_pre<3, 9, 5>": (N9 * (S9 + 8 * W9 + 4 + 9 * C9 + 8 + 1 + 8 + 4 + 1 + 1), "agent ply + opponent's obs + mask"),
    "k_selfplay_pre<12, 19, 5>": (N19 * (S19 + 8 * W19 + 4 + 9 * C19 + 8 + 1 + 8 + 4 + 1 + 1), "agent ply + opponent's obs + mask"),
    "k_selfplay_post<3, 9, 5>": (N9 * (S9 + 8 * W9 + 4 + 9 * C9 + 8 + 1 + 8 + 8 + 2 + 1), "opponent ply + agent's obs + mask"),
    "k_selfplay_post<12, 19, 5>": (N19 * (S19 + 8 * W19 + 4 + 9 * C19 + 8 + 1 + 8 + 8 + 2 + 1), "opponent ply + agent's obs + mask"),
    "k_sample_logits<4, 21, true, float>": (N9 * (5 * C9 + 12), "65 536 rows x 81 f32 logits + mask in, action + log-prob out"),
    "k_sample_logits<4, 21, true, unsigned short>": (N9 * (3 * C9 + 12), "65 536 rows x 81 bf16 logits + mask in"),
    "k_sample_logits<4, 21, true, void>": (N9 * (C9 + 12), "65 536 rows x 81 mask bytes in (uniform draw, RandomPolicy)"),
    "k_sample_logits<16, 23, true, float>": (N19 * (5 * C19 + 12), "32 768 rows x 361 f32 logits + mask in, action + log-prob out"),
    "k_sample_logits<16, 23, true, unsigned short>": (N19 * (3 * C19 + 12), "32 768 rows x 361 bf16 logits + mask in"),
    "k_sample_logits<16, 23, true, void>": (N19 * (C19 + 12), "32 768 rows x 361 mask bytes in (uniform draw)"),
    "k_sample_legal<3, 9, 5>": (N9 * (S9 - 4 + 8), "planes in, action i64 out"),
    "k_sample_legal<12, 19, 5>": (N19 * (S19 - 4 + 8), "planes in, action i64 out"),
    "k_gae": (None, "rewards, values, dones in; advantages, returns out (17 B per step and env; two shapes mixed)"),
}


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").replace(", NoDraw>", ">")
    return re.sub(r"\(.*$", "", name)


def main(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(files[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups = collections.OrderedDict()
    seen = collections.Counter()
    for r in rows:
        name = short(r["Kernel_Name"])
        if not name.startswith("k_"):
            continue
        key, what, nbytes = name, None, None
        if name.startswith("k_step_full<") and name.endswith("true>"):
            big = "<12, 19, 5" in name
            nbytes = (N19 if big else N9) * (b_step(S19 if big else S9, W19 if big else W9, C19 if big else C9, True, False) - 8)
            key, what = f"{name} (mnk_step_random) + legal mask", "state in; draw, mover plane, meta, reward, done, mask out (no action read)"
        elif name.startswith("k_step_full<"):
            block = seen[name] // 55  # exp_kernels.py: 55 consecutive dispatches per output set, in STEP_SETS order
            seen[name] += 1
            if block < len(STEP_SETS):
                label, mask, obs = STEP_SETS[block]
                big = "<12, 19, 5" in name
                nbytes = (N19 if big else N9) * b_step(S19 if big else S9, W19 if big else W9, C19 if big else C9, mask, obs)
                key, what = f"{name} {label}", "action + state in; mover plane, meta, reward, done" + \
                    (", mask" if mask else "") + (", obs" if obs else "") + " out"
            else:
                key, what = f"{name} (other callers)", ""
        elif name in BYTES:
            nbytes, what = BYTES[name]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        g = groups.setdefault(key, {"us": [], "bytes": nbytes, "what": what or ""})
        g["us"].append(us)
    print("# rocprofv3 kernel durations of the API-level kernels (MI355X, gfx950)\n")
    print("`rocprofv3 --kernel-trace --stats -- python3 tools/exp_kernels.py` (9x9x5 with 65 536 envs, 19x19x5 with "
          "32 768 envs; stationary random positions), condensed by `tools/summarize_api_kernels.py` from the per-dispatch "
          "trace.  Bytes = algorithmic bytes of that launch (SURVEY.md section 8d / DESIGN.md section 3); GB/s = bytes / "
          "average duration; frac = GB/s / 8000.\n")
    print("| kernel | calls | avg us | min us | max us | algorithmic MB | GB/s (avg) | frac of 8 TB/s | what moves |")
    print("|---|---|---|---|---|---|---|---|---|")
    for key, g in groups.items():
        us = g["us"]
        avg = sum(us) / len(us)
        b = g["bytes"]
        mb = f"{b / 1e6:.1f}" if b else ""
        gbs = f"{b / avg / 1e3:.0f}" if b else ""
        frac = f"{b / avg / 1e3 / 8000:.2f}" if b else ""
        print(f"| `{key}` | {len(us)} | {avg:.2f} | {min(us):.2f} | {max(us):.2f} | {mb} | {gbs} | {frac} | {g['what']} |")


if __name__ == "__main__":
    main(sys.argv[1])
