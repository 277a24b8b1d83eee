"""Condenses a rocprofv3 kernel trace (and, optionally, FETCH_SIZE / WRITE_SIZE counter passes of the same command)
into a per-kernel table: EVERY kernel that ran, torch's elementwise / copy kernels included -- used to show what runs
between the step kernels of a rollout loop, and for kernels that have no entry in summarize_api_kernels.py.

usage: python tools/summarize_trace.py <trace dir> [<fetch dir> <write dir>] [--title "..."] [--skip N] > profiles/x.md
--skip N drops the first N dispatches of every kernel name (warm-up).
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reads half of a wide coalesced stream
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): doubled here, as in tools/summarize_prof.py."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def trace_rows(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(files[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def counters(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


def main(argv):
    title, skip, dirs = "rocprofv3 kernel trace", 0, []
    it = iter(argv)
    for a in it:
        if a == "--title":
            title = next(it)
        elif a == "--skip":
            skip = int(next(it))
        else:
            dirs.append(a)
    rows = trace_rows(dirs[0])
    fetch = counters(dirs[1], "FETCH_SIZE") if len(dirs) > 2 else {}
    write = counters(dirs[2], "WRITE_SIZE") if len(dirs) > 2 else {}
    groups = collections.OrderedDict()
    for r in rows:
        name = short(r["Kernel_Name"])
        groups.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"# {title}\n")
    pmc = bool(fetch or write)
    print("| kernel | dispatches | avg us | min us | max us | total ms |" + (" read MB | write MB |" if pmc else ""))
    print("|---|---|---|---|---|---|" + ("---|---|" if pmc else ""))
    total = 0.0
    for name, us in sorted(groups.items(), key=lambda kv: -sum(kv[1][skip:] or kv[1])):
        us = us[skip:] or us
        total += sum(us)
        line = f"| `{name}` | {len(us)} | {sum(us) / len(us):.2f} | {min(us):.2f} | {max(us):.2f} | {sum(us) / 1e3:.3f} |"
        if pmc:
            f, w = fetch.get(name, []), write.get(name, [])
            f, w = (f[skip:] or f), (w[skip:] or w)
            rd = sum(f) / max(len(f), 1) * 1024 * 2 / 1e6
            wr = sum(w) / max(len(w), 1) * 1024 / 1e6
            line += f" {rd:.2f} | {wr:.2f} |"
        print(line)
    print(f"\nall kernels: {total / 1e3:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1:])
