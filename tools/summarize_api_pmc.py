"""HBM traffic of the API-level kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of
tools/exp_kernels.py), next to their algorithmic bytes.

usage: python tools/summarize_api_pmc.py <fetch dir> <write dir>  > profiles/r02_api_kernels_pmc.md
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half of a wide coalesced stream
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): doubled here, as in tools/summarize_prof.py."""
import collections, csv, glob, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_api_kernels import BYTES, N9, S9, W9, C9, b_step, short  # noqa: E402


def counters(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


def main(fd, wd):
    fetch, write = counters(fd, "FETCH_SIZE"), counters(wd, "WRITE_SIZE")
    print("# HBM traffic of the API-level kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n")
    print("`rocprofv3 --pmc <counter> --kernel-trace -- python3 tools/exp_kernels.py 9x9x5 65536`; mean per dispatch over all "
          "dispatches of that kernel name; read = FETCH_SIZE x 1024 x 2 (gfx950 correction), write = WRITE_SIZE x 1024.  "
          "State (2.4 MB) and small vectors stay in L2 / MALL between launches, so reads can be below the algorithmic figure; "
          "what the comparison shows is that no kernel moves more than its algorithmic bytes -- nothing is re-read or written twice.  "
          "`k_step_full` and `k_observe` are called with several output sets, so their means mix those.\n")
    print("| kernel | dispatches | read MB | write MB | total MB | algorithmic MB (largest form) |")
    print("|---|---|---|---|---|---|")
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("k_"):
            continue
        rd = sum(fetch.get(name, [0])) / max(len(fetch.get(name, [0])), 1) * 1024 * 2 / 1e6
        wr = sum(write.get(name, [0])) / max(len(write.get(name, [0])), 1) * 1024 / 1e6
        alg = BYTES.get(name, (None, ""))[0]
        if name.startswith("k_step_full<3, 9, 5, false"):
            alg = N9 * b_step(S9, W9, C9, True, True)
        if name.startswith("k_observe<3, 9, 5>"):
            alg = N9 * (32 + 9 * C9)
        print(f"| `{name}` | {len(write.get(name, []))} | {rd:.2f} | {wr:.2f} | {rd + wr:.2f} | {alg / 1e6:.1f} |" if alg else
              f"| `{name}` | {len(write.get(name, []))} | {rd:.2f} | {wr:.2f} | {rd + wr:.2f} | |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
