"""Per-launch durations of the rollout kernel over a bench run, from a rocprofv3 kernel trace (developer tool).
usage: python tools/exp_jitter.py <dir with *_kernel_trace.csv>
Prints the duration distribution, and the series in buckets of 32 consecutive launches (mean / min / max and the gap
to the previous launch), so that a drift with time (clock state), a periodic pattern or isolated outliers can be told
apart."""
import csv, glob, os, sys

def main(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if "k_rollout_random" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    gap = [0.0] + [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
    s = sorted(dur)
    q = lambda p: s[min(len(s) - 1, int(p * len(s)))]
    print(f"{len(dur)} launches: min {s[0]:.1f}  p10 {q(.1):.1f}  p50 {q(.5):.1f}  p90 {q(.9):.1f}  p99 {q(.99):.1f}  max {s[-1]:.1f} us")
    t0 = int(rows[0]["Start_Timestamp"])
    print("bucket  t_ms   mean    min    max   mean_gap_us")
    for b in range(0, len(dur), 32):
        seg, gs = dur[b:b + 32], gap[b:b + 32]
        t = (int(rows[b]["Start_Timestamp"]) - t0) / 1e6
        print(f"{b:6d} {t:6.1f} {sum(seg)/len(seg):6.1f} {min(seg):6.1f} {max(seg):6.1f} {sum(gs)/len(gs):8.1f}")

if __name__ == "__main__":
    main(sys.argv[1])
