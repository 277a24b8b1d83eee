#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 passes over any developer script; counters in their own passes
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains mixed in).
# usage: tools/profile_script.sh <tag> <script.py> [args...]      -> gpurun_out/prof_<tag>/summary.md
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$R/$*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $B > "$OUT/trace.log" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 $B > "$OUT/fetch.log" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 $B > "$OUT/write.log" 2> "$OUT/write.err"
python3 "$R/tools/summarize_trace.py" "$OUT/trace" "$OUT/fetch" "$OUT/write" --skip 20 --title "$* under rocprofv3 (kernel trace + FETCH_SIZE / WRITE_SIZE passes; the first 20 dispatches of every kernel dropped)" > "$OUT/summary.md"
cat "$OUT/trace.log" >> "$OUT/summary.md"
cat "$OUT/summary.md"
