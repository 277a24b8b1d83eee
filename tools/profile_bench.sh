#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 passes over bench.py.  Counter passes are
# separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-api-path $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $B > "$OUT/trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 $B --steps 4 > "$OUT/fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 $B --steps 4 > "$OUT/write.json" 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$OUT/sq" -- python3 $B --steps 4 > "$OUT/sq.json" 2> "$OUT/sq.err"
python3 "$R/tools/summarize_prof.py" "$OUT" "$TAG" ${MNK_PROF_KERNEL:-k_rollout_random} > "$OUT/summary.md"
cat "$OUT/summary.md"
