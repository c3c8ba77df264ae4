#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the three rocprofv3 passes behind profiles/rNN_* .
#   pass 1: kernel trace + stats of the bench command          -> gpurun_out/prof/kt
#   pass 4: kernel trace + stats of the same command with its extras -> gpurun_out/prof/extras
#   pass 2, 3: PMC counters FETCH_SIZE, WRITE_SIZE (own passes) -> gpurun_out/prof/fetch, gpurun_out/prof/write
# tools/summarize_profiles.py turns the CSVs into the committed summaries.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2> "$OUT/write.err"
# pass 4: kernel stats of the extras (scan kernels, non-empty-bin form, toy MC, API-level fit)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/extras" -o extras -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline > "$OUT/bench_extras_under_rocprof.json" 2> "$OUT/extras.err"
# keep what travels back small: the per-dispatch kernel trace is large, the stats and counter files are not
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
ls -la "$OUT" "$OUT"/*/ | head -40
