#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the rocprofv3 passes behind profiles/rNN_* (counter passes never share a run
# with a trace).  tools/summarize_profiles.py turns the outputs into the committed summaries.
#   bench/        plain bench line, and the same command under --kernel-trace --stats
#   fetch/ write/ PMC FETCH_SIZE / WRITE_SIZE of the headline kernel (own passes)
#   extras/       kernel trace + stats of the bench with its legs and extras
#   bb*/          Beeston-Barlow pass (tools/profile/bb_only.py): kernel trace, FETCH_SIZE
#   unb*/         unbinned pass (tools/profile/unbinned_only.py): kernel trace, FETCH_SIZE
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-legs > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extras --no-legs > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extras --no-legs > /dev/null 2> "$OUT/write.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/extras" -o extras -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline > "$OUT/bench_extras_under_rocprof.json" 2> "$OUT/extras.err"
python3 tools/profile/bb_only.py 20 > "$OUT/bb_plain.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bbkt" -o bb -- python3 tools/profile/bb_only.py 20 > "$OUT/bb_kt.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/bbfetch" -o bb -- python3 tools/profile/bb_only.py 8 > "$OUT/bb_fetch.txt" 2>&1
python3 tools/profile/unbinned_only.py 24 > "$OUT/unb_plain.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/unbkt" -o unb -- python3 tools/profile/unbinned_only.py 24 > "$OUT/unb_kt.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/unbfetch" -o unb -- python3 tools/profile/unbinned_only.py 8 > "$OUT/unb_fetch.txt" 2>&1
# keep what travels back small: the per-dispatch kernel trace is large, the stats and counter files are not
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
ls "$OUT" "$OUT"/*/ | head -60
