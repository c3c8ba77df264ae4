#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the toy-MC call of configs[2] (bench.py --config C3: 10^4 device-drawn datasets of C2,
# one parameter point per call) -- a kernel trace for the per-kernel durations and the gaps of a call, and two counter passes
# of the dataset kernel -> gpurun_out/prof_toy/ ; tools/summarize_toy_call.py <round> turns it into profiles/rNN_toy_call.json.
# One counter pass per command, never combined with a trace other than the kernel trace.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_toy
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py --config C3 --no-cpu-baseline --steps 30 > "$OUT/plain.json" 2> "$OUT/plain.err"
rocprofv3 --kernel-trace -d "$OUT/kt" -o c3 -- python3 bench.py --config C3 --no-cpu-baseline --steps 30 > "$OUT/kt.json" 2> "$OUT/kt.err"
python3 tools/profile/c3_call_timeline.py "$(find "$OUT/kt" -name 'c3_results.db' | head -1)" > "$OUT/timeline.txt"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc1" -o pmc -- python3 bench.py --config C3 --no-cpu-baseline --steps 10 > "$OUT/pmc1.json" 2> "$OUT/pmc1.err"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d "$OUT/pmc2" -o pmc -- python3 bench.py --config C3 --no-cpu-baseline --steps 10 > "$OUT/pmc2.json" 2> "$OUT/pmc2.err"
find "$OUT" -name '*.db' -size +16M -delete
cat "$OUT/timeline.txt"
