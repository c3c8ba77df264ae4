#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel traces and counter passes for the kernels round 5 added or changed -- one counter
# pass per command, never combined with a trace other than the kernel trace -> gpurun_out/prof5/ ;
# tools/summarize_round5.py <round> turns it into profiles/rNN_{toy_points,bb_scan_pmc,grad_batch}.json.
#   toy*     bi_eval_datasets_points: 10^4 toys of C2 x 4 hypotheses in one cell / in random cells (tools/profile/toy_points_trace.py)
#   bb*      k_scan_bb: 256 Beeston-Barlow points in one cell of configs[4]            (tools/profile/bb_scan_only.py)
#   grad*    k_grad_mfma: bi_eval_grad over 131 072 points of C2                        (tools/profile/grad_only.py)
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof5
rm -rf "$OUT"; mkdir -p "$OUT"
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES"
python3 tools/profile/toy_points.py 60 "$OUT/toy_points.json" > "$OUT/toy_points.txt" 2>&1
for which in same random; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/toy_kt_$which" -o t -- python3 tools/profile/toy_points_trace.py 20 $which > "$OUT/toy_kt_$which.txt" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/toy_fetch_$which" -o t -- python3 tools/profile/toy_points_trace.py 6 $which > "$OUT/toy_fetch_$which.txt" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/toy_write_$which" -o t -- python3 tools/profile/toy_points_trace.py 6 $which > "$OUT/toy_write_$which.txt" 2>&1
done
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/toy_sq" -o t -- python3 tools/profile/toy_points_trace.py 6 same > "$OUT/toy_sq.txt" 2>&1
rocprofv3 --pmc $SQ2 --output-format csv -d "$OUT/toy_sq2" -o t -- python3 tools/profile/toy_points_trace.py 6 same > "$OUT/toy_sq2.txt" 2>&1
python3 tools/profile/bb_scan_only.py 3 > "$OUT/bb_plain.txt" 2>&1
python3 tools/profile/bb_scan_only.py 2 256 0 >> "$OUT/bb_plain.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bb_kt" -o t -- python3 tools/profile/bb_scan_only.py 3 > "$OUT/bb_kt.txt" 2>&1
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/bb_sq" -o t -- python3 tools/profile/bb_scan_only.py 2 > "$OUT/bb_sq.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/bb_fetch" -o t -- python3 tools/profile/bb_scan_only.py 2 > "$OUT/bb_fetch.txt" 2>&1
python3 tools/profile/grad_throughput.py "$OUT/grad_throughput.json" > "$OUT/grad_throughput.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/grad_kt" -o t -- python3 tools/profile/grad_only.py 3 > "$OUT/grad_kt.txt" 2>&1
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/grad_sq" -o t -- python3 tools/profile/grad_only.py 2 > "$OUT/grad_sq.txt" 2>&1
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
find "$OUT" -name '*.db' -delete
cat "$OUT/bb_plain.txt" "$OUT/grad_kt.txt" | tail -5
