#!/bin/bash
# Run ON THE GPU BOX (through gpurun): counter evidence for the 131 072-point scan of C2 over DENSE data (an event in
# nearly every bin), count-sorted rows (k_scan_sorted<8,false>, round 4) against rows in bin order (k_scan_mfma<2,8,false,0>,
# round 2's path) -> gpurun_out/prof_scan_dense/ ; tools/summarize_scan_dense.py turns it into profiles/rNN_scan_dense_data_*.
# One counter pass per command, never combined with a trace.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_scan_dense
rm -rf "$OUT"; mkdir -p "$OUT"
python3 tools/profile/scan_only.py 2 dense > "$OUT/plain.txt" 2>&1
python3 tools/profile/scan_only.py 2 dense binorder >> "$OUT/plain.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 tools/profile/scan_only.py 2 dense > "$OUT/kt.txt" 2>&1
for order in sorted binorder; do
  extra=""; [ "$order" = binorder ] && extra="binorder"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$order" -o pmc -- python3 tools/profile/scan_only.py 2 dense $extra > "$OUT/pmc_$order.txt" 2>&1
done
# the default path of a scan (non-empty-bin form over the compacted rows, ordered by count since round 3): 10^6 points
python3 tools/profile/sparse_scan_only.py 3 >> "$OUT/plain.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kts" -o kt -- python3 tools/profile/sparse_scan_only.py 3 > "$OUT/kts.txt" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sparse" -o pmc -- python3 tools/profile/sparse_scan_only.py 2 > "$OUT/pmc_sparse.txt" 2>&1
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
cat "$OUT/plain.txt"
