#!/bin/bash
# Run ON THE GPU BOX (through gpurun): counter evidence for the matrix-core scan kernel (k_scan_mfma) on the 131 072-point
# dense scan of C2 -> gpurun_out/prof_scan/ ; tools/summarize_profiles.py turns it into profiles/rNN_scan_*.
#   overlap.txt   tools/micro/mfma_valu_overlap: fp64 MFMA alone / fp64 VALU alone / both on one SIMD, two operand sets
#   kt/           kernel trace + stats
#   pmc1..3/      SQ / GRBM counters, one pass each (never combined with a trace)
#   ktd/ pmcd/    the same scan over dense data (every bin holds events): kernel trace, one counter pass
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_scan
rm -rf "$OUT"; mkdir -p "$OUT"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_valu_overlap tools/micro/mfma_valu_overlap.hip
/tmp/mfma_valu_overlap > "$OUT/overlap.txt" 2>&1
python3 tools/profile/scan_only.py 3 > "$OUT/plain.txt" 2>&1
python3 tools/profile/scan_only.py 2 dense >> "$OUT/plain.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 tools/profile/scan_only.py 3 > "$OUT/kt.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktd" -o kt -- python3 tools/profile/scan_only.py 2 dense > "$OUT/ktd.txt" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc1" -o pmc -- python3 tools/profile/scan_only.py 2 > "$OUT/pmc1.txt" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc2" -o pmc -- python3 tools/profile/scan_only.py 2 > "$OUT/pmc2.txt" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$OUT/pmc3" -o pmc -- python3 tools/profile/scan_only.py 2 > "$OUT/pmc3.txt" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmcd" -o pmc -- python3 tools/profile/scan_only.py 2 dense > "$OUT/pmcd.txt" 2>&1
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name '*.csv' | head -20
cat "$OUT/overlap.txt" "$OUT/plain.txt"
