#!/bin/bash
# Registers, LDS, scratch and occupancy-relevant numbers of every kernel in the library, from the gfx950 code object's
# metadata (no GPU needed):   tools/kernel_resources.sh [name filter]
set -e
cd "$(dirname "$0")/.."
# (the library is six translation units, blueice_amd/csrc/bi_common.h: each is compiled device-only, side by side, and the
#  notes of all code objects are read together; UNITS="tu_scan_sorted" limits the run to one of them)
UNITS=${UNITS:-"blueice_hip tu_morph tu_scan tu_scan_sorted tu_grad tu_prim"}
OUT=${TMPDIR:-/tmp}/blueice_hip_gfx950
for u in $UNITS; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -c -ffp-contract=off -mllvm --amdgpu-mfma-vgpr-form \
        -Wno-unused-function -o "$OUT.$u.bundle" blueice_amd/csrc/$u.hip
      /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$OUT.$u.bundle" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$OUT.$u.co" ) &
done
wait
for u in $UNITS; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$OUT.$u.co"; done | python3 -c '
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows, cur = [], None
for line in sys.stdin:
    if re.match(r"\s+- \.agpr_count:", line):          # first key of every kernel record
        cur = {}
        rows.append(cur)
        line = line.replace("- .", "  .", 1)
    m = re.match(r"\s{4}\.(\w+):\s+(.*)", line)
    if m and cur is not None:
        cur[m.group(1)] = m.group(2).strip().strip("\x27")
names = subprocess.run(["c++filt"], input="\n".join(r.get("name", "?") for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, name in zip(rows, names):
    if flt in name:
        print("%-100s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill_v %s" % (name[:100], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"), r.get("group_segment_fixed_size"), r.get("private_segment_fixed_size"), r.get("vgpr_spill_count")))
' "$1"
