#!/bin/bash
# Run ON THE GPU BOX (through gpurun): k_grad_mfma built five ways (rows fetched at a block's top, behind its last product-1 MFMA, a
# block ahead; one or two waves per SIMD),
# bi_eval_grad over 131 072 points of C2 each -> gpurun_out/grad_variants.txt.  Leaves the library built with the LAST variant on
# the box only (the box is scratch).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/grad_variants.txt
VARIANTS=("-DBI_GRAD_PREFETCH=0" "-DBI_GRAD_PREFETCH=2" "-DBI_GRAD_PREFETCH=3" "-DBI_GRAD_PREFETCH=1" "-DBI_GRAD_PREFETCH=1 -DBI_GRAD_WAVES=1" "-DBI_GRAD_PREFETCH=0 -DBI_GRAD_WAVES=1")
if [ -n "$GRAD_VARIANTS_SHORT" ]; then VARIANTS=("-DBI_GRAD_PREFETCH=0" "-DBI_GRAD_PREFETCH=2" "-DBI_GRAD_PREFETCH=3" "-DBI_GRAD_PREFETCH=0" "-DBI_GRAD_PREFETCH=2" "-DBI_GRAD_PREFETCH=3"); fi
for v in "${VARIANTS[@]}"; do
  BLUEICE_AMD_EXTRA_FLAGS="$v" python -c "import blueice_amd.build as b; b.build()" > /dev/null 2>&1
  echo "== $v" >> gpurun_out/grad_variants.txt
  BLUEICE_AMD_EXTRA_FLAGS="$v" python tools/profile/grad_only.py 4 >> gpurun_out/grad_variants.txt 2>&1
done
cat gpurun_out/grad_variants.txt
