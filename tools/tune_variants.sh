#!/bin/bash
# A/B differently built libraries on the bench workload, interleaved, same box
for round in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    BLUEICE_AMD_LIB=$PWD/blueice_amd/lib/$lib python tools/tune_dense.py 2>&1 | grep "nt=1 blocks_per_cu= 8"
  done
done
