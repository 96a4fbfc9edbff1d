#!/bin/bash
# A/B of library builds on ONE box: tools/ab.sh libA.so libB.so ... (each run = tools/perf_variants.py 3, best of 3 launches); two rounds
for round in 1 2; do
  for lib in "$@"; do
    echo -n "$lib: "
    RT06_LIB=$PWD/$lib python tools/perf_variants.py 3 2>/dev/null | tail -1
  done
done
