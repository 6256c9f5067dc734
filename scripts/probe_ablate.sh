#!/bin/bash
# timing of variant builds (variants/libmst_ab*.so) against the shipped library.  Usage: probe_ablate.sh tag names...
TAG=$1; shift
OUT=gpurun_out/ablate_$TAG.txt; : > $OUT
for w in 3 2; do
  for n in shipped "$@"; do
    echo "=== WPS=$w lib=$n" >> $OUT
    if [ "$n" = shipped ]; then MST_V2_WPS=$w timeout -k 10 200 python scripts/probe_melfeat.py 72 2>&1 | grep "ms/step" >> $OUT
    else MST_LIB=$PWD/variants/libmst_ab$n.so MST_V2_WPS=$w timeout -k 10 200 python scripts/probe_melfeat.py 72 2>&1 | grep "ms/step" >> $OUT; fi
  done
done
cat $OUT
