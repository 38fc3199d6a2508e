#!/bin/bash
# usage: run_modes.sh lib "<mode B N>,<mode B N>..."   (mode "-" = engine default)
lib="$1"; cfgs="$2"
if [ "$lib" != base ]; then export NVW_LIB=$PWD/scripts/ubench/$lib; fi
echo "$cfgs" | tr ',' '\n' | while read M B N; do
  if [ "$M" = "-" ]; then unset NVW_MODE; else export NVW_MODE=$M; fi
  printf "%-20s %-8s " "$lib" "$M"; timeout 300 python scripts/quick_phase.py $B $N 2>&1 | grep "us/sample" | cut -c1-40
done
