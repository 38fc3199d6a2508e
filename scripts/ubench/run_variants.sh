#!/bin/bash
# usage: run_variants.sh "<B N> <B N> ..." lib1 lib2 ...   (base = the in-tree library)
cfgs="$1"; shift
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset NVW_LIB; else export NVW_LIB=$PWD/scripts/ubench/$lib; fi
  echo "$cfgs" | tr ',' '\n' | while read B N; do
    printf "%-28s " "$lib"; timeout 300 python scripts/quick_phase.py $B $N 2>&1 | grep "us/sample" | cut -c1-40
  done
done
