#!/bin/bash
# the reference-definition measurement (nv_wavenet_perf equivalent) on the BASELINE configs, twice
export TMPDIR=/tmp
for rep in 1 2; do
for args in "-r 64 -s 256 -a 256 -l 20 -b 16 -m 1" "-r 64 -s 256 -a 256 -l 20 -b 16 -m 3" "-r 128 -s 256 -a 256 -l 30 -b 8 -m 3" "-r 64 -s 128 -a 256 -l 20 -b 4 -m 3"; do
  echo -n "$args: "
  timeout 300 python scripts/nv_wavenet_perf.py $args -n 16384 -t 2048 2>&1 | grep -E "Sample rate|timed out|rror" | tr '\n' ' '; echo
done; done
