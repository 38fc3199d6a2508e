#!/bin/bash
# A/B builds of the engine for timing experiments: only the C3 fp16 instantiation, with extra compiler flags, into
# scripts/ubench/bld_<name>/libwavenet_infer.so (git-ignored; travels to the GPU box).  Use with NVW_LIB=<that file>.
#   [SHAPE=128_256_256] scripts/build_variant.sh <name> "<extra flags>"
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
out=$here/scripts/ubench/bld_$name
mkdir -p "$out"
make -s -C "$here/nv_wavenet_amd/csrc" -j4 SHAPES=${SHAPE:-64_256_256} PRECS=16 EXTRA_INST= BLD="$out/obj" OUT="$out/libwavenet_infer.so" EXTRA="$*" 2>&1 | grep -E "error|warning: v" || true
ls -la "$out/libwavenet_infer.so"
