#!/bin/bash
# A/B builds of libopenpano_hip.so for timing experiments (never the product build):
#   scripts/build_variant.sh <name> "<file.hip> [file2.hip ...]" "<-D flags>"   ->  ab/libopenpano_hip_<name>.so
# The product sources carry no experiment code.  The instrumented forms of the kernels DESIGN section 6 reports on
# (-DOP_PYR_EXPERIMENT=1..9, -DOP_DESC_EXPERIMENT=1..7, -DOP_MATCH_EXPERIMENT=9: pieces of a kernel compiled out, clock64
# phase traces read through op_debug_*_timers) live as patches under scripts/experiments/; a named file is copied to
# a scratch directory, patched there when its patch exists and still applies, and compiled from the copy.
# Only the linked library lands in the tree (ab/, git-ignored, ~1 MB each): it has to travel to the GPU box with the
# snapshot; objects and patched sources stay under /tmp.  Empty ab/ when an experiment is over.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/openpano_amd/csrc"
name=$1; srcs=$2; flags=$3
work=/tmp/op_variants/$name; mkdir -p $work "$root/ab"
for src in $srcs; do
  cp $src $work/$src
  pt=$root/scripts/experiments/${src%.hip}_timing_experiments.patch
  if [ -f $pt ] && [[ "$flags" == *EXPERIMENT* ]]; then
    (cd $work && patch -s -p4 $src < $pt) || { echo "experiment patch for $src no longer applies"; exit 1; }
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fopenmp -I../../include -I. -Wall -Wno-unused-function -Wno-unused-value $flags -c $work/$src -o $work/${src%.hip}.o &
done
wait
objs="ransac_accept_simd.o"                  # host-only TU, never a variant
for f in *.hip; do
  if [[ " $srcs " == *" $f "* ]]; then objs="$objs $work/${f%.hip}.o"; else objs="$objs ${f%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o "$root/ab/libopenpano_hip_${name}.so" $objs
echo built ab/libopenpano_hip_${name}.so
