#!/bin/bash
# A/B builds of libopenpano_hip.so for timing experiments (never the product build):
#   scripts/build_variant.sh <name> "<file.hip> [file2.hip ...]" "<-D flags>"   ->  openpano_amd/variants/libopenpano_hip_<name>.so
# The product sources carry no experiment code.  The instrumented forms of the three kernels DESIGN section 6 reports on
# (-DOP_PYR_EXPERIMENT=1..9, -DOP_DESC_EXPERIMENT=1..7, -DOP_MATCH_EXPERIMENT=9: pieces of a kernel compiled out, clock64
# phase traces read through op_debug_*_timers) live as patches under scripts/experiments/; a named file is copied to
# openpano_amd/variants/src/, patched there when its patch exists and still applies, and compiled from the copy.
set -e
cd "$(dirname "$0")/../openpano_amd/csrc"
mkdir -p ../variants/src
name=$1; srcs=$2; flags=$3
for src in $srcs; do
  cp $src ../variants/src/$src
  pt=../../scripts/experiments/${src%.hip}_timing_experiments.patch
  if [ -f $pt ] && [[ "$flags" == *EXPERIMENT* ]]; then
    (cd ../variants/src && patch -s -p4 $src < ../../../scripts/experiments/${src%.hip}_timing_experiments.patch) || { echo "experiment patch for $src no longer applies"; exit 1; }
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fopenmp -I../../include -I. -Wall -Wno-unused-function -Wno-unused-value $flags -c ../variants/src/$src -o ../variants/${name}_${src%.hip}.o &
done
wait
objs=""
for f in *.hip; do
  if [[ " $srcs " == *" $f "* ]]; then objs="$objs ../variants/${name}_${f%.hip}.o"; else objs="$objs ${f%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o ../variants/libopenpano_hip_${name}.so $objs
echo built ../variants/libopenpano_hip_${name}.so
