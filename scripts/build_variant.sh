#!/bin/bash
# A/B builds of libopenpano_hip.so with extra defines (timing experiments only; never the product build):
#   scripts/build_variant.sh <name> "<file.hip> [file2.hip ...]" "<-D flags>"   ->  openpano_amd/variants/libopenpano_hip_<name>.so
set -e
cd "$(dirname "$0")/../openpano_amd/csrc"
mkdir -p ../variants
name=$1; srcs=$2; flags=$3
for src in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fopenmp -I../../include -I. -Wall -Wno-unused-function -Wno-unused-value $flags -c $src -o ../variants/${name}_${src%.hip}.o &
done
wait
objs=""
for f in *.hip; do
  if [[ " $srcs " == *" $f "* ]]; then objs="$objs ../variants/${name}_${f%.hip}.o"; else objs="$objs ${f%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o ../variants/libopenpano_hip_${name}.so $objs
echo built ../variants/libopenpano_hip_${name}.so
