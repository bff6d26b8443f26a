#!/bin/bash
# Builds variant libraries with ONE source file of qradiolink_amd/csrc compiled under extra flags (timing experiments, same-box A/B
# through `tools/gpu_call.sh OUT ab ...`): build/libqrl_<name>.so.  One script for every kernel file and for engine.cpp.
# usage: tools/kernel_variants.sh kernels_x.hip|engine.cpp name1 "flags1" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/../qradiolink_amd/csrc"
make -s -j8
mkdir -p ../../build
src=$1; shift
base=${src%.*}
OBJ=$(ls *.o | grep -v "^$base.o$")
extra="--offload-arch=gfx950"
[ "$base" = kernels_decim_pl ] && extra="$extra -fno-slp-vectorize"
[ "${src##*.}" = cpp ] && extra="-I../../include"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra $flags -c $src -o ../../build/${base}_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/libqrl_$name.so $OBJ ../../build/${base}_$name.o -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
  echo built $name
done
