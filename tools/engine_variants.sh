#!/bin/bash
# Builds variant libraries with engine.cpp compiled under extra flags (timing experiments): build/libqrl_<name>.so
# usage: tools/engine_variants.sh name "extra hipcc flags" [name flags ...]
set -e
cd "$(dirname "$0")/../qradiolink_amd/csrc"
make -s -j8
mkdir -p ../../build
OBJ=$(ls *.o | grep -v '^engine.o$')
CXXFLAGS=$(make -pn 2>/dev/null | grep -m1 '^CXXFLAGS' | sed 's/^CXXFLAGS *[:+]*= *//')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include $flags -c engine.cpp -o ../../build/engine_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/libqrl_$name.so $OBJ ../../build/engine_$name.o -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
  echo built $name
done
