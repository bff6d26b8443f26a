#!/bin/bash
# Builds variant libraries with kernels_chan.hip compiled under extra flags (timing experiments): build/libqrl_<name>.so
set -e
cd "$(dirname "$0")/../qradiolink_amd/csrc"
make -s -j8
mkdir -p ../../build
OBJ=$(ls *.o | grep -v '^kernels_chan.o$')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 $flags -c kernels_chan.hip -o ../../build/ch_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/libqrl_$name.so $OBJ ../../build/ch_$name.o -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
  echo built $name
done
