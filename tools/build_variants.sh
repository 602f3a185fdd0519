#!/bin/bash
# Builds libptx_backend_<name>.so variants with extra -D flags:  tools/build_variants.sh name "-DX=1 -DY=2" ...
# SRC=<dir> builds another checkout of the sources (e.g. `git archive HEAD | tar -x -C /tmp/head`) for before / after
# runs with tools/ab.py.
set -e
cd "$(dirname "$0")/.."
SRC=${SRC:-.}
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$SRC/include \
     -fhip-fp32-correctly-rounded-divide-sqrt $flags -shared -o path-tracer_amd/lib/libptx_backend_$name.so $SRC/path-tracer_amd/csrc/ptx_backend.hip &
done
wait
ls -la path-tracer_amd/lib/
