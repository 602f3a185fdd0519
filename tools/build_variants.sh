#!/bin/bash
# Builds libptx_backend_<name>.so variants with extra -D flags:  tools/build_variants.sh name "-DX=1 -DY=2" ...
set -e
cd "$(dirname "$0")/.."
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude \
     -fhip-fp32-correctly-rounded-divide-sqrt $flags -shared -o path-tracer_amd/lib/libptx_backend_$name.so path-tracer_amd/csrc/ptx_backend.hip &
done
wait
ls -la path-tracer_amd/lib/
