#!/bin/bash
# Builds an A/B variant of the backend: tools/variant.sh <name> [-D flags...] -> path-tracer_amd/lib/libptx_backend_<name>.so
# (timed against each other on the GPU box with tools/ab.py).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-$ROOT}   # another checkout of the sources (e.g. git archive HEAD) for before/after runs
NAME=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall \
  -I$SRC/include -fhip-fp32-correctly-rounded-divide-sqrt -shared "$@" \
  -o $ROOT/path-tracer_amd/lib/libptx_backend_$NAME.so $SRC/path-tracer_amd/csrc/ptx_backend.hip
