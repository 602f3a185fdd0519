# Build everything in-tree:  make            (host lib, HIP backend, CLI, oracle)
#                            make host       (no hipcc needed)
ROOT := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
PKG  := path-tracer_amd
HOST := $(PKG)/host
CSRC := $(PKG)/csrc
LIB  := $(PKG)/lib

CXX      = g++
HIPCC    = /opt/rocm/bin/hipcc
# -ffp-contract=off everywhere: the BVH, the instance matrices and the -a heat map are bit-exact
# contracts between x86-64 and gfx950 (DESIGN.md "Floating point").
CXXFLAGS = -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Iinclude
HIPFLAGS = -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Iinclude \
           -fhip-fp32-correctly-rounded-divide-sqrt

HOST_SRC = $(HOST)/scene.cpp $(HOST)/loaders.cpp $(HOST)/sample_scenes.cpp $(HOST)/flatten.cpp \
           $(HOST)/output.cpp $(HOST)/capi.cpp
HOST_HDR = $(wildcard $(HOST)/*.hpp) include/ptx_backend.h include/pt_host.h
HIP_SRC  = $(CSRC)/ptx_backend.hip
HIP_HDR  = $(wildcard $(CSRC)/*.hpp) $(HOST)/ptmath.hpp $(HOST)/rng.hpp $(HOST)/colors.hpp include/ptx_backend.h

all: host backend cli oracle sharedlibm

host: $(LIB)/libpt_host.so
backend: $(LIB)/libptx_backend.so
# test build of the backend whose five libm calls come from tests/shared_libm/shared_libm.h (tests/test_gpu_shared_libm.py)
sharedlibm: $(LIB)/libptx_backend_sharedlibm.so
cli: $(PKG)/PathTracer
oracle:
	$(MAKE) -C oracle

$(LIB)/libpt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIB)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -lz

# the library carries the hash of its sources and flags (ptx_build_id(): tools/source_hash.py)
$(LIB)/libptx_backend.so: $(HIP_SRC) $(HIP_HDR) tools/source_hash.py
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -DPTX_BUILD_ID=\"$(shell python3 tools/source_hash.py "$(HIPFLAGS)")\" -shared -o $@ $(HIP_SRC)

$(LIB)/libptx_backend_sharedlibm.so: $(HIP_SRC) $(HIP_HDR) tests/shared_libm/shared_libm.h
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -DPTX_BUILD_ID=\"sharedlibm\" -DPTX_SHARED_LIBM='"../../tests/shared_libm/shared_libm.h"' -shared -o $@ $(HIP_SRC)

$(PKG)/PathTracer: $(HOST)/main.cpp $(HOST)/gpu_render.cpp $(LIB)/libpt_host.so $(LIB)/libptx_backend.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp $(HOST)/gpu_render.cpp -L$(LIB) -lpt_host -lptx_backend \
	    -Wl,-rpath,'$$ORIGIN/lib' -lpthread

clean:
	rm -rf $(LIB) $(PKG)/PathTracer
	$(MAKE) -C oracle clean

.PHONY: all host backend cli oracle sharedlibm clean
