// PCG hash RNG of the reference (random.hpp:12-29), usable on host and in HIP kernels.
//
// getRandomVec2/3 in the reference are `dvec2(pcg_hash(s), pcg_hash(s)) * 2^-32`: the order in which
// the two (three) draws land in x, y (, z) is unspecified by C++.  The parity target is the
// reference's CPU path, which is host code compiled by g++ (nvcc's host compiler, Makefile:4), and
// g++ evaluates call arguments right to left (verified with g++ 11.4 at -O0 and -O3): the FIRST draw
// goes to the LAST component.  That order is fixed here explicitly, on every compiler.
#pragma once

#include "ptmath.hpp"

namespace pt {

PT_HD uint pcg_hash(uint& seed) {  // random.hpp:12-17
  seed = seed * 747796405u + 2891336453u;
  uint state = seed;
  uint word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

// (double)hash * 2^-32 narrowed to float (random.hpp:9,19-21) == (float)hash * 2^-32 exactly
// (one round-to-nearest of an exactly scaled integer); may return 1.0f.
PT_HD float getRandomFloat(uint& state) { return (float)pcg_hash(state) * 2.3283064365386963e-10f; }

PT_HD vec2 getRandomVec2(uint& state) {
  float y = getRandomFloat(state);
  float x = getRandomFloat(state);
  return vec2(x, y);
}

PT_HD vec3 getRandomVec3(uint& state) {
  float z = getRandomFloat(state);
  float y = getRandomFloat(state);
  float x = getRandomFloat(state);
  return vec3(x, y, z);
}

}  // namespace pt
