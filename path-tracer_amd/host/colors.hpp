// Degree-6 polynomial colour maps used by the reference: afmhot for the `-a` heat map
// (main.cpp:363-374, colors.cpp:6-15) and hsv for scene 1's material albedos
// (sampleScenes.cpp:89, colors.cpp:18-27).  Coefficients are data from the reference (originally
// https://www.shadertoy.com/view/Nd3fR2); evaluation is Horner's scheme per channel, clamped.
// The coefficients are double literals narrowed to float, exactly as the reference's vec3(double...) does.
#pragma once

#include "ptmath.hpp"

namespace pt {

namespace detail {
PT_HD vec3 poly6(const float (&c)[7][3], float t) {
  vec3 acc(c[6][0], c[6][1], c[6][2]);
  for (int k = 5; k >= 0; k--) acc = vec3(c[k][0], c[k][1], c[k][2]) + t * acc;
  return gclamp(acc, 0.0f, 1.0f);
}
}  // namespace detail

PT_HD vec3 afmhot(float t) {
  const float c[7][3] = {{-0.020390, 0.009557, 0.018508},     {3.108226, -0.106297, -1.105891},
                         {-14.539061, -2.943057, 14.548595},  {71.394557, 22.644423, -71.418400},
                         {-152.022488, -31.024563, 152.048692}, {139.593599, 12.411251, -139.604042},
                         {-46.532952, -0.000874, 46.532928}};
  return detail::poly6(c, t);
}

PT_HD vec3 hsv(float t) {
  const float c[7][3] = {{0.834511, -0.153764, -0.139860},     {8.297883, 13.629371, 7.673034},
                         {-80.602944, -80.577977, -90.865764}, {245.028545, 291.294154, 390.181844},
                         {-376.406597, -575.667879, -714.180803}, {306.639709, 538.472148, 596.580595},
                         {-102.934273, -187.108098, -189.286489}};
  return detail::poly6(c, t);
}

}  // namespace pt
