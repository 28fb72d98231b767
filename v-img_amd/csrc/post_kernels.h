// Post chain on the GPU: tonemap -> sRGB OETF -> 8-bit quantise, one lane
// per pixel.  HBM-bound by construction: 12 B read + 3 B written per pixel, no reuse.
//   simple_clamp / sRGB_gamma_correction  reference include/color_utils.h:21-68
//   agx                                    reference src/tonemap/agx.cpp:6-90
//   reinhard_lum                           reference src/tonemap/reinhard.cpp:3-35
//   aces                                   reference src/tonemap/aces.cpp:5-29
//   quantise, NaN -> magenta               reference src/main.cpp:339-356
// Float pow/log2 of the reference are evaluated in double and narrowed once (DESIGN.md Numerics).
#pragma once
#include "device_math.h"

namespace vimg {

VD float F_pow(float x, float y) {
  return static_cast<float>(::pow(static_cast<double>(x), static_cast<double>(y)));
}
VD f3 mat3_mul(const float* m, f3 v) {   // glm column-major mat3 * vec3
  return f3{m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
            m[2] * v.x + m[5] * v.y + m[8] * v.z};
}
VD float lum3(f3 v) { return dot(v, f3{0.212671f, 0.715160f, 0.072169f}); }

VD f3 agx_pixel(f3 val) {
  const float agx_mat[9] = {0.842479062253094, 0.0423282422610123, 0.0423756549057051,
                            0.0784335999999992, 0.878468636469772, 0.0784336,
                            0.0792237451477643, 0.0791661274605434, 0.879142973793104};
  const float agx_mat_inv[9] = {1.19687900512017, -0.0528968517574562, -0.0529716355144438,
                                -0.0980208811401368, 1.15190312990417, -0.0980434501171241,
                                -0.0990297440797205, -0.0989611768448433, 1.15107367264116};
  const float min_ev = -12.47393f, max_ev = 4.026069f;
  val = mat3_mul(agx_mat, val);
  val = f3{clampf(F_log2(val.x), min_ev, max_ev), clampf(F_log2(val.y), min_ev, max_ev),
           clampf(F_log2(val.z), min_ev, max_ev)};
  val = (val + (-min_ev)) / (max_ev - min_ev);
  {
    f3 x = val, x2 = x * x, x4 = x2 * x2;
    val = splat3(+15.5f) * x4 * x2 - splat3(40.14f) * x4 * x + splat3(31.96f) * x4
          - splat3(6.868f) * x2 * x + splat3(0.4298f) * x2 + splat3(0.1191f) * x
          - splat3(0.00232f);
  }
  {
    float luma = lum3(val);
    val = f3{luma + 1.0f * (val.x - luma), luma + 1.0f * (val.y - luma), luma + 1.0f * (val.z - luma)};
  }
  val = mat3_mul(agx_mat_inv, val);
  if (val.x < 0.f) val.x = 0.f;
  if (val.y < 0.f) val.y = 0.f;
  if (val.z < 0.f) val.z = 0.f;
  return f3{F_pow(val.x, 2.2f), F_pow(val.y, 2.2f), F_pow(val.z, 2.2f)};
}
VD f3 aces_pixel(f3 v) {
  const float in_m[9] = {0.59719f, 0.07600f, 0.02840f, 0.35458f, 0.90834f,
                         0.13383f, 0.04823f, 0.01566f, 0.83777f};
  const float out_m[9] = {1.60475f,  -0.10208f, -0.00327f, -0.53108f, 1.10813f,
                          -0.07276f, -0.07367f, -0.00605f, 1.07602f};
  v = mat3_mul(in_m, v);
  f3 a = v * (v + 0.0245786f) + (-0.000090537f);
  f3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
  v = a / b;
  return mat3_mul(out_m, v);
}
VD float srgb_oetf(float x) {
  x = clampf(x, 0.0f, 1.0f);
  if (x < 0.0031308f) return x * 12.92f;
  return 1.055f * F_pow(x, 1.0f / 2.4f) - 0.055f;
}

// largest_luminance (reference src/tonemap/reinhard.cpp:3-15): max over non-NaN luminances;
// non-negative floats order like their bit patterns, so one atomicMax per wave on the bits
__global__ void post_max_luminance_kernel(const float* __restrict__ rgb, size_t n,
                                          unsigned int* __restrict__ max_bits) {
  float best = 0.0f;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
       i += size_t(gridDim.x) * blockDim.x) {
    float l = lum3(f3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]});
    if (l > best) best = l;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    float o = __shfl_xor(best, off);
    if (o > best) best = o;
  }
  if ((threadIdx.x & 63) == 0 && best > 0.0f) atomicMax(max_bits, __float_as_uint(best));
}

__global__ void post_rgb8_kernel(const float* __restrict__ rgb, size_t n, int tonemapper,
                                 const unsigned int* __restrict__ max_bits,
                                 unsigned char* __restrict__ out) {
  const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  f3 c{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
  if (tonemapper == 0) {
    c = f3{clampf(c.x, 0.f, 1.f), clampf(c.y, 0.f, 1.f), clampf(c.z, 0.f, 1.f)};
  } else if (tonemapper == 1) {
    c = agx_pixel(c);
  } else if (tonemapper == 2) {
    const float largest_L = __uint_as_float(*max_bits);
    float in_L = lum3(c);
    float numerator = in_L * (1.0f + (in_L / (largest_L * largest_L)));
    float new_L = numerator / (1.0f + in_L);
    c = (in_L > 0.f) ? c * (new_L / in_L) : f3{0.f, 0.f, 0.f};
  } else {
    c = aces_pixel(c);
  }
  c = f3{srgb_oetf(c.x), srgb_oetf(c.y), srgb_oetf(c.z)};
  unsigned char r, g, b;
  if (is_nan(c.x) || is_nan(c.y) || is_nan(c.z)) {
    r = 255, g = 0, b = 255;
  } else {
    r = static_cast<unsigned char>(clampi(static_cast<int>(255.999 * c.x), 0, 255));
    g = static_cast<unsigned char>(clampi(static_cast<int>(255.999 * c.y), 0, 255));
    b = static_cast<unsigned char>(clampi(static_cast<int>(255.999 * c.z), 0, 255));
  }
  out[3 * i + 0] = r;
  out[3 * i + 1] = g;
  out[3 * i + 2] = b;
}

}  // namespace vimg
