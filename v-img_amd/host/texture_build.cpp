// Host precompute for image textures and environment maps (one-off, not on the timed path):
//   mip chain   : ImageTexture ctor, reference src/image_texture.cpp:60-130 (8-tap filter)
//   bilinear tap: ImageTexture::col_at_uv_mipmap, reference src/image_texture.cpp:132-160
//   wrapping    : handle_wrapping, reference include/texture/texture_common.h:22-53
//   env CDFs    : ArraySampling1D/2D ctors, reference include/rng/sampling.h:113-135,168-197
#include <algorithm>
#include <cmath>
#include <numbers>

#include "host_scene.hpp"

namespace {

float wrap_coord(float coord, uint32_t mode) {
  switch (mode) {
    case VIMG_WRAP_CLAMP:
      return std::clamp(coord, 0.f, 1.f);
    case VIMG_WRAP_REPEAT: {
      float fraction = coord - static_cast<int>(coord);
      return std::signbit(fraction) ? 1.f + fraction : fraction;
    }
    case VIMG_WRAP_MIRROR: {
      int int_part = static_cast<int>(coord);
      float fraction = coord - int_part;
      if (std::signbit(fraction)) return (int_part % 2) ? std::fabs(fraction) : 1.f + fraction;
      return fraction;
    }
    default:
      return std::clamp(coord, 0.f, 1.f);
  }
}

struct Rgb {
  float r, g, b;
};
inline Rgb mix(Rgb x, Rgb y, float a) {
  return {x.r * (1.f - a) + y.r * a, x.g * (1.f - a) + y.g * a, x.b * (1.f - a) + y.b * a};
}

Rgb tap(const float* level, uint32_t mip_w, uint32_t mip_h, uint32_t wrap_u, uint32_t wrap_v,
        float u, float v) {
  float pixel_u = wrap_coord(u, wrap_u) * mip_w;
  float pixel_v = wrap_coord(v, wrap_v) * mip_h;
  int cx = std::clamp(static_cast<int>(pixel_u), 0, static_cast<int>(mip_w) - 1);
  int cy = std::clamp(static_cast<int>(pixel_v), 0, static_cast<int>(mip_h) - 1);
  int nx = std::clamp(cx + 1, 0, static_cast<int>(mip_w) - 1);
  int ny = std::clamp(cy + 1, 0, static_cast<int>(mip_h) - 1);
  float fx = pixel_u - cx, fy = pixel_v - cy;
  auto at = [&](int x, int y) {
    const float* p = level + (static_cast<size_t>(x) + static_cast<size_t>(y) * mip_w) * 3;
    return Rgb{p[0], p[1], p[2]};
  };
  Rgb a = mix(at(cx, cy), at(nx, cy), fx);
  Rgb b = mix(at(cx, ny), at(nx, ny), fx);
  return mix(a, b, fy);
}

// builders installed by vimg_host_set_precompute (e.g. libvimg_hip's GPU kernels); null = the
// loops below
vimg_mip_builder_fn g_mip_builder = nullptr;
vimg_env_cdf_builder_fn g_cdf_builder = nullptr;

}  // namespace

extern "C" void vimg_host_set_precompute(vimg_mip_builder_fn mip, vimg_env_cdf_builder_fn cdf) {
  g_mip_builder = mip;
  g_cdf_builder = cdf;
}

bool build_mip_chain(uint32_t w, uint32_t h, const float* level0, uint32_t wrap_u, uint32_t wrap_v,
                     VimgTexture& tex, std::vector<float>& pool) {
  tex.type = VIMG_TEX_IMAGE;
  tex.width = w;
  tex.height = h;
  tex.wrap_u = wrap_u;
  tex.wrap_v = wrap_v;
  const int num_levels = std::min(
      static_cast<int>(std::ceil(std::log2(static_cast<float>(std::min(w, h))))),
      VIMG_MAX_MIP_LEVELS);

  if (g_mip_builder) {
    // an installed builder fills all levels at once, level 0 first
    uint64_t total = 0, off = pool.size() / 3;
    uint32_t lw = w, lh = h;
    tex.num_levels = static_cast<uint32_t>(std::max(num_levels, 1));
    for (uint32_t l = 0; l < tex.num_levels; ++l) {
      tex.level_offset[l] = off + total;
      total += uint64_t(lw) * lh;
      lw = std::max(lw / 2u, 1u), lh = std::max(lh / 2u, 1u);
    }
    const size_t base = pool.size();
    pool.resize(base + total * 3);
    if (g_mip_builder(w, h, level0, wrap_u, wrap_v, pool.data() + base) == 0) return true;
    // an installed builder that fails is an error of the caller's set-up (no GPU ...): no silent
    // second path
    pool.resize(base);
    host_set_error("the installed mip-chain builder failed");
    return false;
  }

  tex.level_offset[0] = pool.size() / 3;
  pool.insert(pool.end(), level0, level0 + static_cast<size_t>(w) * h * 3);
  tex.num_levels = 1;

  uint32_t prev_w = w, prev_h = h;
  for (int l = 1; l < num_levels; ++l) {
    const uint32_t next_w = std::max(prev_w / 2u, 1u), next_h = std::max(prev_h / 2u, 1u);
    std::vector<float> next(static_cast<size_t>(next_w) * next_h * 3);
    const size_t prev_off = tex.level_offset[l - 1] * 3;
    const float inv_x = 1.f / prev_w, inv_y = 1.f / prev_h;
    // the 8 taps: 4 diagonal positives, 4 axial negatives
    static const float kOff[8][2] = {{-0.75777f, -0.75777f}, {0.75777f, -0.75777f},
                                     {0.75777f, 0.75777f},   {-0.75777f, 0.75777f},
                                     {-2.907f, 0.f},         {2.907f, 0.f},
                                     {0.f, -2.907f},         {0.f, 2.907f}};
    static const float kW[8] = {0.37487566f,  0.37487566f,  0.37487566f,  0.37487566f,
                                -0.12487566f, -0.12487566f, -0.12487566f, -0.12487566f};
#pragma omp parallel for if (next_w * next_h > 4096)
    for (int y = 0; y < static_cast<int>(next_h); ++y) {
      for (uint32_t x = 0; x < next_w; ++x) {
        const float* prev = pool.data() + prev_off;
        float cu = static_cast<float>(2 * x) * inv_x, cv = static_cast<float>(2 * y) * inv_y;
        Rgb sum{0.f, 0.f, 0.f};
        for (int k = 0; k < 8; ++k) {
          Rgb c = tap(prev, prev_w, prev_h, wrap_u, wrap_v, cu + kOff[k][0] * inv_x,
                      cv + kOff[k][1] * inv_y);
          sum.r += kW[k] * c.r;
          sum.g += kW[k] * c.g;
          sum.b += kW[k] * c.b;
        }
        if (sum.r < 0) sum.r = 0.f;
        if (sum.g < 0) sum.g = 0.f;
        if (sum.b < 0) sum.b = 0.f;
        float* o = next.data() + (x + static_cast<size_t>(y) * next_w) * 3;
        o[0] = sum.r;
        o[1] = sum.g;
        o[2] = sum.b;
      }
    }
    tex.level_offset[l] = pool.size() / 3;
    pool.insert(pool.end(), next.begin(), next.end());
    tex.num_levels = l + 1;
    prev_w = next_w;
    prev_h = next_h;
  }
  return true;
}

namespace {
// ArraySampling1D ctor: cdf of |f|, normalised; uniform when the integral is 0.  Returns integral.
float build_cdf1d(const float* f, size_t n, float* cdf) {
  cdf[0] = 0.f;
  for (size_t x = 1; x < n + 1; ++x) cdf[x] = cdf[x - 1] + std::abs(f[x - 1]);
  float func_int = cdf[n];
  if (func_int == 0)
    for (size_t i = 0; i < n + 1; ++i) cdf[i] = static_cast<float>(i) / static_cast<float>(n);
  else
    for (size_t i = 0; i < n + 1; ++i) cdf[i] /= func_int;
  return func_int;
}
}  // namespace

bool build_env_cdfs(const float* img, uint32_t w, uint32_t h, std::vector<float>& pool,
                    uint64_t& row_off, uint64_t& col_off) {
  if (g_cdf_builder) {
    const size_t base = pool.size();
    row_off = base;
    col_off = base + (h + 1);
    pool.resize(base + (h + 1) + static_cast<size_t>(h) * (w + 1));
    if (g_cdf_builder(img, w, h, pool.data() + row_off, pool.data() + col_off) == 0) return true;
    pool.resize(base);
    host_set_error("the installed env-map CDF builder failed");
    return false;
  }
  std::vector<float> lum(static_cast<size_t>(w) * h);
  for (size_t y = 0; y < h; ++y) {
    float v = (static_cast<float>(y) + 0.5f) / static_cast<float>(h);
    float sin_elevation = std::sin(std::numbers::pi * v);
    for (size_t x = 0; x < w; ++x) {
      const float* p = img + (y * w + x) * 3;
      float l = p[0] * 0.212671f + p[1] * 0.715160f + p[2] * 0.072169f;  // luminance()
      lum[y * w + x] = l * sin_elevation;
    }
  }
  row_off = pool.size();
  pool.resize(pool.size() + (h + 1));
  col_off = pool.size();
  pool.resize(pool.size() + static_cast<size_t>(h) * (w + 1));
  std::vector<float> row_int(h);
  for (size_t y = 0; y < h; ++y)
    row_int[y] = build_cdf1d(lum.data() + y * w, w, pool.data() + col_off + y * (w + 1));
  build_cdf1d(row_int.data(), h, pool.data() + row_off);
  return true;
}

// ---- 8-bit image conversions of the reference's texture loaders ------------------------------
// convert_sRGB_to_linear (src/image_texture.cpp:257-263, include/color_utils.h:28-47): value/255,
// then the sRGB EOTF; a function of the 256 possible inputs
extern "C" void vimg_host_srgb8_lut(float lut[256]) {
  for (int i = 0; i < 256; ++i) {
    float p = static_cast<float>(i) / 255.f;
    lut[i] = (p <= 0.04045f) ? p / 12.92f : std::pow((p + 0.055f) / 1.055f, 2.4f);
  }
}
extern "C" void vimg_host_srgb8_to_linear(const uint8_t* in, uint64_t n, float* out) {
  float lut[256];
  vimg_host_srgb8_lut(lut);
#pragma omp parallel for
  for (int64_t i = 0; i < static_cast<int64_t>(n); ++i) out[i] = lut[in[i]];
}
// convert_RGB_to_normal (src/image_texture.cpp:265-275)
extern "C" void vimg_host_rgb8_to_normal(const uint8_t* rgb8, uint64_t n_pixels, float scale, float* out) {
#pragma omp parallel for
  for (int64_t i = 0; i < static_cast<int64_t>(n_pixels); ++i) {
    float x = static_cast<float>(rgb8[i * 3 + 0]) / 127.5f - 1.f;
    float y = static_cast<float>(rgb8[i * 3 + 1]) / 127.5f - 1.f;
    float z = static_cast<float>(rgb8[i * 3 + 2]) / 127.5f - 1.f;
    x *= scale;
    y *= scale;
    const float inv = 1.0f / std::sqrt(x * x + y * y + z * z);   // glm::normalize
    out[i * 3 + 0] = x * inv;
    out[i * 3 + 1] = y * inv;
    out[i * 3 + 2] = z * inv;
  }
}
