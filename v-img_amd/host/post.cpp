// Post chain on the host: tonemap -> sRGB OETF -> 8-bit quantise -> PNG.
//   clamp / sRGB / quantise : reference include/color_utils.h:21-68, src/main.cpp:339-356
//   AgX                     : reference src/tonemap/agx.cpp:6-90
//   ACES (fitted)           : reference src/tonemap/aces.cpp:5-29
//   Reinhard (luminance)    : reference src/tonemap/reinhard.cpp:3-35
// The PNG writer emits stored (uncompressed) deflate blocks; the reference uses stb_image_write.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "host_scene.hpp"

namespace {

struct C3 {
  float r, g, b;
};
inline float lum(C3 c) { return c.r * 0.212671f + c.g * 0.715160f + c.b * 0.072169f; }
// glm mat3 (column-major, 9 scalars in column order) times vec3
inline C3 mat3_mul(const float m[9], C3 v) {
  return {m[0] * v.r + m[3] * v.g + m[6] * v.b, m[1] * v.r + m[4] * v.g + m[7] * v.b,
          m[2] * v.r + m[5] * v.g + m[8] * v.b};
}

C3 agx_pixel(C3 val) {
  static const float agx_mat[9] = {0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f,
                                   0.0784335999999992f, 0.878468636469772f,  0.0784336f,
                                   0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f};
  static const float agx_mat_inv[9] = {
      1.19687900512017f,   -0.0528968517574562f, -0.0529716355144438f,
      -0.0980208811401368f, 1.15190312990417f,   -0.0980434501171241f,
      -0.0990297440797205f, -0.0989611768448433f, 1.15107367264116f};
  const float min_ev = -12.47393f, max_ev = 4.026069f;
  val = mat3_mul(agx_mat, val);
  auto enc = [&](float x) {
    x = std::min(std::max(std::log2(x), min_ev), max_ev);
    return (x - min_ev) / (max_ev - min_ev);
  };
  auto contrast = [](float x) {
    float x2 = x * x, x4 = x2 * x2;
    return 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 +
           0.1191f * x - 0.00232f;
  };
  val = {contrast(enc(val.r)), contrast(enc(val.g)), contrast(enc(val.b))};
  // agxLook with the default (identity) CDL: pow(val*1+0, 1), then luma + 1*(val - luma)
  val = {std::pow(val.r * 1.f + 0.f, 1.f), std::pow(val.g * 1.f + 0.f, 1.f),
         std::pow(val.b * 1.f + 0.f, 1.f)};
  float l = lum(val);
  val = {l + 1.f * (val.r - l), l + 1.f * (val.g - l), l + 1.f * (val.b - l)};
  val = mat3_mul(agx_mat_inv, val);
  if (val.r < 0.f) val.r = 0.f;
  if (val.g < 0.f) val.g = 0.f;
  if (val.b < 0.f) val.b = 0.f;
  return {std::pow(val.r, 2.2f), std::pow(val.g, 2.2f), std::pow(val.b, 2.2f)};
}

C3 aces_pixel(C3 v) {
  static const float in_m[9] = {0.59719f, 0.07600f, 0.02840f, 0.35458f, 0.90834f,
                                0.13383f, 0.04823f, 0.01566f, 0.83777f};
  static const float out_m[9] = {1.60475f,  -0.10208f, -0.00327f, -0.53108f, 1.10813f,
                                 -0.07276f, -0.07367f, -0.00605f, 1.07602f};
  v = mat3_mul(in_m, v);
  auto fit = [](float x) {
    float a = x * (x + 0.0245786f) - 0.000090537f;
    float b = x * (0.983729f * x + 0.4329510f) + 0.238081f;
    return a / b;
  };
  v = {fit(v.r), fit(v.g), fit(v.b)};
  return mat3_mul(out_m, v);
}

float srgb_oetf(float x) {
  x = std::min(std::max(x, 0.0f), 1.0f);
  if (x < 0.0031308f) return x * 12.92f;
  return 1.055f * std::pow(x, 1.0f / 2.4f) - 0.055f;
}

uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
  if (!crc_ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      crc_table[i] = c;
    }
    crc_ready = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}
void put_u32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back(x >> 24);
  v.push_back(x >> 16);
  v.push_back(x >> 8);
  v.push_back(x);
}
void put_chunk(std::vector<uint8_t>& out, const char tag[4], const std::vector<uint8_t>& data) {
  put_u32(out, static_cast<uint32_t>(data.size()));
  std::vector<uint8_t> body(tag, tag + 4);
  body.insert(body.end(), data.begin(), data.end());
  out.insert(out.end(), body.begin(), body.end());
  put_u32(out, crc32(body.data(), body.size()));
}

}  // namespace

extern "C" int vimg_host_tonemap_to_rgb8(const float* rgb, int w, int h, int tonemapper,
                                          uint8_t* out) {
  if (!rgb || !out || w <= 0 || h <= 0 || tonemapper < 0 || tonemapper > 3) {
    host_set_error("tonemap: bad arguments");
    return -1;
  }
  const size_t n = static_cast<size_t>(w) * h;
  float largest_l = 0.f;
  if (tonemapper == 2)
    for (size_t i = 0; i < n; ++i) {
      float l = lum(C3{rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]});
      if (l > largest_l) largest_l = l;
    }
  for (size_t i = 0; i < n; ++i) {
    C3 c{rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]};
    switch (tonemapper) {
      case 0:
        c = {std::min(std::max(c.r, 0.f), 1.f), std::min(std::max(c.g, 0.f), 1.f),
             std::min(std::max(c.b, 0.f), 1.f)};
        break;
      case 1: c = agx_pixel(c); break;
      case 2: {
        float in_l = lum(c);
        float new_l = (in_l * (1.0f + (in_l / (largest_l * largest_l)))) / (1.0f + in_l);
        if (in_l > 0.f) {
          float k = new_l / in_l;
          c = {c.r * k, c.g * k, c.b * k};
        } else {
          c = {0.f, 0.f, 0.f};
        }
        break;
      }
      case 3: c = aces_pixel(c); break;
    }
    c = {srgb_oetf(c.r), srgb_oetf(c.g), srgb_oetf(c.b)};
    if (std::isnan(c.r) || std::isnan(c.g) || std::isnan(c.b)) {
      out[i * 3] = 255, out[i * 3 + 1] = 0, out[i * 3 + 2] = 255;
    } else {
      out[i * 3] = static_cast<uint8_t>(std::clamp(static_cast<int>(255.999 * c.r), 0, 255));
      out[i * 3 + 1] = static_cast<uint8_t>(std::clamp(static_cast<int>(255.999 * c.g), 0, 255));
      out[i * 3 + 2] = static_cast<uint8_t>(std::clamp(static_cast<int>(255.999 * c.b), 0, 255));
    }
  }
  return 0;
}

extern "C" int vimg_host_write_png(const char* path, const uint8_t* rgb8, int w, int h) {
  if (!path || !rgb8 || w <= 0 || h <= 0) {
    host_set_error("write_png: bad arguments");
    return -1;
  }
  std::vector<uint8_t> raw;
  raw.reserve(static_cast<size_t>(h) * (w * 3 + 1));
  for (int y = 0; y < h; ++y) {
    raw.push_back(0);  // filter: none
    raw.insert(raw.end(), rgb8 + static_cast<size_t>(y) * w * 3,
               rgb8 + static_cast<size_t>(y + 1) * w * 3);
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;
  for (uint8_t c : raw) {
    a = (a + c) % 65521;
    b = (b + a) % 65521;
  }
  for (size_t pos = 0; pos < raw.size();) {
    size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back(n & 0xff);
    z.push_back(n >> 8);
    z.push_back(~n & 0xff);
    z.push_back((~n >> 8) & 0xff);
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  }
  put_u32(z, (b << 16) | a);

  std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  put_u32(ihdr, w);
  put_u32(ihdr, h);
  ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
  put_chunk(png, "IHDR", ihdr);
  put_chunk(png, "IDAT", z);
  put_chunk(png, "IEND", {});
  FILE* f = std::fopen(path, "wb");
  if (!f) {
    host_set_error(std::string("write_png: cannot open ") + path);
    return -1;
  }
  size_t wr = std::fwrite(png.data(), 1, png.size(), f);
  std::fclose(f);
  if (wr != png.size()) {
    host_set_error("write_png: short write");
    return -1;
  }
  return 0;
}
