// TEST INFRASTRUCTURE (see oracle.h): CPU restatement of the reference's per-pixel path-tracing
// hot path, function by function.  Scalar, AoS, one virtual-dispatch site of the reference = one
// switch here.  Not the product; never linked into libvimg_hip.so.
//
// Reference = atom501/v-img at /root/reference; every function cites the file:line it follows.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <bit>
#include <cmath>
#include <cstring>
#include <thread>
#include <utility>
#include <vector>

#include "omath.h"

#if defined(ORACLE_AVX2_TIMING) && ORACLE_AVX2_TIMING
#include <immintrin.h>
#endif

using namespace om;

namespace {

// ============================================================================ RNG
// pcg32_random_t / pcg32_random_r / pcg32_srandom_r — reference include/rng/pcg_rand.h:5-33
struct Pcg {
  uint64_t state;
  uint64_t inc;
};
inline uint32_t pcg32_random_r(Pcg* rng) {
  uint64_t oldstate = rng->state;
  rng->state = oldstate * 6364136223846793005ULL + rng->inc;
  uint32_t xorshifted = static_cast<uint32_t>(((oldstate >> 18u) ^ oldstate) >> 27u);
  uint32_t rot = static_cast<uint32_t>(oldstate >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
inline void pcg32_srandom_r(Pcg* rng, uint64_t initstate, uint64_t initseq) {
  rng->state = 0U;
  rng->inc = (initseq << 1u) | 1u;
  pcg32_random_r(rng);
  rng->state += initstate;
  pcg32_random_r(rng);
}
// rand_float — reference include/rng/sampling.h:85-105 (dense float in [0,1) from 64 bits)
inline float rand_float(Pcg& pcg) {
  uint64_t r1 = pcg32_random_r(&pcg);
  uint64_t r2 = pcg32_random_r(&pcg);
  uint64_t u = (r1 << 32ull) | r2;
  uint32_t z = static_cast<uint32_t>(std::countl_zero(u));
  if (z <= 40) {
    uint32_t e = 126 - z;
    uint32_t m = static_cast<uint32_t>(u) & 0x7fffff;
    uint32_t float_bits = e << 23 | m;
    return std::bit_cast<float>(float_bits);
  }
  return 0x1.0p-64f * static_cast<float>(static_cast<uint32_t>(u));
}
// random_x_y_r2 — reference include/rng/sampling.h:228-239
inline vec2 random_x_y_r2(uint32_t n) {
  constexpr float g = 1.32471795724474602596;
  constexpr float a1 = 1.0 - (1.0 / g);
  constexpr float a2 = 1.0 - (1.0 / (g * g));
  float x = a1 * n;
  float y = a2 * n;
  return vec2{x - std::floor(x), y - std::floor(y)};
}

// ============================================================================ warps
// sample_disk — reference include/rng/sampling.h:15-22
inline vec2 sample_disk(float rand1, float rand2) {
  float r = std::sqrt(rand1);
  float phi = 2.f * kPi * rand2;
  return vec2{r * F_cos(phi), r * F_sin(phi)};
}
// sample_sphere — reference include/rng/sampling.h:26-36 (unqualified cos/sin -> double)
inline vec3 sample_sphere(float rand1, float rand2) {
  float phi = 2 * kPi * rand1;
  float cos_theta = 2 * rand2 - 1;
  float sin_theta = static_cast<float>(::sqrt(static_cast<double>(1 - cos_theta * cos_theta)));
  float x = ::cos(static_cast<double>(phi)) * sin_theta;
  float y = ::sin(static_cast<double>(phi)) * sin_theta;
  return vec3{x, y, cos_theta};
}
// std::lerp(float a, float b, float t) — libstdc++ <cmath> __lerp (Q17: exact at the ends,
// monotonic); restated because the GPU has no std::lerp.
inline float std_lerp(float a, float b, float t) {
  if ((a <= 0 && b >= 0) || (a >= 0 && b <= 0)) return t * b + (1 - t) * a;
  if (t == 1) return b;
  const float x = a + t * (b - a);
  return (t > 1) == (b > a) ? (b < x ? x : b) : (b > x ? x : b);
}
// sample_sphere_cap — reference include/rng/sampling.h:40-51
inline vec3 sample_sphere_cap(float rand1, float rand2, float cos_theta_max) {
  float phi = 2 * kPi * rand1;
  float cos_theta = std_lerp(cos_theta_max, 1.0f, rand2);
  float sin_theta = sqrtf(1 - cos_theta * cos_theta);
  float x = ::cos(static_cast<double>(phi)) * sin_theta;
  float y = ::sin(static_cast<double>(phi)) * sin_theta;
  return vec3{x, y, cos_theta};
}
// sample_hemisphere_cosine — reference include/rng/sampling.h:69-79
inline vec3 sample_hemisphere_cosine(float rand1, float rand2) {
  float phi = 2 * kPi * rand1;
  float cos_theta = std::sqrt(rand2);
  float sin_theta = std::sqrt(1 - cos_theta * cos_theta);
  float x = F_cos(phi) * sin_theta;
  float y = F_sin(phi) * sin_theta;
  return vec3{x, y, cos_theta};
}

// ============================================================================ records
// RayCone, Ray — reference include/ray.h:11-42
struct RayCone {
  float cone_width;
  float spread_angle;
};
struct Ray {
  vec3 dir{1.f, 1.f, 1.f};
  vec3 o{0.f, 0.f, 0.f};
  float minT = 0.0001f;
  float maxT = kInf;
  RayCone ray_cone{0.f, 0.f};   // the reference leaves it uninitialised for shadow rays (Q11)
  Ray() = default;
  Ray(vec3 o_, vec3 d_) : dir(d_), o(o_) {}
  Ray(vec3 o_, vec3 d_, RayCone c) : dir(d_), o(o_), ray_cone(c) {}
};
// ONB, HitInfo, ForHitInfo, EmitterInfo — reference include/hit_utils.h:17-81
struct ONB {
  vec3 u, v, w;
};
struct HitInfo {
  uint32_t mat;    // Material* -> index
  uint32_t prim;   // const Emitter* obj -> index into prims[]
  vec3 hit_p, hit_n_s, hit_n_g;
  vec2 uv, metal_rough_uv;
  ONB n_frame;
  float primitive_area, tex_coord_area, mean_curvature;
};
struct ForHitInfo {
  float e0, e1, e2, invDet;
  uint32_t prim;
  bool valid;
};
struct EmitterInfo {
  vec3 wi;
  float pdf, dist, G;
};
// ScatterInfo — reference include/material/material.h:13-17
struct ScatterInfo {
  vec3 wo;
  float eta;
  bool is_specular;
  bool valid;
};
inline ScatterInfo no_scatter() { return ScatterInfo{vec3{0, 0, 0}, 0.f, false, false}; }

struct Counters {
  uint64_t closest = 0, shadow = 0, internal = 0, leaf = 0, prim = 0, sphere = 0;
};

// xform_with_onb / project_onto_onb / GramSchmidt / get_axis / init_onb
// — reference include/hit_utils.h:32-59
inline vec3 xform_with_onb(const ONB& onb, vec3 v) { return onb.u * v.x + onb.v * v.y + onb.w * v.z; }
inline vec3 project_onto_onb(const ONB& onb, vec3 v) {
  return vec3{dot(v, onb.u), dot(v, onb.v), dot(v, onb.w)};
}
inline vec3 GramSchmidt(vec3 v, vec3 w) { return v - dot(v, w) * w; }
inline void get_axis(vec3 n, vec3& a, vec3& b) {
  if (n.z < (-0.9999999f)) {
    a = vec3{0, -1, 0};
    b = vec3{-1, 0, 0};
  } else {
    float aa = 1.f / (1.f + n.z);
    float bb = -n.x * n.y * aa;
    a = vec3{1.f - n.x * n.x * aa, bb, -n.x};
    b = vec3{bb, 1 - n.y * n.y * aa, -n.y};
  }
}
inline ONB init_onb(vec3 n) {
  ONB o;
  get_axis(n, o.u, o.v);
  o.w = n;
  return o;
}
inline float luminance(vec3 v) {   // reference include/color_utils.h:9-11
  return dot(v, vec3{0.212671f, 0.715160f, 0.072169f});
}
inline float raise_to_power_5(float b) { return b * b * b * b * b; }   // material.h:75

// ============================================================================ ray cones
// reference include/ray.h:44-174
inline RayCone raycone_for_primary_ray(float vfov, uint32_t pixel_height) {
  float spread_angle = F_atan(2.f * (F_tan(vfov / 2.f)) / static_cast<float>(pixel_height));
  return RayCone{0.f, spread_angle};
}
inline float float_sign(float in) { return in > 0.f ? 1.f : -1.f; }
inline float spread_angle_from_curvature(float mean_curvature, float rayConeWidth, vec3 rayDir,
                                         vec3 normal) {
  float dn = -dot(rayDir, normal);
  dn = std::abs(dn) < 1.0e-5 ? float_sign(dn) * 1.0e-5 : dn;   // double literals: compare and
                                                               // product in double, stored float
  float deltaPhi = (mean_curvature * rayConeWidth / dn);
  return deltaPhi;
}
inline RayCone propagate_reflect_cone(const RayCone& cone, float surface_spread_angle,
                                      float hit_dist) {
  float new_cone_width = std::abs(cone.spread_angle * hit_dist + cone.cone_width);
  float new_spread_angle = cone.spread_angle + surface_spread_angle;
  return RayCone{new_cone_width, new_spread_angle};
}
inline bool refract_with_TIR2D(vec2 rayDir, vec2 normal, float eta, vec2& out) {
  float NdotD = dot(normal, rayDir);
  float k = 1.0f - eta * eta * (1.0f - NdotD * NdotD);
  if (k < 0.0f) return false;
  out = rayDir * eta - normal * (eta * NdotD + std::sqrt(k));
  return true;
}
inline void rotate2DPlusMinus(vec2 v, float angle, vec2& plus, vec2& minus) {
  float c = ::cos(static_cast<double>(angle));   // unqualified cos(float) -> double
  float s = ::sin(static_cast<double>(angle));
  float cx = c * v.x, sy = s * v.y, sx = s * v.x, cy = c * v.y;
  plus = vec2{cx - sy, +sx + cy};
  minus = vec2{cx + sy, -sx + cy};
}
inline vec2 orthogonal(vec2 v) { return vec2{-v.y, v.x}; }
inline RayCone propagate_refract_cone(const RayCone& rayCone, vec3 ray_in_dir, vec3 /*hitPoint*/,
                                      float surface_spread_angle, float eta,
                                      vec3 refractedRayDir) {
  vec3 normal = -(eta * refractedRayDir + ray_in_dir) / length(eta * refractedRayDir + ray_in_dir);
  vec3 xAxis = normalize(ray_in_dir - normal * dot(normal, ray_in_dir));
  vec3 yAxis = normal;
  vec2 refractedDir2D{dot(refractedRayDir, xAxis), dot(refractedRayDir, yAxis)};
  vec2 incidentDir2D{dot(ray_in_dir, xAxis), dot(ray_in_dir, yAxis)};
  vec2 incidentDirOrtho2D = orthogonal(incidentDir2D);
  float widthSign = rayCone.cone_width > 0.0f ? 1.0f : -1.0f;
  vec2 incidentDir2D_u, incidentDir2D_l;
  rotate2DPlusMinus(incidentDir2D, rayCone.spread_angle * widthSign * 0.5f, incidentDir2D_u,
                    incidentDir2D_l);
  vec2 tu = incidentDirOrtho2D * rayCone.cone_width * 0.5f;
  vec2 tl = -tu;
  float hitPoint_u_x = tu.x + incidentDir2D_u.x * (-tu.y / incidentDir2D_u.y);
  float hitPoint_l_x = tl.x + incidentDir2D_l.x * (-tl.y / incidentDir2D_l.y);
  float normalSign = hitPoint_u_x > hitPoint_l_x ? +1.0f : -1.0f;
  vec2 normal2D{0.0f, 1.0f};
  vec2 normal2D_u, normal2D_l;
  rotate2DPlusMinus(normal2D, -surface_spread_angle * normalSign * 0.5f, normal2D_u, normal2D_l);
  vec2 refractedDir2D_u, refractedDir2D_l;
  if (!refract_with_TIR2D(incidentDir2D_u, normal2D_u, eta, refractedDir2D_u)) {
    refractedDir2D_u = incidentDir2D_u - normal2D_u * dot(normal2D_u, incidentDir2D_u);
    refractedDir2D_u = normalize(refractedDir2D_u);
  }
  if (!refract_with_TIR2D(incidentDir2D_l, normal2D_l, eta, refractedDir2D_l)) {
    refractedDir2D_l = incidentDir2D_l - normal2D_l * dot(normal2D_l, incidentDir2D_l);
    refractedDir2D_l = normalize(refractedDir2D_l);
  }
  float signA = (refractedDir2D_u.x * refractedDir2D_l.y - refractedDir2D_u.y * refractedDir2D_l.x)
                            * normalSign < 0.0f
                    ? +1.0f
                    : -1.0f;
  float spreadAngle = F_acos(dot(refractedDir2D_u, refractedDir2D_l)) * signA;
  if (std::isnan(spreadAngle)) spreadAngle = 0.f;
  vec2 refractDirOrtho2D = orthogonal(refractedDir2D);
  float width = (-hitPoint_u_x * refractedDir2D_u.y)
                / dot(refractDirOrtho2D, orthogonal(refractedDir2D_u));
  width += (hitPoint_l_x * refractedDir2D_l.y)
           / dot(refractDirOrtho2D, orthogonal(refractedDir2D_l));
  return RayCone{width, spreadAngle};
}

// ============================================================================ scene access
struct Ctx {
  const VimgScene* s;
  // TLCam derived members — reference src/tl_camera.cpp:6-23
  float p_size0, p_size1;
  Counters* cnt;
};

inline vec3 load3(const float* p) { return vec3{p[0], p[1], p[2]}; }

struct TriVerts {
  vec3 p0, p1, p2;
  const VimgMesh* mesh;
  uint32_t i0, i1, i2;   // global vertex ids
};
inline TriVerts tri_verts(const VimgScene* s, uint32_t tri) {
  const VimgMesh* m = &s->meshes[s->tri_mesh[tri]];
  uint32_t i0 = m->first_vertex + s->tri_indices[tri * 3 + 0];
  uint32_t i1 = m->first_vertex + s->tri_indices[tri * 3 + 1];
  uint32_t i2 = m->first_vertex + s->tri_indices[tri * 3 + 2];
  return TriVerts{load3(s->vertices + 3 * size_t{i0}), load3(s->vertices + 3 * size_t{i1}),
                  load3(s->vertices + 3 * size_t{i2}), m, i0, i1, i2};
}
inline vec2 mesh_uv(const VimgScene* s, const VimgMesh* m, uint32_t set, uint32_t global_vertex) {
  const float* p = s->uvs + 2 * (size_t{m->uv_offset[set]} + (global_vertex - m->first_vertex));
  return vec2{p[0], p[1]};
}

// ============================================================================ textures
// handle_wrapping — reference include/texture/texture_common.h:22-53
inline float handle_wrapping(float coord, uint32_t mode) {
  switch (mode) {
    case VIMG_WRAP_CLAMP:
      return clampf(coord, 0.f, 1.f);
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
      return clampf(coord, 0.f, 1.f);
  }
}
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (hi < v ? hi : v); }

// ImageTexture::col_at_uv_mipmap — reference src/image_texture.cpp:132-160
vec3 col_at_uv_mipmap(const VimgScene* s, const VimgTexture& t, int level, vec2 uv) {
  uint32_t mip_w = std::max(t.width >> level, 1u);
  uint32_t mip_h = std::max(t.height >> level, 1u);
  float pixel_u = handle_wrapping(uv.x, t.wrap_u) * mip_w;
  float pixel_v = handle_wrapping(uv.y, t.wrap_v) * mip_h;
  int curr_x = clampi(static_cast<int>(pixel_u), 0, static_cast<int>(mip_w) - 1);
  int curr_y = clampi(static_cast<int>(pixel_v), 0, static_cast<int>(mip_h) - 1);
  int next_x = clampi(curr_x + 1, 0, static_cast<int>(mip_w) - 1);
  int next_y = clampi(curr_y + 1, 0, static_cast<int>(mip_h) - 1);
  float x_fraction = pixel_u - curr_x;
  float y_fraction = pixel_v - curr_y;
  const float* base = s->texels + 3 * t.level_offset[level];
  auto at = [&](int x, int y) { return load3(base + 3 * (size_t(x) + size_t(y) * mip_w)); };
  vec3 a = mix(at(curr_x, curr_y), at(next_x, curr_y), x_fraction);
  vec3 b = mix(at(curr_x, next_y), at(next_x, next_y), x_fraction);
  return mix(a, b, y_fraction);
}
// ImageTexture::col_mipmap_interpolate — reference src/image_texture.cpp:174-189
vec3 col_mipmap_interpolate(const VimgScene* s, const VimgTexture& t, float lambda, vec2 uv) {
  const int last = static_cast<int>(t.num_levels - 1);
  lambda = clampf(lambda, 0.f, static_cast<float>(t.num_levels - 1));
  int level0 = clampi(static_cast<int>(std::floor(lambda)), 0, last);
  int level1 = clampi(level0 + 1, 0, last);
  float fraction = lambda - std::floor(lambda);
  vec3 col0 = col_at_uv_mipmap(s, t, level0, uv);
  vec3 col1 = col_at_uv_mipmap(s, t, level1, uv);
  return mix(col0, col1, fraction);
}
// ImageTexture::compute_texture_LOD — reference include/texture/texture_RGB.h:138-149
// (CompileConsts::mipmap0 == false, include/comptime_settings.h:5)
inline float compute_texture_LOD(const VimgTexture& t, vec3 ray_dir, const RayCone& cone,
                                 const HitInfo& surf) {
  float lambda = 0.5f * F_log2((surf.tex_coord_area) / surf.primitive_area);
  lambda += F_log2(std::abs(cone.cone_width) / std::abs(dot(ray_dir, surf.hit_n_g)));
  lambda += 0.5f * ::log2(static_cast<double>(t.width * t.height));   // log2(uint) -> double
  return std::isnan(lambda) ? 0.f : lambda;
}
// TextureRGB::col_at_ray_hit for ConstColor / Checkerboard / ImageTexture
// — reference include/texture/texture_RGB.h:45-81, src/image_texture.cpp:162-172
vec3 col_at_ray_hit(const VimgScene* s, int tex, vec3 ray_in_dir, const RayCone& cone,
                    const HitInfo& hit) {
  const VimgTexture& t = s->textures[tex];
  switch (t.type) {
    case VIMG_TEX_CONST:
      return load3(t.col_a);
    case VIMG_TEX_CHECKER: {
      uint32_t u_board = static_cast<uint32_t>(std::floor(hit.uv.x * t.width));
      uint32_t v_board = static_cast<uint32_t>(std::floor(hit.uv.y * t.height));
      return ((u_board + v_board) % 2 == 0) ? load3(t.col_a) : load3(t.col_b);
    }
    default: {
      float lambda = compute_texture_LOD(t, ray_in_dir, cone, hit) - 2.f;
      return col_mipmap_interpolate(s, t, lambda, hit.uv);
    }
  }
}
// ImageTexture::get_normal — reference src/image_texture.cpp:277-279
inline vec3 get_normal(const VimgScene* s, int tex, vec2 uv) {
  return normalize(col_at_uv_mipmap(s, s->textures[tex], 0, uv));
}
// TextureRG::get_at_uv — reference include/texture/texture_RG.h:32-57, including the
// "* height" indexing of the +x neighbours (Q6)
vec2 rg_get_at_uv(const VimgScene* s, int tex, vec2 uv) {
  const VimgTextureRG& t = s->rg_textures[tex];
  float pixel_u = handle_wrapping(uv.x, t.wrap_u) * t.width;
  float pixel_v = handle_wrapping(uv.y, t.wrap_v) * t.height;
  int curr_x = clampi(static_cast<int>(pixel_u), 0, static_cast<int>(t.width) - 1);
  int curr_y = clampi(static_cast<int>(pixel_v), 0, static_cast<int>(t.height) - 1);
  int next_x = clampi(curr_x + 1, 0, static_cast<int>(t.width) - 1);
  int next_y = clampi(curr_y + 1, 0, static_cast<int>(t.height) - 1);
  float x_fraction = pixel_u - curr_x;
  float y_fraction = pixel_v - curr_y;
  const float* base = s->rg_texels + 2 * t.offset;
  auto at = [&](size_t i) { return vec2{base[2 * i], base[2 * i + 1]}; };
  vec2 x0 = at(curr_x + size_t(curr_y) * t.width);
  vec2 x1 = at(next_x + size_t(curr_y) * t.height);
  vec2 a = mix(x0, x1, x_fraction);
  vec2 y0 = at(curr_x + size_t(next_y) * t.width);
  vec2 y1 = at(next_x + size_t(next_y) * t.height);
  vec2 b = mix(y0, y1, x_fraction);
  return mix(a, b, y_fraction);
}

// ============================================================================ camera
// TLCam::generate_ray — reference src/tl_camera.cpp:25-53; Ray::xform_ray include/ray.h:36-41
Ray generate_ray(const Ctx& c, float x, float y, float rand1, float rand2) {
  const VimgCamera& cam = c.s->camera;
  float x_dir = (c.p_size0 * (x / cam.res_x)) - (c.p_size0 / 2.0f);
  float y_dir = (c.p_size1 * (y / cam.res_y)) - (c.p_size1 / 2.0f);
  vec3 ray_dir = normalize(vec3{x_dir, y_dir, -1.0f});
  Ray r(vec3{0.f, 0.f, 0.f}, ray_dir);
  if (cam.aperture_radius > 0.f) {
    vec2 d = cam.aperture_radius * sample_disk(rand1, rand2);
    vec3 ray_origin{d.x, d.y, 0.f};
    float ft = cam.focal_dist / std::abs(ray_dir.z);
    vec3 focal_plane_p = ray_dir * ft;
    r.o = ray_origin;
    r.dir = normalize(focal_plane_p - ray_origin);
  }
  vec4 d4 = mat_mul(cam.cam_to_world, vec4{r.dir.x, r.dir.y, r.dir.z, 0.0f});
  r.dir = vec3{d4.x, d4.y, d4.z};
  vec4 o4 = mat_mul(cam.cam_to_world, vec4{r.o.x, r.o.y, r.o.z, 1.0f});
  r.o = vec3{o4.x / o4.w, o4.y / o4.w, o4.z / o4.w};
  r.dir = normalize(r.dir);
  r.ray_cone = raycone_for_primary_ray((cam.vfov_deg * kPi) / 180.f,
                                       static_cast<uint32_t>(cam.res_y));
  return r;
}

// ============================================================================ intersection
// slab_intersect_aabb_array — reference include/hit_utils.h:134-151 (scalar path, exact 1/x: Q9)
inline float slab_intersect_aabb_array(const Ray& ray, vec3 inv, const float* bb_min,
                                       const float* bb_max) {
  vec3 tLower = (load3(bb_min) - ray.o) * inv;
  vec3 tUpper = (load3(bb_max) - ray.o) * inv;
  vec3 lo = vmin(tLower, tUpper), hi = vmax(tLower, tUpper);
  float tBoxMin = std::max(lo.x, std::max(lo.y, std::max(lo.z, ray.minT)));
  float tBoxMax = std::min(hi.x, std::min(hi.y, std::min(hi.z, ray.maxT)));
  return (tBoxMin <= tBoxMax) ? tBoxMin : kInf;
}

#if defined(ORACLE_AVX2_TIMING) && ORACLE_AVX2_TIMING
// ---- TIMING BUILD ONLY (liboracle_avx2.so, `make oracle-avx2`): the reference's AVX2 slab path,
// which is what its release binary runs on an AVX2 machine - ray_1aabb_slab / ray_2aabb_slab
// (reference include/simd_hit.h:37-156) on a ~12-bit reciprocal of the direction
// (_mm256_rcp_ps, include/bvh.h:109-116).  Its results differ from the scalar path's (SURVEY
// quirk Q9: 27 % of config-2 pixels at 4 spp), so this build is never a parity partner: bench.py
// times it as the CPU baseline SURVEY.md 8d defines, next to the scalar build.
struct SimdRay {
  __m256 o, inv;
};
inline SimdRay simd_ray(const Ray& ray) {
  SimdRay r;
  __m256 d = _mm256_set_ps(1.f, ray.dir.z, ray.dir.y, ray.dir.x, 1.f, ray.dir.z, ray.dir.y, ray.dir.x);
  r.inv = _mm256_rcp_ps(d);
  r.o = _mm256_set_ps(0.f, ray.o.z, ray.o.y, ray.o.x, 0.f, ray.o.z, ray.o.y, ray.o.x);
  return r;
}
inline float hmax4(__m128 x) {   // max of the four lanes (horizontal_max_128, simd_hit.h:13-20)
  __m128 s1 = _mm_shuffle_ps(x, x, _MM_SHUFFLE(0, 0, 3, 2));
  __m128 m1 = _mm_max_ps(x, s1);
  __m128 s2 = _mm_shuffle_ps(m1, m1, _MM_SHUFFLE(0, 0, 0, 1));
  return _mm_cvtss_f32(_mm_max_ps(m1, s2));
}
inline float hmin4(__m128 x) {   // horizontal_min_128, simd_hit.h:23-33
  __m128 s1 = _mm_shuffle_ps(x, x, _MM_SHUFFLE(0, 0, 3, 2));
  __m128 m1 = _mm_min_ps(x, s1);
  __m128 s2 = _mm_shuffle_ps(m1, m1, _MM_SHUFFLE(0, 0, 0, 1));
  return _mm_cvtss_f32(_mm_min_ps(m1, s2));
}
// ray_1aabb_slab, simd_hit.h:37-68: one box, the fourth lane carries minT / maxT
inline float simd_slab1(const float* mins, const float* maxs, const SimdRay& sr, const Ray& r) {
  const __m128 o = _mm256_castps256_ps128(sr.o), inv = _mm256_castps256_ps128(sr.inv);
  __m128 lo = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(mins), o), inv);
  __m128 hi = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(maxs), o), inv);
  __m128 mn = _mm_min_ps(lo, hi), mx = _mm_max_ps(lo, hi);
  float t_min = hmax4(_mm_insert_ps(mn, _mm_set_ss(r.minT), 0b00110000));
  float t_max = hmin4(_mm_insert_ps(mx, _mm_set_ss(r.maxT), 0b00110000));
  return (t_min <= t_max) ? t_min : kInf;
}
// ray_2aabb_slab, simd_hit.h:121-156: both sibling boxes in one 8-lane pass.  `mins` points at
// left.min (3 floats) followed by right.min, `maxs` at left.max followed by right.max, as
// BB_mins_maxes stores them; the permute sorts the eight loaded floats into two 4-lane halves.
inline void simd_slab2(const float* mins, const float* maxs, const SimdRay& sr, const Ray& r, float& t1,
                       float& t2) {
  const __m256i perm = _mm256_set_epi32(6, 5, 4, 3, 7, 2, 1, 0);
  __m256 lo = _mm256_permutevar8x32_ps(_mm256_loadu_ps(mins), perm);
  lo = _mm256_mul_ps(_mm256_sub_ps(lo, sr.o), sr.inv);
  __m256 hi = _mm256_permutevar8x32_ps(_mm256_loadu_ps(maxs), perm);
  hi = _mm256_mul_ps(_mm256_sub_ps(hi, sr.o), sr.inv);
  __m256 mn = _mm256_min_ps(lo, hi), mx = _mm256_max_ps(lo, hi);
  mn = _mm256_blend_ps(mn, _mm256_set1_ps(r.minT), 0b10001000);
  mx = _mm256_blend_ps(mx, _mm256_set1_ps(r.maxT), 0b10001000);
  const float lo1 = hmax4(_mm256_castps256_ps128(mn)), lo2 = hmax4(_mm256_extractf128_ps(mn, 1));
  const float hi1 = hmin4(_mm256_castps256_ps128(mx)), hi2 = hmin4(_mm256_extractf128_ps(mx, 1));
  t1 = (lo1 <= hi1) ? lo1 : kInf;
  t2 = (lo2 <= hi2) ? lo2 : kInf;
}
#endif

// difference_of_products(_double) — reference include/geometry/triangle.h:11-21
inline float difference_of_products(float a, float b, float c, float d) {
  float cd = c * d;
  return std::fma(a, b, -cd);
}
inline double difference_of_products_double(float a, float b, float c, float d) {
  double cd = c * d;
  return static_cast<double>(::fmal(a, b, -cd));
}
// Triangle::tri_hit_template — reference include/geometry/triangle.h:74-180
bool tri_hit(const VimgScene* s, uint32_t tri, Ray& ray, ForHitInfo* out) {
  TriVerts tv = tri_verts(s, tri);
  vec3 p0 = tv.p0, p1 = tv.p1, p2 = tv.p2;
  vec3 edge1 = p1 - p0, edge2 = p2 - p0;
  if (length2(cross(edge2, edge1)) == 0.f) return false;
  vec3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
  // max_componenet_index
  vec3 a = vabs(ray.dir);
  int kz = 0;
  float max_val = a.x;
  if (a.y > max_val) { kz = 1; max_val = a.y; }
  if (a.z > max_val) { kz = 2; max_val = a.z; }
  int kx = kz + 1; if (kx == 3) kx = 0;
  int ky = kx + 1; if (ky == 3) ky = 0;
  auto permute = [&](vec3 v) { return vec3{v[kx], v[ky], v[kz]}; };
  vec3 d = permute(ray.dir);
  p0t = permute(p0t); p1t = permute(p1t); p2t = permute(p2t);
  float Sx = -d.x / d.z, Sy = -d.y / d.z, Sz = 1.f / d.z;
  p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
  p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
  p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
  float e0 = difference_of_products(p1t.x, p2t.y, p1t.y, p2t.x);
  float e1 = difference_of_products(p2t.x, p0t.y, p2t.y, p0t.x);
  float e2 = difference_of_products(p0t.x, p1t.y, p0t.y, p1t.x);
  if (e0 == 0.f || e1 == 0.f || e2 == 0.f) {
    e0 = static_cast<float>(difference_of_products_double(p1t.x, p2t.y, p1t.y, p2t.x));
    e1 = static_cast<float>(difference_of_products_double(p2t.x, p0t.y, p2t.y, p0t.x));
    e2 = static_cast<float>(difference_of_products_double(p0t.x, p1t.y, p0t.y, p1t.x));
  }
  if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
  float det = e0 + e1 + e2;
  if (det == 0) return false;
  p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
  float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
  if (det < 0 && (tScaled >= 0 || tScaled < ray.maxT * det || tScaled > ray.minT * det))
    return false;
  else if (det > 0 && (tScaled <= 0 || tScaled > ray.maxT * det || tScaled < ray.minT * det))
    return false;
  float invDet = 1.f / det;
  float t = tScaled * invDet;
  ray.maxT = t;
  if (out) {
    out->e0 = e0; out->e1 = e1; out->e2 = e2; out->invDet = invDet;
  }
  return true;
}

// solveQuadratic + Sphere::sphere_hit_template — reference include/geometry/sphere.h:13-100
// (`sqrt(a*discriminant)` is unqualified: double sqrt, and `q` is a double)
bool sphere_hit(const VimgSphere& sp, Ray& r) {
  float t0, t1;
  const float radius = sp.radius;
  const vec3 center = load3(sp.center);
  const float radius_squared = radius * radius;
  vec3 f = r.o - center;
  const float a = dot(r.dir, r.dir);
  const float b_prime = dot(-1.0f * f, r.dir);
  const float c = dot(f, f) - radius_squared;
  const vec3 temp = f + (b_prime / a) * r.dir;
  const float discriminant = radius_squared - (dot(temp, temp));
  if (discriminant < 0) return false;
  {
    float sign = (b_prime > 0) ? 1.0f : -1.0f;
    double q = b_prime + sign * (::sqrt(static_cast<double>(a * discriminant)));
    if (discriminant == 0) {
      t0 = t1 = c / q;
    } else {
      t0 = c / q;
      t1 = q / a;
    }
    if (t0 > t1) std::swap(t0, t1);
  }
  if (t0 < r.minT || t0 > r.maxT) {
    t0 = t1;
    if (t0 < r.minT || t0 > r.maxT) return false;
  }
  r.maxT = t0;
  return true;
}

// Triangle::hit_info — reference src/geometry/triangle.cpp:13-153
HitInfo tri_hit_info(const VimgScene* s, uint32_t prim_id, uint32_t tri, const Ray& /*r*/,
                     const ForHitInfo& pre) {
  TriVerts tv = tri_verts(s, tri);
  const VimgMesh* mesh = tv.mesh;
  vec3 p0 = tv.p0, p1 = tv.p1, p2 = tv.p2;
  vec3 edge1 = p1 - p0, edge2 = p2 - p0;
  float u = pre.e0 * pre.invDet, v = pre.e1 * pre.invDet, w = pre.e2 * pre.invDet;
  const vec3 tri_normal = normalize(cross(edge1, edge2));
  vec3 n0, n1, n2, shading_normal;
  if (mesh->has_normals) {
    n0 = load3(s->normals + 3 * size_t{tv.i0});
    n1 = load3(s->normals + 3 * size_t{tv.i1});
    n2 = load3(s->normals + 3 * size_t{tv.i2});
    shading_normal = normalize(u * n0 + v * n1 + w * n2);
  } else {
    n0 = tri_normal, n1 = tri_normal, n2 = tri_normal;
    shading_normal = tri_normal;
  }
  const vec3 hit_p = u * p0 + v * p1 + w * p2;
  vec2 uv{u, v};
  vec2 uv0{0, 0}, uv1{1, 0}, uv2{1, 1};
  if (mesh->color_tex_uv != VIMG_NO_UV) {
    uv0 = mesh_uv(s, mesh, mesh->color_tex_uv, tv.i0);
    uv1 = mesh_uv(s, mesh, mesh->color_tex_uv, tv.i1);
    uv2 = mesh_uv(s, mesh, mesh->color_tex_uv, tv.i2);
    uv = u * uv0 + v * uv1 + w * uv2;
  }
  vec2 metallic_roughness_uv = uv;
  if (mesh->metallic_roughness_tex_uv != VIMG_NO_UV) {
    vec2 m0 = mesh_uv(s, mesh, mesh->metallic_roughness_tex_uv, tv.i0);
    vec2 m1 = mesh_uv(s, mesh, mesh->metallic_roughness_tex_uv, tv.i1);
    vec2 m2 = mesh_uv(s, mesh, mesh->metallic_roughness_tex_uv, tv.i2);
    metallic_roughness_uv = u * m0 + v * m1 + w * m2;
  }
  vec2 duvds = uv2 - uv0;
  vec2 duvdt = uv2 - uv1;
  float det = duvds.x * duvdt.y - duvdt.x * duvds.y;
  float dsdu = 0.f, dtdu = 0.f, dsdv = 0.f, dtdv = 0.f;
  vec3 dpdu, dpdv;
  if (std::abs(det) > 1e-8f && !std::isnan(det)) {
    dsdu = duvdt.y / det;
    dtdu = -duvds.y / det;
    dsdv = duvdt.x / det;
    dtdv = -duvds.x / det;
    vec3 dpds = p2 - p0;
    vec3 dpdt = p2 - p1;
    dpdu = dpds * dsdu + dpdt * dtdu;
    dpdv = dpds * dsdv + dpdt * dtdv;
  } else {
    get_axis(shading_normal, dpdu, dpdv);
  }
  const VimgMaterial& mat = s->materials[mesh->material];
  if (mat.normal_map >= 0) {
    vec2 n_uv{u, v};
    if (mesh->normal_tex_uv != VIMG_NO_UV) {
      vec2 q0 = mesh_uv(s, mesh, mesh->normal_tex_uv, tv.i0);
      vec2 q1 = mesh_uv(s, mesh, mesh->normal_tex_uv, tv.i1);
      vec2 q2 = mesh_uv(s, mesh, mesh->normal_tex_uv, tv.i2);
      n_uv = u * q0 + v * q1 + w * q2;
    }
    vec3 n_tangent_space = get_normal(s, mat.normal_map, n_uv);
    ONB onb_n_map = init_onb(shading_normal);
    vec3 local_space_normal = xform_with_onb(onb_n_map, n_tangent_space);
    float ulen = length(dpdu), vlen = length(dpdv);
    dpdu = normalize(GramSchmidt(dpdu, local_space_normal)) * ulen;
    dpdv = normalize(cross(local_space_normal, dpdu)) * vlen;
    shading_normal = local_space_normal;
  }
  vec3 tangent = normalize(dpdu - shading_normal * dot(shading_normal, dpdu));
  vec3 dnds = n2 - n0;
  vec3 dndt = n2 - n1;
  vec3 dndu = dnds * dsdu + dndt * dtdu;
  vec3 dndv = dnds * dsdv + dndt * dtdv;
  vec3 bitangent = normalize(cross(shading_normal, tangent));
  float mean_curvature = (dot(dndu, tangent) + dot(dndv, bitangent)) / 2.f;
  float twice_tri_area = length(cross(p1 - p0, p2 - p0));
  float uv_area = std::abs((uv1.x - uv0.x) * (uv2.y - uv0.y) - (uv2.x - uv0.x) * (uv1.y - uv0.y));
  HitInfo h;
  h.mat = mesh->material;
  h.prim = prim_id;
  h.hit_p = hit_p;
  h.hit_n_s = shading_normal;
  h.hit_n_g = tri_normal;
  h.uv = uv;
  h.metal_rough_uv = metallic_roughness_uv;
  h.n_frame = ONB{tangent, bitangent, shading_normal};
  h.primitive_area = twice_tri_area;   // Q13: twice the area
  h.tex_coord_area = uv_area;
  h.mean_curvature = mean_curvature;
  return h;
}

// Sphere::hit_info — reference src/geometry/sphere.cpp:12-45 (dpdv is computed there and never
// read; omitted)
HitInfo sphere_hit_info(const VimgScene* s, uint32_t prim_id, const VimgSphere& sp, const Ray& r) {
  const vec3 center = load3(sp.center);
  const vec3 hit_p = r.o + r.dir * r.maxT;
  const vec3 normal = normalize(hit_p - center);
  float theta = F_acos(-normal.y);
  float phi = F_atan2(-normal.z, normal.x) + kPi;
  float u = phi / (2.f * kPi);
  float v = theta / kPi;
  vec3 dpdu{-sp.radius * normal.y, sp.radius * normal.x, 0.f};
  vec3 tangent = normalize(dpdu - normal * dot(normal, dpdu));
  HitInfo h;
  h.mat = sp.material;
  h.prim = prim_id;
  h.hit_p = hit_p;
  h.hit_n_s = normal;
  h.hit_n_g = normal;
  h.uv = vec2{u, v};
  h.metal_rough_uv = vec2{u, v};
  h.n_frame = ONB{tangent, normalize(cross(normal, tangent)), normal};
  h.primitive_area = 1.f;
  h.tex_coord_area = 0.000001f;
  h.mean_curvature = 1.f / sp.radius;
  (void)s;
  return h;
}

// BVH::hit<T> — reference include/bvh.h:83-225, scalar slab path (:119-122,:184-187).
// closest: returns true and fills `out`; any_hit: returns true on the first hit.
// The traversal stack is per query here; the reference shares one per thread and leaves stale
// entries after an any-hit early return (Q7: result-neutral).
template <bool ANY_HIT>
bool bvh_hit(const Ctx& c, Ray& ray, HitInfo* out) {
  const VimgScene* s = c.s;
  const VimgBVH& bvh = s->bvh;
  if (bvh.num_nodes == 0) return false;
  if (c.cnt) (ANY_HIT ? c.cnt->shadow : c.cnt->closest)++;
  const float* bb = bvh.bb_mins_maxes;
#if defined(ORACLE_AVX2_TIMING) && ORACLE_AVX2_TIMING
  const SimdRay sr = simd_ray(ray);
  float root_hit = simd_slab1(bb + 0, bb + 6, sr, ray);
#else
  vec3 inv{1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z};
  float root_hit = slab_intersect_aabb_array(ray, inv, bb + 0, bb + 6);
#endif
  if (std::isinf(root_hit)) return false;
  uint32_t stack[128];
  int sp = 0;
  stack[sp++] = 0;
  ForHitInfo inter{0, 0, 0, 0, 0, false};
  while (sp > 0) {
    const VimgBVHNode& node = bvh.nodes[stack[--sp]];
    if (node.obj_count != 0) {
      if (c.cnt) c.cnt->leaf++;
      for (uint32_t i = 0; i < node.obj_count; ++i) {
        uint32_t prim_index = bvh.obj_indices[node.first_index + i];
        const VimgPrim& p = s->prims[prim_index];
        if (c.cnt) {
          c.cnt->prim++;
          if (p.type == VIMG_PRIM_SPHERE) c.cnt->sphere++;
        }
        ForHitInfo tmp{0.f, 0.f, 0.f, 0.f, prim_index, true};
        bool hit = (p.type == VIMG_PRIM_TRIANGLE) ? tri_hit(s, p.index, ray, &tmp)
                                                  : sphere_hit(s->spheres[p.index], ray);
        if (hit) {
          if (ANY_HIT) return true;
          inter = tmp;   // "last success wins", ties included (Q8)
        }
      }
    } else {
      if (c.cnt) c.cnt->internal++;
      uint32_t first_child = node.first_index;
      uint32_t sec_child = first_child + 1;
      size_t l_min = size_t{first_child} * 2 + 2, l_max = l_min + 2;
#if defined(ORACLE_AVX2_TIMING) && ORACLE_AVX2_TIMING
      float bb_hit1, bb_hit2;
      simd_slab2(bb + 3 * l_min, bb + 3 * l_max, sr, ray, bb_hit1, bb_hit2);
#else
      size_t r_min = l_min + 1, r_max = l_max + 1;
      float bb_hit1 = slab_intersect_aabb_array(ray, inv, bb + 3 * l_min, bb + 3 * l_max);
      float bb_hit2 = slab_intersect_aabb_array(ray, inv, bb + 3 * r_min, bb + 3 * r_max);
#endif
      if (ANY_HIT) {
        if (!std::isinf(bb_hit1)) stack[sp++] = first_child;
        if (!std::isinf(bb_hit2)) stack[sp++] = sec_child;
      } else {
        if (!std::isinf(bb_hit2)) {
          if (!std::isinf(bb_hit1)) {
            if (bb_hit2 > bb_hit1) std::swap(first_child, sec_child);
            stack[sp++] = first_child;
          }
          stack[sp++] = sec_child;
        } else if (!std::isinf(bb_hit1)) {
          stack[sp++] = first_child;
        }
      }
    }
  }
  if (!ANY_HIT && inter.valid) {
    const VimgPrim& p = s->prims[inter.prim];
    *out = (p.type == VIMG_PRIM_TRIANGLE) ? tri_hit_info(s, inter.prim, p.index, ray, inter)
                                          : sphere_hit_info(s, inter.prim, s->spheres[p.index], ray);
    return true;
  }
  return false;
}

// BVH::hit<float> — reference include/bvh.h:83-225 with T = float, scalar slab path: the cost of
// one closest-hit style traversal (root test 0.5, a pair of sibling boxes 2 x 0.5 (:189-192), a
// primitive test 1 (:160-162); hit_check shortens ray.maxT as it goes).
float bvh_cost(const Ctx& c, Ray& ray) {
  const VimgScene* s = c.s;
  const VimgBVH& bvh = s->bvh;
  const float intersection_cost = 1.f, traversal_cost = 0.5f;   // BVHConst, bvh.h:17-20
  float return_variable = 0.f;
  if (bvh.num_nodes == 0) return return_variable;
  vec3 inv{1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z};
  const float* bb = bvh.bb_mins_maxes;
  float root_hit = slab_intersect_aabb_array(ray, inv, bb + 0, bb + 6);
  return_variable += traversal_cost;
  if (std::isinf(root_hit)) return return_variable;
  uint32_t stack[128];
  int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    const VimgBVHNode& node = bvh.nodes[stack[--sp]];
    if (node.obj_count != 0) {
      for (uint32_t i = 0; i < node.obj_count; ++i) {
        uint32_t prim_index = bvh.obj_indices[node.first_index + i];
        const VimgPrim& p = s->prims[prim_index];
        ForHitInfo tmp{0.f, 0.f, 0.f, 0.f, prim_index, true};
        if (p.type == VIMG_PRIM_TRIANGLE) tri_hit(s, p.index, ray, &tmp);
        else sphere_hit(s->spheres[p.index], ray);
        return_variable += intersection_cost;
      }
    } else {
      uint32_t first_child = node.first_index;
      uint32_t sec_child = first_child + 1;
      size_t l_min = size_t{first_child} * 2 + 2, l_max = l_min + 2;
      size_t r_min = l_min + 1, r_max = l_max + 1;
      float bb_hit1 = slab_intersect_aabb_array(ray, inv, bb + 3 * l_min, bb + 3 * l_max);
      float bb_hit2 = slab_intersect_aabb_array(ray, inv, bb + 3 * r_min, bb + 3 * r_max);
      return_variable += traversal_cost * 2.f;
      if (!std::isinf(bb_hit2)) {
        if (!std::isinf(bb_hit1)) {
          if (bb_hit2 > bb_hit1) std::swap(first_child, sec_child);
          stack[sp++] = first_child;
        }
        stack[sp++] = sec_child;
      } else if (!std::isinf(bb_hit1)) {
        stack[sp++] = first_child;
      }
    }
  }
  return return_variable;
}

// turbo_colormap — reference src/integrators/heatmap.cpp:21-36; glm::dot of vec4 is
// (x + y) + (z + w) of the products, of vec2 x + y (glm 1.0.1 detail/func_geometric.inl)
vec3 turbo_colormap(float x) {
  const float kR4[4] = {static_cast<float>(0.13572138), static_cast<float>(4.61539260),
                        static_cast<float>(-42.66032258), static_cast<float>(132.13108234)};
  const float kG4[4] = {static_cast<float>(0.09140261), static_cast<float>(2.19418839),
                        static_cast<float>(4.84296658), static_cast<float>(-14.18503333)};
  const float kB4[4] = {static_cast<float>(0.10667330), static_cast<float>(12.64194608),
                        static_cast<float>(-60.58204836), static_cast<float>(110.36276771)};
  const float kR2[2] = {static_cast<float>(-152.94239396), static_cast<float>(59.28637943)};
  const float kG2[2] = {static_cast<float>(4.27729857), static_cast<float>(2.82956604)};
  const float kB2[2] = {static_cast<float>(-89.90310912), static_cast<float>(27.34824973)};
  x = clampf(x, 0.f, 1.f);
  const float v4[4] = {1.0f, x, x * x, x * x * x};
  const float v2[2] = {v4[2] * v4[2], v4[3] * v4[2]};
  auto dot4 = [&](const float* k) { return (v4[0] * k[0] + v4[1] * k[1]) + (v4[2] * k[2] + v4[3] * k[3]); };
  auto dot2 = [&](const float* k) { return v2[0] * k[0] + v2[1] * k[1]; };
  return vec3{dot4(kR4) + dot2(kR2), dot4(kG4) + dot2(kG2), dot4(kB4) + dot2(kB2)};
}

// ============================================================================ materials
inline bool mat_is_emissive(const VimgMaterial& m) { return m.type == VIMG_MAT_DIFFUSE_LIGHT; }
// is_delta: base true (material.h:71), Lambertian/DiffuseLight/Principled false, Dielectric true
inline bool mat_is_delta(const VimgMaterial& m) { return m.type == VIMG_MAT_DIELECTRIC; }
// emitted: base 0 (material.h:63-66); DiffuseLight one-sided (diffuse_light.h:30-38)
inline vec3 mat_emitted(const VimgMaterial& m, vec3 ray_dir, vec3 shading_normal) {
  if (m.type != VIMG_MAT_DIFFUSE_LIGHT) return vec3{0, 0, 0};
  bool front_face = dot(shading_normal, ray_dir) < 0;
  return front_face ? load3(m.emit) : vec3{0.0f, 0.0f, 0.0f};
}

// ---- Lambertian — reference src/material/lambertian.cpp:5-54
ScatterInfo lambertian_sample(const HitInfo& hit, vec3 wi, Pcg& rng) {
  float rand1 = rand_float(rng);
  float rand2 = rand_float(rng);
  bool front_face = dot(wi, hit.hit_n_s) < 0;
  vec3 shading_normal = front_face ? hit.hit_n_s : -hit.hit_n_s;
  ONB onb = init_onb(shading_normal);
  vec3 dir = xform_with_onb(onb, sample_hemisphere_cosine(rand1, rand2));
  if (front_face) return ScatterInfo{dir, 0.f, false, true};
  return no_scatter();
}
void lambertian_eval_pdf(const VimgScene* s, const VimgMaterial& m, vec3 wi, vec3 wo,
                         const HitInfo& hit, const RayCone& cone, vec3& f, float& pdf) {
  float dot_product = static_cast<float>(std::max(0.0f, dot(wo, hit.hit_n_s)) / kPi);
  f = col_at_ray_hit(s, m.tex, wi, cone, hit) * dot_product;
  pdf = dot_product;
}

// ---- Dielectric — reference src/material/dielectric.cpp:5-69
inline vec3 reflect_dir(vec3 wi, vec3 n) { return wi - (2.f * dot(wi, n) * n); }
inline float schlick_apprx(float cosine, float in_ior, float out_ior) {
  float r0 = (in_ior - out_ior) / (in_ior + out_ior);
  r0 = r0 * r0;
  return r0 + (1.f - r0) * raise_to_power_5(1.f - cosine);
}
inline vec3 refract_dir(vec3 wi, vec3 n, float i_over_o, float cos_thetaI,
                        float sin_thetaT_square) {
  float normal_mul = (i_over_o * cos_thetaI) - sqrtf(1.f - sin_thetaT_square);
  return (i_over_o * wi) + (normal_mul * n);
}
ScatterInfo dielectric_sample(const VimgMaterial& m, const HitInfo& hit, vec3 wi, Pcg& rng) {
  const float ior = m.ior;
  vec3 wo;
  float eta;
  bool front_face = dot(wi, hit.hit_n_s) < 0;
  vec3 shading_normal = front_face ? hit.hit_n_s : -hit.hit_n_s;
  const float cos_thetaI = -1.f * (dot(wi, shading_normal));
  float randf = rand_float(rng);
  if (front_face) {
    eta = ior;
    const float schlick = schlick_apprx(cos_thetaI, 1.0f, ior);
    if (schlick > randf) {
      wo = reflect_dir(wi, shading_normal);
    } else {
      const float i_over_o = 1.f / ior;
      const float sin2 = (i_over_o * i_over_o) * (1.f - (cos_thetaI * cos_thetaI));
      wo = refract_dir(wi, shading_normal, i_over_o, cos_thetaI, sin2);
    }
  } else {
    eta = 1.f / ior;
    const float i_over_o = ior;
    const float sin2 = (i_over_o * i_over_o) * (1.f - (cos_thetaI * cos_thetaI));
    if ((sin2 > 1.f) || (schlick_apprx(sqrtf(1.f - sin2), ior, 1.f) > randf)) {
      wo = reflect_dir(wi, shading_normal);
    } else {
      wo = refract_dir(wi, shading_normal, i_over_o, cos_thetaI, sin2);
    }
  }
  return ScatterInfo{wo, eta, true, true};
}

// ---- Disney helpers — reference include/material/disney_helpers/*.h
// G_w — disney_common.h:6-14 (double-promoted by the 1. / 2. literals)
inline float G_w(vec3 w, float alphax, float alphay, const ONB& frame) {
  const vec3 wl = project_onto_onb(frame, w);
  float vec_alpha = ((wl.x * alphax) * (wl.x * alphax) + (wl.y * alphay) * (wl.y * alphay))
                    / (wl.z * wl.z);
  float caret = (::sqrt(1. + static_cast<double>(vec_alpha)) - 1.) / 2.;
  return 1. / (1. + static_cast<double>(caret));
}
// anisotropic_sample_visible_normals — disney_common.h:16-52
vec3 anisotropic_sample_visible_normals(vec3 local_dir_in, float alphax, float alphay, Pcg& rng) {
  float sign = 1.f;
  vec3 top = local_dir_in;
  if (local_dir_in.z < 0.f) {
    sign = -1.f;
    top = -top;
  }
  vec3 hemi_dir_in = normalize(vec3{alphax * top.x, alphay * top.y, top.z});
  float rand_x = rand_float(rng);
  float rand_y = rand_float(rng);
  float phi = 2 * kPi * rand_x;
  float z = std::fma((1.0f - rand_y), (1.0f + hemi_dir_in.z), -hemi_dir_in.z);
  float sinTheta = std::sqrt(clampf(1.0f - z * z, 0.0f, 1.0f));
  float x = sinTheta * ::cos(static_cast<double>(phi));   // unqualified cos/sin -> double
  float y = sinTheta * ::sin(static_cast<double>(phi));
  const vec3 cc{x, y, z};
  const vec3 hemi_N = cc + hemi_dir_in;
  return sign * normalize(vec3{alphax * hemi_N.x, alphay * hemi_N.y, std::max(0.f, hemi_N.z)});
}
// fresnel_dielectric — disney_common.h:54-68
inline float fresnel_dielectric(float n_dot_i, float eta) {
  float n_dot_t_sq = 1.f - (1.f - n_dot_i * n_dot_i) / (eta * eta);
  if (n_dot_t_sq < 0) return 1;
  float n_dot_t = std::sqrt(n_dot_t_sq);
  n_dot_i = std::abs(n_dot_i);
  float rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
  float rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
  return (rs * rs + rp * rp) / 2;
}
// FD — disney_diffuse.h:9-11
inline float FD(vec3 n, vec3 w, float FD_90) {
  return 1.f + (FD_90 - 1.f) * raise_to_power_5(1.f - std::max(dot(n, w), 0.f));
}
// eval_pdf_disney_diffuse — disney_diffuse.h:73-104
void eval_pdf_disney_diffuse(vec3 dir_in, vec3 dir_out, const HitInfo& hit, vec3 base_col,
                             float subsurface, float roughness, vec3 half_vec, const ONB& frame,
                             vec3& f, float& pdf) {
  if (dot(hit.hit_n_g, dir_in) < 0 || dot(hit.hit_n_g, dir_out) < 0) {
    f = vec3{0.f, 0.f, 0.f};
    pdf = 0.f;
    return;
  }
  float normal_dirout_dot = dot(frame.w, dir_out);
  const float cos_theta_out = std::max(normal_dirout_dot, 0.f);
  const float cos_theta_in = std::max(dot(frame.w, dir_in), 0.f);
  const float dot_h_out = std::max(dot(half_vec, dir_out), 0.f);
  const float FD_90 = 0.5 + 2.0 * roughness * dot_h_out * dot_h_out;   // double expression
  const vec3 base_diffuse = base_col * kInvPiF * FD(frame.w, dir_in, FD_90)
                            * FD(frame.w, dir_out, FD_90) * cos_theta_out;
  const float FSS_90 = roughness * dot_h_out * dot_h_out;
  vec3 ss_diffuse = base_col * 1.25f * kInvPiF
                    * (FD(frame.w, dir_in, FSS_90) * FD(frame.w, dir_out, FSS_90)
                           * ((1.f / (cos_theta_out + cos_theta_in)) - 0.5f)
                       + 0.5f)
                    * cos_theta_out;
  f = (1.f - subsurface) * base_diffuse + subsurface * ss_diffuse;
  pdf = std::max(normal_dirout_dot, 0.f) * kInvPiF;
}
// sample_disney_diffuse — disney_diffuse.h:52-71
ScatterInfo sample_disney_diffuse(vec3 dir_in, const HitInfo& hit, const ONB& frame, Pcg& rng) {
  if (dot(hit.hit_n_g, dir_in) < 0) return no_scatter();
  float rand1 = rand_float(rng);
  float rand2 = rand_float(rng);
  vec3 dir_out = xform_with_onb(frame, sample_hemisphere_cosine(rand1, rand2));
  if (dot(hit.hit_n_g, dir_out) <= 0) return no_scatter();
  return ScatterInfo{dir_out, 0.f, false, true};
}
// eval_disney_sheen — disney_sheen.h:10-27
vec3 eval_disney_sheen(vec3 dir_in, vec3 dir_out, const HitInfo& hit, vec3 base_col,
                       float sheen_tint, vec3 half_vec, const ONB& frame) {
  if (dot(hit.hit_n_g, dir_in) < 0 || dot(hit.hit_n_g, dir_out) < 0) return vec3{0, 0, 0};
  const float base_lum = luminance(base_col);
  vec3 C_tint = base_lum > 0 ? base_col / base_lum : v3(1.f);
  vec3 C_sheen = (v3(1.f) - v3(sheen_tint)) + sheen_tint * C_tint;
  return C_sheen * raise_to_power_5(1.f - std::max(dot(half_vec, dir_out), 0.f))
         * std::max(dot(frame.w, dir_out), 0.f);
}
// eval_pdf_disney_clearcoat — disney_clearcoat.h:107-139
void eval_pdf_disney_clearcoat(vec3 dir_in, vec3 dir_out, const HitInfo& hit, float alpha_g,
                               vec3 half_vec, const ONB& frame, vec3& f, float& pdf) {
  if (dot(hit.hit_n_g, dir_in) < 0 || dot(hit.hit_n_g, dir_out) < 0) {
    f = vec3{0.f, 0.f, 0.f};
    pdf = 0.f;
    return;
  }
  constexpr float R0 = ((1.5f - 1.f) * (1.5f - 1.f)) / ((1.5f + 1.f) * (1.5f + 1.f));
  float h_dirout_dot = std::abs(dot(half_vec, dir_out));
  float Fresenl = R0 + (1. - R0) * raise_to_power_5(1.f - h_dirout_dot);
  float G = G_w(dir_in, 0.25, 0.25, frame) * G_w(dir_out, 0.25, 0.25, frame);
  const float alpha_g_square = alpha_g * alpha_g;
  const vec3 local_H = project_onto_onb(frame, half_vec);
  float D = (alpha_g_square - 1.f)
            / (kPi * F_log(alpha_g_square)
               * (1. + (alpha_g_square - 1.) * local_H.z * local_H.z));
  float clearcoat_eval = (Fresenl * D * G) / (4.f * std::abs(dot(frame.w, dir_in)));
  pdf = (D * std::abs(dot(frame.w, half_vec))) / (4.f * h_dirout_dot);
  f = v3(clearcoat_eval);
}
// sample_local_h_clearcoat + sample_disney_clearcoat — disney_clearcoat.h:60-105
ScatterInfo sample_disney_clearcoat(vec3 dir_in, const HitInfo& hit, ONB frame,
                                    float clearcoat_gloss, Pcg& rng, bool regularize) {
  if (dot(hit.hit_n_g, dir_in) < 0) return no_scatter();
  float alpha_g = (1.f - clearcoat_gloss) * 0.1f + clearcoat_gloss * 0.001f;
  if (regularize && alpha_g < 0.1f) alpha_g = clampf(2.f * alpha_g, 0.03f, 0.1f);
  vec3 local_h;
  {
    const float alpha = alpha_g;
    float rand1 = rand_float(rng);
    float rand2 = rand_float(rng);
    float cos_square_elevation
        = (1.f - ::pow(static_cast<double>(alpha * alpha), 1. - rand1)) / (1.f - (alpha * alpha));
    float cos_elevation = std::sqrt(cos_square_elevation);
    float sin_elevation = std::sqrt(1 - cos_square_elevation);
    float h_azimuth = 2.f * kPi * rand2;
    local_h = vec3{sin_elevation * F_cos(h_azimuth), sin_elevation * F_sin(h_azimuth),
                   cos_elevation};
  }
  if (dot(frame.w, dir_in) < 0) {
    frame.u = -frame.u;
    frame.v = -frame.v;
    frame.w = -frame.w;
  }
  const vec3 H = normalize(xform_with_onb(frame, local_h));
  vec3 reflected = normalize(-dir_in + 2 * dot(dir_in, H) * H);
  if (dot(hit.hit_n_g, reflected) <= 0) return no_scatter();
  return ScatterInfo{reflected, 0, true, true};
}
// eval_pdf_disney_metal — disney_metal.h:122-152 (signed dot(h, wo) in the Fresnel: Q14)
void eval_pdf_disney_metal(vec3 dir_in, vec3 dir_out, const HitInfo& hit, vec3 base_col,
                           float spec_tint, float specular, float eta, float metallic,
                           vec3 half_vec, const ONB& frame, float G, float G_in, float alphax,
                           float alphay, vec3& f, float& pdf) {
  if (dot(hit.hit_n_g, dir_in) < 0 || dot(hit.hit_n_g, dir_out) < 0) {
    f = vec3{0.f, 0.f, 0.f};
    pdf = 0.f;
    return;
  }
  float base_lum = luminance(base_col);
  vec3 C_tint = base_lum > 0 ? base_col / base_lum : v3(1.f);
  vec3 K_s = (v3(1.f) - v3(spec_tint)) + spec_tint * C_tint;
  float R0 = ((eta - 1.f) * (eta - 1.f)) / ((eta + 1.f) * (eta + 1.f));
  vec3 C_0 = (specular * R0 * (1.f - metallic)) * K_s + metallic * base_col;
  vec3 Fresnel = C_0 + (v3(1.f) - C_0) * raise_to_power_5(1.f - dot(half_vec, dir_out));
  const vec3 local_H = project_onto_onb(frame, half_vec);
  float h_alpha_denominator = (local_H.x * local_H.x) / (alphax * alphax)
                              + (local_H.y * local_H.y) / (alphay * alphay)
                              + (local_H.z * local_H.z);
  float D = 1. / (kPi * alphax * alphay * (h_alpha_denominator * h_alpha_denominator));
  float D_mul_denominator = D / (4.f * std::abs(dot(frame.w, dir_in)));
  f = Fresnel * G * D_mul_denominator;
  pdf = G_in * D_mul_denominator;
}
inline void regularize_alpha(float& alphax, float& alphay) {   // MatConst, material.h:19-23
  alphax = alphax < 0.1f ? clampf(2.f * alphax, 0.03f, 0.1f) : alphax;
  alphay = alphay < 0.1f ? clampf(2.f * alphay, 0.03f, 0.1f) : alphay;
}
// sample_disney_metal — disney_metal.h:78-120 (note: roughness is NOT clamped here)
ScatterInfo sample_disney_metal(vec3 dir_in, const HitInfo& hit, float roughness,
                                float anisotropic, const ONB& frame, Pcg& rng, bool regularize) {
  if (dot(hit.hit_n_g, dir_in) < 0) return no_scatter();
  vec3 local_dir_in = project_onto_onb(frame, dir_in);
  constexpr float alpha_min = 0.0001;
  float aspect = std::sqrt(1.f - 0.9f * anisotropic);
  float roughness_square = roughness * roughness;
  float alphax = std::max(alpha_min, roughness_square / aspect);
  float alphay = std::max(alpha_min, roughness_square * aspect);
  if (regularize) regularize_alpha(alphax, alphay);
  vec3 local_micro_normal = anisotropic_sample_visible_normals(local_dir_in, alphax, alphay, rng);
  vec3 half_vector = normalize(xform_with_onb(frame, local_micro_normal));
  vec3 reflected = normalize(-dir_in + 2 * dot(dir_in, half_vector) * half_vector);
  if (dot(reflected, hit.hit_n_g) <= 0) return no_scatter();
  return ScatterInfo{reflected, 0.f, true, true};
}
// eval_pdf_disney_rough_glass — disney_glass.h:188-234
void eval_pdf_disney_rough_glass(vec3 dir_in, vec3 dir_out, const HitInfo& hit, vec3 base_col,
                                 float mat_eta, vec3 half_vec, const ONB& frame, float G,
                                 float G_in, float alphax, float alphay, vec3& eval, float& pdf) {
  float in_geo_dot = dot(dir_in, hit.hit_n_g);
  bool reflect = (in_geo_dot * dot(hit.hit_n_g, dir_out)) >= 0;
  float eta = in_geo_dot >= 0 ? mat_eta : 1.f / mat_eta;
  if (!reflect) half_vec = normalize(dir_in + dir_out * eta);
  float h_dot_in = dot(half_vec, dir_in);
  float F = fresnel_dielectric(h_dot_in, eta);
  const vec3 local_H = project_onto_onb(frame, half_vec);
  float h_alpha_denominator = (local_H.x * local_H.x) / (alphax * alphax)
                              + (local_H.y * local_H.y) / (alphay * alphay)
                              + (local_H.z * local_H.z);
  float D = 1. / (kPi * alphax * alphay * (h_alpha_denominator * h_alpha_denominator));
  float normal_in_dot = dot(frame.w, dir_in);
  if (reflect) {
    eval = base_col * (F * D * G) / (4.f * std::abs(normal_in_dot));
    pdf = (F * D * G_in) / (4.f * std::abs(normal_in_dot));
  } else {
    float eta_factor = 1.f / (eta * eta);
    float h_dot_out = dot(half_vec, dir_out);
    float sqrt_denom = h_dot_in + eta * h_dot_out;
    eval = vec3{std::sqrt(base_col.x), std::sqrt(base_col.y), std::sqrt(base_col.z)}
           * (eta_factor * (1 - F) * D * G * eta * eta * std::abs(h_dot_out * h_dot_in))
           / (std::abs(normal_in_dot) * sqrt_denom * sqrt_denom);
    float dh_dout = eta * eta * h_dot_out / (sqrt_denom * sqrt_denom);
    pdf = (1.f - F) * D * G_in * std::abs(dh_dout * h_dot_in / normal_in_dot);
  }
}
// sample_disney_rough_glass — disney_glass.h:108-186
ScatterInfo sample_disney_rough_glass(vec3 dir_in, const HitInfo& hit, float mat_eta,
                                      float anisotropic, float roughness, const ONB& frame,
                                      Pcg& rng, bool regularize) {
  float in_geo_dot = dot(dir_in, hit.hit_n_g);
  float eta = in_geo_dot >= 0 ? mat_eta : 1.f / mat_eta;
  constexpr float alpha_min = 0.0001;
  float aspect = std::sqrt(1.f - 0.9f * anisotropic);
  roughness = clampf(roughness, 0.01f, 1.f);
  float roughness_square = roughness * roughness;
  float alphax = std::max(alpha_min, roughness_square / aspect);
  float alphay = std::max(alpha_min, roughness_square * aspect);
  if (regularize) regularize_alpha(alphax, alphay);
  vec3 local_dir_in = project_onto_onb(frame, dir_in);
  vec3 local_micro_normal = anisotropic_sample_visible_normals(local_dir_in, alphax, alphay, rng);
  vec3 half_vec = xform_with_onb(frame, local_micro_normal);
  float h_dot_in = dot(half_vec, dir_in);
  float F = fresnel_dielectric(h_dot_in, eta);
  float rand = rand_float(rng);
  if (rand <= F) {
    vec3 reflected = normalize(-dir_in + 2 * dot(dir_in, half_vec) * half_vec);
    if (dot(reflected, hit.hit_n_g) * dot(dir_in, hit.hit_n_g) <= 0) return no_scatter();
    return ScatterInfo{reflected, 0.f, true, true};
  }
  float h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (eta * eta);
  if (h_dot_out_sq <= 0) return no_scatter();
  if (h_dot_in < 0) half_vec = -half_vec;
  float h_dot_out = static_cast<float>(::sqrt(static_cast<double>(h_dot_out_sq)));
  vec3 refracted = -dir_in / eta + (std::abs(h_dot_in) / eta - h_dot_out) * half_vec;
  if (dot(refracted, hit.hit_n_g) * dot(dir_in, hit.hit_n_g) >= 0) return no_scatter();
  vec3 generalized_h = normalize(dir_in + refracted * eta);
  float g_h_dot_in = dot(generalized_h, dir_in);
  if ((1 - (1 - g_h_dot_in * g_h_dot_in) / (eta * eta)) <= 0) return no_scatter();
  return ScatterInfo{refracted, eta, true, true};
}

struct PrincipledCommon {
  vec3 dir_in;
  ONB frame;
  float metallic, roughness;
};
// shared prologue of Principled::eval_pdf / sample_mat — principled.h:103-119, principled.cpp:5-21
PrincipledCommon principled_prologue(const VimgScene* s, const VimgMaterial& m, vec3 wi,
                                     const HitInfo& hit) {
  PrincipledCommon p;
  p.dir_in = -wi;
  p.frame = hit.n_frame;
  if ((dot(hit.hit_n_s, p.dir_in) * dot(hit.hit_n_g, p.dir_in)) < 0) {
    p.frame.u = -p.frame.u;
    p.frame.v = -p.frame.v;
    p.frame.w = -p.frame.w;
  }
  vec2 m_r{1.f, 1.f};
  if (m.mr_tex >= 0) m_r = rg_get_at_uv(s, m.mr_tex, hit.metal_rough_uv);
  m_r = m_r * vec2{m.metallic_factor, m.roughness_factor};
  p.metallic = m_r.x;
  p.roughness = m_r.y;
  return p;
}
// Principled::eval_pdf<std::pair<vec3,float>> — reference include/material/principled.h:100-205
void principled_eval_pdf(const VimgScene* s, const VimgMaterial& m, vec3 wi, vec3 wo,
                         const HitInfo& hit, const RayCone& cone, bool regularize, vec3& f_out,
                         float& pdf_out) {
  PrincipledCommon pc = principled_prologue(s, m, wi, hit);
  const vec3 dir_in = pc.dir_in;
  const ONB& frame = pc.frame;
  const float metallic = pc.metallic, roughness = pc.roughness;
  vec3 base_color = col_at_ray_hit(s, m.tex, wi, cone, hit);
  vec3 half_vector = normalize(dir_in + wo);
  constexpr float alpha_min = 0.0001;
  float aspect = std::sqrt(1.f - 0.9f * m.anisotropic);
  float roughness_clamp = clampf(roughness, 0.01f, 1.f);
  float roughness_square = roughness_clamp * roughness_clamp;
  float alphax = std::max(alpha_min, roughness_square / aspect);
  float alphay = std::max(alpha_min, roughness_square * aspect);
  if (regularize) regularize_alpha(alphax, alphay);
  float G_in = G_w(dir_in, alphax, alphay, frame);
  float G = G_in * G_w(wo, alphax, alphay, frame);
  vec3 eval_glass;
  float pdf_glass;
  eval_pdf_disney_rough_glass(dir_in, wo, hit, base_color, m.eta, half_vector, frame, G, G_in,
                              alphax, alphay, eval_glass, pdf_glass);
  if (dot(hit.hit_n_g, dir_in) < 0) {
    f_out = (1.f - metallic) * m.specular_transmission * eval_glass;
    pdf_out = pdf_glass;
    return;
  }
  vec3 eval_sheen = eval_disney_sheen(dir_in, wo, hit, base_color, m.sheen_tint, half_vector, frame);
  vec3 eval_diff;
  float pdf_diff;
  eval_pdf_disney_diffuse(dir_in, wo, hit, base_color, m.subsurface, roughness, half_vector, frame,
                          eval_diff, pdf_diff);
  float alpha_g = (1.f - m.clearcoat_gloss) * 0.1f + m.clearcoat_gloss * 0.001f;
  alpha_g = regularize && (alpha_g < 0.1f) ? clampf(2.f * alpha_g, 0.03f, 0.1f) : alpha_g;
  vec3 eval_clearcoat;
  float pdf_clearcoat;
  eval_pdf_disney_clearcoat(dir_in, wo, hit, alpha_g, half_vector, frame, eval_clearcoat,
                            pdf_clearcoat);
  vec3 eval_metal;
  float pdf_metal;
  eval_pdf_disney_metal(dir_in, wo, hit, base_color, m.specular_tint, m.specular, m.eta, metallic,
                        half_vector, frame, G, G_in, alphax, alphay, eval_metal, pdf_metal);
  const float st = m.specular_transmission;
  vec3 eval_principled = ((1.f - st) * (1.f - metallic) * eval_diff)
                         + ((1.f - metallic) * m.sheen * eval_sheen)
                         + (0.25f * m.clearcoat * eval_clearcoat)
                         + ((1.f - st * (1.f - metallic)) * eval_metal)
                         + ((1.f - metallic) * st * eval_glass);
  float diffuse_weight = (1.f - metallic) * (1.f - st);
  float clearcoat_weight = 0.25f * m.clearcoat;
  float metal_weight = (1.f - st * (1.f - metallic));
  float glass_weight = (1.f - metallic) * st;
  float total_w = diffuse_weight + clearcoat_weight + metal_weight + glass_weight;
  float choose_diff = diffuse_weight / total_w;
  float choose_clearcoat = clearcoat_weight / total_w;
  float choose_metal = metal_weight / total_w;
  float choose_glass = glass_weight / total_w;
  pdf_out = choose_diff * pdf_diff + choose_clearcoat * pdf_clearcoat + choose_metal * pdf_metal
            + choose_glass * pdf_glass;
  f_out = eval_principled;
}
// Principled::sample_mat — reference src/material/principled.cpp:3-58
ScatterInfo principled_sample(const VimgScene* s, const VimgMaterial& m, vec3 wi,
                              const HitInfo& hit, Pcg& rng, bool regularize) {
  PrincipledCommon pc = principled_prologue(s, m, wi, hit);
  const vec3 dir_in = pc.dir_in;
  const float metallic = pc.metallic, roughness = pc.roughness;
  if (dot(hit.hit_n_g, dir_in) < 0)
    return sample_disney_rough_glass(dir_in, hit, m.eta, m.anisotropic, roughness, pc.frame, rng,
                                     regularize);
  const float st = m.specular_transmission;
  float diffuse_weight = (1.f - metallic) * (1.f - st);
  float clearcoat_weight = 0.25f * m.clearcoat;
  float metal_weight = (1.f - st * (1.f - metallic));
  float glass_weight = (1.f - metallic) * st;
  float total_w = diffuse_weight + clearcoat_weight + metal_weight + glass_weight;
  float choose_diff = diffuse_weight / total_w;
  float choose_clearcoat = clearcoat_weight / total_w;
  float choose_metal = metal_weight / total_w;
  float choose_glass = glass_weight / total_w;
  float rnd = rand_float(rng);
  if (rnd <= choose_diff) {
    return sample_disney_diffuse(dir_in, hit, pc.frame, rng);
  } else if (rnd > choose_diff && rnd <= (choose_diff + choose_clearcoat)) {
    return sample_disney_clearcoat(dir_in, hit, pc.frame, m.clearcoat_gloss, rng, regularize);
  } else if (rnd > (choose_diff + choose_clearcoat)
             && rnd <= (choose_diff + choose_clearcoat + choose_metal)) {
    return sample_disney_metal(dir_in, hit, roughness, m.anisotropic, pc.frame, rng, regularize);
  } else if (rnd > (choose_diff + choose_clearcoat + choose_metal)
             && rnd <= (choose_diff + choose_clearcoat + choose_metal + choose_glass)) {
    return sample_disney_rough_glass(dir_in, hit, m.eta, m.anisotropic, roughness, pc.frame, rng,
                                     regularize);
  }
  return no_scatter();
}

// Material::sample_mat dispatch (virtual in the reference, include/material/material.h:37-40)
ScatterInfo sample_mat(const VimgScene* s, const HitInfo& hit, vec3 wi, Pcg& rng, bool regularize) {
  const VimgMaterial& m = s->materials[hit.mat];
  switch (m.type) {
    case VIMG_MAT_LAMBERTIAN: return lambertian_sample(hit, wi, rng);
    case VIMG_MAT_DIELECTRIC: return dielectric_sample(m, hit, wi, rng);
    case VIMG_MAT_PRINCIPLED: return principled_sample(s, m, wi, hit, rng, regularize);
    default: return no_scatter();   // DiffuseLight: base class, nullopt
  }
}
// Material::eval_pdf_pair dispatch; base class returns (0, 1) (material.h:56-60) — Dielectric
// and DiffuseLight do not override it (Q1)
void eval_pdf_pair(const VimgScene* s, const HitInfo& hit, vec3 wi, vec3 wo, const RayCone& cone,
                   bool regularize, vec3& f, float& pdf) {
  const VimgMaterial& m = s->materials[hit.mat];
  switch (m.type) {
    case VIMG_MAT_LAMBERTIAN: lambertian_eval_pdf(s, m, wi, wo, hit, cone, f, pdf); return;
    case VIMG_MAT_PRINCIPLED: principled_eval_pdf(s, m, wi, wo, hit, cone, regularize, f, pdf); return;
    default: f = vec3{0.f, 0.f, 0.f}; pdf = 1.0f; return;
  }
}

// ============================================================================ emitters
// Triangle::sample — reference src/geometry/triangle.cpp:178-233
void tri_light_sample(const VimgScene* s, uint32_t tri, vec3 look_from, Pcg& rng, vec3& Le,
                      EmitterInfo& info) {
  TriVerts tv = tri_verts(s, tri);
  vec3 p0 = tv.p0, p1 = tv.p1, p2 = tv.p2;
  const vec3 edge1 = p1 - p0, edge2 = p2 - p0;
  vec3 tri_normal = normalize(cross(edge1, edge2));
  vec3 n0, n1, n2;
  if (tv.mesh->has_normals) {
    n0 = load3(s->normals + 3 * size_t{tv.i0});
    n1 = load3(s->normals + 3 * size_t{tv.i1});
    n2 = load3(s->normals + 3 * size_t{tv.i2});
  } else {
    n0 = tri_normal, n1 = tri_normal, n2 = tri_normal;
  }
  float rand1 = rand_float(rng);
  float rand2 = rand_float(rng);
  float u, v;
  if (rand1 < rand2) {
    u = rand1 / 2.f;
    v = rand2 - u;
  } else {
    v = rand2 / 2.f;
    u = rand1 - v;
  }
  float w = 1.f - u - v;
  const vec3 hit_p = p0 * u + p1 * v + p2 * w;
  vec3 hit_n = normalize(u * n0 + v * n1 + w * n2);
  vec3 dir_vec = hit_p - look_from;
  float dist2 = length2(dir_vec);
  dir_vec = normalize(dir_vec);
  float area = length(cross(edge2, edge1)) / 2.0f;
  float pdf = 1.f / area;
  float cosine = std::abs(dot(hit_n, -dir_vec));
  float G = cosine / dist2;
  info = EmitterInfo{dir_vec, pdf, sqrtf(dist2), G};
  Le = mat_emitted(s->materials[tv.mesh->material], info.wi, hit_n);
}
// Triangle::surf_pdf — reference src/geometry/triangle.cpp:235-248
float tri_surf_pdf(const VimgScene* s, uint32_t tri) {
  TriVerts tv = tri_verts(s, tri);
  vec3 edge1 = tv.p1 - tv.p0, edge2 = tv.p2 - tv.p0;
  float area = length(cross(edge2, edge1)) / 2.0f;
  return 1.f / area;
}
// Sphere::sample — reference src/geometry/sphere.cpp:58-118 (Q16 cone construction kept)
void sphere_light_sample(const VimgScene* s, const VimgSphere& sp, vec3 look_from, Pcg& rng,
                         vec3& Le, EmitterInfo& info) {
  const vec3 center = load3(sp.center);
  const float radius = sp.radius;
  float rand1 = rand_float(rng);
  float rand2 = rand_float(rng);
  vec3 shading_normal;
  if (length2(look_from - center) <= radius * radius) {
    vec3 point_on_unit_sphere = sample_sphere(rand1, rand2);
    vec3 point_on_sphere = (point_on_unit_sphere * radius) + center;
    vec3 vec_from_lf_to_pos = point_on_sphere - look_from;
    shading_normal = point_on_unit_sphere;
    const float sphere_sa = 4.f * kPi * radius * radius;
    vec3 dir_to_surf = normalize(vec_from_lf_to_pos);
    float dist2 = length2(vec_from_lf_to_pos);
    float cosine = std::abs(dot(shading_normal, -dir_to_surf));
    float G = cosine / dist2;
    float pdf = 1.f / sphere_sa;
    info = EmitterInfo{dir_to_surf, pdf, sqrtf(dist2), G};
  } else {
    float cos_theta_max = static_cast<float>(
        ::sqrt(static_cast<double>(1.0f - ((radius * radius) / length2(look_from - center)))));
    vec3 dir_center_to_lf = normalize(look_from - center);
    ONB onb = init_onb(dir_center_to_lf);
    vec3 sample_z_dir = sample_sphere_cap(rand1, rand2, cos_theta_max);
    vec3 sampled_point = normalize(xform_with_onb(onb, sample_z_dir)) * radius + center;
    float dist2 = length2(sampled_point - look_from);
    shading_normal = normalize(sampled_point - center);
    vec3 sampled_dir = normalize(sampled_point - look_from);
    float cosine = std::abs(dot(shading_normal, -sampled_dir));
    float G = cosine / dist2;
    float pdf_solid_angle = 1.0f / (2.f * kPi * (1.0f - cos_theta_max));
    float pdf = pdf_solid_angle * G;
    info = EmitterInfo{sampled_dir, pdf, sqrtf(dist2), G};
  }
  Le = mat_emitted(s->materials[sp.material], info.wi, shading_normal);
}
// Sphere::surf_pdf — reference src/geometry/sphere.cpp:120-139
float sphere_surf_pdf(const VimgSphere& sp, vec3 look_from, vec3 point_on_light, vec3 dir) {
  const vec3 center = load3(sp.center);
  const float radius = sp.radius;
  if (length2(look_from - center) <= radius * radius) {
    const float sphere_sa = 4.f * kPi * radius * radius;
    return 1.f / sphere_sa;
  }
  float cos_theta_max = static_cast<float>(
      ::sqrt(static_cast<double>(1.0f - ((radius * radius) / length2(look_from - center)))));
  float pdf_solid_angle = 1.0f / (2.f * kPi * (1.0f - cos_theta_max));
  vec3 shading_normal = normalize(point_on_light - center);
  float cosine = std::abs(dot(shading_normal, -dir));
  float dist2 = length2(point_on_light - look_from);
  return pdf_solid_angle * cosine / dist2;
}

// ---- Background — reference include/background.h:25-179
inline bool background_is_emissive(const VimgBackground& bg) {
  if (bg.type == VIMG_BG_ENVMAP) return true;
  return !(load3(bg.col) == vec3{0.f, 0.f, 0.f});
}
inline vec3 mat_dir(const float* m, vec3 d) {
  vec4 r = mat_mul(m, vec4{d.x, d.y, d.z, 0.0f});
  return vec3{r.x, r.y, r.z};
}
inline void env_dir_to_uv(const VimgBackground& bg, vec3 in_dir, float& u, float& v) {
  vec3 dir = normalize(mat_dir(bg.world_to_env, in_dir));
  u = (1.f + F_atan2(-dir.x, dir.z) * kInvPi) * 0.5f;
  v = F_acos(dir.y) * kInvPi;
}
vec3 background_emit(const VimgScene* s, vec3 in_dir, const RayCone& cone) {
  const VimgBackground& bg = s->background;
  if (bg.type == VIMG_BG_CONST) return load3(bg.col);
  const VimgTexture& img = s->textures[bg.env_tex];
  float u, v;
  env_dir_to_uv(bg, in_dir, u, v);
  float lambda = ::log2(std::abs(cone.spread_angle) * (img.height / kPi));   // double
  lambda = std::isnan(lambda) ? 0.f : lambda;
  return col_mipmap_interpolate(s, img, lambda - 2.f, vec2{u, v}) * bg.radiance_scale;
}
float background_pdf(const VimgScene* s, vec3 in_dir) {
  const VimgBackground& bg = s->background;
  if (bg.type == VIMG_BG_CONST) return 1.f / (4 * kPi);
  const VimgTexture& img = s->textures[bg.env_tex];
  float u, v;
  env_dir_to_uv(bg, in_dir, u, v);
  int pixel_u = u * img.width;
  int pixel_v = v * img.height;
  int column_index = clampi(pixel_u, 0, static_cast<int>(img.width) - 1);
  int row_index = clampi(pixel_v, 0, static_cast<int>(img.height) - 1);
  const float* row_cdf = s->cdf_pool + bg.row_cdf_offset;
  const float* col_cdf = s->cdf_pool + bg.col_cdf_offset + size_t(row_index) * (img.width + 1);
  float pdf_y = row_cdf[row_index + 1] - row_cdf[row_index];
  float pdf_x = col_cdf[column_index + 1] - col_cdf[column_index];
  float sin_elevation = ::sin(kPi * v);
  return (pdf_y * pdf_x * img.width * img.height) / (2.f * kPi * kPi * sin_elevation);
}
// ArraySampling1D::sample — reference include/rng/sampling.h:144-155 (upper_bound, then -1)
inline void cdf_sample(const float* cdf, size_t n_plus_1, float u, size_t& index, float& du) {
  const float* itr = std::upper_bound(cdf, cdf + n_plus_1, u);
  index = static_cast<size_t>(itr - cdf) - 1;
  du = u - cdf[index];
  if (cdf[index + 1] - cdf[index] > 0) du /= cdf[index + 1] - cdf[index];
}
void background_sample(const VimgScene* s, Pcg& rng, vec3& Le, EmitterInfo& info) {
  const VimgBackground& bg = s->background;
  float r1 = rand_float(rng);
  float r2 = rand_float(rng);
  if (bg.type == VIMG_BG_CONST) {
    vec3 wi = sample_sphere(r1, r2);
    constexpr float pdf = 1.f / (4 * kPi);
    Le = load3(bg.col);
    info = EmitterInfo{wi, pdf, kInf, 1.f};
    return;
  }
  const VimgTexture& img = s->textures[bg.env_tex];
  const float* row_cdf = s->cdf_pool + bg.row_cdf_offset;
  size_t row_index, column_index;
  float dv, du;
  cdf_sample(row_cdf, img.height + 1, r1, row_index, dv);
  const float* col_cdf = s->cdf_pool + bg.col_cdf_offset + row_index * (img.width + 1);
  cdf_sample(col_cdf, img.width + 1, r2, column_index, du);
  const float u_env = (static_cast<float>(column_index) + du) / img.width;
  const float v_env = (static_cast<float>(row_index) + dv) / img.height;
  float pdf_y = row_cdf[row_index + 1] - row_cdf[row_index];
  float pdf_x = col_cdf[column_index + 1] - col_cdf[column_index];
  float choose_sample_pdf = pdf_y * pdf_x;
  float elevation = v_env * kPi;
  float y = ::cos(v_env * kPi);
  const float azimuth = u_env * 2.f * kPi;
  float x = F_sin(azimuth) * F_sin(elevation);
  float z = -1 * F_cos(azimuth) * F_sin(elevation);
  vec3 wi = normalize(mat_dir(bg.env_to_world, vec3{x, y, z}));
  float sin_elevation = F_sin(elevation);
  float pdf = (choose_sample_pdf * img.width * img.height) / (2.f * kPi * kPi * sin_elevation);
  Le = col_at_uv_mipmap(s, img, 0, vec2{u_env, v_env}) * bg.radiance_scale;
  info = EmitterInfo{wi, pdf, kInf, 1.f};
}

// GroupOfEmitters::sample — reference include/geometry/emitters.h:39-56
void lights_sample(const VimgScene* s, vec3 look_from, Pcg& rng, vec3& Le, EmitterInfo& info) {
  float rand = rand_float(rng);
  float sx = rand * s->num_lights;
  const int index_obj = clampi(static_cast<int>(sx), 0, static_cast<int>(s->num_lights) - 1);
  const float prob_obj = 1.f / s->num_lights;
  const VimgLight& l = s->lights[index_obj];
  if (l.type == VIMG_LIGHT_BACKGROUND) {
    background_sample(s, rng, Le, info);
  } else {
    const VimgPrim& p = s->prims[l.prim];
    if (p.type == VIMG_PRIM_TRIANGLE)
      tri_light_sample(s, p.index, look_from, rng, Le, info);
    else
      sphere_light_sample(s, s->spheres[p.index], look_from, rng, Le, info);
  }
  info.pdf *= prob_obj;
}
// Emitter::surf_pdf on the object a bounce ray hit (virtual, emitters.h:22-24)
float surf_pdf(const VimgScene* s, uint32_t prim_id, vec3 look_from, vec3 look_at, vec3 dir) {
  const VimgPrim& p = s->prims[prim_id];
  if (p.type == VIMG_PRIM_TRIANGLE) return tri_surf_pdf(s, p.index);
  return sphere_surf_pdf(s->spheres[p.index], look_from, look_at, dir);
}

// ============================================================================ integrators
inline float balance_heuristic(float pdf1, float pdf2) { return pdf1 / (pdf1 + pdf2); }
// geometric_term — reference src/integrators/mis_integrator.cpp:7-16
inline float geometric_term(vec3 look_from, vec3 point_on_surface, vec3 surface_normal) {
  vec3 dir_from_surf = look_from - point_on_surface;
  float distance2 = length2(dir_from_surf);
  dir_from_surf = normalize(dir_from_surf);
  float cosine = std::abs(dot(surface_normal, dir_from_surf));
  return cosine / distance2;
}

// mis_integrator — reference src/integrators/mis_integrator.cpp:18-189
vec3 mis_integrator(const Ctx& c, Ray& input_ray, Pcg& rng, uint32_t depth) {
  const VimgScene* s = c.s;
  Ray test_ray = input_ray;
  float light_pdf;
  vec3 bounce_result{0.f, 0.f, 0.f};
  vec3 throughput{1.f, 1.f, 1.f};
  bool non_specular_bounce = false;
  float eta_scale = 1;
  constexpr uint32_t roulette_threshold = 5;

  HitInfo hit;
  if (!bvh_hit<false>(c, test_ray, &hit))
    return background_emit(s, test_ray.dir, test_ray.ray_cone);
  if (mat_is_emissive(s->materials[hit.mat]))
    return mat_emitted(s->materials[hit.mat], test_ray.dir, hit.hit_n_s);

  for (size_t d = 0; d < depth; d++) {
    const VimgMaterial& mat = s->materials[hit.mat];
    bool is_delta = mat_is_delta(mat);
    float hit_dist = length(test_ray.o - hit.hit_p);
    float surface_spread_angle = spread_angle_from_curvature(
        hit.mean_curvature, test_ray.ray_cone.cone_width, test_ray.dir, hit.hit_n_s);

    if (!is_delta) {
      vec3 light_col;
      EmitterInfo li;
      lights_sample(s, hit.hit_p, rng, light_col, li);
      if (li.pdf != 0.f) {
        Ray shadow_ray(hit.hit_p, li.wi);
        shadow_ray.maxT = li.dist - 0.0001f;   // Q15: absolute epsilon
        bool light_is_occluded = bvh_hit<true>(c, shadow_ray, nullptr);
        if (!light_is_occluded) {
          vec3 mat_eval;
          float mat_pdf;
          eval_pdf_pair(s, hit, test_ray.dir, li.wi,
                        propagate_reflect_cone(test_ray.ray_cone, surface_spread_angle * 2.f,
                                               hit_dist),
                        non_specular_bounce, mat_eval, mat_pdf);
          if (mat_pdf != 0 && !std::isnan(mat_pdf)) {
            float G = li.G;
            float mis_weight = balance_heuristic(li.pdf, mat_pdf * G);
            bounce_result += throughput * mat_eval * mis_weight * G * light_col / li.pdf;
          }
        }
      }
    }

    ScatterInfo sc = sample_mat(s, hit, test_ray.dir, rng, non_specular_bounce);
    if (!sc.valid) return bounce_result;
    if (!sc.is_specular) non_specular_bounce = true;
    if (sc.eta != 0.f) {
      eta_scale /= (sc.eta * sc.eta);
      test_ray.ray_cone = propagate_refract_cone(test_ray.ray_cone, test_ray.dir, hit.hit_p,
                                                 surface_spread_angle, sc.eta, sc.wo);
    } else {
      test_ray.ray_cone
          = propagate_reflect_cone(test_ray.ray_cone, surface_spread_angle * 2.f, hit_dist);
    }
    vec3 mat_sample_eval;
    float mat_sample_pdf;
    eval_pdf_pair(s, hit, test_ray.dir, sc.wo, test_ray.ray_cone, non_specular_bounce,
                  mat_sample_eval, mat_sample_pdf);
    if (std::isnan(mat_sample_pdf)) return bounce_result;
    throughput *= (mat_sample_eval / mat_sample_pdf);

    Ray direct_light_ray(hit.hit_p, sc.wo, test_ray.ray_cone);
    HitInfo next;
    if (bvh_hit<false>(c, direct_light_ray, &next)) {
      const VimgMaterial& nmat = s->materials[next.mat];
      if (mat_is_emissive(nmat)) {
        if (mat_sample_pdf != 0) {
          light_pdf = surf_pdf(s, next.prim, hit.hit_p, next.hit_p, sc.wo) / s->num_lights;
          float G = geometric_term(hit.hit_p, next.hit_p, next.hit_n_g);
          float mis_weight = balance_heuristic(mat_sample_pdf * G, light_pdf);
          bounce_result += throughput * mis_weight
                           * mat_emitted(nmat, direct_light_ray.dir, next.hit_n_s);
        } else {
          bounce_result += throughput * mat_emitted(nmat, direct_light_ray.dir, next.hit_n_s);
        }
        return bounce_result;
      } else {
        if (d > roulette_threshold) {
          float rr = static_cast<float>(pcg32_random_r(&rng))
                     / std::numeric_limits<uint32_t>::max();   // float(u32)/float(UINT32_MAX)
          vec3 rr_throughput = (1.f / eta_scale) * throughput;
          float max_val = std::min(
              std::max(std::max(rr_throughput.x, rr_throughput.y), rr_throughput.z), 0.95f);
          if (rr > max_val) break;
          throughput /= max_val;
        }
        hit = next;
        test_ray = direct_light_ray;
      }
    } else {
      if (mat_sample_pdf != 0 && background_is_emissive(s->background)) {
        light_pdf = background_pdf(s, direct_light_ray.dir) / s->num_lights;
        float mis_weight = balance_heuristic(mat_sample_pdf, light_pdf);
        bounce_result += throughput * mis_weight
                         * background_emit(s, direct_light_ray.dir, direct_light_ray.ray_cone);
      }
      return bounce_result;
    }
  }
  return bounce_result;
}

// Material::eval_div_pdf dispatch (virtual, include/material/material.h:51-54): Lambertian returns
// the texture colour (src/material/lambertian.cpp:42-45), Dielectric 1 (dielectric.cpp:86-88),
// Principled eval/pdf through the same eval_pdf template (principled.cpp:155-158), base 0
vec3 eval_div_pdf(const VimgScene* s, const HitInfo& hit, vec3 wi, vec3 wo, const RayCone& cone,
                  bool regularize) {
  const VimgMaterial& m = s->materials[hit.mat];
  switch (m.type) {
    case VIMG_MAT_LAMBERTIAN: return col_at_ray_hit(s, m.tex, wi, cone, hit);
    case VIMG_MAT_DIELECTRIC: return v3(1.f);
    case VIMG_MAT_PRINCIPLED: {
      vec3 f;
      float pdf;
      principled_eval_pdf(s, m, wi, wo, hit, cone, regularize, f, pdf);
      return f / pdf;
    }
    default: return vec3{0.f, 0.f, 0.f};
  }
}

// material_integrator — reference src/integrators/mat_integrator.cpp:4-85
vec3 material_integrator(const Ctx& c, Ray& input_ray, Pcg& rng, uint32_t depth) {
  const VimgScene* s = c.s;
  Ray test_ray = input_ray;
  vec3 throughput{1.f, 1.f, 1.f};
  constexpr uint32_t roulette_threshold = 5;
  bool non_specular_bounce = false;
  float eta_scale = 1;
  for (size_t d = 0; d < depth; d++) {
    HitInfo hit;
    if (!bvh_hit<false>(c, test_ray, &hit))
      return throughput * background_emit(s, test_ray.dir, test_ray.ray_cone);
    vec3 emitted_col = mat_emitted(s->materials[hit.mat], test_ray.dir, hit.hit_n_s);
    ScatterInfo sc = sample_mat(s, hit, test_ray.dir, rng, non_specular_bounce);
    if (!sc.valid) return throughput * emitted_col;
    if (!sc.is_specular) non_specular_bounce = true;
    float hit_dist = length(test_ray.o - hit.hit_p);
    float surface_spread_angle = spread_angle_from_curvature(
        hit.mean_curvature, test_ray.ray_cone.cone_width, test_ray.dir, hit.hit_n_s);
    if (sc.eta != 0.f) {
      eta_scale /= (sc.eta * sc.eta);
      test_ray.ray_cone = propagate_refract_cone(test_ray.ray_cone, test_ray.dir, hit.hit_p,
                                                 surface_spread_angle, sc.eta, sc.wo);
    } else {
      test_ray.ray_cone
          = propagate_reflect_cone(test_ray.ray_cone, surface_spread_angle * 2.f, hit_dist);
    }
    throughput *= emitted_col
                  + eval_div_pdf(s, hit, test_ray.dir, sc.wo, test_ray.ray_cone,
                                 non_specular_bounce);
    if (d > roulette_threshold) {
      float rr = static_cast<float>(pcg32_random_r(&rng)) / std::numeric_limits<uint32_t>::max();
      vec3 rr_throughput = (1.f / eta_scale) * throughput;
      float max_val = std::min(
          std::max(std::max(rr_throughput.x, rr_throughput.y), rr_throughput.z), 0.95f);
      if (rr > max_val) break;
      throughput /= max_val;
    }
    test_ray = Ray(hit.hit_p, sc.wo, test_ray.ray_cone);
  }
  return vec3{0.f, 0.f, 0.f};
}

// shading_normal_integrator / geometric_normal_integrator — reference src/integrators/normals.cpp
vec3 normal_integrator(const Ctx& c, Ray& input_ray, bool geometric) {
  HitInfo hit;
  if (bvh_hit<false>(c, input_ray, &hit)) {
    vec3 n = geometric ? hit.hit_n_g : hit.hit_n_s;
    return (n + 1.0f) / 2.0f;
  }
  vec3 unit_dir = normalize(input_ray.dir);
  float a = 0.5 * (unit_dir.y + 1.0);
  return (1.0f - a) * vec3{1.0f, 1.0f, 1.0f} + a * vec3{0.5f, 0.7f, 1.0f};
}

vec3 run_integrator(const Ctx& c, uint32_t func, Ray& ray, Pcg& rng, uint32_t depth) {
  switch (func) {
    case VIMG_INTEGRATOR_S_NORMAL: return normal_integrator(c, ray, false);
    case VIMG_INTEGRATOR_G_NORMAL: return normal_integrator(c, ray, true);
    case VIMG_INTEGRATOR_MATERIAL: return material_integrator(c, ray, rng, depth);
    default: return mis_integrator(c, ray, rng, depth);
  }
}

Ctx make_ctx(const VimgScene* s, Counters* cnt) {
  Ctx c;
  c.s = s;
  c.cnt = cnt;
  // TLCam ctor — reference src/tl_camera.cpp:6-23 (tan is unqualified: double)
  const VimgCamera& cam = s->camera;
  float theta = (cam.vfov_deg * kPi) / 180.0;
  float ratio = static_cast<float>(cam.res_x) / cam.res_y;
  float img_height = 2.0f * (::tan(static_cast<double>(theta / 2.0f)));
  float img_width = ratio * img_height;
  c.p_size0 = img_width;
  c.p_size1 = img_height;
  return c;
}

// the per-pixel body shared by scene_integrator and trace_pixel
// — reference include/integrators.h:109-138 / :195-219
vec3 pixel_body(const Ctx& c, const VimgRenderParams& p, size_t x, size_t y, uint64_t* nan_count) {
  const uint32_t W = c.s->camera.res_x, H = c.s->camera.res_y;
  vec3 acc{0.f, 0.f, 0.f};
  size_t image_index = x + ((H - 1 - y) * size_t{W});
  size_t image_seq_start = x + y;
  Pcg rng;
  pcg32_srandom_r(&rng, image_index, 0);
  for (size_t smp = 0; smp < p.samples; smp++) {
    vec2 off = random_x_y_r2(static_cast<uint32_t>(image_seq_start + smp));
    // Q4: g++ evaluates the two rand_float arguments right to left
    float rand2 = rand_float(rng);
    float rand1 = rand_float(rng);
    Ray cam_ray = generate_ray(c, x + off.x, y + off.y, rand1, rand2);
    vec3 col = run_integrator(c, p.integrator, cam_ray, rng, p.depth);
    if (std::isnan(col.x) || std::isnan(col.y) || std::isnan(col.z)) (*nan_count)++;
    acc += col;
  }
  acc /= static_cast<float>(p.samples);   // vec3 /= uint32_t: glm converts the scalar to float
  return acc;
}

bool params_ok(const VimgScene* s, const VimgRenderParams* p) {
  return s && p && p->tile_world >= 1 && p->tile_rank < p->tile_world && s->camera.res_x > 0
         && s->camera.res_y > 0 && p->integrator <= VIMG_INTEGRATOR_MIS && s->bvh.max_depth + 2 <= 128;
}

}  // namespace

extern "C" {

// ---- post chain: tonemap -> sRGB OETF -> 8-bit quantise
//   simple_clamp / sRGB_gamma_correction  reference include/color_utils.h:21-68
//   agx                                    reference src/tonemap/agx.cpp:6-90
//   reinhard_lum                           reference src/tonemap/reinhard.cpp:3-35
//   aces                                   reference src/tonemap/aces.cpp:5-29
//   quantise, NaN -> magenta               reference src/main.cpp:339-356
}  // extern "C" (helpers below are C++)
namespace {
#if defined(ORACLE_LIBM_FLOAT) && ORACLE_LIBM_FLOAT
inline float F_pow(float x, float y) { return ::powf(x, y); }
#else
inline float F_pow(float x, float y) {
  return static_cast<float>(::pow(static_cast<double>(x), static_cast<double>(y)));
}
#endif
inline vec3 mat3_mul(const float m[9], vec3 v) {   // glm column-major mat3 * vec3
  return vec3{m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
              m[2] * v.x + m[5] * v.y + m[8] * v.z};
}
vec3 agx_pixel(vec3 val) {
  static const float agx_mat[9] = {0.842479062253094, 0.0423282422610123, 0.0423756549057051,
                                   0.0784335999999992, 0.878468636469772, 0.0784336,
                                   0.0792237451477643, 0.0791661274605434, 0.879142973793104};
  static const float agx_mat_inv[9] = {1.19687900512017, -0.0528968517574562, -0.0529716355144438,
                                       -0.0980208811401368, 1.15190312990417, -0.0980434501171241,
                                       -0.0990297440797205, -0.0989611768448433, 1.15107367264116};
  const float min_ev = -12.47393f, max_ev = 4.026069f;
  val = mat3_mul(agx_mat, val);
  val = vec3{clampf(F_log2(val.x), min_ev, max_ev), clampf(F_log2(val.y), min_ev, max_ev),
             clampf(F_log2(val.z), min_ev, max_ev)};
  val = (val - min_ev) / (max_ev - min_ev);
  {   // agxDefaultContrastApprox
    vec3 x = val, x2 = x * x, x4 = x2 * x2;
    val = v3(+15.5f) * x4 * x2 - v3(40.14f) * x4 * x + v3(31.96f) * x4 - v3(6.868f) * x2 * x
          + v3(0.4298f) * x2 + v3(0.1191f) * x - v3(0.00232f);
  }
  {   // agxLook, default look: pow(val*1+0, 1) is val; luma + 1*(val - luma)
    float luma = luminance(val);
    val = vec3{luma + 1.0f * (val.x - luma), luma + 1.0f * (val.y - luma), luma + 1.0f * (val.z - luma)};
  }
  val = mat3_mul(agx_mat_inv, val);
  if (val.x < 0.f) val.x = 0.f;
  if (val.y < 0.f) val.y = 0.f;
  if (val.z < 0.f) val.z = 0.f;
  return vec3{F_pow(val.x, 2.2f), F_pow(val.y, 2.2f), F_pow(val.z, 2.2f)};
}
vec3 aces_pixel(vec3 v) {
  static const float in_m[9] = {0.59719f, 0.07600f, 0.02840f, 0.35458f, 0.90834f,
                                0.13383f, 0.04823f, 0.01566f, 0.83777f};
  static const float out_m[9] = {1.60475f,  -0.10208f, -0.00327f, -0.53108f, 1.10813f,
                                 -0.07276f, -0.07367f, -0.00605f, 1.07602f};
  v = mat3_mul(in_m, v);
  vec3 a = v * (v + 0.0245786f) - 0.000090537f;
  vec3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
  v = a / b;
  return mat3_mul(out_m, v);
}
inline float srgb_oetf(float x) {
  x = clampf(x, 0.0f, 1.0f);
  if (x < 0.0031308f) return x * 12.92f;
  return 1.055f * F_pow(x, 1.0f / 2.4f) - 0.055f;
}
}  // namespace
extern "C" {

int oracle_post_rgb8(const float* rgb, int w, int h, int tonemapper, uint8_t* out) {
  if (!rgb || !out || w <= 0 || h <= 0 || tonemapper < 0 || tonemapper > 3) return -1;
  const size_t n = static_cast<size_t>(w) * h;
  float largest_L = 0.0f;
  if (tonemapper == 2)
    for (size_t i = 0; i < n; ++i) {
      float l = luminance(load3(rgb + 3 * i));
      if (l > largest_L) largest_L = l;
    }
  for (size_t i = 0; i < n; ++i) {
    vec3 c = load3(rgb + 3 * i);
    switch (tonemapper) {
      case 0: c = vec3{clampf(c.x, 0.f, 1.f), clampf(c.y, 0.f, 1.f), clampf(c.z, 0.f, 1.f)}; break;
      case 1: c = agx_pixel(c); break;
      case 2: {
        float in_L = luminance(c);
        float numerator = in_L * (1.0f + (in_L / (largest_L * largest_L)));
        float new_L = numerator / (1.0f + in_L);
        c = (in_L > 0.f) ? c * (new_L / in_L) : vec3{0.f, 0.f, 0.f};
        break;
      }
      default: c = aces_pixel(c); break;
    }
    c = vec3{srgb_oetf(c.x), srgb_oetf(c.y), srgb_oetf(c.z)};
    uint8_t* o = out + 3 * i;
    if (std::isnan(c.x) || std::isnan(c.y) || std::isnan(c.z)) {
      o[0] = 255, o[1] = 0, o[2] = 255;
    } else {
      o[0] = static_cast<uint8_t>(clampi(static_cast<int>(255.999 * c.x), 0, 255));
      o[1] = static_cast<uint8_t>(clampi(static_cast<int>(255.999 * c.y), 0, 255));
      o[2] = static_cast<uint8_t>(clampi(static_cast<int>(255.999 * c.z), 0, 255));
    }
  }
  return 0;
}

int oracle_uses_float_libm(void) {
#if defined(ORACLE_LIBM_FLOAT) && ORACLE_LIBM_FLOAT
  return 1;
#else
  return 0;
#endif
}

int oracle_render(const VimgScene* scene, const VimgRenderParams* params, int num_threads,
                  float* out_rgb, VimgRenderStats* stats) {
  if (!params_ok(scene, params) || !out_rgb) return -1;
  const uint32_t W = scene->camera.res_x, H = scene->camera.res_y;
  const unsigned cores
      = num_threads > 0 ? num_threads : std::max(1u, std::thread::hardware_concurrency());
  // work_list of 8x8 tiles, x-major — reference include/integrators.h:57-65
  struct Tile { uint32_t x0, y0, x1, y1; };
  std::vector<Tile> work;
  uint32_t tile_id = 0;
  for (uint32_t x = 0; x < W; x += 8)
    for (uint32_t y = 0; y < H; y += 8, ++tile_id)
      if (tile_id % params->tile_world == params->tile_rank)
        work.push_back(Tile{x, y, std::min(x + 7, W - 1), std::min(y + 7, H - 1)});

  std::vector<Counters> counters(cores);
  std::vector<uint64_t> nans(cores, 0);
  std::vector<std::thread> workers;
  for (unsigned t = 0; t < cores; ++t) {
    workers.emplace_back([&, t]() {
      Ctx c = make_ctx(scene, stats ? &counters[t] : nullptr);
      for (size_t i = t; i < work.size(); i += cores) {   // static interleave, :101
        const Tile& tl = work[i];
        for (size_t y = tl.y0; y <= tl.y1; y++)
          for (size_t x = tl.x0; x <= tl.x1; x++) {
            vec3 col = pixel_body(c, *params, x, y, &nans[t]);
            size_t image_index = x + ((H - 1 - y) * size_t{W});
            out_rgb[image_index * 3 + 0] = col.x;
            out_rgb[image_index * 3 + 1] = col.y;
            out_rgb[image_index * 3 + 2] = col.z;
          }
      }
    });
  }
  for (auto& th : workers) th.join();
  if (stats) {
    *stats = VimgRenderStats{};
    for (unsigned t = 0; t < cores; ++t) {
      stats->closest_rays += counters[t].closest;
      stats->shadow_rays += counters[t].shadow;
      stats->internal_visits += counters[t].internal;
      stats->leaf_visits += counters[t].leaf;
      stats->prim_tests += counters[t].prim;
      stats->sphere_tests += counters[t].sphere;
      stats->nan_samples += nans[t];
    }
    uint64_t px = 0;
    for (const Tile& tl : work) px += uint64_t{tl.x1 - tl.x0 + 1} * (tl.y1 - tl.y0 + 1);
    stats->paths = px * params->samples;
  }
  return static_cast<int>(cores);
}

// heatmap_img — reference src/integrators/heatmap.cpp:38-147.  Per pixel: the traversal cost of
// its camera rays summed over the samples, averaged, truncated to uint32, colour-mapped with
// turbo(value / factor) (factor <= 0 -> 20).  out_counts (optional, test hook): the truncated
// averages before the colour map.
int oracle_heatmap(const VimgScene* scene, const VimgRenderParams* params, float factor,
                   int num_threads, float* out_rgb, float* out_counts) {
  if (!params_ok(scene, params) || !out_rgb) return -1;
  const uint32_t W = scene->camera.res_x, H = scene->camera.res_y;
  const unsigned cores
      = num_threads > 0 ? num_threads : std::max(1u, std::thread::hardware_concurrency());
  if (factor <= 0) factor = 20.f;
  std::vector<std::thread> workers;
  for (unsigned t = 0; t < cores; ++t) {
    workers.emplace_back([&, t]() {
      Ctx c = make_ctx(scene, nullptr);
      uint32_t tile_id = 0, mine = 0;
      for (uint32_t x0 = 0; x0 < W; x0 += 8)
        for (uint32_t y0 = 0; y0 < H; y0 += 8, ++tile_id) {
          if (tile_id % params->tile_world != params->tile_rank) continue;
          if (mine++ % cores != t) continue;
          for (size_t y = y0; y <= std::min(y0 + 7, H - 1); y++)
            for (size_t x = x0; x <= std::min(x0 + 7, W - 1); x++) {
              float pixel_hit_accumulator = 0.f;
              size_t image_index = x + ((H - 1 - y) * size_t{W});
              size_t image_seq_start = x + y;
              Pcg rng;
              pcg32_srandom_r(&rng, image_index, 0);
              for (size_t smp = 0; smp < params->samples; smp++) {
                vec2 off = random_x_y_r2(static_cast<uint32_t>(image_seq_start + smp));
                float rand2 = rand_float(rng);   // Q4: right-to-left argument evaluation
                float rand1 = rand_float(rng);
                Ray cam_ray = generate_ray(c, x + off.x, y + off.y, rand1, rand2);
                pixel_hit_accumulator += bvh_cost(c, cam_ray);
              }
              float v = static_cast<float>(
                  static_cast<uint32_t>(pixel_hit_accumulator / static_cast<float>(params->samples)));
              if (out_counts) out_counts[image_index] = v;
              vec3 col = turbo_colormap(v / factor);
              out_rgb[image_index * 3 + 0] = col.x;
              out_rgb[image_index * 3 + 1] = col.y;
              out_rgb[image_index * 3 + 2] = col.z;
            }
        }
    });
  }
  for (auto& th : workers) th.join();
  return static_cast<int>(cores);
}

int oracle_trace_pixel(const VimgScene* scene, const VimgRenderParams* params, int x, int y,
                       float* out_rgb3) {
  if (!params_ok(scene, params) || !out_rgb3 || x < 0 || y < 0 || x >= scene->camera.res_x
      || y >= scene->camera.res_y)
    return -1;
  Ctx c = make_ctx(scene, nullptr);
  uint64_t nans = 0;
  vec3 col = pixel_body(c, *params, x, y, &nans);
  out_rgb3[0] = col.x;
  out_rgb3[1] = col.y;
  out_rgb3[2] = col.z;
  return 0;
}

void oracle_pcg32_srandom(uint64_t si[2], uint64_t initstate, uint64_t initseq) {
  Pcg p;
  pcg32_srandom_r(&p, initstate, initseq);
  si[0] = p.state;
  si[1] = p.inc;
}
uint32_t oracle_pcg32_random(uint64_t si[2]) {
  Pcg p{si[0], si[1]};
  uint32_t r = pcg32_random_r(&p);
  si[0] = p.state;
  return r;
}
float oracle_rand_float(uint64_t si[2]) {
  Pcg p{si[0], si[1]};
  float r = rand_float(p);
  si[0] = p.state;
  return r;
}
void oracle_random_x_y_r2(uint32_t n, float out_xy[2]) {
  vec2 v = random_x_y_r2(n);
  out_xy[0] = v.x;
  out_xy[1] = v.y;
}

// Probe layouts (floats).  Integers (prim ids, seeds) travel as exactly representable floats.
//  CLOSEST_HIT out[28]: hit t prim mat | hit_p3 | n_s3 | n_g3 | uv2 | mr_uv2 | tangent3 |
//                       bitangent3 | prim_area tex_area curvature
int oracle_probe(const VimgScene* scene, int kind, int n, const float* in, float* out) {
  if (!scene || !in || !out || n < 0) return -1;
  Ctx c = make_ctx(scene, nullptr);
  auto trace = [&](const float* p, HitInfo& h, Ray& r) {
    r = Ray(vec3{p[0], p[1], p[2]}, vec3{p[3], p[4], p[5]});
    return bvh_hit<false>(c, r, &h);
  };
  for (int i = 0; i < n; ++i) {
    switch (kind) {
      case ORACLE_PROBE_CAMERA_RAY: {
        const float* p = in + 4 * i;
        float* o = out + 8 * i;
        Ray r = generate_ray(c, p[0], p[1], p[2], p[3]);
        o[0] = r.o.x, o[1] = r.o.y, o[2] = r.o.z, o[3] = r.dir.x, o[4] = r.dir.y, o[5] = r.dir.z;
        o[6] = r.ray_cone.cone_width, o[7] = r.ray_cone.spread_angle;
        break;
      }
      case ORACLE_PROBE_CLOSEST_HIT: {
        float* o = out + 28 * i;
        std::fill(o, o + 28, 0.f);
        HitInfo h;
        Ray r;
        if (trace(in + 6 * i, h, r)) {
          float v[28] = {1.f, r.maxT, float(h.prim), float(h.mat), h.hit_p.x, h.hit_p.y, h.hit_p.z,
                         h.hit_n_s.x, h.hit_n_s.y, h.hit_n_s.z, h.hit_n_g.x, h.hit_n_g.y,
                         h.hit_n_g.z, h.uv.x, h.uv.y, h.metal_rough_uv.x, h.metal_rough_uv.y,
                         h.n_frame.u.x, h.n_frame.u.y, h.n_frame.u.z, h.n_frame.v.x, h.n_frame.v.y,
                         h.n_frame.v.z, h.primitive_area, h.tex_coord_area, h.mean_curvature, 0.f,
                         0.f};
          std::memcpy(o, v, sizeof(v));
        }
        break;
      }
      case ORACLE_PROBE_OCCLUDED: {
        const float* p = in + 7 * i;
        Ray r(vec3{p[0], p[1], p[2]}, vec3{p[3], p[4], p[5]});
        r.maxT = p[6];
        out[i] = bvh_hit<true>(c, r, nullptr) ? 1.f : 0.f;
        break;
      }
      case ORACLE_PROBE_BSDF_EVAL: {
        const float* p = in + 12 * i;
        float* o = out + 5 * i;
        std::fill(o, o + 5, 0.f);
        HitInfo h;
        Ray r;
        if (trace(p, h, r)) {
          vec3 f;
          float pdf;
          eval_pdf_pair(scene, h, r.dir, vec3{p[6], p[7], p[8]}, RayCone{p[9], p[10]},
                        p[11] != 0.f, f, pdf);
          o[0] = 1.f, o[1] = f.x, o[2] = f.y, o[3] = f.z, o[4] = pdf;
        }
        break;
      }
      case ORACLE_PROBE_BSDF_SAMPLE: {
        const float* p = in + 8 * i;
        float* o = out + 7 * i;
        std::fill(o, o + 7, 0.f);
        HitInfo h;
        Ray r;
        if (trace(p, h, r)) {
          Pcg rng;
          pcg32_srandom_r(&rng, static_cast<uint64_t>(p[6]), 0);
          ScatterInfo sc = sample_mat(scene, h, r.dir, rng, p[7] != 0.f);
          o[0] = 1.f;
          o[1] = sc.valid ? 1.f : 0.f;
          if (sc.valid) {
            o[2] = sc.wo.x, o[3] = sc.wo.y, o[4] = sc.wo.z, o[5] = sc.eta;
            o[6] = sc.is_specular ? 1.f : 0.f;
          }
        }
        break;
      }
      case ORACLE_PROBE_LIGHT_SAMPLE: {
        const float* p = in + 4 * i;
        float* o = out + 10 * i;
        std::fill(o, o + 10, 0.f);
        if (scene->num_lights == 0) break;
        Pcg rng;
        pcg32_srandom_r(&rng, static_cast<uint64_t>(p[3]), 0);
        vec3 Le;
        EmitterInfo li;
        lights_sample(scene, vec3{p[0], p[1], p[2]}, rng, Le, li);
        o[0] = Le.x, o[1] = Le.y, o[2] = Le.z, o[3] = li.wi.x, o[4] = li.wi.y, o[5] = li.wi.z;
        o[6] = li.pdf, o[7] = li.dist, o[8] = li.G;
        break;
      }
      case ORACLE_PROBE_BACKGROUND: {
        const float* p = in + 5 * i;
        float* o = out + 4 * i;
        vec3 d{p[0], p[1], p[2]};
        vec3 e = background_emit(scene, d, RayCone{p[3], p[4]});
        o[0] = e.x, o[1] = e.y, o[2] = e.z, o[3] = background_pdf(scene, d);
        break;
      }
      default:
        return -1;
    }
  }
  return 0;
}

}  // extern "C"
