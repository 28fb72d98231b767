// Hand-written gfx950 kernels for v-img's per-pixel path-tracing hot path.
//
// Shape of the computation (DESIGN.md "Kernel"):
//  * one lane owns one pixel for all of its samples, because the reference seeds ONE PCG stream
//    per pixel and draws from it sequentially across samples (include/integrators.h:116-127);
//  * persistent waves: a lane that finishes its pixel pulls the next one from a global counter
//    (one aggregated atomic per wave, __ballot + popcount), so the grid is sized by the machine,
//    not by the image;
//  * the wave advances as a phase machine — regenerate camera rays, closest-hit traversal,
//    hit record, next-event estimation (any-hit traversal), BSDF sample/eval, Russian roulette —
//    and lanes sitting at different bounces of different samples execute the same phase together;
//  * the BVH2 traversal keeps its per-lane stack in LDS ([depth][lane] so a wave's accesses are
//    bank-conflict free) and reads the top of the tree from an LDS copy staged per workgroup;
//  * nothing here is a dense contraction: no MFMA.
//
// Every function cites the reference code whose result it must reproduce.
#pragma once
#include "device_math.h"
#include "device_scene.h"

namespace vimg {

// Register budget of the render kernel = its WPS template argument: waves per SIMD the compiler
// must leave room for (2 -> 256 VGPRs, 3 -> 168, 4 -> 128).  Measured on MI355X (DESIGN.md
// "Occupancy"): scenes that live in LDS/L1 are VALU-bound and fastest at 2 (least spilling);
// scenes of hundreds of MB are latency-bound and fastest at 3.

// ================================================================================ RNG
// pcg32 with initseq = 0 (inc = 1): reference include/rng/pcg_rand.h:15-33, seeded per pixel by
// pcg32_srandom_r(&state, image_index, 0) (include/integrators.h:116)
struct Rng {
  uint64_t s;
};
VD uint32_t pcg_next(Rng& r) {
  uint64_t old = r.s;
  r.s = old * 6364136223846793005ULL + 1ULL;
  uint32_t xorshifted = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = static_cast<uint32_t>(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
VD void pcg_seed(Rng& r, uint64_t initstate) {
  r.s = 0;
  pcg_next(r);
  r.s += initstate;
  pcg_next(r);
}
// Six draws ahead in one step: state * a^6 + (a^5 + a^4 + a^3 + a^2 + a + 1) (the stream's increment is 1).
// A light sample always takes six draws (GroupOfEmitters::sample: one rand_float for the pick, two in
// every emitter's sample(), two pcg draws each: include/geometry/emitters.h:39-56, src/geometry/
// {triangle,sphere}.cpp, include/background.h), so the BSDF sample that follows it in the stream can
// be computed first from the state six draws on.
constexpr uint64_t kPcgA = 6364136223846793005ULL;
constexpr uint64_t pcg_a_pow(int n) { uint64_t r = 1; for (int i = 0; i < n; ++i) r *= kPcgA; return r; }
constexpr uint64_t pcg_c_sum(int n) { uint64_t r = 0; for (int i = 0; i < n; ++i) r += pcg_a_pow(i); return r; }
constexpr uint64_t pcg_step_n(uint64_t s, int n) { for (int i = 0; i < n; ++i) s = s * kPcgA + 1ULL; return s; }
static_assert(pcg_step_n(0x853c49e6748fea9bULL, 6) == 0x853c49e6748fea9bULL * pcg_a_pow(6) + pcg_c_sum(6), "six LCG steps in one");
VD void pcg_skip6(Rng& r) { r.s = r.s * pcg_a_pow(6) + pcg_c_sum(6); }
// rand_float: reference include/rng/sampling.h:85-105
VD float rand_float(Rng& r) {
  uint64_t r1 = pcg_next(r);
  uint64_t r2 = pcg_next(r);
  uint64_t u = (r1 << 32) | r2;
  uint32_t z = (u == 0) ? 64u : static_cast<uint32_t>(__builtin_clzll(u));
  if (z <= 40) {
    uint32_t e = 126 - z;
    uint32_t m = static_cast<uint32_t>(u) & 0x7fffffu;
    return __uint_as_float((e << 23) | m);
  }
  return 0x1.0p-64f * static_cast<float>(static_cast<uint32_t>(u));
}
// random_x_y_r2: reference include/rng/sampling.h:228-239 (a1, a2 are the float constants the
// reference's constexpr expressions evaluate to)
VD f2 random_x_y_r2(uint32_t n) {
  constexpr float g = 1.32471795724474602596;
  constexpr float a1 = 1.0 - (1.0 / g);
  constexpr float a2 = 1.0 - (1.0 / (g * g));
  float x = a1 * static_cast<float>(n);
  float y = a2 * static_cast<float>(n);
  return f2{x - __builtin_floorf(x), y - __builtin_floorf(y)};
}

// ================================================================================ warps
// reference include/rng/sampling.h:15-79
VD f2 sample_disk(float rand1, float rand2) {
  float r = sqrt_f(rand1);
  float phi = 2.f * kPi * rand2;
  double sn, cs;
  D_sincos(phi, sn, cs);
  return f2{r * static_cast<float>(cs), r * static_cast<float>(sn)};
}
VD f3 sample_sphere(float rand1, float rand2) {
  float phi = 2 * kPi * rand1;
  float cos_theta = 2 * rand2 - 1;
  float sin_theta = static_cast<float>(__builtin_sqrt(static_cast<double>(1 - cos_theta * cos_theta)));
  double sn, cs;
  D_sincos(phi, sn, cs);
  float x = cs * sin_theta;   // (double product, one rounding: std::cos(float) is the double function here)
  float y = sn * sin_theta;
  return f3{x, y, cos_theta};
}
// std::lerp(float,float,float) as libstdc++ implements it (exact ends, monotonic)
VD float std_lerp(float a, float b, float t) {
  if ((a <= 0 && b >= 0) || (a >= 0 && b <= 0)) return t * b + (1 - t) * a;
  if (t == 1) return b;
  const float x = a + t * (b - a);
  return (t > 1) == (b > a) ? (b < x ? x : b) : (b > x ? x : b);
}
VD f3 sample_sphere_cap(float rand1, float rand2, float cos_theta_max) {
  float phi = 2 * kPi * rand1;
  float cos_theta = std_lerp(cos_theta_max, 1.0f, rand2);
  float sin_theta = sqrt_f(1 - cos_theta * cos_theta);
  double sn, cs;
  D_sincos(phi, sn, cs);
  float x = cs * sin_theta;
  float y = sn * sin_theta;
  return f3{x, y, cos_theta};
}
VD f3 sample_hemisphere_cosine(float rand1, float rand2) {
  float phi = 2 * kPi * rand1;
  float cos_theta = sqrt_f(rand2);
  float sin_theta = sqrt_f(1 - cos_theta * cos_theta);
  double sn, cs;
  D_sincos(phi, sn, cs);
  float x = static_cast<float>(cs) * sin_theta;
  float y = static_cast<float>(sn) * sin_theta;
  return f3{x, y, cos_theta};
}

// ================================================================================ records
struct RayCone {
  float cone_width, spread_angle;
};
struct Onb {
  f3 u, v, w;
};
// HitInfo (reference include/hit_utils.h:61-74).  Fields only the textured build reads are still
// members; the lean build never computes or uses them and the compiler drops them.
struct Hit {
  f3 p, ns, ng;
  f3 tu, tv;            // n_frame.u, n_frame.v (n_frame.w == ns)
  f2 uv, mr_uv;
  float prim_area, tex_area, curvature;
  uint32_t mat, prim;
};
struct EmitterInfo {
  f3 wi;
  float pdf, dist, G;
};
struct Scatter {
  f3 wo;
  float eta;
  bool is_specular, valid;
};
VD Scatter no_scatter() { return Scatter{f3{0.f, 0.f, 0.f}, 0.f, false, false}; }

struct Counters {
  uint32_t closest, shadow, internal, leaf, prim, sphere;
  uint32_t trip_descend, trip_prim;   // wave-level loop trips (credited to the first active lane)
};
VD bool first_active_lane() {
  const unsigned long long m = __ballot(1);
  return (threadIdx.x & 63u) == static_cast<uint32_t>(__ffsll(static_cast<long long>(m)) - 1);
}

// ONB helpers: reference include/hit_utils.h:32-59
VD f3 xform_with_onb(const Onb& o, f3 v) { return o.u * v.x + o.v * v.y + o.w * v.z; }
VD f3 project_onto_onb(const Onb& o, f3 v) { return f3{dot(v, o.u), dot(v, o.v), dot(v, o.w)}; }
VD f3 gram_schmidt(f3 v, f3 w) { return v - dot(v, w) * w; }
VD void get_axis(f3 n, f3& a, f3& b) {
  if (n.z < (-0.9999999f)) {
    a = f3{0.f, -1.f, 0.f};
    b = f3{-1.f, 0.f, 0.f};
  } else {
    float aa = 1.f / (1.f + n.z);
    float bb = -n.x * n.y * aa;
    a = f3{1.f - n.x * n.x * aa, bb, -n.x};
    b = f3{bb, 1 - n.y * n.y * aa, -n.y};
  }
}
VD Onb init_onb(f3 n) {
  Onb o;
  get_axis(n, o.u, o.v);
  o.w = n;
  return o;
}
VD float luminance(f3 v) { return dot(v, f3{0.212671f, 0.715160f, 0.072169f}); }
VD float pow5(float b) { return b * b * b * b * b; }
VD f3 load3(gptr<float> p) { return f3{p[0], p[1], p[2]}; }
VD f3 load3k(const float* p) { return f3{p[0], p[1], p[2]}; }   // kernel-argument arrays
VD VimgPrim load_prim(gptr<VimgPrim> p) {
  v2u v = *reinterpret_cast<const VIMG_GLOBAL v2u*>(p);
  return VimgPrim{v.x, v.y};
}
VD VimgLight load_light(gptr<VimgLight> p) {
  v2u v = *reinterpret_cast<const VIMG_GLOBAL v2u*>(p);
  return VimgLight{v.x, v.y};
}

// ================================================================================ ray cones
// reference include/ray.h:52-174 (only the textured build carries cones: nothing else reads them)
VD float float_sign(float in) { return in > 0.f ? 1.f : -1.f; }
VD float spread_angle_from_curvature(float mean_curvature, float cone_width, f3 ray_dir, f3 normal) {
  float dn = -dot(ray_dir, normal);
  dn = absf(dn) < 1.0e-5 ? float_sign(dn) * 1.0e-5 : dn;
  return (mean_curvature * cone_width / dn);
}
VD RayCone propagate_reflect_cone(RayCone cone, float surface_spread_angle, float hit_dist) {
  float w = absf(cone.spread_angle * hit_dist + cone.cone_width);
  float a = cone.spread_angle + surface_spread_angle;
  return RayCone{w, a};
}
VD bool refract_with_tir_2d(f2 ray_dir, f2 normal, float eta, f2& out) {
  float n_dot_d = dot(normal, ray_dir);
  float k = 1.0f - eta * eta * (1.0f - n_dot_d * n_dot_d);
  if (k < 0.0f) return false;
  out = ray_dir * eta - normal * (eta * n_dot_d + sqrt_f(k));
  return true;
}
VD void rotate_2d_plus_minus(f2 v, float angle, f2& plus, f2& minus) {
  double sn, cs;
  D_sincos(angle, sn, cs);
  float c = static_cast<float>(cs);
  float s = static_cast<float>(sn);
  float cx = c * v.x, sy = s * v.y, sx = s * v.x, cy = c * v.y;
  plus = f2{cx - sy, +sx + cy};
  minus = f2{cx + sy, -sx + cy};
}
VD f2 orthogonal(f2 v) { return f2{-v.y, v.x}; }
VD RayCone propagate_refract_cone(RayCone rc, f3 ray_in_dir,
                                                       float surface_spread_angle, float eta,
                                                       f3 refracted) {
  f3 normal = -(eta * refracted + ray_in_dir) / length(eta * refracted + ray_in_dir);
  f3 x_axis = normalize(ray_in_dir - normal * dot(normal, ray_in_dir));
  f3 y_axis = normal;
  f2 refracted_2d{dot(refracted, x_axis), dot(refracted, y_axis)};
  f2 incident_2d{dot(ray_in_dir, x_axis), dot(ray_in_dir, y_axis)};
  f2 incident_ortho = orthogonal(incident_2d);
  float width_sign = rc.cone_width > 0.0f ? 1.0f : -1.0f;
  f2 inc_u, inc_l;
  rotate_2d_plus_minus(incident_2d, rc.spread_angle * width_sign * 0.5f, inc_u, inc_l);
  f2 tu = incident_ortho * rc.cone_width * 0.5f;
  f2 tl = -tu;
  float hit_u_x = tu.x + inc_u.x * (-tu.y / inc_u.y);
  float hit_l_x = tl.x + inc_l.x * (-tl.y / inc_l.y);
  float normal_sign = hit_u_x > hit_l_x ? +1.0f : -1.0f;
  f2 n_u, n_l;
  rotate_2d_plus_minus(f2{0.0f, 1.0f}, -surface_spread_angle * normal_sign * 0.5f, n_u, n_l);
  f2 ref_u, ref_l;
  if (!refract_with_tir_2d(inc_u, n_u, eta, ref_u)) {
    ref_u = inc_u - n_u * dot(n_u, inc_u);
    ref_u = normalize(ref_u);
  }
  if (!refract_with_tir_2d(inc_l, n_l, eta, ref_l)) {
    ref_l = inc_l - n_l * dot(n_l, inc_l);
    ref_l = normalize(ref_l);
  }
  float sign_a = (ref_u.x * ref_l.y - ref_u.y * ref_l.x) * normal_sign < 0.0f ? +1.0f : -1.0f;
  float spread = F_acos(dot(ref_u, ref_l)) * sign_a;
  if (is_nan(spread)) spread = 0.f;
  f2 refract_ortho = orthogonal(refracted_2d);
  float width = (-hit_u_x * ref_u.y) / dot(refract_ortho, orthogonal(ref_u));
  width += (hit_l_x * ref_l.y) / dot(refract_ortho, orthogonal(ref_l));
  return RayCone{width, spread};
}

// ================================================================================ textures
// handle_wrapping: reference include/texture/texture_common.h:22-53
VD float handle_wrapping(float coord, uint32_t mode) {
  if (mode == VIMG_WRAP_REPEAT) {
    float fraction = coord - static_cast<float>(static_cast<int>(coord));
    return __builtin_signbit(fraction) ? 1.f + fraction : fraction;
  }
  if (mode == VIMG_WRAP_MIRROR) {
    int int_part = static_cast<int>(coord);
    float fraction = coord - static_cast<float>(int_part);
    if (__builtin_signbit(fraction)) return (int_part % 2) ? absf(fraction) : 1.f + fraction;
    return fraction;
  }
  return clampf(coord, 0.f, 1.f);
}
VD uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
// ImageTexture::col_at_uv_mipmap: reference src/image_texture.cpp:132-160
__device__ __noinline__ f3 col_at_uv_mipmap(const DScene& g, gptr<VimgTexture> t, int level, f2 uv) {
  uint32_t mip_w = umax(t->width >> level, 1u);
  uint32_t mip_h = umax(t->height >> level, 1u);
  float pixel_u = handle_wrapping(uv.x, t->wrap_u) * mip_w;
  float pixel_v = handle_wrapping(uv.y, t->wrap_v) * mip_h;
  int cx = clampi(static_cast<int>(pixel_u), 0, static_cast<int>(mip_w) - 1);
  int cy = clampi(static_cast<int>(pixel_v), 0, static_cast<int>(mip_h) - 1);
  int nx = clampi(cx + 1, 0, static_cast<int>(mip_w) - 1);
  int ny = clampi(cy + 1, 0, static_cast<int>(mip_h) - 1);
  float fx = pixel_u - cx;
  float fy = pixel_v - cy;
  gptr<float> base = g.texels + 3 * t->level_offset[level];
  f3 x0 = load3(base + 3 * (size_t(cx) + size_t(cy) * mip_w));
  f3 x1 = load3(base + 3 * (size_t(nx) + size_t(cy) * mip_w));
  f3 a = mix3(x0, x1, fx);
  f3 y0 = load3(base + 3 * (size_t(cx) + size_t(ny) * mip_w));
  f3 y1 = load3(base + 3 * (size_t(nx) + size_t(ny) * mip_w));
  f3 b = mix3(y0, y1, fx);
  return mix3(a, b, fy);
}
// ImageTexture::col_mipmap_interpolate: reference src/image_texture.cpp:174-189
VD f3 col_mipmap_interpolate(const DScene& g, gptr<VimgTexture> t, float lambda, f2 uv) {
  const int last = static_cast<int>(t->num_levels - 1);
  lambda = clampf(lambda, 0.f, static_cast<float>(t->num_levels - 1));
  int level0 = clampi(static_cast<int>(__builtin_floorf(lambda)), 0, last);
  int level1 = clampi(level0 + 1, 0, last);
  float fraction = lambda - __builtin_floorf(lambda);
  f3 col0 = col_at_uv_mipmap(g, t, level0, uv);
  f3 col1 = col_at_uv_mipmap(g, t, level1, uv);
  return mix3(col0, col1, fraction);
}
// TextureRGB::col_at_ray_hit: ConstColor / Checkerboard (include/texture/texture_RGB.h:45-81),
// ImageTexture (src/image_texture.cpp:162-172 + compute_texture_LOD texture_RGB.h:138-149)
template <bool TEX>
VD f3 col_at_ray_hit(const DScene& g, int tex, f3 ray_in_dir, RayCone cone, const Hit& hit) {
  gptr<VimgTexture> t = g.textures + tex;
  const uint32_t type = t->type;
  if (type == VIMG_TEX_CONST) return load3(t->col_a);
  if (type == VIMG_TEX_CHECKER) {
    uint32_t u_board = static_cast<uint32_t>(__builtin_floorf(hit.uv.x * t->width));
    uint32_t v_board = static_cast<uint32_t>(__builtin_floorf(hit.uv.y * t->height));
    return ((u_board + v_board) % 2 == 0) ? load3(t->col_a) : load3(t->col_b);
  }
  if constexpr (TEX) {
    float lambda = 0.5f * F_log2((hit.tex_area) / hit.prim_area);
    lambda += F_log2(absf(cone.cone_width) / absf(dot(ray_in_dir, hit.ng)));
    lambda += 0.5f * ::log2(static_cast<double>(t->width * t->height));
    if (is_nan(lambda)) lambda = 0.f;
    return col_mipmap_interpolate(g, t, lambda - 2.f, hit.uv);
  }
  return f3{0.f, 0.f, 0.f};
}
// TextureRG::get_at_uv: reference include/texture/texture_RG.h:32-57 (keeps the "* height" index
// of the +x neighbours)
VD f2 rg_get_at_uv(const DScene& g, int tex, f2 uv) {
  gptr<VimgTextureRG> t = g.rg_textures + tex;
  const uint32_t w = t->width, h = t->height;
  float pixel_u = handle_wrapping(uv.x, t->wrap_u) * w;
  float pixel_v = handle_wrapping(uv.y, t->wrap_v) * h;
  int cx = clampi(static_cast<int>(pixel_u), 0, static_cast<int>(w) - 1);
  int cy = clampi(static_cast<int>(pixel_v), 0, static_cast<int>(h) - 1);
  int nx = clampi(cx + 1, 0, static_cast<int>(w) - 1);
  int ny = clampi(cy + 1, 0, static_cast<int>(h) - 1);
  float fx = pixel_u - cx;
  float fy = pixel_v - cy;
  gptr<float> base = g.rg_texels + 2 * t->offset;
  auto at = [&](size_t i) { return f2{base[2 * i], base[2 * i + 1]}; };
  f2 x0 = at(cx + size_t(cy) * w);
  f2 x1 = at(nx + size_t(cy) * h);
  f2 a = mix2(x0, x1, fx);
  f2 y0 = at(cx + size_t(ny) * w);
  f2 y1 = at(nx + size_t(ny) * h);
  f2 b = mix2(y0, y1, fx);
  return mix2(a, b, fy);
}

// ================================================================================ camera
// TLCam::generate_ray (reference src/tl_camera.cpp:25-53) + Ray::xform_ray (include/ray.h:36-41)
VD void generate_ray(const DScene& g, float x, float y, float rand1, float rand2, f3& o, f3& d) {
  float x_dir = (g.p_size0 * (x / static_cast<float>(g.res_x))) - (g.p_size0 / 2.0f);
  float y_dir = (g.p_size1 * (y / static_cast<float>(g.res_y))) - (g.p_size1 / 2.0f);
  f3 ray_dir = normalize(f3{x_dir, y_dir, -1.0f});
  f3 ray_o{0.f, 0.f, 0.f};
  if (g.aperture_radius > 0.f) {
    f2 disk = g.aperture_radius * sample_disk(rand1, rand2);
    f3 ray_origin{disk.x, disk.y, 0.f};
    float ft = g.focal_dist / absf(ray_dir.z);
    f3 focal_plane_p = ray_dir * ft;
    ray_o = ray_origin;
    ray_dir = normalize(focal_plane_p - ray_origin);
  }
  float r[4];
  mat_mul4(g.cam_to_world, ray_dir.x, ray_dir.y, ray_dir.z, 0.0f, r);
  d = f3{r[0], r[1], r[2]};
  mat_mul4(g.cam_to_world, ray_o.x, ray_o.y, ray_o.z, 1.0f, r);
  o = f3{r[0] / r[3], r[1] / r[3], r[2] / r[3]};
  d = normalize(d);
}

// ================================================================================ traversal
// Per-workgroup LDS: SoA planes of the first `n_nodes` (breadth-first = top of tree) nodes and
// the per-lane traversal stacks.
struct Lds {
  const VIMG_LDS v4f* na;
  const VIMG_LDS v4f* nb;
  const VIMG_LDS v4f* nc;
  const VIMG_LDS v2u* nm;
  VIMG_LDS uint32_t* stack;   // already offset to this lane: entry k lives at stack[k * 64]
  uint32_t n_nodes;
};

struct TravRay {
  f3 o, d;
  float min_t, max_t;
};
struct HitRec {
  float e0, e1, e2, inv_det;
  uint32_t prim;     // 0xffffffff = no hit
  uint32_t kind;
};

// slab_intersect_aabb_array: reference include/hit_utils.h:134-151 (scalar path; exact 1/x)
VD float slab(f3 bmin, f3 bmax, f3 o, f3 inv, float min_t, float max_t) {
  f3 lower = (bmin - o) * inv;
  f3 upper = (bmax - o) * inv;
  float lo_x = sel_min(lower.x, upper.x), lo_y = sel_min(lower.y, upper.y),
        lo_z = sel_min(lower.z, upper.z);
  float hi_x = sel_max(lower.x, upper.x), hi_y = sel_max(lower.y, upper.y),
        hi_z = sel_max(lower.z, upper.z);
  float t_min = sel_max(lo_x, sel_max(lo_y, sel_max(lo_z, min_t)));
  float t_max = sel_min(hi_x, sel_min(hi_y, sel_min(hi_z, max_t)));
  return (t_min <= t_max) ? t_min : VIMG_INF;
}

// per-ray constants of the watertight triangle test (reference include/geometry/triangle.h:98-117
// recomputes them per triangle from the ray alone)
struct TriRayConst {
  float sx, sy, sz;
  int kz;
};
VD f3 permute(f3 v, int kz) {
  return kz == 0 ? f3{v.y, v.z, v.x} : (kz == 1 ? f3{v.z, v.x, v.y} : v);
}
VD TriRayConst tri_ray_const(f3 dir) {
  float ax = absf(dir.x), ay = absf(dir.y), az = absf(dir.z);
  int kz = 0;
  float mx = ax;
  if (ay > mx) { kz = 1; mx = ay; }
  if (az > mx) { kz = 2; mx = az; }
  f3 d = permute(dir, kz);
  return TriRayConst{-d.x / d.z, -d.y / d.z, 1.f / d.z, kz};
}
VD float diff_of_products(float a, float b, float c, float d) {
  float cd = c * d;
  return __builtin_fmaf(a, b, -cd);
}
// the reference's fallback uses fmal (x87 long double); binary64 fma is the widest the GPU has —
// reached only when an edge function is exactly 0
VD float diff_of_products_double(float a, float b, float c, float d) {
  double cd = c * d;
  return static_cast<float>(__builtin_fma(static_cast<double>(a), static_cast<double>(b), -cd));
}
// Triangle::tri_hit_template: reference include/geometry/triangle.h:74-180
VD bool tri_test(f3 p0, f3 p1, f3 p2, const TravRay& ray, const TriRayConst& rc, float& t_out,
                 float& e0o, float& e1o, float& e2o, float& inv_det_o) {
  f3 p0t = permute(p0 - ray.o, rc.kz);
  f3 p1t = permute(p1 - ray.o, rc.kz);
  f3 p2t = permute(p2 - ray.o, rc.kz);
  p0t.x += rc.sx * p0t.z;
  p0t.y += rc.sy * p0t.z;
  p1t.x += rc.sx * p1t.z;
  p1t.y += rc.sy * p1t.z;
  p2t.x += rc.sx * p2t.z;
  p2t.y += rc.sy * p2t.z;
  float e0 = diff_of_products(p1t.x, p2t.y, p1t.y, p2t.x);
  float e1 = diff_of_products(p2t.x, p0t.y, p2t.y, p0t.x);
  float e2 = diff_of_products(p0t.x, p1t.y, p0t.y, p1t.x);
  if (e0 == 0.f || e1 == 0.f || e2 == 0.f) {
    e0 = diff_of_products_double(p1t.x, p2t.y, p1t.y, p2t.x);
    e1 = diff_of_products_double(p2t.x, p0t.y, p2t.y, p0t.x);
    e2 = diff_of_products_double(p0t.x, p1t.y, p0t.y, p1t.x);
  }
  if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
  float det = e0 + e1 + e2;
  if (det == 0) return false;
  p0t.z *= rc.sz;
  p1t.z *= rc.sz;
  p2t.z *= rc.sz;
  float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
  if (det < 0 && (t_scaled >= 0 || t_scaled < ray.max_t * det || t_scaled > ray.min_t * det))
    return false;
  else if (det > 0 && (t_scaled <= 0 || t_scaled > ray.max_t * det || t_scaled < ray.min_t * det))
    return false;
  float inv_det = 1.f / det;
  t_out = t_scaled * inv_det;
  e0o = e0, e1o = e1, e2o = e2, inv_det_o = inv_det;
  return true;
}
// The same test with its rejections folded into one predicate (one exec-mask region for the
// hit instead of one per early return: the primitive loop of the walk is dominated by mask
// bookkeeping otherwise).  Comparisons keep the reference's form, so NaN operands take the same
// way out (every ordered comparison false -> accepted, as in the reference).
VD bool tri_test_flat(f3 p0, f3 p1, f3 p2, const TravRay& ray, const TriRayConst& rc, float& t_out,
                      float& e0o, float& e1o, float& e2o, float& inv_det_o) {
  f3 p0t = permute(p0 - ray.o, rc.kz);
  f3 p1t = permute(p1 - ray.o, rc.kz);
  f3 p2t = permute(p2 - ray.o, rc.kz);
  p0t.x += rc.sx * p0t.z;
  p0t.y += rc.sy * p0t.z;
  p1t.x += rc.sx * p1t.z;
  p1t.y += rc.sy * p1t.z;
  p2t.x += rc.sx * p2t.z;
  p2t.y += rc.sy * p2t.z;
  float e0 = diff_of_products(p1t.x, p2t.y, p1t.y, p2t.x);
  float e1 = diff_of_products(p2t.x, p0t.y, p2t.y, p0t.x);
  float e2 = diff_of_products(p0t.x, p1t.y, p0t.y, p1t.x);
  if (e0 == 0.f || e1 == 0.f || e2 == 0.f) {
    e0 = diff_of_products_double(p1t.x, p2t.y, p1t.y, p2t.x);
    e1 = diff_of_products_double(p2t.x, p0t.y, p2t.y, p0t.x);
    e2 = diff_of_products_double(p0t.x, p1t.y, p0t.y, p1t.x);
  }
  const bool mixed = (e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0);
  const float det = e0 + e1 + e2;
  const float t_scaled = e0 * (p0t.z * rc.sz) + e1 * (p1t.z * rc.sz) + e2 * (p2t.z * rc.sz);
  const float hi = ray.max_t * det, lo = ray.min_t * det;
  const bool out_neg = det < 0 && (t_scaled >= 0 || t_scaled < hi || t_scaled > lo);
  const bool out_pos = det > 0 && (t_scaled <= 0 || t_scaled > hi || t_scaled < lo);
  const bool hit = !mixed && !(det == 0) && !out_neg && !out_pos;
  if (hit) {
    const float inv_det = 1.f / det;
    t_out = t_scaled * inv_det;
    e0o = e0, e1o = e1, e2o = e2, inv_det_o = inv_det;
  }
  return hit;
}
// Sphere::sphere_hit_template + solveQuadratic: reference include/geometry/sphere.h:13-100.
// `q` is a double there (unqualified sqrt) — kept, FP64 is half rate on this chip.
VD bool sphere_test(f3 center, float radius, const TravRay& r, float a, float& t_out) {
  float t0, t1;
  const float radius_squared = radius * radius;
  f3 f = r.o - center;
  const float b_prime = dot(-1.0f * f, r.d);
  const float c = dot(f, f) - radius_squared;
  const f3 temp = f + (b_prime / a) * r.d;
  const float discriminant = radius_squared - (dot(temp, temp));
  if (discriminant < 0) return false;
  float sign = (b_prime > 0) ? 1.0f : -1.0f;
  double q = b_prime + sign * (__builtin_sqrt(static_cast<double>(a * discriminant)));
  if (discriminant == 0) {
    t0 = t1 = c / q;
  } else {
    t0 = c / q;
    t1 = q / a;
  }
  if (t0 > t1) {
    float tmp = t0;
    t0 = t1;
    t1 = tmp;
  }
  if (t0 < r.min_t || t0 > r.max_t) {
    t0 = t1;
    if (t0 < r.min_t || t0 > r.max_t) return false;
  }
  t_out = t0;
  return true;
}

// Fast form of the slab test for rays whose direction has no zero component: then no
// (b - o) * inv product can be 0 * inf = NaN, and for NaN-free operands v_min/v_max return
// exactly what the reference's (a<b)?b:a selects return (a +-0 difference cannot survive the
// `max(..., minT)` with minT = 1e-4 > 0 nor flip the <= below).  One max3/min3 tree instead
// of twelve compare+select pairs.
VD float slab_fast(f3 bmin, f3 bmax, f3 o, f3 inv, float min_t, float max_t) {
  f3 lower = (bmin - o) * inv;
  f3 upper = (bmax - o) * inv;
  float t_min = __builtin_fmaxf(
      __builtin_fmaxf(__builtin_fminf(lower.x, upper.x), __builtin_fminf(lower.y, upper.y)),
      __builtin_fmaxf(__builtin_fminf(lower.z, upper.z), min_t));
  float t_max = __builtin_fminf(
      __builtin_fminf(__builtin_fmaxf(lower.x, upper.x), __builtin_fmaxf(lower.y, upper.y)),
      __builtin_fminf(__builtin_fmaxf(lower.z, upper.z), max_t));
  return (t_min <= t_max) ? t_min : VIMG_INF;
}

// Child references: bits [31:25] = primitive count (0 = internal node), [24:0] = index of the
// internal node, or of the leaf's first slot in leaf_prims.  Checked at upload.
constexpr uint32_t REF_DONE = 0xffffffffu;
VD uint32_t ref_count(uint32_t r) { return r >> 25; }
VD uint32_t ref_index(uint32_t r) { return r & 0x1ffffffu; }

// BVH::hit<T>: reference include/bvh.h:83-225.  Node visiting ORDER is the reference's: closest
// hit descends into the nearer child first and keeps the farther one on the stack; any-hit
// visits the second sibling first; a leaf tests its primitives in obj_indices order and, for
// closest hit, the last success wins (ties at t == maxT included).
//
// Wave structure ("while-while"): all lanes first descend through internal nodes until each
// holds a leaf (or is finished), then all lanes holding a leaf intersect it.  The box loop and
// the primitive loop therefore each run with most lanes enabled, instead of every loop trip
// paying for both bodies.
template <bool ANY_HIT>
VD bool traverse(const DScene& g, const Lds& L, TravRay& ray, HitRec& rec, Counters& cnt,
                 bool full_stats) {
  rec.prim = 0xffffffffu;
  const f3 inv{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
  // exact-select slab path only when 0 * inf is possible for this ray
  const bool exact_slab = (ray.d.x == 0.f) || (ray.d.y == 0.f) || (ray.d.z == 0.f);
  float root = slab(load3k(g.root_min), load3k(g.root_max), ray.o, inv, ray.min_t, ray.max_t);
  if (is_inf(root)) return false;
  const TriRayConst rc = tri_ray_const(ray.d);
  const float dir_len2 = dot(ray.d, ray.d);
  uint32_t sp = 0;
  uint32_t cur = g.root_ref;
  bool found = false;
  const uint32_t stat_inc = full_stats ? 1u : 0u;   // event counters without a branch in the loops
  // the box loop in two builds (see render_pool_kernel.h): only a wave that carries a ray with a
  // zero direction component runs the one with the exact select form of the slab test
  auto box_loop = [&](auto exact_possible) {
    while (cur != REF_DONE && ref_count(cur) == 0) {
      v4f na, nb, nc;
      v2u refs;
      if (cur < L.n_nodes) {
        na = L.na[cur], nb = L.nb[cur], nc = L.nc[cur];
        refs = L.nm[cur];
      } else {
        gptr<DNode> n = g.nodes + cur;
        na = n->a, nb = n->b, nc = n->c;
        refs = v2u{n->left_ref, n->right_ref};
      }
      const uint32_t sp_below = sp != 0 ? sp - 1 : 0u;
      const uint32_t popped = L.stack[sp_below * 64];   // what a pop would return, read ahead
      cnt.internal += stat_inc;
      float h1, h2;
      if (decltype(exact_possible)::value && exact_slab) {
        h1 = slab(f3{na.x, na.y, na.z}, f3{na.w, nb.x, nb.y}, ray.o, inv, ray.min_t, ray.max_t);
        h2 = slab(f3{nb.z, nb.w, nc.x}, f3{nc.y, nc.z, nc.w}, ray.o, inv, ray.min_t, ray.max_t);
      } else {
        h1 = slab_fast(f3{na.x, na.y, na.z}, f3{na.w, nb.x, nb.y}, ray.o, inv, ray.min_t, ray.max_t);
        h2 = slab_fast(f3{nb.z, nb.w, nc.x}, f3{nc.y, nc.z, nc.w}, ray.o, inv, ray.min_t, ray.max_t);
      }
      const bool in1 = !is_inf(h1), in2 = !is_inf(h2);
      const uint32_t c1 = refs.x, c2 = refs.y;
      const bool both = in1 && in2, any = in1 || in2;
      const bool first_is_near = ANY_HIT ? false : (h2 > h1);
      const uint32_t near_c = first_is_near ? c1 : c2;
      const uint32_t far_c = first_is_near ? c2 : c1;
      L.stack[sp * 64] = far_c;   // above the top of the stack: kept only when both were hit
      const uint32_t one_c = in1 ? c1 : c2;
      cur = both ? near_c : (any ? one_c : (sp != 0 ? popped : REF_DONE));
      sp = both ? sp + 1 : (any ? sp : sp_below);
    }
  };
  while (cur != REF_DONE) {
    // ---- descend: box tests until this lane holds a leaf
    if (__any(exact_slab))
      box_loop(std::true_type{});
    else
      box_loop(std::false_type{});
    // ---- intersect the leaf this lane holds
    if (cur != REF_DONE) {
      const uint32_t first = ref_index(cur), count = ref_count(cur);
      cnt.leaf += stat_inc;
      for (uint32_t i = 0; i < count; ++i) {
        gptr<DLeafPrim> lp = g.leaf_prims + (first + i);
        const v4f a = lp->a, b = lp->b;
        const float c0 = lp->c0;
        const uint32_t kind = lp->kind;
        cnt.prim += stat_inc;
        bool hit = false;
        float t = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f, idet = 0.f;
        if (kind == 0) {
          hit = tri_test_flat(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c0}, ray, rc, t, e0,
                              e1, e2, idet);
        } else if (kind == 1) {
          cnt.sphere += stat_inc;
          hit = sphere_test(f3{a.x, a.y, a.z}, a.w, ray, dir_len2, t);
        }
        if (hit) {
          ray.max_t = t;
          found = true;
          if (ANY_HIT) break;
          rec.e0 = e0, rec.e1 = e1, rec.e2 = e2, rec.inv_det = idet;
          rec.prim = lp->prim;
          rec.kind = kind;
        }
      }
      if (ANY_HIT && found) {
        cur = REF_DONE;   // exit on first hit
      } else if (sp != 0) {
        --sp;
        cur = L.stack[sp * 64];
      } else {
        cur = REF_DONE;
      }
    }
  }
  return found;
}

// ================================================================================ hit records
VD f2 load_uv(const DScene& g, gptr<VimgMesh> m, uint32_t set, uint32_t vtx) {
  gptr<float> p = g.uvs + 2 * (size_t(m->uv_offset[set]) + (vtx - m->first_vertex));
  return f2{p[0], p[1]};
}
// Triangle::hit_info: reference src/geometry/triangle.cpp:13-153
template <bool TEX>
VD void tri_hit_info(const DScene& g, uint32_t tri, const HitRec& rec, uint32_t prim_id, Hit& h) {
  gptr<DTriShade> ts = g.tri_shade + tri;
  const f3 p0{ts->p[0], ts->p[1], ts->p[2]}, p1{ts->p[3], ts->p[4], ts->p[5]},
      p2{ts->p[6], ts->p[7], ts->p[8]};
  gptr<VimgMesh> mesh = g.meshes + ts->mesh;
  const uint32_t i0 = ts->i0, i1 = ts->i1, i2 = ts->i2;
  const uint32_t mat = mesh->material;
  const uint32_t mflags = g.material_flags[mat];
  f3 edge1 = p1 - p0, edge2 = p2 - p0;
  float u = rec.e0 * rec.inv_det, v = rec.e1 * rec.inv_det, w = rec.e2 * rec.inv_det;
  const f3 tri_normal{ts->n[0], ts->n[1], ts->n[2]};   // normalize(cross(edge1, edge2)), baked
  f3 n0 = tri_normal, n1 = tri_normal, n2 = tri_normal, shading_normal = tri_normal;
  if (mesh->has_normals) {
    n0 = load3(g.normals + 3 * size_t(i0));
    n1 = load3(g.normals + 3 * size_t(i1));
    n2 = load3(g.normals + 3 * size_t(i2));
    shading_normal = normalize(u * n0 + v * n1 + w * n2);
  }
  h.p = u * p0 + v * p1 + w * p2;
  h.ng = tri_normal;
  h.mat = mat;
  h.prim = prim_id;
  h.ns = shading_normal;
  // the frame / uv block feeds Principled (n_frame) and non-constant textures only; the normal
  // map (textured build) also changes the shading normal, so it cannot be skipped there
  const bool need = TEX || (mflags & (MATF_NEEDS_UV | MATF_NEEDS_FRAME));
  if (!need) return;
  f2 uv{u, v};
  f2 uv0{0.f, 0.f}, uv1{1.f, 0.f}, uv2{1.f, 1.f};
  const uint32_t cset = mesh->color_tex_uv;
  if (cset != VIMG_NO_UV) {
    uv0 = load_uv(g, mesh, cset, i0), uv1 = load_uv(g, mesh, cset, i1), uv2 = load_uv(g, mesh, cset, i2);
    uv = u * uv0 + v * uv1 + w * uv2;
  }
  h.uv = uv;
  h.mr_uv = uv;
  if constexpr (TEX) {
    const uint32_t mset = mesh->metallic_roughness_tex_uv;
    if (mset != VIMG_NO_UV) {
      f2 m0 = load_uv(g, mesh, mset, i0), m1 = load_uv(g, mesh, mset, i1),
         m2 = load_uv(g, mesh, mset, i2);
      h.mr_uv = u * m0 + v * m1 + w * m2;
    }
  }
  f2 duvds = uv2 - uv0;
  f2 duvdt = uv2 - uv1;
  float det = duvds.x * duvdt.y - duvdt.x * duvds.y;
  float dsdu = 0.f, dtdu = 0.f, dsdv = 0.f, dtdv = 0.f;
  f3 dpdu, dpdv;
  if (absf(det) > 1e-8f && !is_nan(det)) {
    dsdu = duvdt.y / det;
    dtdu = -duvds.y / det;
    dsdv = duvdt.x / det;
    dtdv = -duvds.x / det;
    f3 dpds = p2 - p0;
    f3 dpdt = p2 - p1;
    dpdu = dpds * dsdu + dpdt * dtdu;
    dpdv = dpds * dsdv + dpdt * dtdv;
  } else {
    get_axis(shading_normal, dpdu, dpdv);
  }
  if constexpr (TEX) {
    const int nmap = g.materials[mat].normal_map;
    if (nmap >= 0) {
      f2 n_uv{u, v};
      const uint32_t nset = mesh->normal_tex_uv;
      if (nset != VIMG_NO_UV) {
        f2 q0 = load_uv(g, mesh, nset, i0), q1 = load_uv(g, mesh, nset, i1),
           q2 = load_uv(g, mesh, nset, i2);
        n_uv = u * q0 + v * q1 + w * q2;
      }
      f3 n_tangent_space = normalize(col_at_uv_mipmap(g, g.textures + nmap, 0, n_uv));
      Onb onb_n_map = init_onb(shading_normal);
      f3 local_space_normal = xform_with_onb(onb_n_map, n_tangent_space);
      float ulen = length(dpdu), vlen = length(dpdv);
      dpdu = normalize(gram_schmidt(dpdu, local_space_normal)) * ulen;
      dpdv = normalize(cross(local_space_normal, dpdu)) * vlen;
      shading_normal = local_space_normal;
      h.ns = shading_normal;
    }
  }
  f3 tangent = normalize(dpdu - shading_normal * dot(shading_normal, dpdu));
  f3 bitangent = normalize(cross(shading_normal, tangent));
  h.tu = tangent;
  h.tv = bitangent;
  if constexpr (TEX) {
    f3 dnds = n2 - n0;
    f3 dndt = n2 - n1;
    f3 dndu = dnds * dsdu + dndt * dtdu;
    f3 dndv = dnds * dsdv + dndt * dtdv;
    h.curvature = (dot(dndu, tangent) + dot(dndv, bitangent)) / 2.f;
    h.prim_area = length(cross(p1 - p0, p2 - p0));   // twice the triangle area, as the reference
    h.tex_area = absf((uv1.x - uv0.x) * (uv2.y - uv0.y) - (uv2.x - uv0.x) * (uv1.y - uv0.y));
  }
}
// Sphere::hit_info: reference src/geometry/sphere.cpp:12-45
template <bool TEX>
VD void sphere_hit_info(const DScene& g, uint32_t sphere, const TravRay& r, uint32_t prim_id,
                        Hit& h) {
  gptr<VimgSphere> sp = g.spheres + sphere;
  const f3 center = load3(sp->center);
  const float radius = sp->radius;
  const uint32_t mat = sp->material;
  const uint32_t mflags = g.material_flags[mat];
  h.p = r.o + r.d * r.max_t;
  const f3 normal = normalize(h.p - center);
  h.ns = normal;
  h.ng = normal;
  h.mat = mat;
  h.prim = prim_id;
  if (TEX || (mflags & MATF_NEEDS_UV)) {
    float theta = F_acos(-normal.y);
    float phi = F_atan2(-normal.z, normal.x) + kPi;
    float u = phi / (2.f * kPi);
    float v = theta / kPi;
    h.uv = f2{u, v};
    h.mr_uv = f2{u, v};
  }
  if (TEX || (mflags & MATF_NEEDS_FRAME)) {
    f3 dpdu{-radius * normal.y, radius * normal.x, 0.f};
    f3 tangent = normalize(dpdu - normal * dot(normal, dpdu));
    h.tu = tangent;
    h.tv = normalize(cross(normal, tangent));
  }
  if constexpr (TEX) {
    // (the same fields in the same order as the triangle's record: the compiler sinks the two branches' last
    // stores into one, and a store through a choice of two fields keeps both fields in scratch memory)
    h.curvature = 1.f / radius;
    h.prim_area = 1.f;
    h.tex_area = 0.000001f;
  }
}
template <bool TEX>
VD void make_hit_info(const DScene& g, const HitRec& rec, const TravRay& ray, Hit& h) {
  const VimgPrim p = load_prim(g.prims + rec.prim);
  if (p.type == VIMG_PRIM_TRIANGLE)
    tri_hit_info<TEX>(g, p.index, rec, rec.prim, h);
  else
    sphere_hit_info<TEX>(g, p.index, ray, rec.prim, h);
}

// ================================================================================ materials
// emitted: base 0 (include/material/material.h:63-66), DiffuseLight one-sided
// (include/material/diffuse_light.h:30-38)
VD f3 mat_emitted(gptr<VimgMaterial> m, f3 ray_dir, f3 shading_normal) {
  if (m->type != VIMG_MAT_DIFFUSE_LIGHT) return f3{0.f, 0.f, 0.f};
  return (dot(shading_normal, ray_dir) < 0) ? load3(m->emit) : f3{0.f, 0.f, 0.f};
}

// ---- Dielectric: reference src/material/dielectric.cpp:5-69
VD f3 reflect_dir(f3 wi, f3 n) { return wi - (2.f * dot(wi, n) * n); }
VD float schlick_apprx(float cosine, float in_ior, float out_ior) {
  float r0 = (in_ior - out_ior) / (in_ior + out_ior);
  r0 = r0 * r0;
  return r0 + (1.f - r0) * pow5(1.f - cosine);
}
VD f3 refract_dir(f3 wi, f3 n, float i_over_o, float cos_i, float sin2_t) {
  float normal_mul = (i_over_o * cos_i) - sqrt_f(1.f - sin2_t);
  return (i_over_o * wi) + (normal_mul * n);
}
VD Scatter dielectric_sample(gptr<VimgMaterial> m, const Hit& hit, f3 wi, Rng& rng) {
  const float ior = m->ior;
  f3 wo;
  float eta;
  bool front_face = dot(wi, hit.ns) < 0;
  f3 n = front_face ? hit.ns : -hit.ns;
  const float cos_i = -1.f * (dot(wi, n));
  float randf = rand_float(rng);
  if (front_face) {
    eta = ior;
    const float schlick = schlick_apprx(cos_i, 1.0f, ior);
    if (schlick > randf) {
      wo = reflect_dir(wi, n);
    } else {
      const float i_over_o = 1.f / ior;
      const float sin2 = (i_over_o * i_over_o) * (1.f - (cos_i * cos_i));
      wo = refract_dir(wi, n, i_over_o, cos_i, sin2);
    }
  } else {
    eta = 1.f / ior;
    const float i_over_o = ior;
    const float sin2 = (i_over_o * i_over_o) * (1.f - (cos_i * cos_i));
    if ((sin2 > 1.f) || (schlick_apprx(sqrt_f(1.f - sin2), ior, 1.f) > randf)) {
      wo = reflect_dir(wi, n);
    } else {
      wo = refract_dir(wi, n, i_over_o, cos_i, sin2);
    }
  }
  return Scatter{wo, eta, true, true};
}

// ---- Disney lobes: reference include/material/disney_helpers/*.h
// G_w: disney_common.h:6-14 (double-promoted by its 1. / 2. literals)
VD float g_w(f3 w, float alphax, float alphay, const Onb& frame) {
  const f3 wl = project_onto_onb(frame, w);
  float vec_alpha = ((wl.x * alphax) * (wl.x * alphax) + (wl.y * alphay) * (wl.y * alphay))
                    / (wl.z * wl.z);
  float caret = (__builtin_sqrt(1. + static_cast<double>(vec_alpha)) - 1.) / 2.;
  return 1. / (1. + static_cast<double>(caret));
}
// ---------------------------------------------------------------------------- BSDF sampling
// Material::sample_mat for every material (virtual in the reference,
// include/material/material.h:37-40), restated as ONE staged routine instead of one routine per
// lobe.  All continuous lobes have the same skeleton — early-out tests, two rand_float draws, an
// azimuth phi = 2*pi*r evaluated through cos/sin, a few lobe-specific lines — so a wave whose
// lanes sit on different materials and lobes executes the expensive part (the RNG and the
// double-precision sincos) once for everybody and only the short tails per lobe:
//   Lambertian::sample_mat            src/material/lambertian.cpp:5-30
//   Dielectric::sample_mat            src/material/dielectric.cpp:29-69
//   Principled::sample_mat            src/material/principled.cpp:3-58
//   sample_disney_diffuse             disney_diffuse.h:52-71
//   sample_disney_clearcoat           disney_clearcoat.h:60-105
//   sample_disney_metal               disney_metal.h:78-120
//   sample_disney_rough_glass         disney_glass.h:108-186
//   anisotropic_sample_visible_normals disney_common.h:16-52
// The order of RNG draws of each branch is the reference's.

// sin and cos of an azimuth in [0, 2*pi] (float 2*pi*r rounds to at most 6.2831855) in double.
// Reduction by multiples of pi/2 (two-term Cody-Waite, k <= 4) and the fdlibm kernels; error
// < 1 ulp(double), i.e. after narrowing to float the same value as libm's cos/sin in all but
// ~1e-8 of arguments — the same statement that holds for OCML's routines, at a third of the cost.
VD void sincos_azimuth(float phi_f, double& s_out, double& c_out) {
  const double x = static_cast<double>(phi_f);
  const double kd = __builtin_rint(x * 6.36619772367581382433e-01);
  const int k = static_cast<int>(kd);
  double r = __builtin_fma(-kd, 1.57079632673412561417e+00, x);
  r = __builtin_fma(-kd, 6.07710050650619224932e-11, r);
  const double z = r * r;
  // __kernel_sin
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  const double sn = __builtin_fma(z * r, __builtin_fma(z, ps, -1.66666666666666324348e-01), r);
  // __kernel_cos
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double wv = 1.0 - hz;
  const double cs = wv + (((1.0 - wv) - hz) + z * (z * pc));
  const bool swap = (k & 1) != 0;
  double s = swap ? cs : sn;
  double c = swap ? sn : cs;
  if (k & 2) s = -s;
  if ((k + 1) & 2) c = -c;
  s_out = s, c_out = c;
}

enum : uint32_t {
  LOBE_NONE = 0,      // no scatter (base class / early-out)
  LOBE_COSINE = 1,    // cosine hemisphere about a frame: Lambertian and the Disney diffuse lobe
  LOBE_CLEARCOAT = 2,
  LOBE_METAL = 3,     // visible-normal sampling, reflect
  LOBE_GLASS = 4,     // visible-normal sampling, Fresnel reflect/refract (one more draw)
  LOBE_DIELECTRIC = 5
};

// fresnel_dielectric: disney_common.h:54-68
VD float fresnel_dielectric(float n_dot_i, float eta) {
  float n_dot_t_sq = 1.f - (1.f - n_dot_i * n_dot_i) / (eta * eta);
  if (n_dot_t_sq < 0) return 1;
  float n_dot_t = sqrt_f(n_dot_t_sq);
  n_dot_i = absf(n_dot_i);
  float rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
  float rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
  return (rs * rs + rp * rp) / 2;
}
VD float fd_term(f3 n, f3 w, float fd90) {   // FD, disney_diffuse.h:9-11
  return 1.f + (fd90 - 1.f) * pow5(1.f - sel_max(dot(n, w), 0.f));
}
VD void regularize_alpha(float& ax, float& ay) {   // MatConst, include/material/material.h:19-23
  ax = ax < 0.1f ? clampf(2.f * ax, 0.03f, 0.1f) : ax;
  ay = ay < 0.1f ? clampf(2.f * ay, 0.03f, 0.1f) : ay;
}

struct PrincipledCommon {
  f3 dir_in;
  Onb frame;
  float metallic, roughness;
};
// shared prologue of Principled::eval_pdf / sample_mat: principled.h:103-119, principled.cpp:5-21
template <bool TEX>
VD PrincipledCommon principled_prologue(const DScene& g, gptr<VimgMaterial> m, f3 wi, const Hit& hit) {
  PrincipledCommon p;
  p.dir_in = -wi;
  p.frame = Onb{hit.tu, hit.tv, hit.ns};
  if ((dot(hit.ns, p.dir_in) * dot(hit.ng, p.dir_in)) < 0) {
    p.frame.u = -p.frame.u;
    p.frame.v = -p.frame.v;
    p.frame.w = -p.frame.w;
  }
  f2 m_r{1.f, 1.f};
  if constexpr (TEX) {
    if (m->mr_tex >= 0) m_r = rg_get_at_uv(g, m->mr_tex, hit.mr_uv);
  }
  m_r = m_r * f2{m->metallic_factor, m->roughness_factor};
  p.metallic = m_r.x;
  p.roughness = m_r.y;
  return p;
}

// Principled::eval_pdf<pair>: reference include/material/principled.h:100-205 with the lobe
// helpers eval_pdf_disney_{rough_glass,diffuse,clearcoat,metal} and eval_disney_sheen
// (disney_glass.h:188-234, disney_diffuse.h:73-104, disney_clearcoat.h:107-139,
// disney_metal.h:122-152, disney_sheen.h:10-27)
template <bool TEX>
VD void principled_eval_pdf(const DScene& g, gptr<VimgMaterial> m, f3 wi,
                                                 f3 wo, const Hit& hit, RayCone cone,
                                                 bool regularize, f3& f_out, float& pdf_out) {
  const PrincipledCommon pc = principled_prologue<TEX>(g, m, wi, hit);
  const f3 dir_in = pc.dir_in;
  const Onb& frame = pc.frame;
  const float metallic = pc.metallic, roughness = pc.roughness;
  const f3 base_color = col_at_ray_hit<TEX>(g, m->tex, wi, cone, hit);
  const f3 half_vector = normalize(dir_in + wo);
  constexpr float alpha_min = 0.0001;
  const float aspect = sqrt_f(1.f - 0.9f * m->anisotropic);
  const float roughness_clamp = clampf(roughness, 0.01f, 1.f);
  const float roughness_square = roughness_clamp * roughness_clamp;
  float alphax = sel_max(alpha_min, roughness_square / aspect);
  float alphay = sel_max(alpha_min, roughness_square * aspect);
  if (regularize) regularize_alpha(alphax, alphay);
  const float g_in = g_w(dir_in, alphax, alphay, frame);
  const float G = g_in * g_w(wo, alphax, alphay, frame);
  const float ng_in = dot(hit.ng, dir_in);
  const float ng_out = dot(hit.ng, wo);

  // ---- rough glass (always evaluated first, as the reference does)
  f3 eval_glass;
  float pdf_glass;
  {
    const float mat_eta = m->eta;
    float in_geo_dot = dot(dir_in, hit.ng);
    bool reflect = (in_geo_dot * dot(hit.ng, wo)) >= 0;
    float eta = in_geo_dot >= 0 ? mat_eta : 1.f / mat_eta;
    f3 hv = half_vector;
    if (!reflect) hv = normalize(dir_in + wo * eta);
    float h_dot_in = dot(hv, dir_in);
    float F = fresnel_dielectric(h_dot_in, eta);
    const f3 lh = project_onto_onb(frame, hv);
    float had = (lh.x * lh.x) / (alphax * alphax) + (lh.y * lh.y) / (alphay * alphay)
                + (lh.z * lh.z);
    float D = 1. / (kPi * alphax * alphay * (had * had));
    float normal_in_dot = dot(frame.w, dir_in);
    if (reflect) {
      eval_glass = base_color * (F * D * G) / (4.f * absf(normal_in_dot));
      pdf_glass = (F * D * g_in) / (4.f * absf(normal_in_dot));
    } else {
      float eta_factor = 1.f / (eta * eta);
      float h_dot_out = dot(hv, wo);
      float sqrt_denom = h_dot_in + eta * h_dot_out;
      eval_glass = f3{sqrt_f(base_color.x), sqrt_f(base_color.y), sqrt_f(base_color.z)}
                   * (eta_factor * (1 - F) * D * G * eta * eta * absf(h_dot_out * h_dot_in))
                   / (absf(normal_in_dot) * sqrt_denom * sqrt_denom);
      float dh_dout = eta * eta * h_dot_out / (sqrt_denom * sqrt_denom);
      pdf_glass = (1.f - F) * D * g_in * absf(dh_dout * h_dot_in / normal_in_dot);
    }
  }
  const float st = m->specular_transmission;
  if (ng_in < 0) {
    f_out = (1.f - metallic) * st * eval_glass;
    pdf_out = pdf_glass;
    return;
  }
  const bool below = (ng_in < 0 || ng_out < 0);   // "No light below the surface" guards

  // ---- sheen
  f3 eval_sheen{0.f, 0.f, 0.f};
  // ---- diffuse
  f3 eval_diff{0.f, 0.f, 0.f};
  float pdf_diff = 0.f;
  // ---- clearcoat
  f3 eval_clearcoat{0.f, 0.f, 0.f};
  float pdf_clearcoat = 0.f;
  // ---- metal
  f3 eval_metal{0.f, 0.f, 0.f};
  float pdf_metal = 0.f;

  float alpha_g = (1.f - m->clearcoat_gloss) * 0.1f + m->clearcoat_gloss * 0.001f;
  alpha_g = regularize && (alpha_g < 0.1f) ? clampf(2.f * alpha_g, 0.03f, 0.1f) : alpha_g;

  if (!below) {
    const float base_lum = luminance(base_color);
    const f3 c_tint = base_lum > 0 ? base_color / base_lum : splat3(1.f);
    {
      const float sheen_tint = m->sheen_tint;
      f3 c_sheen = (splat3(1.f) - splat3(sheen_tint)) + sheen_tint * c_tint;
      eval_sheen = c_sheen * pow5(1.f - sel_max(dot(half_vector, wo), 0.f))
                   * sel_max(dot(frame.w, wo), 0.f);
    }
    {
      const float subsurface = m->subsurface;
      float normal_dirout_dot = dot(frame.w, wo);
      const float cos_theta_out = sel_max(normal_dirout_dot, 0.f);
      const float cos_theta_in = sel_max(dot(frame.w, dir_in), 0.f);
      const float dot_h_out = sel_max(dot(half_vector, wo), 0.f);
      const float fd90 = 0.5 + 2.0 * roughness * dot_h_out * dot_h_out;
      const f3 base_diffuse = base_color * kInvPiF * fd_term(frame.w, dir_in, fd90)
                              * fd_term(frame.w, wo, fd90) * cos_theta_out;
      const float fss90 = roughness * dot_h_out * dot_h_out;
      f3 ss_diffuse = base_color * 1.25f * kInvPiF
                      * (fd_term(frame.w, dir_in, fss90) * fd_term(frame.w, wo, fss90)
                             * ((1.f / (cos_theta_out + cos_theta_in)) - 0.5f)
                         + 0.5f)
                      * cos_theta_out;
      eval_diff = (1.f - subsurface) * base_diffuse + subsurface * ss_diffuse;
      pdf_diff = sel_max(normal_dirout_dot, 0.f) * kInvPiF;
    }
    {
      constexpr float R0 = ((1.5f - 1.f) * (1.5f - 1.f)) / ((1.5f + 1.f) * (1.5f + 1.f));
      float h_dirout_dot = absf(dot(half_vector, wo));
      float fresnel = R0 + (1. - R0) * pow5(1.f - h_dirout_dot);
      float Gc = g_w(dir_in, 0.25, 0.25, frame) * g_w(wo, 0.25, 0.25, frame);
      const float ag2 = alpha_g * alpha_g;
      const f3 lh = project_onto_onb(frame, half_vector);
      float D = (ag2 - 1.f) / (kPi * F_log(ag2) * (1. + (ag2 - 1.) * lh.z * lh.z));
      float clearcoat_eval = (fresnel * D * Gc) / (4.f * absf(dot(frame.w, dir_in)));
      pdf_clearcoat = (D * absf(dot(frame.w, half_vector))) / (4.f * h_dirout_dot);
      eval_clearcoat = splat3(clearcoat_eval);
    }
    {
      const float spec_tint = m->specular_tint, specular = m->specular, eta = m->eta;
      f3 k_s = (splat3(1.f) - splat3(spec_tint)) + spec_tint * c_tint;
      float R0 = ((eta - 1.f) * (eta - 1.f)) / ((eta + 1.f) * (eta + 1.f));
      f3 c_0 = (specular * R0 * (1.f - metallic)) * k_s + metallic * base_color;
      f3 fresnel = c_0 + (splat3(1.f) - c_0) * pow5(1.f - dot(half_vector, wo));
      const f3 lh = project_onto_onb(frame, half_vector);
      float had = (lh.x * lh.x) / (alphax * alphax) + (lh.y * lh.y) / (alphay * alphay)
                  + (lh.z * lh.z);
      float D = 1. / (kPi * alphax * alphay * (had * had));
      float d_mul_denominator = D / (4.f * absf(dot(frame.w, dir_in)));
      eval_metal = fresnel * G * d_mul_denominator;
      pdf_metal = g_in * d_mul_denominator;
    }
  }
  const float clearcoat = m->clearcoat, sheen = m->sheen;
  f3 eval_principled = ((1.f - st) * (1.f - metallic) * eval_diff)
                       + ((1.f - metallic) * sheen * eval_sheen)
                       + (0.25f * clearcoat * eval_clearcoat)
                       + ((1.f - st * (1.f - metallic)) * eval_metal)
                       + ((1.f - metallic) * st * eval_glass);
  float diffuse_weight = (1.f - metallic) * (1.f - st);
  float clearcoat_weight = 0.25f * clearcoat;
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

// MT: material type known at compile time (a batch of one material class), -1 = read it
template <bool TEX, int MT = -1>
VD Scatter sample_mat(const DScene& g, const Hit& hit, f3 wi, Rng& rng, bool regularize) {
  gptr<VimgMaterial> m = g.materials + hit.mat;
  const uint32_t type = MT >= 0 ? uint32_t(MT) : m->type;
  const f3 dir_in = -wi;
  uint32_t lobe = LOBE_NONE;
  Onb frame{f3{1.f, 0.f, 0.f}, f3{0.f, 1.f, 0.f}, f3{0.f, 0.f, 1.f}};
  float alphax = 0.f, alphay = 0.f, eta = 0.f, alpha_g = 0.f;
  bool lambert_front = true;

  // ---- stage A: which lobe, its frame and parameters; early-outs that precede any draw
  if (type == VIMG_MAT_LAMBERTIAN) {
    lobe = LOBE_COSINE;   // draws first, front-face test afterwards (lambertian.cpp:7-29)
    lambert_front = dot(wi, hit.ns) < 0;
    frame = init_onb(lambert_front ? hit.ns : -hit.ns);
  } else if (type == VIMG_MAT_DIELECTRIC) {
    lobe = LOBE_DIELECTRIC;
  } else if (type == VIMG_MAT_PRINCIPLED) {
    const PrincipledCommon pc = principled_prologue<TEX>(g, m, wi, hit);
    frame = pc.frame;
    const float metallic = pc.metallic;
    float roughness = pc.roughness;
    const float ng_in = dot(hit.ng, dir_in);
    if (ng_in < 0) {
      lobe = LOBE_GLASS;   // inside the surface: glass only, no lobe draw (principled.cpp:23-26)
    } else {
      const float st = m->specular_transmission;
      float diffuse_weight = (1.f - metallic) * (1.f - st);
      float clearcoat_weight = 0.25f * m->clearcoat;
      float metal_weight = (1.f - st * (1.f - metallic));
      float glass_weight = (1.f - metallic) * st;
      float total_w = diffuse_weight + clearcoat_weight + metal_weight + glass_weight;
      float choose_diff = diffuse_weight / total_w;
      float choose_clearcoat = clearcoat_weight / total_w;
      float choose_metal = metal_weight / total_w;
      float choose_glass = glass_weight / total_w;
      float rnd = rand_float(rng);
      if (rnd <= choose_diff) {
        lobe = LOBE_COSINE;   // ng_in >= 0 here, so sample_disney_diffuse's early-out cannot fire
      } else if (rnd > choose_diff && rnd <= (choose_diff + choose_clearcoat)) {
        lobe = LOBE_CLEARCOAT;
      } else if (rnd > (choose_diff + choose_clearcoat)
                 && rnd <= (choose_diff + choose_clearcoat + choose_metal)) {
        lobe = LOBE_METAL;
      } else if (rnd > (choose_diff + choose_clearcoat + choose_metal)
                 && rnd <= (choose_diff + choose_clearcoat + choose_metal + choose_glass)) {
        lobe = LOBE_GLASS;
      }
    }
    if (lobe == LOBE_CLEARCOAT) {
      const float gloss = m->clearcoat_gloss;
      alpha_g = (1.f - gloss) * 0.1f + gloss * 0.001f;
      if (regularize && alpha_g < 0.1f) alpha_g = clampf(2.f * alpha_g, 0.03f, 0.1f);
    } else if (lobe == LOBE_METAL || lobe == LOBE_GLASS) {
      constexpr float alpha_min = 0.0001;
      float aspect = sqrt_f(1.f - 0.9f * m->anisotropic);
      // the glass lobe clamps roughness, the metal lobe does not (disney_glass.h:118 vs
      // disney_metal.h:92-96)
      if (lobe == LOBE_GLASS) roughness = clampf(roughness, 0.01f, 1.f);
      float roughness_square = roughness * roughness;
      alphax = sel_max(alpha_min, roughness_square / aspect);
      alphay = sel_max(alpha_min, roughness_square * aspect);
      if (regularize) regularize_alpha(alphax, alphay);
      const float mat_eta = m->eta;
      eta = ng_in >= 0 ? mat_eta : 1.f / mat_eta;
    }
  }
  if (lobe == LOBE_DIELECTRIC) return dielectric_sample(m, hit, wi, rng);
  if (lobe == LOBE_NONE) return no_scatter();

  // ---- stage B: the two draws and the azimuth every continuous lobe needs
  const float ra = rand_float(rng);
  const float rb = rand_float(rng);
  const float phi = 2 * kPi * ((lobe == LOBE_CLEARCOAT) ? rb : ra);
  double sin_phi, cos_phi;
  sincos_azimuth(phi, sin_phi, cos_phi);

  // ---- stage C: lobe tails
  if (lobe == LOBE_COSINE) {
    // sample_hemisphere_cosine (include/rng/sampling.h:69-79): std::cos/std::sin of a float
    float cos_theta = sqrt_f(rb);
    float sin_theta = sqrt_f(1 - cos_theta * cos_theta);
    f3 local{static_cast<float>(cos_phi) * sin_theta, static_cast<float>(sin_phi) * sin_theta,
             cos_theta};
    f3 dir = xform_with_onb(frame, local);
    if (type == VIMG_MAT_LAMBERTIAN) {
      if (lambert_front) return Scatter{dir, 0.f, false, true};
      return no_scatter();
    }
    if (dot(hit.ng, dir) <= 0) return no_scatter();
    return Scatter{dir, 0.f, false, true};
  }
  if (lobe == LOBE_CLEARCOAT) {
    const float alpha = alpha_g;
    float cos2 = (1.f - ::pow(static_cast<double>(alpha * alpha), 1. - ra)) / (1.f - (alpha * alpha));
    float cos_elevation = sqrt_f(cos2);
    float sin_elevation = sqrt_f(1 - cos2);
    f3 local_h{sin_elevation * static_cast<float>(cos_phi), sin_elevation * static_cast<float>(sin_phi),
               cos_elevation};
    if (dot(frame.w, dir_in) < 0) {
      frame.u = -frame.u;
      frame.v = -frame.v;
      frame.w = -frame.w;
    }
    const f3 H = normalize(xform_with_onb(frame, local_h));
    f3 reflected = normalize(-dir_in + 2 * dot(dir_in, H) * H);
    if (dot(hit.ng, reflected) <= 0) return no_scatter();
    return Scatter{reflected, 0.f, true, true};
  }
  // visible-normal sampling shared by the metal and the glass lobe (spherical caps):
  // unqualified cos/sin there -> double products, narrowed once
  const f3 local_dir_in = project_onto_onb(frame, dir_in);
  f3 micro;
  {
    float sign = 1.f;
    f3 top = local_dir_in;
    if (local_dir_in.z < 0.f) {
      sign = -1.f;
      top = -top;
    }
    f3 hemi_dir_in = normalize(f3{alphax * top.x, alphay * top.y, top.z});
    float z = __builtin_fmaf((1.0f - rb), (1.0f + hemi_dir_in.z), -hemi_dir_in.z);
    float sin_theta = sqrt_f(clampf(1.0f - z * z, 0.0f, 1.0f));
    float x = sin_theta * cos_phi;
    float y = sin_theta * sin_phi;
    const f3 hemi_n = f3{x, y, z} + hemi_dir_in;
    micro = sign * normalize(f3{alphax * hemi_n.x, alphay * hemi_n.y, sel_max(0.f, hemi_n.z)});
  }
  if (lobe == LOBE_METAL) {
    f3 half_vector = normalize(xform_with_onb(frame, micro));
    f3 reflected = normalize(-dir_in + 2 * dot(dir_in, half_vector) * half_vector);
    if (dot(reflected, hit.ng) <= 0) return no_scatter();
    return Scatter{reflected, 0.f, true, true};
  }
  // LOBE_GLASS
  f3 half_vec = xform_with_onb(frame, micro);
  float h_dot_in = dot(half_vec, dir_in);
  float F = fresnel_dielectric(h_dot_in, eta);
  float rnd = rand_float(rng);
  if (rnd <= F) {
    f3 reflected = normalize(-dir_in + 2 * dot(dir_in, half_vec) * half_vec);
    if (dot(reflected, hit.ng) * dot(dir_in, hit.ng) <= 0) return no_scatter();
    return Scatter{reflected, 0.f, true, true};
  }
  float h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (eta * eta);
  if (h_dot_out_sq <= 0) return no_scatter();
  if (h_dot_in < 0) half_vec = -half_vec;
  float h_dot_out = static_cast<float>(__builtin_sqrt(static_cast<double>(h_dot_out_sq)));
  f3 refracted = -dir_in / eta + (absf(h_dot_in) / eta - h_dot_out) * half_vec;
  if (dot(refracted, hit.ng) * dot(dir_in, hit.ng) >= 0) return no_scatter();
  f3 generalized_h = normalize(dir_in + refracted * eta);
  float g_h_dot_in = dot(generalized_h, dir_in);
  if ((1 - (1 - g_h_dot_in * g_h_dot_in) / (eta * eta)) <= 0) return no_scatter();
  return Scatter{refracted, eta, true, true};
}
// Material::eval_pdf_pair dispatch; the base class returns (0, 1) (material.h:56-60), which is
// what Dielectric and DiffuseLight inherit
template <bool TEX, int MT = -1>
VD void eval_pdf_pair(const DScene& g, const Hit& hit, f3 wi, f3 wo, RayCone cone, bool regularize,
                      f3& f, float& pdf) {
  gptr<VimgMaterial> m = g.materials + hit.mat;
  const uint32_t type = MT >= 0 ? uint32_t(MT) : m->type;
  if (type == VIMG_MAT_LAMBERTIAN) {
    // Lambertian::eval_pdf_pair, reference src/material/lambertian.cpp:47-54
    float dot_product = static_cast<float>(sel_max(0.0f, dot(wo, hit.ns)) / kPi);
    f = col_at_ray_hit<TEX>(g, m->tex, wi, cone, hit) * dot_product;
    pdf = dot_product;
  } else if (type == VIMG_MAT_PRINCIPLED) {
    principled_eval_pdf<TEX>(g, m, wi, wo, hit, cone, regularize, f, pdf);
  } else {
    f = f3{0.f, 0.f, 0.f};
    pdf = 1.0f;
  }
}

// ================================================================================ emitters
// Triangle::sample: reference src/geometry/triangle.cpp:178-233
// (the sampling itself, given the triangle: corners, vertex normals, area pdf)
VD void tri_light_sample_with(f3 p0, f3 p1, f3 p2, f3 n0, f3 n1, f3 n2, float pdf, f3 look_from, Rng& rng, f3& hit_n,
                              EmitterInfo& info) {
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
  const f3 hit_p = p0 * u + p1 * v + p2 * w;
  hit_n = normalize(u * n0 + v * n1 + w * n2);
  f3 dir_vec = hit_p - look_from;
  float dist2 = length2(dir_vec);
  dir_vec = normalize(dir_vec);
  float cosine = absf(dot(hit_n, -dir_vec));
  float G = cosine / dist2;
  info = EmitterInfo{dir_vec, pdf, sqrt_f(dist2), G};
}
VD void tri_light_sample_at(f3 p0, f3 p1, f3 p2, f3 tri_normal, float pdf, f3 look_from, Rng& rng, f3& hit_n, EmitterInfo& info) {
  tri_light_sample_with(p0, p1, p2, tri_normal, tri_normal, tri_normal, pdf, look_from, rng, hit_n, info);
}
VD void tri_light_sample(const DScene& g, uint32_t tri, f3 look_from, Rng& rng, f3& le,
                         EmitterInfo& info) {
  gptr<DTriShade> ts = g.tri_shade + tri;
  const f3 p0{ts->p[0], ts->p[1], ts->p[2]}, p1{ts->p[3], ts->p[4], ts->p[5]},
      p2{ts->p[6], ts->p[7], ts->p[8]};
  gptr<VimgMesh> mesh = g.meshes + ts->mesh;
  f3 tri_normal{ts->n[0], ts->n[1], ts->n[2]};   // normalize(cross(edge1, edge2)), baked
  f3 n0 = tri_normal, n1 = tri_normal, n2 = tri_normal;
  if (mesh->has_normals) {
    n0 = load3(g.normals + 3 * size_t(ts->i0));
    n1 = load3(g.normals + 3 * size_t(ts->i1));
    n2 = load3(g.normals + 3 * size_t(ts->i2));
  }
  f3 hit_n;
  tri_light_sample_with(p0, p1, p2, n0, n1, n2, g.tri_area_pdf[tri] /* 1.f / (length(cross(edge2, edge1)) / 2.0f), baked */, look_from,
                        rng, hit_n, info);
  le = mat_emitted(g.materials + mesh->material, info.wi, hit_n);
}
// Triangle::surf_pdf: reference src/geometry/triangle.cpp:235-248
VD float tri_surf_pdf(const DScene& g, uint32_t tri) { return g.tri_area_pdf[tri]; }
// Sphere::sample: reference src/geometry/sphere.cpp:58-118 (cone construction of quirk Q16 kept)
VD void sphere_light_sample_at(f3 center, float radius, f3 look_from, Rng& rng, f3& shading_normal, EmitterInfo& info) {
  float rand1 = rand_float(rng);
  float rand2 = rand_float(rng);
  if (length2(look_from - center) <= radius * radius) {
    f3 unit = sample_sphere(rand1, rand2);
    f3 point_on_sphere = (unit * radius) + center;
    f3 to_pos = point_on_sphere - look_from;
    shading_normal = unit;
    const float sphere_sa = 4.f * kPi * radius * radius;
    f3 dir_to_surf = normalize(to_pos);
    float dist2 = length2(to_pos);
    float cosine = absf(dot(shading_normal, -dir_to_surf));
    float G = cosine / dist2;
    float pdf = 1.f / sphere_sa;
    info = EmitterInfo{dir_to_surf, pdf, sqrt_f(dist2), G};
  } else {
    float cos_theta_max = static_cast<float>(__builtin_sqrt(
        static_cast<double>(1.0f - ((radius * radius) / length2(look_from - center)))));
    f3 dir_center_to_lf = normalize(look_from - center);
    Onb onb = init_onb(dir_center_to_lf);
    f3 sample_z_dir = sample_sphere_cap(rand1, rand2, cos_theta_max);
    f3 sampled_point = normalize(xform_with_onb(onb, sample_z_dir)) * radius + center;
    float dist2 = length2(sampled_point - look_from);
    shading_normal = normalize(sampled_point - center);
    f3 sampled_dir = normalize(sampled_point - look_from);
    float cosine = absf(dot(shading_normal, -sampled_dir));
    float G = cosine / dist2;
    float pdf_solid_angle = 1.0f / (2.f * kPi * (1.0f - cos_theta_max));
    float pdf = pdf_solid_angle * G;
    info = EmitterInfo{sampled_dir, pdf, sqrt_f(dist2), G};
  }
}
// Sphere::surf_pdf: reference src/geometry/sphere.cpp:120-139
VD float sphere_surf_pdf(gptr<VimgSphere> sp, f3 look_from, f3 point_on_light, f3 dir) {
  const f3 center = load3(sp->center);
  const float radius = sp->radius;
  if (length2(look_from - center) <= radius * radius) {
    const float sphere_sa = 4.f * kPi * radius * radius;
    return 1.f / sphere_sa;
  }
  float cos_theta_max = static_cast<float>(__builtin_sqrt(
      static_cast<double>(1.0f - ((radius * radius) / length2(look_from - center)))));
  float pdf_solid_angle = 1.0f / (2.f * kPi * (1.0f - cos_theta_max));
  f3 shading_normal = normalize(point_on_light - center);
  float cosine = absf(dot(shading_normal, -dir));
  float dist2 = length2(point_on_light - look_from);
  return pdf_solid_angle * cosine / dist2;
}

// ---- Background: reference include/background.h:25-179
VD void env_dir_to_uv(const DScene& g, f3 in_dir, float& u, float& v) {
  f3 dir = normalize(mat_dir(g.background.world_to_env, in_dir));
  u = (1.f + F_atan2(-dir.x, dir.z) * kInvPi) * 0.5f;
  v = F_acos(dir.y) * kInvPi;
}
template <bool TEX>
VD f3 background_emit(const DScene& g, f3 in_dir, RayCone cone) {
  if constexpr (TEX) {
    if (g.background.type == VIMG_BG_ENVMAP) {
      gptr<VimgTexture> img = g.textures + g.background.env_tex;
      float u, v;
      env_dir_to_uv(g, in_dir, u, v);
      float lambda = ::log2(absf(cone.spread_angle) * (img->height / kPi));
      lambda = is_nan(lambda) ? 0.f : lambda;
      return col_mipmap_interpolate(g, img, lambda - 2.f, f2{u, v}) * g.background.radiance_scale;
    }
  }
  return load3k(g.background.col);
}
template <bool TEX>
VD float background_pdf(const DScene& g, f3 in_dir) {
  if constexpr (TEX) {
    if (g.background.type == VIMG_BG_ENVMAP) {
      gptr<VimgTexture> img = g.textures + g.background.env_tex;
      const uint32_t W = img->width, H = img->height;
      float u, v;
      env_dir_to_uv(g, in_dir, u, v);
      int pixel_u = u * W;
      int pixel_v = v * H;
      int column_index = clampi(pixel_u, 0, static_cast<int>(W) - 1);
      int row_index = clampi(pixel_v, 0, static_cast<int>(H) - 1);
      gptr<float> row_cdf = g.cdf_pool + g.background.row_cdf_offset;
      gptr<float> col_cdf = g.cdf_pool + g.background.col_cdf_offset + size_t(row_index) * (W + 1);
      float pdf_y = row_cdf[row_index + 1] - row_cdf[row_index];
      float pdf_x = col_cdf[column_index + 1] - col_cdf[column_index];
      float sin_elevation = ::sin(kPi * v);
      return (pdf_y * pdf_x * W * H) / (2.f * kPi * kPi * sin_elevation);
    }
  }
  return 1.f / (4 * kPi);
}
// ArraySampling1D::sample: reference include/rng/sampling.h:144-155 — std::upper_bound (first
// element > u), then -1
VD void cdf_sample(gptr<float> cdf, uint32_t n_plus_1, float u, uint32_t& index, float& du) {
  uint32_t lo = 0, len = n_plus_1;
  while (len > 0) {
    uint32_t half = len >> 1;
    uint32_t mid = lo + half;
    if (!(u < cdf[mid])) {
      lo = mid + 1;
      len = len - half - 1;
    } else {
      len = half;
    }
  }
  index = lo - 1;
  du = u - cdf[index];
  if (cdf[index + 1] - cdf[index] > 0) du /= cdf[index + 1] - cdf[index];
}
template <bool TEX>
VD void background_sample(const DScene& g, Rng& rng, f3& le, EmitterInfo& info) {
  float r1 = rand_float(rng);
  float r2 = rand_float(rng);
  if constexpr (TEX) {
    if (g.background.type == VIMG_BG_ENVMAP) {
      gptr<VimgTexture> img = g.textures + g.background.env_tex;
      const uint32_t W = img->width, H = img->height;
      gptr<float> row_cdf = g.cdf_pool + g.background.row_cdf_offset;
      uint32_t row_index, column_index;
      float dv, du;
      cdf_sample(row_cdf, H + 1, r1, row_index, dv);
      gptr<float> col_cdf = g.cdf_pool + g.background.col_cdf_offset + size_t(row_index) * (W + 1);
      cdf_sample(col_cdf, W + 1, r2, column_index, du);
      const float u_env = (static_cast<float>(column_index) + du) / W;
      const float v_env = (static_cast<float>(row_index) + dv) / H;
      float pdf_y = row_cdf[row_index + 1] - row_cdf[row_index];
      float pdf_x = col_cdf[column_index + 1] - col_cdf[column_index];
      float choose_sample_pdf = pdf_y * pdf_x;
      float elevation = v_env * kPi;
      float y = ::cos(v_env * kPi);
      const float azimuth = u_env * 2.f * kPi;
      float x = F_sin(azimuth) * F_sin(elevation);
      float z = -1 * F_cos(azimuth) * F_sin(elevation);
      f3 wi = normalize(mat_dir(g.background.env_to_world, f3{x, y, z}));
      float sin_elevation = F_sin(elevation);
      float pdf = (choose_sample_pdf * W * H) / (2.f * kPi * kPi * sin_elevation);
      le = col_at_uv_mipmap(g, img, 0, f2{u_env, v_env}) * g.background.radiance_scale;
      info = EmitterInfo{wi, pdf, VIMG_INF, 1.f};
      return;
    }
  }
  f3 wi = sample_sphere(r1, r2);
  constexpr float pdf = 1.f / (4 * kPi);
  le = load3k(g.background.col);
  info = EmitterInfo{wi, pdf, VIMG_INF, 1.f};
}
// GroupOfEmitters::sample: reference include/geometry/emitters.h:39-56
template <bool TEX>
VD void lights_sample(const DScene& g, f3 look_from, Rng& rng, f3& le, EmitterInfo& info) {
  float rnd = rand_float(rng);
  float sx = rnd * g.num_lights;
  const int index_obj = clampi(static_cast<int>(sx), 0, static_cast<int>(g.num_lights) - 1);
  const float prob_obj = 1.f / g.num_lights;
  // the emitter's baked record (device_scene.h: DLight): one fetch, then arithmetic
  gptr<DLight> L = g.dlights + index_obj;
  const uint32_t kind = L->kind;
  if (kind == 0u) {
    background_sample<TEX>(g, rng, le, info);
  } else if (kind == 2u) {
    tri_light_sample(g, L->index, look_from, rng, le, info);   // (vertex normals: through the tables)
  } else {
    f3 shading_normal;
    const v4f la = L->a;
    if (kind == 1u) {
      const v4f lb = L->b, lc = L->c;
      tri_light_sample_at(f3{la.x, la.y, la.z}, f3{la.w, lb.x, lb.y}, f3{lb.z, lb.w, lc.x}, f3{lc.y, lc.z, lc.w}, L->d.w, look_from,
                          rng, shading_normal, info);
    } else {
      sphere_light_sample_at(f3{la.x, la.y, la.z}, la.w, look_from, rng, shading_normal, info);
    }
    // Material::emitted of the emitter's material (DiffuseLight::emitted: one-sided; zero for the others)
    const v4f ld = L->d;
    le = (dot(shading_normal, info.wi) < 0) ? f3{ld.x, ld.y, ld.z} : f3{0.f, 0.f, 0.f};
  }
  info.pdf *= prob_obj;
}
VD float surf_pdf(const DScene& g, uint32_t prim_id, f3 look_from, f3 look_at, f3 dir) {
  const VimgPrim p = load_prim(g.prims + prim_id);
  if (p.type == VIMG_PRIM_TRIANGLE) return tri_surf_pdf(g, p.index);
  return sphere_surf_pdf(g.spheres + p.index, look_from, look_at, dir);
}

VD float balance_heuristic(float pdf1, float pdf2) { return pdf1 / (pdf1 + pdf2); }
// geometric_term: reference src/integrators/mis_integrator.cpp:7-16
VD float geometric_term(f3 look_from, f3 point_on_surface, f3 surface_normal) {
  f3 d = look_from - point_on_surface;
  float distance2 = length2(d);
  d = normalize(d);
  return absf(dot(surface_normal, d)) / distance2;
}

// ================================================================================ LDS set-up
VD uint32_t lds_node_bytes(uint32_t n) { return n * (3 * 16 + 8); }
// where the node planes and this lane's stack live (pointer arithmetic only)
VD Lds lds_layout(const RenderArgs& A, VIMG_LDS unsigned char* lds_raw) {
  const uint32_t n = A.lds_nodes;
  VIMG_LDS v4f* na = reinterpret_cast<VIMG_LDS v4f*>(lds_raw);
  VIMG_LDS v4f* nb = na + n;
  VIMG_LDS v4f* nc = nb + n;
  VIMG_LDS v2u* nm = reinterpret_cast<VIMG_LDS v2u*>(nc + n);
  // stacks start at the next 256-byte boundary: [wave][entry][lane]
  uint32_t stack_off = (lds_node_bytes(n) + 255u) & ~255u;
  VIMG_LDS uint32_t* stacks = reinterpret_cast<VIMG_LDS uint32_t*>(lds_raw + stack_off);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  Lds L;
  L.na = na, L.nb = nb, L.nc = nc, L.nm = nm;
  L.stack = stacks + size_t(wave) * A.stack_entries * 64 + lane;
  L.n_nodes = n;
  return L;
}
VD Lds stage_lds(const DScene& g, const RenderArgs& A, VIMG_LDS unsigned char* lds_raw) {
  const uint32_t n = A.lds_nodes;
  VIMG_LDS v4f* na = reinterpret_cast<VIMG_LDS v4f*>(lds_raw);
  VIMG_LDS v4f* nb = na + n;
  VIMG_LDS v4f* nc = nb + n;
  VIMG_LDS v2u* nm = reinterpret_cast<VIMG_LDS v2u*>(nc + n);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    gptr<DNode> src = g.nodes + i;
    na[i] = src->a;
    nb[i] = src->b;
    nc[i] = src->c;
    nm[i] = v2u{src->left_ref, src->right_ref};
  }
  __syncthreads();
  return lds_layout(A, lds_raw);
}

// ================================================================================ render kernel
// scene_integrator + mis_integrator / normal integrators:
// reference include/integrators.h:36-153, src/integrators/mis_integrator.cpp:18-189,
// src/integrators/normals.cpp:4-46
template <bool TEX, int WPS>
__global__ void __launch_bounds__(256, WPS)
render_kernel(const DScene g, const RenderArgs A, float* __restrict__ out,
              DeviceStats* __restrict__ stats, unsigned int* __restrict__ work_counter) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Lds L = stage_lds(g, A, (VIMG_LDS unsigned char*)lds_raw);
  const uint32_t lane = threadIdx.x & 63;
  const bool full_stats = A.full_stats != 0;
  const uint32_t W = static_cast<uint32_t>(g.res_x), H = static_cast<uint32_t>(g.res_y);
  const bool single = A.single_x >= 0;
  const uint32_t total_items = single ? 1u : A.num_local_tiles * 64u;
  constexpr uint32_t roulette_threshold = 5;

  Counters cnt{0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t nan_samples = 0, iter_wave = 0;

  // per-pixel state
  bool alive = true, need_pixel = true;
  uint32_t item = 0, px = 0, py = 0, smp = 0;
  Rng rng{0};
  f3 acc{0.f, 0.f, 0.f};
  // per-path state
  bool new_path = true, primary = true, non_specular_bounce = false;
  f3 ray_o{0.f, 0.f, 0.f}, ray_d{0.f, 0.f, 1.f};
  RayCone cone{0.f, 0.f};
  f3 throughput{1.f, 1.f, 1.f}, result{0.f, 0.f, 0.f};
  float eta_scale = 1.f, prev_pdf = 0.f;
  uint32_t bounce = 0;

  for (;;) {
    // ------------------------------------------------------------------ pixel fetch
    {
      const bool want = alive && need_pixel;
      const unsigned long long mask = __ballot(want);
      if (mask != 0ull) {
        const uint32_t n = __popcll(mask);
        uint32_t base = 0;
        const uint32_t leader = __ffsll(static_cast<long long>(mask)) - 1;
        if (lane == leader) base = atomicAdd(work_counter, n);
        base = __shfl(base, leader);
        if (want) {
          const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
          item = base + rank;
          if (item >= total_items) {
            alive = false;
          } else {
            bool valid = true;
            if (single) {
              px = static_cast<uint32_t>(A.single_x), py = static_cast<uint32_t>(A.single_y);
            } else {
              // 8x8 tiles in the reference's x-major work_list order (integrators.h:57-65);
              // shard r of n owns tiles t with t % n == r
              const uint32_t tile = (item >> 6) * A.tile_world + A.tile_rank;
              const uint32_t within = item & 63u;
              const uint32_t tx = tile / A.tiles_y, ty = tile - tx * A.tiles_y;
              px = tx * 8 + (within & 7u);
              py = ty * 8 + (within >> 3);
              valid = (tx < A.tiles_x) && (px < W) && (py < H);
            }
            if (valid) {
              const uint64_t image_index = uint64_t(px) + uint64_t(H - 1 - py) * W;
              pcg_seed(rng, image_index);
              smp = 0;
              acc = f3{0.f, 0.f, 0.f};
              new_path = true;
              need_pixel = false;
            }
          }
        }
      }
    }
    if (!__any(alive)) break;
    const bool active = alive && !need_pixel;
    if (full_stats && lane == 0) iter_wave++;

    // ------------------------------------------------------------------ camera ray
    if (active && new_path) {
      const f2 off = random_x_y_r2(px + py + smp);
      // the reference's call evaluates its two rand_float arguments right to left (g++):
      // rand2 receives the first draw
      const float rand2 = rand_float(rng);
      const float rand1 = rand_float(rng);
      generate_ray(g, static_cast<float>(px) + off.x, static_cast<float>(py) + off.y, rand1, rand2,
                   ray_o, ray_d);
      cone = RayCone{0.f, g.cone_spread};
      throughput = f3{1.f, 1.f, 1.f};
      result = f3{0.f, 0.f, 0.f};
      eta_scale = 1.f;
      non_specular_bounce = false;
      primary = true;
      bounce = 0;
      new_path = false;
    }

    // ------------------------------------------------------------------ closest hit
    TravRay tr{ray_o, ray_d, 0.0001f, VIMG_INF};
    HitRec rec;
    rec.prim = 0xffffffffu;
    bool hit_any = false;
    if (active) {
      cnt.closest++;
      hit_any = traverse<false>(g, L, tr, rec, cnt, full_stats);
    }
    Hit hit;
    hit.p = f3{0.f, 0.f, 0.f};
    if (hit_any) make_hit_info<TEX>(g, rec, tr, hit);

    bool finish = false;     // this path's radiance is final
    bool at_vertex = false;  // continue with next-event estimation + BSDF sampling at `hit`

    const bool material_mode = (A.integrator == VIMG_INTEGRATOR_MATERIAL);   // wave-uniform
    if (active) {
      if (material_mode) {
        // material_integrator (mat_integrator.cpp:16-23,79-81): a miss ends the path with the
        // background; every hit, emissive or not, is a vertex
        if (!hit_any) {
          result = throughput * background_emit<TEX>(g, ray_d, cone);
          finish = true;
        } else {
          at_vertex = true;
        }
      } else if (A.integrator != VIMG_INTEGRATOR_MIS) {
        // shading_normal_integrator / geometric_normal_integrator
        if (hit_any) {
          f3 n = (A.integrator == VIMG_INTEGRATOR_G_NORMAL) ? hit.ng : hit.ns;
          result = (n + 1.0f) / 2.0f;
        } else {
          f3 unit_dir = normalize(ray_d);
          float a = 0.5 * (unit_dir.y + 1.0);
          result = (1.0f - a) * f3{1.0f, 1.0f, 1.0f} + a * f3{0.5f, 0.7f, 1.0f};
        }
        finish = true;
      } else if (primary) {
        if (!hit_any) {
          result = background_emit<TEX>(g, ray_d, cone);
          finish = true;
        } else {
          gptr<VimgMaterial> m = g.materials + hit.mat;
          if (m->type == VIMG_MAT_DIFFUSE_LIGHT) {
            result = mat_emitted(m, ray_d, hit.ns);
            finish = true;
          } else {
            bounce = 0;
            at_vertex = true;
          }
        }
      } else {
        // the ray sampled from the BSDF at the previous vertex (mis_integrator.cpp:120-186)
        if (hit_any) {
          gptr<VimgMaterial> m = g.materials + hit.mat;
          if (m->type == VIMG_MAT_DIFFUSE_LIGHT) {
            const f3 le = mat_emitted(m, ray_d, hit.ns);
            if (prev_pdf != 0) {
              float light_pdf = surf_pdf(g, hit.prim, ray_o, hit.p, ray_d) / g.num_lights;
              float G = geometric_term(ray_o, hit.p, hit.ng);
              float mis_weight = balance_heuristic(prev_pdf * G, light_pdf);
              result = result + throughput * mis_weight * le;
            } else {
              result = result + throughput * le;
            }
            finish = true;
          } else {
            bool survive = true;
            if (bounce > roulette_threshold) {
              float rr = static_cast<float>(pcg_next(rng)) / 4294967296.0f;   // float(UINT32_MAX)
              f3 rr_t = (1.f / eta_scale) * throughput;
              float max_val = sel_min(sel_max(sel_max(rr_t.x, rr_t.y), rr_t.z), 0.95f);
              if (rr > max_val)
                survive = false;
              else
                throughput = throughput / max_val;
            }
            if (survive) {
              bounce += 1;
              at_vertex = true;
            } else {
              finish = true;
            }
          }
        } else {
          if (prev_pdf != 0 && g.background_emissive) {
            float light_pdf = background_pdf<TEX>(g, ray_d) / g.num_lights;
            float mis_weight = balance_heuristic(prev_pdf, light_pdf);
            result = result + throughput * mis_weight * background_emit<TEX>(g, ray_d, cone);
          }
          finish = true;
        }
      }
      if (at_vertex && !(bounce < A.depth)) {   // for (d = 0; d < depth; d++)
        at_vertex = false;
        finish = true;
      }
    }

    // ------------------------------------------------------------------ vertex, material_integrator
    // mat_integrator.cpp:24-78: BSDF sampling only, throughput *= emitted + eval/pdf
    if (material_mode && at_vertex) {
      gptr<VimgMaterial> m = g.materials + hit.mat;
      const f3 emitted_col = mat_emitted(m, ray_d, hit.ns);
      Scatter sc = sample_mat<TEX>(g, hit, ray_d, rng, non_specular_bounce);
      if (!sc.valid) {
        result = throughput * emitted_col;
        finish = true;
      } else {
        if (!sc.is_specular) non_specular_bounce = true;
        if constexpr (TEX) {
          const float hd = length(ray_o - hit.p);
          const float ssa = spread_angle_from_curvature(hit.curvature, cone.cone_width, ray_d, hit.ns);
          if (sc.eta != 0.f)
            cone = propagate_refract_cone(cone, ray_d, ssa, sc.eta, sc.wo);
          else
            cone = propagate_reflect_cone(cone, ssa * 2.f, hd);
        }
        if (sc.eta != 0.f) eta_scale /= (sc.eta * sc.eta);
        // Material::eval_div_pdf (material.h:51-54): Lambertian -> texture colour, Dielectric -> 1,
        // Principled -> eval / pdf of the same eval_pdf template, base -> 0
        f3 fdiv{0.f, 0.f, 0.f};
        const uint32_t type = m->type;
        if (type == VIMG_MAT_LAMBERTIAN) {
          fdiv = col_at_ray_hit<TEX>(g, m->tex, ray_d, cone, hit);
        } else if (type == VIMG_MAT_DIELECTRIC) {
          fdiv = splat3(1.f);
        } else if (type == VIMG_MAT_PRINCIPLED) {
          f3 f;
          float pdf;
          principled_eval_pdf<TEX>(g, m, ray_d, sc.wo, hit, cone, non_specular_bounce, f, pdf);
          fdiv = f / pdf;
        }
        throughput = throughput * (emitted_col + fdiv);
        bool survive = true;
        if (bounce > roulette_threshold) {
          float rr = static_cast<float>(pcg_next(rng)) / 4294967296.0f;
          f3 rr_t = (1.f / eta_scale) * throughput;
          float max_val = sel_min(sel_max(sel_max(rr_t.x, rr_t.y), rr_t.z), 0.95f);
          if (rr > max_val)
            survive = false;
          else
            throughput = throughput / max_val;
        }
        bounce += 1;
        if (!survive || !(bounce < A.depth)) {
          result = f3{0.f, 0.f, 0.f};   // roulette break / depth limit: return vec3(0)
          finish = true;
        } else {
          ray_o = hit.p;
          ray_d = sc.wo;
          primary = false;
        }
      }
      at_vertex = false;
    }

    // ------------------------------------------------------------------ vertex: NEE + BSDF
    // mis_integrator.cpp:45-122.  Draw order: light pick + emitter sample, then sample_mat.
    float hit_dist = 0.f, surface_spread_angle = 0.f;
    bool nee = false;
    f3 light_col{0.f, 0.f, 0.f};
    EmitterInfo li{f3{0.f, 0.f, 1.f}, 0.f, 0.f, 0.f};
    uint32_t mat_type = 0xffffffffu;
    if (at_vertex) {
      mat_type = g.materials[hit.mat].type;
      if constexpr (TEX) {
        hit_dist = length(ray_o - hit.p);
        surface_spread_angle =
            spread_angle_from_curvature(hit.curvature, cone.cone_width, ray_d, hit.ns);
      }
      const bool is_delta = (mat_type == VIMG_MAT_DIELECTRIC);
      if (!is_delta) {
        lights_sample<TEX>(g, hit.p, rng, light_col, li);
        nee = (li.pdf != 0.f);
      }
    }
    bool occluded = false;
    {
      TravRay sr{hit.p, li.wi, 0.0001f, li.dist - 0.0001f};   // absolute epsilon (quirk Q15)
      HitRec srec;
      if (nee) {
        cnt.shadow++;
        occluded = traverse<true>(g, L, sr, srec, cnt, full_stats);
      }
    }
    Scatter sc = no_scatter();
    const bool reg_before = non_specular_bounce;
    RayCone nee_cone = cone;
    if (at_vertex) {
      sc = sample_mat<TEX>(g, hit, ray_d, rng, reg_before);
      if constexpr (TEX) {
        nee_cone = propagate_reflect_cone(cone, surface_spread_angle * 2.f, hit_dist);
      }
      if (sc.valid) {
        if (!sc.is_specular) non_specular_bounce = true;
        if (sc.eta != 0.f) {
          eta_scale /= (sc.eta * sc.eta);
          if constexpr (TEX) {
            cone = propagate_refract_cone(cone, ray_d, surface_spread_angle, sc.eta, sc.wo);
          }
        } else {
          if constexpr (TEX) cone = nee_cone;
        }
      }
    }
    // the two BSDF evaluations of a vertex (light direction, then sampled direction) share one
    // copy of the evaluation code; the regularisation flag of the first is the one from BEFORE
    // this bounce, of the second the updated one
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
      const bool run = at_vertex && (k == 0 ? (nee && !occluded) : sc.valid);
      if (__any(run)) {
        f3 f{0.f, 0.f, 0.f};
        float pdf = 0.f;
        if (run) {
          const f3 wo = (k == 0) ? li.wi : sc.wo;
          const RayCone c = (k == 0) ? nee_cone : cone;
          const bool reg = (k == 0) ? reg_before : non_specular_bounce;
          eval_pdf_pair<TEX>(g, hit, ray_d, wo, c, reg, f, pdf);
          if (k == 0) {
            if (pdf != 0 && !is_nan(pdf)) {
              float G = li.G;
              float mis_weight = balance_heuristic(li.pdf, pdf * G);
              result = result + throughput * f * mis_weight * G * light_col / li.pdf;
            }
          } else {
            if (is_nan(pdf)) {
              sc.valid = false;   // NaN pdf terminates the path (mis_integrator.cpp:108-114)
            } else {
              throughput = throughput * (f / pdf);
              prev_pdf = pdf;
            }
          }
        }
      }
    }
    if (at_vertex) {
      if (sc.valid) {
        ray_o = hit.p;
        ray_d = sc.wo;
        primary = false;
      } else {
        finish = true;
      }
    }

    // ------------------------------------------------------------------ sample / pixel done
    if (finish) {
      if (is_nan(result.x) || is_nan(result.y) || is_nan(result.z)) nan_samples++;
      acc = acc + result;
      smp += 1;
      new_path = true;
      if (smp == A.samples) {
        const f3 px_col = acc / static_cast<float>(A.samples);
        size_t o;
        if (single)
          o = 0;
        else if (A.tile_world == 1)
          o = (size_t(px) + size_t(H - 1 - py) * W) * 3;
        else
          o = size_t(item) * 3;
        out[o + 0] = px_col.x;
        out[o + 1] = px_col.y;
        out[o + 2] = px_col.z;
        need_pixel = true;
      }
    }
  }

  // ---- flush event counts: one atomic per wave and counter
  if (stats) {
    auto wave_sum = [&](uint32_t v) {
      unsigned long long s = v;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      return s;
    };
    unsigned long long c0 = wave_sum(cnt.closest), c1 = wave_sum(cnt.shadow),
                       c2 = wave_sum(cnt.internal), c3 = wave_sum(cnt.leaf),
                       c4 = wave_sum(cnt.prim), c5 = wave_sum(nan_samples),
                       c6 = wave_sum(cnt.sphere), c7 = wave_sum(cnt.trip_descend),
                       c8 = wave_sum(cnt.trip_prim), c9 = wave_sum(iter_wave);
    if (lane == 0) {
      atomicAdd(&stats->closest, c0);
      atomicAdd(&stats->shadow, c1);
      if (full_stats) {
        atomicAdd(&stats->internal, c2);
        atomicAdd(&stats->leaf, c3);
        atomicAdd(&stats->prim, c4);
        atomicAdd(&stats->sphere, c6);
        atomicAdd(&stats->trip_descend, c7);
        atomicAdd(&stats->trip_prim, c8);
        atomicAdd(&stats->iterations, c9);
      }
      if (c5) atomicAdd(&stats->nan_samples, c5);
    }
  }
}

}  // namespace vimg
