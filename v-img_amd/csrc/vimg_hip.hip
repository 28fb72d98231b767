// libvimg_hip.so — C ABI (include/vimg_hip.h) over the gfx950 kernels in render_kernels.h.
// Host-side work here is limited to validating and baking the scene tables into the device
// layout (once per scene) and launching kernels; there is no CPU render path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numbers>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/vimg_hip.h"
#include "post_kernels.h"
#include "pre_kernels.h"
#include "aux_kernels.h"
#include "kernel_tus.h"
#include "render_pool_kernel.h"
#include "render_stage_kernel.h"
#include "render_pool4_kernel.h"
#include "render_cu_kernel.h"
#include "heatmap_kernel.h"

using namespace vimg;

namespace {

thread_local std::string g_err;
hipStream_t g_stream = nullptr;
int g_device = -1;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(VIMG_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

struct VimgDeviceScene {
  DScene d{};
  std::vector<void*> allocs;
  size_t total_bytes = 0;
  bool textured = false;       // needs the TEX=true kernels (cones, image textures, env map)
  VimgHipOptions opt{};        // the caller's options (VIMG_OPT_AUTO where the policy decides)
  int waves_per_simd = 2;      // LANE register budget by policy (scene size)
  bool too_wide = false;       // resolution beyond the 16-bit pixel coordinates of the slot records
  uint32_t num_cus = 0;
  uint32_t num_leaf_prims = 0;   // records in d.leaf_prims (= primitives of the scene)
  // scratch owned by the scene: stats, work counter, host-render framebuffer
  DeviceStats* d_stats = nullptr;
  unsigned int* d_counter = nullptr;
  float* d_frame = nullptr;
  void* d_pool_cold = nullptr;   // pooled kernel: cold slot records of every resident wave
  size_t pool_cold_bytes = 0;
  void* d_stack_ovf = nullptr;   // pool4, deep trees: the stack entries beyond the LDS part, per resident wave
  size_t stack_ovf_bytes = 0;
  void* d_pool_state = nullptr;  // pooled kernel: per-pixel record between sample segments
  size_t pool_state_bytes = 0;
  uint32_t pool_epoch = 0;       // bumped per launch: tags of earlier launches never match
  size_t frame_floats = 0;
  // staged kernel: control block, queue rings, ready-pixel ring, per-pixel records, slot records
  void* d_stage_ctl = nullptr;
  void* d_stage_kargs = nullptr;   // StageKArgs block of the launch in flight
  void* d_stage_rings = nullptr;
  size_t stage_rings_bytes = 0;
  void* d_stage_pix_ring = nullptr;
  size_t stage_pix_ring_bytes = 0;
  void* d_stage_pix_state = nullptr;
  size_t stage_pix_state_bytes = 0;
  void* d_stage_slots = nullptr;
  size_t stage_slots_bytes = 0;
};

namespace {

template <typename T>
int upload(VimgDeviceScene* s, const T* host, size_t count, const T** out) {
  *out = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  s->allocs.push_back(p);
  s->total_bytes += bytes;
  if (count) HIP_TRY(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(p);
  return VIMG_OK;
}

// Shape checks so that no kernel ever indexes outside its tables.
int validate(const VimgScene* sc) {
  if (!sc) return fail(VIMG_E_INVALID, "scene is null");
  if (sc->camera.res_x <= 0 || sc->camera.res_y <= 0) return fail(VIMG_E_INVALID, "bad resolution");
  if (sc->num_prims == 0 || !sc->prims) return fail(VIMG_E_INVALID, "scene has no primitives");
  const VimgBVH& b = sc->bvh;
  if (b.num_nodes == 0 || !b.nodes || !b.bb_mins_maxes || !b.obj_indices)
    return fail(VIMG_E_INVALID, "scene has no BVH");
  if (b.max_depth + 2 > 96) return fail(VIMG_E_INVALID, "BVH deeper than the 94-level stack bound");
  for (uint32_t i = 0; i < sc->num_prims; ++i) {
    const VimgPrim& p = sc->prims[i];
    if (p.type == VIMG_PRIM_TRIANGLE) {
      if (p.index >= sc->num_tris) return fail(VIMG_E_INVALID, "prim: triangle index out of range");
    } else if (p.type == VIMG_PRIM_SPHERE) {
      if (p.index >= sc->num_spheres) return fail(VIMG_E_INVALID, "prim: sphere index out of range");
    } else {
      return fail(VIMG_E_INVALID, "prim: unknown type");
    }
  }
  for (uint32_t t = 0; t < sc->num_tris; ++t) {
    if (sc->tri_mesh[t] >= sc->num_meshes) return fail(VIMG_E_INVALID, "tri: mesh out of range");
    const VimgMesh& m = sc->meshes[sc->tri_mesh[t]];
    for (int k = 0; k < 3; ++k)
      if (sc->tri_indices[t * 3 + k] >= m.num_vertices)
        return fail(VIMG_E_INVALID, "tri: vertex index out of range");
  }
  for (uint32_t i = 0; i < sc->num_meshes; ++i) {
    const VimgMesh& m = sc->meshes[i];
    if (uint64_t(m.first_vertex) + m.num_vertices > sc->num_vertices)
      return fail(VIMG_E_INVALID, "mesh: vertex range out of bounds");
    if (m.material >= sc->num_materials) return fail(VIMG_E_INVALID, "mesh: material out of range");
    if (m.num_uv_sets > VIMG_MAX_UV_SETS) return fail(VIMG_E_INVALID, "mesh: too many uv sets");
    for (uint32_t k = 0; k < m.num_uv_sets; ++k)
      if (uint64_t(m.uv_offset[k]) + m.num_vertices > sc->num_uvs)
        return fail(VIMG_E_INVALID, "mesh: uv set out of bounds");
    auto ok = [&](uint32_t u) { return u == VIMG_NO_UV || u < m.num_uv_sets; };
    if (!ok(m.color_tex_uv) || !ok(m.normal_tex_uv) || !ok(m.metallic_roughness_tex_uv))
      return fail(VIMG_E_INVALID, "mesh: uv selector out of range");
  }
  for (uint32_t i = 0; i < sc->num_spheres; ++i)
    if (sc->spheres[i].material >= sc->num_materials)
      return fail(VIMG_E_INVALID, "sphere: material out of range");
  for (uint32_t i = 0; i < sc->num_textures; ++i) {
    const VimgTexture& t = sc->textures[i];
    if (t.type > VIMG_TEX_IMAGE) return fail(VIMG_E_INVALID, "texture: unknown type");
    if (t.type == VIMG_TEX_IMAGE) {
      if (t.num_levels == 0 || t.num_levels > VIMG_MAX_MIP_LEVELS || t.width == 0 || t.height == 0)
        return fail(VIMG_E_INVALID, "texture: bad mip chain");
      for (uint32_t l = 0; l < t.num_levels; ++l) {
        uint64_t w = std::max(t.width >> l, 1u), h = std::max(t.height >> l, 1u);
        if (t.level_offset[l] + w * h > sc->num_texels)
          return fail(VIMG_E_INVALID, "texture: level out of bounds");
      }
    }
  }
  for (uint32_t i = 0; i < sc->num_rg_textures; ++i) {
    const VimgTextureRG& t = sc->rg_textures[i];
    if (t.width == 0 || t.height == 0 || t.wrap_u > 2 || t.wrap_v > 2)
      return fail(VIMG_E_INVALID, "rg texture: bad size or wrap mode");
    // The reference indexes the +x neighbours with "* height" instead of "* width" (quirk Q6,
    // include/texture/texture_RG.h:47,52).  For width >= height the largest such index,
    // (w-1) + (h-1) h, stays inside the w x h array: the WRONG texel is read, reproducibly, and the
    // kernels and the oracle reproduce it.  For height > width the reference reads beyond its
    // vector (undefined there): refused.
    if (t.height > t.width)
      return fail(VIMG_E_UNSUPPORTED,
                  "metallic-roughness map taller than wide: the reference reads outside the image there");
    if (t.offset + uint64_t(t.width) * t.height > sc->num_rg_texels)
      return fail(VIMG_E_INVALID, "rg texture out of bounds");
  }
  for (uint32_t i = 0; i < sc->num_materials; ++i) {
    const VimgMaterial& m = sc->materials[i];
    if (m.type > VIMG_MAT_PRINCIPLED) return fail(VIMG_E_INVALID, "material: unknown type");
    auto tex_ok = [&](int32_t t) { return t >= -1 && t < int32_t(sc->num_textures); };
    if (!tex_ok(m.tex) || !tex_ok(m.normal_map) || m.mr_tex < -1 ||
        m.mr_tex >= int32_t(sc->num_rg_textures))
      return fail(VIMG_E_INVALID, "material: texture index out of range");
    if ((m.type == VIMG_MAT_LAMBERTIAN || m.type == VIMG_MAT_PRINCIPLED) && m.tex < 0)
      return fail(VIMG_E_INVALID, "material: missing colour texture");
    if (m.normal_map >= 0 && sc->textures[m.normal_map].type != VIMG_TEX_IMAGE)
      return fail(VIMG_E_INVALID, "material: normal map must be an image");
  }
  for (uint32_t i = 0; i < sc->num_lights; ++i) {
    const VimgLight& l = sc->lights[i];
    if (l.type == VIMG_LIGHT_PRIM) {
      if (l.prim >= sc->num_prims) return fail(VIMG_E_INVALID, "light: prim out of range");
    } else if (l.type != VIMG_LIGHT_BACKGROUND) {
      return fail(VIMG_E_INVALID, "light: unknown type");
    }
  }
  if (sc->background.type == VIMG_BG_ENVMAP) {
    const int32_t t = sc->background.env_tex;
    if (t < 0 || t >= int32_t(sc->num_textures) || sc->textures[t].type != VIMG_TEX_IMAGE)
      return fail(VIMG_E_INVALID, "background: env_tex must be an image texture");
    const VimgTexture& img = sc->textures[t];
    if (sc->background.row_cdf_offset + img.height + 1 > sc->num_cdf ||
        sc->background.col_cdf_offset + uint64_t(img.height) * (img.width + 1) > sc->num_cdf)
      return fail(VIMG_E_INVALID, "background: cdf out of bounds");
  } else if (sc->background.type != VIMG_BG_CONST) {
    return fail(VIMG_E_INVALID, "background: unknown type");
  }
  // BVH: every node reachable from the root exactly once, children and leaf ranges in bounds
  std::vector<uint8_t> seen(b.num_nodes, 0);
  std::vector<uint32_t> todo{0};
  seen[0] = 1;
  while (!todo.empty()) {
    uint32_t n = todo.back();
    todo.pop_back();
    const VimgBVHNode& node = b.nodes[n];
    if (node.obj_count != 0) {
      if (uint64_t(node.first_index) + node.obj_count > sc->num_prims)
        return fail(VIMG_E_INVALID, "bvh: leaf range out of bounds");
      for (uint32_t i = 0; i < node.obj_count; ++i)
        if (b.obj_indices[node.first_index + i] >= sc->num_prims)
          return fail(VIMG_E_INVALID, "bvh: obj index out of range");
    } else {
      if (uint64_t(node.first_index) + 1 >= b.num_nodes || node.first_index == 0)
        return fail(VIMG_E_INVALID, "bvh: child index out of range");
      for (uint32_t c = node.first_index; c <= node.first_index + 1; ++c) {
        if (seen[c]) return fail(VIMG_E_INVALID, "bvh: node reachable twice (not a tree)");
        seen[c] = 1;
        todo.push_back(c);
      }
    }
  }
  return VIMG_OK;
}

// tools/ only: VIMG_HIP_* environment variables override single option fields at upload (sweeps
// and profiles without a rebuild of the caller); tests and the product pass VimgHipOptions
void options_from_env(VimgHipOptions* o) {
  if (const char* e = getenv("VIMG_HIP_SCHED")) {
    const std::string v(e);
    o->scheduler = v == "lane" ? VIMG_SCHED_LANE : v == "pool" ? VIMG_SCHED_POOL : v == "stage" ? VIMG_SCHED_STAGE
                 : v == "pool4" ? VIMG_SCHED_POOL4 : v == "pool4g" ? VIMG_SCHED_POOL4G : v == "cu" ? VIMG_SCHED_CU : atoi(e);
  }
  struct { const char* name; int32_t* field; } vars[] = {
      {"VIMG_HIP_WAVES_PER_SIMD", &o->waves_per_simd}, {"VIMG_HIP_LDS_BUDGET_KB", &o->lds_budget_kb},
      {"VIMG_HIP_POOL_SLOTS", &o->pool_slots},         {"VIMG_HIP_POOL_SEGMENTS", &o->pool_segments},
      {"VIMG_HIP_POOL_REFILL", &o->pool_refill},       {"VIMG_HIP_POOL_VBATCH", &o->pool_vbatch},
      {"VIMG_HIP_POOL_CLASSES", &o->pool_classes},     {"VIMG_HIP_POOL_STARVE", &o->pool_starve},
      {"VIMG_HIP_POOL_BOXMIN", &o->pool_boxmin},       {"VIMG_HIP_LDS_LEAF", &o->lds_leaf},
      {"VIMG_HIP_STAGE_SLOTS", &o->stage_slots},       {"VIMG_HIP_STAGE_SEG_LEN", &o->stage_seg_len},
      {"VIMG_HIP_STAGE_WCHUNK", &o->stage_wchunk},     {"VIMG_HIP_STAGE_WALK_QUOTA", &o->stage_walk_quota},
      {"VIMG_HIP_POOL4_RAYS", &o->pool4_rays},             {"VIMG_HIP_LDS_STACK", &o->lds_stack},
      {"VIMG_HIP_POOL_GBREAK", &o->pool_gbreak},       {"VIMG_HIP_CU_WAVES", &o->cu_waves},
      {"VIMG_HIP_CU_WALKERS", &o->cu_walkers},         {"VIMG_HIP_CU_FLEX", &o->cu_flex},
      {"VIMG_HIP_CU_LOWWATER", &o->cu_lowwater},       {"VIMG_HIP_CU_PATIENCE", &o->cu_patience},
      {"VIMG_HIP_CU_JOIN", &o->cu_join},               {"VIMG_HIP_CU_SLEEP", &o->cu_sleep}};
  for (auto& v : vars)
    if (const char* e = getenv(v.name)) *v.field = atoi(e);
}

uint32_t tiles_of(int n) { return (static_cast<uint32_t>(n) + 7u) / 8u; }

uint32_t local_tiles(const VimgDeviceScene* s, const VimgRenderParams* p) {
  const uint32_t total = tiles_of(s->d.res_x) * tiles_of(s->d.res_y);
  if (p->tile_rank >= total) return 0;
  return (total - p->tile_rank + p->tile_world - 1) / p->tile_world;
}

int check_params(const VimgDeviceScene* s, const VimgRenderParams* p) {
  if (!s || !p) return fail(VIMG_E_INVALID, "null scene or params");
  if (p->tile_world == 0 || p->tile_rank >= p->tile_world)
    return fail(VIMG_E_INVALID, "tile_rank must be < tile_world");
  if (p->samples == 0) return fail(VIMG_E_INVALID, "samples must be > 0");
  if (p->integrator == VIMG_INTEGRATOR_MATERIAL && p->depth == 0)
    return fail(VIMG_E_INVALID, "material integrator with depth 0 renders nothing");
  if (p->integrator > VIMG_INTEGRATOR_MIS) return fail(VIMG_E_INVALID, "unknown integrator");
  if (p->integrator == VIMG_INTEGRATOR_MIS && s->d.num_lights == 0)
    return fail(VIMG_E_INVALID, "mis integrator needs at least one light (the reference's "
                                "GroupOfEmitters::sample is undefined without one)");
  return VIMG_OK;
}

struct LaunchCfg {
  RenderArgs args;
  StageArgs stage;
  uint32_t grid, lds_bytes;
  int sched;     // VIMG_SCHED_* of this launch
  bool pooled;   // render_pool_kernel for this launch
  int wps;       // register-budget build (waves per SIMD of __launch_bounds__)
  int rays;      // pool4: rays a lane walks at the same time (1; two measured slower and are not built)
  bool group;    // pool4: one pool and one set of queues per workgroup (VIMG_SCHED_POOL4G) instead of per wave
  bool deep;     // pooled / staged kernel: build whose box loop yields to waiting leaves (tree beyond the LDS node cache)
  int cu_waves;  // CU scheduler: waves per workgroup (16 or 8)
};

RenderKernel pick_kernel(const VimgDeviceScene* s, bool pooled, int wps, bool deep) {
  return pooled ? vimg_pool_kernel(s->textured, wps, deep) : vimg_lane_kernel(s->textured, wps);
}
StageKernel pick_stage_kernel(const VimgDeviceScene* s, bool deep) { return vimg_stage_kernel(s->textured, deep); }
Pool4Kernel pick_pool4_kernel(const VimgDeviceScene* s, bool deep, int wps, bool group) {
  return vimg_pool4_kernel(s->textured, deep, wps, group);
}
CuKernel pick_cu_kernel(const VimgDeviceScene* s, bool deep, int nw, bool diag = false, bool early = false) {
  return diag ? vimg_cu_kernel_diag(s->textured, deep, nw)
              : (early ? vimg_cu_kernel_early(s->textured, deep, nw) : vimg_cu_kernel(s->textured, deep, nw));
}
const void* kernel_of(const VimgDeviceScene* s, const LaunchCfg& c) {
  if (c.sched == VIMG_SCHED_CU) return reinterpret_cast<const void*>(pick_cu_kernel(s, c.deep, c.cu_waves));
  if (c.sched == VIMG_SCHED_STAGE) return reinterpret_cast<const void*>(pick_stage_kernel(s, c.deep));
  if (c.sched == VIMG_SCHED_POOL4) return reinterpret_cast<const void*>(pick_pool4_kernel(s, c.deep, c.wps, c.group));
  return reinterpret_cast<const void*>(pick_kernel(s, c.pooled, c.wps, c.deep));
}

uint32_t opt_or(int32_t v, uint32_t dflt) { return v == VIMG_OPT_AUTO ? dflt : static_cast<uint32_t>(v); }
uint32_t ceil_pow2(uint64_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}
uint32_t log2_of(uint32_t pow2) {
  uint32_t k = 0;
  while ((1u << k) < pow2) ++k;
  return k;
}

// The policy of one launch.  `sched_override`: 0 = by options / policy, else the scheduler to build
// the configuration for (the fall-back from a scheduler that cannot take this launch).
// n / d == mulhi(n, magic) >> shift for every n < 2^31 (Granlund & Montgomery, "Division by invariant
// integers using multiplication", fig. 4.1 with N = 31): the ring index of render_cu_kernel's tickets
void magic_div(uint32_t d, uint32_t* magic, uint32_t* shift) {
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;   // ceil(log2 d), d >= 2
  *magic = static_cast<uint32_t>((1ull << (31u + l)) / d + 1ull);
  *shift = l - 1u;
}

// The launch of the CU-wide scheduler (render_cu_kernel.h): one workgroup of 16 waves per compute unit
// (or two of 8), a pool of as many slots as the CU's LDS holds behind the top of the tree, the
// walking waves' stacks and the rings - never more than the launch has pixels per workgroup.
LaunchCfg make_launch_cu(const VimgDeviceScene* s, const VimgRenderParams* p, int sx, int sy) {
  LaunchCfg c{};
  const VimgHipOptions& o = s->opt;
  const uint64_t items = (sx >= 0) ? 1 : uint64_t(local_tiles(s, p)) * 64u;
  c.sched = VIMG_SCHED_CU;
  c.pooled = true;
  c.group = false;
  c.rays = 1;
  c.wps = 4;
  c.cu_waves = 16;   // (a build with two 8-wave workgroups per CU halves the pool a batch draws from; not built)
  const uint32_t nw = uint32_t(c.cu_waves);
  RenderArgs& a = c.args;
  a.integrator = p->integrator;
  a.samples = p->samples;
  a.depth = p->depth;
  a.tile_rank = p->tile_rank;
  a.tile_world = p->tile_world;
  a.tiles_x = tiles_of(s->d.res_x);
  a.tiles_y = tiles_of(s->d.res_y);
  a.num_local_tiles = local_tiles(s, p);
  a.full_stats = 0;
  a.single_x = sx;
  a.single_y = sy;
  a.stack_entries = s->d.max_depth + 2;
  a.stack_lds = std::min(a.stack_entries, std::max(1u, opt_or(o.lds_stack, 32u)));
  a.stack_ovf = nullptr;
  // walking waves: five of eight by policy (config 2: the walk is 60 % of the wave cycles); when every
  // wave walks, every wave must be allowed to shade too
  a.cu_walkers = 0;   // (set below, once the tree's place is known)
  a.cu_flex = opt_or(o.cu_flex, 1u);   // (bit 1 / 2: shading / walking at wave priority 1; bit 4: no split batches; bit 5: early rays, by policy below)
  a.cu_lowwater = std::max(1u, opt_or(o.cu_lowwater, 64u));
  a.cu_patience = opt_or(o.cu_patience, 4u);
  a.cu_join = std::max(1u, opt_or(o.cu_join, 1u));
  a.cu_sleep = std::min(127u, std::max(1u, opt_or(o.cu_sleep, 4u)));
  a.pool_refill = 16u;   // (set below, once the tree's place is known)
  a.pool_vbatch = std::min(64u, std::max(1u, opt_or(o.pool_vbatch, 64u)));
  a.pool_boxmin = std::min(64u, opt_or(o.pool_boxmin, 16u));
  a.pool_starve = std::min(64u, std::max(1u, opt_or(o.pool_starve, 16u)));   // smallest partial batch worth a wave at once
  a.pool_classes = std::min(3u, std::max(1u, opt_or(o.pool_classes, 3u)));
  a.pool_gbreak = 0;
  // LDS of the workgroup: the whole CU's (16 waves) or half of it, minus a margin
  const uint32_t share = (160u * 1024u) / (16u / nw) - 1024u;
  uint32_t node_budget = 4608u;
  if (o.lds_budget_kb != VIMG_OPT_AUTO) node_budget = uint32_t(std::max(1, o.lds_budget_kb)) * 1024u;
  a.lds_nodes = std::min(node_budget / 56u, s->d.num_nodes);
  const uint32_t node_bytes = (a.lds_nodes * 56u + 255u) & ~255u;
  // the build with the overflow path and the global node fetch serves both trees beyond the LDS
  // node cache and stacks deeper than their LDS rows
  c.deep = a.lds_nodes < s->d.num_nodes || a.stack_lds < a.stack_entries;
  // walking waves (profiles/r3_cu/sweeps.txt): trees in LDS 9 of 16 (config 2 at 512 spp: 8 waves 307,
  // 9: 304, 10: 325, 11: 337 ms), trees in global memory 10 (stand-ins of configs 4 / 5 at 32 spp:
  // 8 waves 2 650 / 3 436, 10: 2 796 / 4 019, 12: 2 649 / 3 762 Mrays/s); when every wave walks, every
  // wave must be allowed to shade too
  a.cu_walkers = std::min(nw, std::max(1u, opt_or(o.cu_walkers, c.deep ? 10u : 9u)));
  // finished rays that send a walking wave to its rings (hand-over, then refill): 16 on trees in LDS; on
  // trees in global memory, where a pass waits for memory and a finished ray would wait with it, 2
  // (stand-ins of configs 4 / 5: an eighth of the frame at 128 spp 82.8 / 77.9 against 93.3 / 82.3 ms,
  // a quarter 87.7 against 92.2, a half 101.9 against 107.8, the whole frame unchanged)
  a.pool_refill = std::max(1u, opt_or(o.pool_refill, c.deep ? 2u : 16u));
  const uint32_t stack_rows = pool4_stack_rows_of(a.stack_entries, a.stack_lds);
  uint32_t leaf_bytes = 0;
  a.lds_leaf = 0;
  if (s->num_leaf_prims * 48u <= 4096u && o.lds_leaf != 0) {
    a.lds_leaf = s->num_leaf_prims;
    leaf_bytes = a.lds_leaf * 48u;
  }
  auto slots_with = [&](uint32_t walkers) {
    const uint32_t fixed = node_bytes + walkers * stack_rows * 256u + cu_pool_bytes(0, nw) + leaf_bytes + 64u;
    uint32_t n = share > fixed ? (share - fixed) / CU_LDS_BYTES : 0u;
    n = std::min(n, 4096u);
    // (trees in LDS: the rate is flat from 1 152 slots on - config 2 at 512 spp: 896 slots 323, 1 024: 305,
    // 1 152: 297.7, 1 280: 297.7, 1 408: 298.4, all 1 490 the LDS holds: 300.2 ms; smaller cold regions stay in L2)
    if (!c.deep && o.pool_slots == VIMG_OPT_AUTO) n = std::min(n, 1280u);
    if (o.pool_slots != VIMG_OPT_AUTO) n = std::min(n, uint32_t(std::max(0, o.pool_slots)));
    return n;
  };
  // never more slots than the launch has pixels per workgroup (a thin shard's pixels each own a slot
  // from the first sample to the last)
  const uint32_t groups = s->num_cus * (16u / nw);
  const uint64_t per_group = (items + groups - 1) / groups;
  // a launch whose pixels all own a slot, on a tree in LDS: every wave walks AND shades (a quarter of
  // config 2: 135.0 against 142.4 ms; an eighth and a third: no difference)
  if (o.cu_walkers == VIMG_OPT_AUTO && !c.deep && o.pool_slots == VIMG_OPT_AUTO && per_group + 8u <= slots_with(nw)) a.cu_walkers = nw;
  if (a.cu_walkers == nw) a.cu_flex |= 1u;
  const uint32_t stack_bytes = a.cu_walkers * stack_rows * 256u;
  uint32_t slots = slots_with(a.cu_walkers);
  // ... and when the pixels are more than the slots but fewer than 2.7 pools' worth (half a frame of
  // config 2), a pool of pixels / 2.7: the segments of a pixel are handed from slot to slot, and a
  // slot that draws a segment whose predecessor is still running can only wait - with 1.65
  // generations of slots per pixel half of config 2 took 253 ms, with 2.7 (1 040 slots) 176, with 3.5 180
  if (o.pool_slots == VIMG_OPT_AUTO && per_group > slots && per_group * 10u < uint64_t(slots) * 27u)
    slots = static_cast<uint32_t>(per_group * 10u / 27u);
  slots = static_cast<uint32_t>(std::min<uint64_t>(slots, per_group + 8u));
  slots = std::max(slots & ~7u, 8u);
  // EARLY rays (cu_flex bit 5: a vertex stage queues each ray as soon as it is known and finishes beside
  // the walk) wherever a slot's hop latency is on the frame's critical path: launches of fewer than three
  // pools' worth of pixels, and trees in global memory (whose walks are long).  Config 2: an eighth 110.4 ->
  // 99.9 ms, a quarter 132.5 -> 118.3, a half 170.6 -> 164.5, the whole frame 300.4 -> 309.7 (it only pays the
  // two extra ring operations per vertex: off there); stand-ins of configs 4 / 5: whole frame 121.4 -> 118.4 /
  // 127.0 -> 124.1, an eighth 81.3 -> 79.4 / 76.8 -> 69.9
  if (o.cu_flex == VIMG_OPT_AUTO && (c.deep || per_group * 10u <= uint64_t(slots) * 30u)) a.cu_flex |= 32u;
  // a tree in global memory on a launch whose pixels all own a slot: the box loop yields to waiting leaves
  // below 8 descending lanes instead of 16 (stand-ins of configs 4 / 5, a quarter at 128 spp: 79.3 / 69.9
  // against 80.8 / 72.7 ms, an eighth 76.7 / 65.5 against 76.9 / 70.1; halves and whole frames want 16)
  if (c.deep && o.pool_boxmin == VIMG_OPT_AUTO && per_group + 8u >= slots && per_group <= slots) a.pool_boxmin = 8u;
  a.pool_slots = slots;
  magic_div(slots, &a.cu_magic_v, &a.cu_shift_v);
  magic_div(2u * slots, &a.cu_magic_w, &a.cu_shift_w);
  c.lds_bytes = node_bytes + stack_bytes + cu_pool_bytes(slots, nw) + leaf_bytes;
  int per_cu = 0;
  hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel_of(s, c), int(nw * 64u), c.lds_bytes);
  if (oe != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min<int>(per_cu, int(16u / nw));
  const uint64_t need_blocks = (items + slots - 1) / slots;
  c.grid = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(need_blocks, uint64_t(s->num_cus) * per_cu)));
  // segments: as the group build of render_pool4_kernel (the tail of a frame is one segment long)
  a.pool_segments = 1;
  a.pool_seg_len = p->samples;
  a.cu_watchdog = static_cast<uint32_t>(1000000000ull >> 20);
  if (sx < 0) {
    const uint64_t in_flight = uint64_t(c.grid) * slots;
    const double gens = double(items) / double(in_flight);
    uint32_t k = gens >= 10.0 ? 1u : uint32_t(std::min(64.0, std::max(1.0, std::floor(176.0 / gens + 0.5))));
    k = std::min<uint32_t>(k, std::max<uint32_t>(p->samples / 4u, 1u));
    if (items * 2u < in_flight * 3u) k = 1u;
    if (o.pool_segments != VIMG_OPT_AUTO) k = uint32_t(std::max(1, o.pool_segments));
    k = std::min<uint32_t>(k, 4096u);
    while (k > 1u && items * k >= 0xfff00000ull) --k;
    const uint32_t len = std::max<uint32_t>((p->samples + k - 1) / k, 1u);
    a.pool_seg_len = len;
    a.pool_segments = std::max<uint32_t>((p->samples + len - 1) / len, 1u);
    a.cu_watchdog = static_cast<uint32_t>(std::min<uint64_t>((1000000000ull + 5000000ull * len) >> 20, 0x7fffffffull));   // 10 s + 50 ms per sample of a segment
  }
  return c;
}

LaunchCfg make_launch(const VimgDeviceScene* s, const VimgRenderParams* p, int sx, int sy,
                      bool for_render = true, int sched_override = 0, bool lds_stack_all = false) {
  LaunchCfg c{};
  const VimgHipOptions& o = s->opt;
  // Policy (AUTO): the CU-wide scheduler for every launch - whole frames, thin shards, trace_pixel.  The
  // lane-bound kernel runs when asked for by name and for frames wider than the 16-bit pixel
  // coordinates of the slot records; the schedulers of rounds 1 and 2 by name, in the development build.
  if (for_render && !sched_override && (o.scheduler == VIMG_SCHED_CU || o.scheduler == VIMG_OPT_AUTO) && !s->too_wide)
    return make_launch_cu(s, p, sx, sy);
  const uint64_t items = (sx >= 0) ? 1 : uint64_t(local_tiles(s, p)) * 64u;
  // ---- which scheduler.  Policy (AUTO): the pooled scheduler with its vertex stage as calls
  // (pool4); launches with too few pixels per wave for pools of 64 slots - test images,
  // trace_pixel, thin shards - go to the lane-bound kernel (decided below, where the pool is
  // sized).  The first pooled kernel and the staged kernel (global queues) run when asked for by name.
  int sched = sched_override ? sched_override : (o.scheduler == VIMG_OPT_AUTO ? 0 : o.scheduler);
  const bool by_policy = (sched == 0);
  if (!for_render) sched = VIMG_SCHED_LANE;   // probes and the heatmap only need the LDS layout
  if (sched == 0) sched = VIMG_SCHED_LANE;   // (AUTO comes here only for frames too wide for the slot records)
  c.group = (sched == VIMG_SCHED_POOL4G);
  if (c.group) sched = VIMG_SCHED_POOL4;   // the same launch in everything but the pool's layout and the kernel build
  if (s->too_wide && sched != VIMG_SCHED_LANE) sched = VIMG_SCHED_LANE;        // slots pack pixel coordinates in 16 bits
  if (sched == VIMG_SCHED_STAGE && items > (1ull << 26)) sched = VIMG_SCHED_POOL4;   // 32-bit byte offsets of the pixel records
  if (sched != VIMG_SCHED_POOL4) c.group = false;
  c.sched = sched;
  c.pooled = (sched == VIMG_SCHED_POOL || sched == VIMG_SCHED_POOL4);
  // register budget: the lane-bound kernel wants 3 waves per SIMD on scenes beyond the on-chip
  // caches (latency-bound) and 2 on small ones (VALU-bound, fewest spills); the pooled kernel
  // hides latency with its slots and always takes the 256-register build (config 4/5: 2 waves
  // 1.02 / 1.70 Grays/s, 3 waves 0.66 / 0.91); the staged kernel has one build (128 registers)
  c.wps = c.pooled ? 2 : s->waves_per_simd;
  if (o.waves_per_simd != VIMG_OPT_AUTO) c.wps = o.waves_per_simd >= 3 ? 3 : 2;
  if (sched == VIMG_SCHED_STAGE) c.wps = 4;
  // pool4: three waves per SIMD by policy (config 2: 12.2 Grays/s at three, 11.3 at four; the stand-ins
  // of configs 3 / 4 / 5: 6.6 / 1.56 / 2.62 against 6.1 / 1.15 / 1.52 - a wave's LDS share, i.e. its
  // pool, shrinks faster than the fourth wave pays, most of all under the deep trees' stacks)
  if (sched == VIMG_SCHED_POOL4) c.wps = (o.waves_per_simd == 4) ? 4 : 3;
  c.rays = 1;
  RenderArgs& a = c.args;
  a.integrator = p->integrator;
  a.samples = p->samples;
  a.depth = p->depth;
  a.tile_rank = p->tile_rank;
  a.tile_world = p->tile_world;
  a.tiles_x = tiles_of(s->d.res_x);
  a.tiles_y = tiles_of(s->d.res_y);
  a.num_local_tiles = local_tiles(s, p);
  a.full_stats = 0;
  a.stack_entries = s->d.max_depth + 2;
  a.stack_lds = a.stack_entries;
  a.stack_ovf = nullptr;
  // pool4 on trees that do not fit in LDS: the first `lds_stack` (AUTO 32) entries of a lane's stack in LDS, the rest in
  // global memory (the LDS goes to path slots instead); `lds_stack_all`: second pass, when the tree
  // turned out to fit (the build without the overflow path)
  if (sched == VIMG_SCHED_POOL4 && !lds_stack_all)
    a.stack_lds = std::min(a.stack_entries, std::max(1u, opt_or(o.lds_stack, 32u)));
  const uint32_t stack_rows = (sched == VIMG_SCHED_POOL4) ? pool4_stack_rows_of(a.stack_entries, a.stack_lds) : a.stack_entries;
  a.single_x = sx;
  a.single_y = sy;
  // LDS budget per 256-thread workgroup: stacks first, then as much of the top of the tree as
  // fits in 40 KiB total (keeps >= 4 workgroups per CU inside the 160 KiB)
  const uint32_t stack_bytes = 4u * stack_rows * 64u * 4u * uint32_t(c.rays);
  // (the pooled and staged kernels spend LDS on path slots / queue chunks instead: they keep the
  // first six levels of the tree, 4 KiB - config 5: 40 KiB budget 1.69, 28 KiB 1.78 Grays/s)
  uint32_t budget = (sched != VIMG_SCHED_LANE) ? std::min(40u * 1024u, stack_bytes + 4608u) : 40u * 1024u;
  if (o.lds_budget_kb != VIMG_OPT_AUTO) budget = uint32_t(std::max(1, o.lds_budget_kb)) * 1024u;
  uint32_t nodes = 0;
  if (stack_bytes + 512 < budget) nodes = (budget - stack_bytes - 256) / 56u;
  a.lds_nodes = std::min(nodes, s->d.num_nodes);
  c.lds_bytes = ((a.lds_nodes * 56u + 255u) & ~255u) + stack_bytes;
  a.pool_slots = 0;
  a.pool_refill = 16u;   // (set below, once the tree's place is known)
  a.pool_vbatch = std::min(64u, std::max(1u, opt_or(o.pool_vbatch, 64u)));
  // config 4 / 5 stand-ins: never 1.02 / 1.74, 8 lanes 1.19 / 2.10, 16: 1.18 / 2.13, 24: 1.20 / 2.13,
  // 40: 1.15 / 1.93 Grays/s
  a.pool_boxmin = std::min(64u, opt_or(o.pool_boxmin, 16u));
  a.pool_gbreak = std::min(64u, opt_or(o.pool_gbreak, 32u));

  c.deep = (sched != VIMG_SCHED_LANE) && a.lds_nodes < s->d.num_nodes;   // the other build reads every node from LDS
  if (!c.deep && a.stack_lds < a.stack_entries) return make_launch(s, p, sx, sy, for_render, sched_override, true);
  // Vertex queues and the starvation threshold.  Trees in LDS (pools of 150-190 slots): one queue per
  // material class, a partial batch when 24 walk lanes idle.  Trees in global memory leave a pool of
  // about 100 slots, which three class queues drain to 21-27 slots per batch and 20 rays per walk
  // pass: there ONE queue of shading vertices (next to the finishers') and 32 idle lanes measure best
  // (stand-ins of configs 4 / 5, 32 spp: 1.60 -> 1.72, 2.68 -> 2.81 Grays/s;
  // profiles/r2_pool4/deep_policy_sweeps.txt)
  a.pool_classes = std::min(3u, std::max(1u, opt_or(o.pool_classes, c.deep ? 1u : 3u)));
  a.pool_starve = std::min(64u, std::max(1u, opt_or(o.pool_starve, c.deep ? 32u : 24u)));
  // small scenes: all leaf records in LDS too (they cost a few slots, the walk gains more)
  uint32_t leaf_bytes = 0;
  a.lds_leaf = 0;
  if (sched != VIMG_SCHED_LANE && s->num_leaf_prims * 48u <= 4096u && o.lds_leaf != 0) {
    a.lds_leaf = s->num_leaf_prims;
    leaf_bytes = a.lds_leaf * 48u;
  }
  if (c.pooled) {
    // Pixels are the unit of parallelism (one sequential RNG stream per pixel): a launch with few
    // pixels per wave is fastest with pools of about pixels / 2.4 slots, and with very few the
    // lane-bound kernel wins - the pooled scheduler's hop latency times the longest pixel's chain of
    // path vertices is then the whole frame time.  By policy only; what is asked for by name stands.
    const bool policy_pool4 = sched == VIMG_SCHED_POOL4 && by_policy && !sched_override && o.pool_slots == VIMG_OPT_AUTO;
    uint64_t want = ~0ull;
    if (policy_pool4) {
      if (sx >= 0) return make_launch(s, p, sx, sy, for_render, VIMG_SCHED_LANE);   // trace_pixel: one path
      const uint64_t waves = uint64_t(s->num_cus) * 3u * 4u;   // three workgroups per CU (checked against the runtime below)
      want = items * 10u / (waves * 24u);
      // (trees in global memory never go there: the lane-bound kernel pays a memory round trip per
      // phase of its machine - quarter / eighth of the config-4 stand-in, 128 spp: 261 / 243 ms against
      // 134 / 112 ms with group pools of 32 slots per wave)
      if (want < 40u && !c.deep) return make_launch(s, p, sx, sy, for_render, VIMG_SCHED_LANE);
      // A full frame on a tree in LDS: FOUR waves per SIMD with group pools (config 2, 512 spp: 13.4
      // against 12.7 Grays/s with three waves and per-wave pools - the fourth wave's issue slots pay
      // now that its smaller LDS share no longer thins the batches; config 3: 7.63 against 7.47).
      // Shards keep three (half of config 2: 209 against 226 ms), and so do trees in global memory
      // (their stacks leave a four-wave workgroup no LDS for slots: 1.34 against 2.09 Grays/s).
      if (!c.deep && o.waves_per_simd == VIMG_OPT_AUTO && want >= 160u) {
        c.wps = 4;
        want = want * 3u / 4u;   // per wave of the larger grid
      }
    }
    // the pool takes what is left of this workgroup's share of the CU's 160 KiB
    const uint32_t share = (160u * 1024u) / uint32_t(c.wps) - 1024u;
    if (sched == VIMG_SCHED_POOL4) c.lds_bytes += 4u * uint32_t(sizeof(Pool4Wave) + sizeof(Pool4Diag));
    auto slots_for = [&](uint32_t slot_bytes, uint32_t extra) {
      const uint32_t used = c.lds_bytes + extra + leaf_bytes + 64u;
      return std::min(share > used ? (share - used) / (slot_bytes * 4u) : 0u, 256u);
    };
    uint32_t slots = slots_for((sched == VIMG_SCHED_POOL4) ? P4_LDS_BYTES : POOL_LDS_BYTES, 0);
    // Which pool4 build (by policy).  One pool per WAVE only when three waves per SIMD are asked for
    // on a full frame of a tree in LDS (config 2 12.3 against 11.9 Grays/s - there the group's lock
    // costs more than its fuller batches earn).  One pool per WORKGROUP otherwise: at four waves per
    // SIMD (above), and wherever pools are small - trees in global memory, whose
    // stacks take half the LDS (stand-ins of configs 4 / 5, 32 spp: 1.72 -> 2.07, 2.81 -> 3.22
    // Grays/s), and frames with few pixels per wave (half of config 2: 238 -> 209 ms; a quarter:
    // lane-bound 205 -> 180 ms with 64 slots; an eighth stays with the lane-bound kernel, 139 ms).
    if (policy_pool4) c.group = c.wps == 4 || c.deep || want < slots;
    if (c.group) {
      c.lds_bytes += pool4g_group_bytes(0);   // group record, batch rows
      slots = slots_for(P4G_LDS_BYTES, 0);
    }
    const uint32_t slot_bytes = c.group ? P4G_LDS_BYTES : (sched == VIMG_SCHED_POOL4) ? P4_LDS_BYTES : POOL_LDS_BYTES;
    if (o.pool_slots != VIMG_OPT_AUTO) slots = std::min(slots, uint32_t(std::max(0, o.pool_slots)));
    a.pool_slots = std::max(slots, 8u);
    if (sched == VIMG_SCHED_POOL4) {
      if (policy_pool4)
        a.pool_slots = static_cast<uint32_t>(std::min<uint64_t>(a.pool_slots, std::max<uint64_t>(want, c.deep ? 32u : 64u)));
      a.pool_slots &= ~1u;   // even: every wave's cold region starts on a 64-byte line (and a group's tables on 16 bytes)
    }
    c.lds_bytes += (c.group ? slot_bytes * 4u * a.pool_slots : 4u * ((slot_bytes * a.pool_slots + 15u) & ~15u)) + leaf_bytes;
    // vertex queues and thresholds of the group build: one queue per material class again (the
    // group's queues fill), 32 idle lanes before a partial batch, a full batch taken by a wave with
    // at most 32 rays in its lanes
    if (c.group) {
      a.pool_classes = std::min(3u, std::max(1u, opt_or(o.pool_classes, 3u)));
      a.pool_gbreak = std::min(64u, opt_or(o.pool_gbreak, 32u));
    }
  }
  StageArgs& g = c.stage;
  if (sched == VIMG_SCHED_STAGE) {
    g.wchunk = std::min(STAGE_WCHUNK_MAX, std::max(128u, opt_or(o.stage_wchunk, 128u)));
    g.walk_quota = std::max(g.wchunk, opt_or(o.stage_walk_quota, 2048u));
    g.seg_len = std::max(1u, opt_or(o.stage_seg_len, 4u));
    c.lds_bytes += 4u * (5u * g.wchunk * 4u + 256u) + leaf_bytes;
  }
  // persistent grid: as many 4-wave workgroups as the kernel's registers and LDS let a CU hold
  // (asked of the runtime), never more than the work
  int per_cu = 0;
  hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel_of(s, c), 256, c.lds_bytes);
  if (oe != hipSuccess || per_cu < 1) per_cu = 1;
  const uint64_t need_blocks = (items + 255) / 256;
  c.grid = static_cast<uint32_t>(
      std::max<uint64_t>(1, std::min<uint64_t>(need_blocks, uint64_t(s->num_cus) * per_cu)));
  if (sched == VIMG_SCHED_STAGE) {
    // slots in flight: twice the resident lanes (every stage then finds full batches queued while
    // as many paths are being worked on), never more than the pixels of the launch, which are the
    // unit of parallelism (one sequential RNG stream per pixel, include/integrators.h:116-127)
    const uint64_t lanes = uint64_t(c.grid) * 256u;
    uint64_t n = opt_or(o.stage_slots, static_cast<uint32_t>(std::min<uint64_t>(lanes * 2u, STAGE_MAX_SLOTS)));
    n = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n, items), STAGE_MAX_SLOTS));
    g.n_slots = static_cast<uint32_t>(n);
    g.ring_cap = std::max(64u, ceil_pow2(n));
    g.ring_shift = log2_of(g.ring_cap);
    g.pix_cap = std::max(64u, ceil_pow2(items));
    g.pix_shift = log2_of(g.pix_cap);
    g.rings_bytes = GQ_COUNT * GQ_SHARDS * g.ring_cap * 4u;
    g.pix_ring_bytes = g.pix_cap * 4u;
    g.pix_state_bytes = static_cast<uint32_t>(items * 32u);
    g.slots_bytes = g.n_slots * GR_BYTES;
  }
  // pooled kernel: split every pixel's samples into segments handed out as separate work items
  // when the image is large against the slots in flight (then the previous segment of a pixel
  // has long been published when its next one is drawn); small images keep one segment
  a.pool_segments = 1;
  a.pool_seg_len = p->samples;
  if (a.pool_slots && sx < 0) {
    const uint64_t in_flight = uint64_t(c.grid) * 4u * a.pool_slots;
    // Segments: the tail of a frame is one segment long, and every hand-over costs a little
    // (config 2, 3.5 pool generations per frame: 1 segment 6.8, 4: 7.5, 8: 7.6, 16-32: 7.6 Grays/s;
    // 3600x1600, 14 generations: 1 segment 7.9, 4: 7.7) - about 56 segments per generation count,
    // at most 16, of at least 4 samples; frames of 10 generations and more keep their pixels whole
    const double gens = double(items) / double(in_flight);
    // (group pools at four waves, config 2 at 512 spp, 2.75 generations: 8 segments 322.5 ms, 16: 315.3,
    // 32: 312.9, 64: 311.4 - the pooled kernels before it were flat from 16 on)
    const bool more = c.group && c.wps == 4;
    uint32_t k = gens >= 10.0 ? 1u : uint32_t(std::min(more ? 64.0 : 16.0, std::max(1.0, std::floor((more ? 176.0 : 56.0) / gens + 0.5))));
    k = std::min<uint32_t>(k, std::max<uint32_t>(p->samples / 4u, 1u));
    if (items * 2u < in_flight * 3u) k = 1u;
    if (o.pool_segments != VIMG_OPT_AUTO) k = uint32_t(std::max(1, o.pool_segments));
    k = std::min<uint32_t>(k, 4096u);
    while (k > 1u && items * k >= 0xfff00000ull) --k;   // (segment, pixel) items must fit the 32-bit counter
    const uint32_t len = std::max<uint32_t>((p->samples + k - 1) / k, 1u);
    a.pool_seg_len = len;
    a.pool_segments = std::max<uint32_t>((p->samples + len - 1) / len, 1u);
  }
  return c;
}

int grow(void** p, size_t* have, size_t need) {
  if (need <= *have) return VIMG_OK;
  if (*p) HIP_TRY(hipFree(*p));
  *p = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(p, need));
  *have = need;
  return VIMG_OK;
}
// The pooled kernel keeps the cold records of its path slots in global memory: one region per
// resident wave, owned by the scene and grown on demand (42 MB for config 2 on 256 CUs).
int ensure_pool(VimgDeviceScene* s, LaunchCfg& c) {
  c.args.pool_cold = nullptr;
  if (!c.pooled || c.args.pool_slots == 0) return VIMG_OK;
  const bool cu = c.sched == VIMG_SCHED_CU;
  const size_t ncold = (c.sched == VIMG_SCHED_POOL4 || cu) ? pool4_cold_records(s->textured)
                                                            : (s->textured ? SC_COUNT : SC_COUNT - 1u);
  const size_t need = size_t(c.grid) * (cu ? 1u : 4u) * ncold * c.args.pool_slots * 16u;
  if (need > s->pool_cold_bytes) {
    if (s->d_pool_cold) HIP_TRY(hipFree(s->d_pool_cold));
    s->d_pool_cold = nullptr;
    s->pool_cold_bytes = 0;
    HIP_TRY(hipMalloc(&s->d_pool_cold, need));
    s->pool_cold_bytes = need;
  }
  c.args.pool_cold = (VIMG_GLOBAL v4u*)s->d_pool_cold;
  if (c.args.stack_lds < c.args.stack_entries) {
    if (int rc = grow(&s->d_stack_ovf, &s->stack_ovf_bytes,
                      size_t(c.grid) * (cu ? c.args.cu_walkers : 4u) * uint32_t(c.rays) * (c.args.stack_entries - c.args.stack_lds) * 256u))
      return rc;
    c.args.stack_ovf = (VIMG_GLOBAL uint32_t*)s->d_stack_ovf;
  }
  c.args.pool_state = nullptr;
  c.args.pool_epoch = 0;
  if (c.args.pool_segments > 1) {
    const size_t want = size_t(c.args.num_local_tiles) * 64u * 32u;
    if (want > s->pool_state_bytes) {
      if (s->d_pool_state) HIP_TRY(hipFree(s->d_pool_state));
      s->d_pool_state = nullptr;
      s->pool_state_bytes = 0;
      HIP_TRY(hipMalloc(&s->d_pool_state, want));
      HIP_TRY(hipMemset(s->d_pool_state, 0, want));
      s->pool_state_bytes = want;
      s->pool_epoch = 0;
    }
    // tags are epoch + segment index (< 4096): one epoch step per launch, wrap with a wipe
    s->pool_epoch += 4096u;
    if (s->pool_epoch >= 0xffff0000u) {
      HIP_TRY(hipMemset(s->d_pool_state, 0, s->pool_state_bytes));
      s->pool_epoch = 4096u;
    }
    c.args.pool_state = (VIMG_GLOBAL v4u*)s->d_pool_state;
    c.args.pool_epoch = s->pool_epoch;
  }
  return VIMG_OK;
}

// The staged kernel keeps all path state in global memory, owned by the scene and grown on demand:
// control block, queue rings, ready-pixel ring, per-pixel records, slot records (config 2 on 256
// CUs: 0.01 + 42 + 8 + 46 + 50 MB).  Counters, rings and the ready-pixel ring are cleared per launch.
int ensure_stage(VimgDeviceScene* s, LaunchCfg& c, hipStream_t st) {
  if (c.sched != VIMG_SCHED_STAGE) return VIMG_OK;
  StageArgs& g = c.stage;
  if (!s->d_stage_ctl) HIP_TRY(hipMalloc(&s->d_stage_ctl, sizeof(StageCtl)));
  if (int rc = grow(&s->d_stage_rings, &s->stage_rings_bytes, g.rings_bytes)) return rc;
  if (int rc = grow(&s->d_stage_pix_ring, &s->stage_pix_ring_bytes, g.pix_ring_bytes)) return rc;
  if (int rc = grow(&s->d_stage_pix_state, &s->stage_pix_state_bytes, g.pix_state_bytes)) return rc;
  if (int rc = grow(&s->d_stage_slots, &s->stage_slots_bytes, g.slots_bytes)) return rc;
  g.ctl = (VIMG_GLOBAL StageCtl*)s->d_stage_ctl;
  g.rings = (VIMG_GLOBAL uint32_t*)s->d_stage_rings;
  g.pix_ring = (VIMG_GLOBAL uint32_t*)s->d_stage_pix_ring;
  g.pix_state = (VIMG_GLOBAL v4u*)s->d_stage_pix_state;
  g.slots = (VIMG_GLOBAL v4u*)s->d_stage_slots;
  HIP_TRY(hipMemsetAsync(s->d_stage_ctl, 0, sizeof(StageCtl), st));
  HIP_TRY(hipMemsetAsync(s->d_stage_rings, 0, g.rings_bytes, st));
  HIP_TRY(hipMemsetAsync(s->d_stage_pix_ring, 0, g.pix_ring_bytes, st));
  // pixels nobody has started: all of them but one per slot (the slots start "fresh")
  const uint64_t items = (c.args.single_x >= 0) ? 1 : uint64_t(c.args.num_local_tiles) * 64u;
  const uint32_t surplus = static_cast<uint32_t>(items - g.n_slots);
  StageCtl* ctl = static_cast<StageCtl*>(s->d_stage_ctl);
  HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&ctl->surplus.v), static_cast<int>(surplus), 1, st));
  return VIMG_OK;
}

// Enqueues one render on `st` (counter / queue resets, then the kernel); ev0 / ev1, when given, are
// recorded right before and right after the kernel itself.
int enqueue_render(VimgDeviceScene* s, const VimgRenderParams* p, float* d_out, hipStream_t st,
                   bool full_stats, bool want_stats, int sx, int sy, hipEvent_t ev0 = nullptr,
                   hipEvent_t ev1 = nullptr) {
  LaunchCfg c = make_launch(s, p, sx, sy);
  if (int rc = ensure_pool(s, c)) return rc;
  c.args.full_stats = full_stats ? 1u : 0u;
  if (c.args.num_local_tiles == 0 && sx < 0) return VIMG_OK;
  if (int rc = ensure_stage(s, c, st)) return rc;
  if (!s->d_stage_kargs) HIP_TRY(hipMalloc(&s->d_stage_kargs, std::max(sizeof(StageKArgs), sizeof(Pool4KArgs))));
  // (the work counter only: the error word behind it is sticky until a blocking call or vimg_hip_check reads it)
  HIP_TRY(hipMemsetAsync(s->d_counter, 0, sizeof(unsigned int), st));
  if (want_stats) HIP_TRY(hipMemsetAsync(s->d_stats, 0, sizeof(DeviceStats), st));
  DeviceStats* stats = want_stats ? s->d_stats : nullptr;
  if (c.lds_bytes > 48u * 1024u) {   // ask for the large dynamic-LDS carve-out
    const void* kfn = (c.sched == VIMG_SCHED_CU) ? reinterpret_cast<const void*>(pick_cu_kernel(s, c.deep, c.cu_waves, full_stats, (c.args.cu_flex & 32u) != 0u))
                                                 : kernel_of(s, c);
    HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, int(c.lds_bytes)));
  }
  if (ev0 && c.sched != VIMG_SCHED_STAGE && c.sched != VIMG_SCHED_POOL4) HIP_TRY(hipEventRecord(ev0, st));
  if (c.sched == VIMG_SCHED_STAGE)
  {
    // scene + launch parameters go to the block the stage functions read (stream-ordered, by value)
    StageKArgs* blk = static_cast<StageKArgs*>(s->d_stage_kargs);
    hipLaunchKernelGGL(stage_args_kernel, dim3(1), dim3(64), 0, st, StageKArgs{s->d, c.args, c.stage, d_out, stats}, blk);
    if (ev0) HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(pick_stage_kernel(s, c.deep), dim3(c.grid), dim3(256), c.lds_bytes, st,
                       static_cast<const StageKArgs*>(blk));
  }
  else if (c.sched == VIMG_SCHED_POOL4) {
    Pool4KArgs* blk = static_cast<Pool4KArgs*>(s->d_stage_kargs);
    hipLaunchKernelGGL(pool4_args_kernel, dim3(1), dim3(64), 0, st, Pool4KArgs{s->d, c.args, d_out, stats, s->d_counter}, blk);
    if (ev0) HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(pick_pool4_kernel(s, c.deep, c.wps, c.group), dim3(c.grid), dim3(256), c.lds_bytes, st,
                       static_cast<const Pool4KArgs*>(blk));
  } else if (c.sched == VIMG_SCHED_CU)
    hipLaunchKernelGGL(pick_cu_kernel(s, c.deep, c.cu_waves, full_stats, (c.args.cu_flex & 32u) != 0u), dim3(c.grid),
                       dim3(uint32_t(c.cu_waves) * 64u), c.lds_bytes, st,
                       CuKArgs{s->d, c.args, d_out, stats, s->d_counter});
  else
    hipLaunchKernelGGL(pick_kernel(s, c.pooled, c.wps, c.deep), dim3(c.grid), dim3(256), c.lds_bytes, st, s->d, c.args,
                       d_out, stats, s->d_counter);
  if (ev1) HIP_TRY(hipEventRecord(ev1, st));
  HIP_TRY(hipGetLastError());
  return VIMG_OK;
}
int launch_render(VimgDeviceScene* s, const VimgRenderParams* p, float* d_out, hipStream_t st,
                  bool full_stats, bool want_stats, int sx, int sy) {
  return enqueue_render(s, p, d_out, st, full_stats, want_stats, sx, sy);
}

// d_counter[1] is the error word of the last launch (raised by the pooled kernel's watchdog); the
// staged kernel has its own in its control block
int check_kernel_error(VimgDeviceScene* s) {
  unsigned int words[2] = {0, 0};
  HIP_TRY(hipMemcpy(words, s->d_counter, sizeof(words), hipMemcpyDeviceToHost));
  unsigned int stage_err = 0;
  if (s->d_stage_ctl)
    HIP_TRY(hipMemcpy(&stage_err, &static_cast<StageCtl*>(s->d_stage_ctl)->error.v, sizeof(stage_err), hipMemcpyDeviceToHost));
  if (words[1] != 0) HIP_TRY(hipMemset(s->d_counter + 1, 0, sizeof(unsigned int)));   // read once
  if (words[1] != 0 || stage_err != 0) {
    // bits of the launch's error word (render_cu_kernel.h: raise): 1 a wave found nothing to do for ten seconds
    // while slots were live, 2 a group lock timed out (development build), 4 a ring entry was reserved and never
    // written, 8 a compute unit queued more than 2^31 rays or slots in one launch
    const std::string what = (words[1] & 8u) ? "a compute unit queued more than 2^31 rays in one launch: render fewer samples per launch"
                                             : "a wave waited for work that never came";
    return fail(VIMG_E_DEVICE, "render kernel watchdog: " + what + " (the frame is incomplete), code " +
                                   std::to_string(words[1] | (stage_err << 8)));
  }
  return VIMG_OK;
}

int fetch_stats(VimgDeviceScene* s, const VimgRenderParams* p, VimgRenderStats* out) {
  DeviceStats ds{};
  HIP_TRY(hipMemcpy(&ds, s->d_stats, sizeof(ds), hipMemcpyDeviceToHost));
  *out = VimgRenderStats{};
  out->closest_rays = ds.closest;
  out->shadow_rays = ds.shadow;
  out->internal_visits = ds.internal;
  out->leaf_visits = ds.leaf;
  out->prim_tests = ds.prim;
  out->sphere_tests = ds.sphere;
  out->nan_samples = ds.nan_samples;
  // pixels owned by this shard (ragged edge tiles counted exactly)
  const uint32_t W = s->d.res_x, H = s->d.res_y, ty_n = tiles_of(H), total = tiles_of(W) * ty_n;
  uint64_t px = 0;
  for (uint32_t t = p->tile_rank; t < total; t += p->tile_world) {
    uint32_t tx = t / ty_n, ty = t % ty_n;
    px += uint64_t(std::min(8u, W - tx * 8)) * std::min(8u, H - ty * 8);
  }
  out->paths = px * p->samples;
  if (getenv("VIMG_HIP_DIAG"))
    std::fprintf(stderr, "[vimg diag] wave trips: descend %llu (lane visits %llu, util %.3f)  prim %llu (lane tests %llu, util %.3f)  main-loop iterations %llu\n",
                 ds.trip_descend, ds.internal, ds.trip_descend ? double(ds.internal) / (64.0 * ds.trip_descend) : 0.0,
                 ds.trip_prim, ds.prim, ds.trip_prim ? double(ds.prim) / (64.0 * ds.trip_prim) : 0.0, ds.iterations);
#ifndef VIMG_PROFILE
  if (getenv("VIMG_HIP_DIAG") && ds.prof[6 + 4]) {   // staged kernel: cycles and batch fill per stage
    static const char* st_names[6] = {"finisher", "lambertian", "principled", "other", "walk", "looking for work"};   // (pool4: walk includes waiting)
    unsigned long long total = 0;
    for (int k = 0; k < 6; ++k) total += ds.prof[k];
    for (int k = 0; k < 6; ++k)
      std::fprintf(stderr, "[vimg stage] %-18s %14llu cyc %6.2f %%  batches %10llu  slots/batch %7.2f\n", st_names[k],
                   ds.prof[k], 100.0 * double(ds.prof[k]) / double(total ? total : 1), k < 5 ? ds.prof[6 + k] : 0ull,
                   (k < 5 && ds.prof[6 + k]) ? double(ds.prof[11 + k]) / double(ds.prof[6 + k]) : 0.0);
    if (ds.prof[16])   // render_cu_kernel: passes of the walk loops and the lanes that took part
      std::fprintf(stderr, "[vimg walk] box passes %llu lanes/pass %.2f   leaf rounds %llu lanes/round %.2f   sessions %llu   refills %llu rays/refill %.2f\n",
                   ds.prof[16], double(ds.prof[17]) / double(ds.prof[16]), ds.prof[18], double(ds.prof[19]) / double(ds.prof[18] ? ds.prof[18] : 1),
                   ds.prof[20], ds.prof[21], double(ds.prof[22]) / double(ds.prof[21] ? ds.prof[21] : 1));
    if (ds.wait_n[4])
      std::fprintf(stderr, "[vimg wait] mean cycles in a ring before a wave takes the slot: finisher %.0f  lambertian %.0f  principled %.0f  other %.0f  walk %.0f\n",
                   double(ds.wait_cyc[0]) / double(ds.wait_n[0] ? ds.wait_n[0] : 1), double(ds.wait_cyc[1]) / double(ds.wait_n[1] ? ds.wait_n[1] : 1),
                   double(ds.wait_cyc[2]) / double(ds.wait_n[2] ? ds.wait_n[2] : 1), double(ds.wait_cyc[3]) / double(ds.wait_n[3] ? ds.wait_n[3] : 1),
                   double(ds.wait_cyc[4]) / double(ds.wait_n[4]));
    if (ds.ray_cyc[1])
      std::fprintf(stderr, "[vimg wait] mean cycles of a ray in a walking lane (pop to hand-over) %.0f; of a vertex batch: finisher %.0f lambertian %.0f principled %.0f other %.0f\n",
                   double(ds.ray_cyc[0]) / double(ds.ray_cyc[1]), double(ds.prof[0]) / double(ds.prof[6] ? ds.prof[6] : 1), double(ds.prof[1]) / double(ds.prof[7] ? ds.prof[7] : 1),
                   double(ds.prof[2]) / double(ds.prof[8] ? ds.prof[8] : 1), double(ds.prof[3]) / double(ds.prof[9] ? ds.prof[9] : 1));
    if (ds.pv_cyc[6])
      std::fprintf(stderr, "[vimg wait] a Principled batch, mean cycles: state loads %.0f  hit record + path logic %.0f  light sample %.0f  BSDF sample %.0f  evaluations %.0f  stores + hand-over %.0f\n",
                   double(ds.pv_cyc[0]) / double(ds.pv_cyc[6]), double(ds.pv_cyc[1]) / double(ds.pv_cyc[6]), double(ds.pv_cyc[2]) / double(ds.pv_cyc[6]),
                   double(ds.pv_cyc[3]) / double(ds.pv_cyc[6]), double(ds.pv_cyc[4]) / double(ds.pv_cyc[6]), double(ds.pv_cyc[5]) / double(ds.pv_cyc[6]));
    if (ds.px_done[1])
      std::fprintf(stderr, "[vimg wait] pixels finished (last sample written) after: mean %.3f ms, latest %.3f ms\n",
                   double(ds.px_done[0]) / double(ds.px_done[1]) * 1e-5, double(ds.px_done[2]) * 1e-5);
    if (ds.px_done[1])
      std::fprintf(stderr, "[vimg wait] vertex-stage visits per pixel: mean %.1f, most %llu; the pixel that finished last: %llu visits in %.3f ms = %.2f us per visit (walks included)\n",
                   double(ds.px_hops[0]) / double(ds.px_done[1]), ds.px_hops[1], ds.px_hops[2] & 0xffffffull, double(ds.px_hops[2] >> 24) * 1e-5,
                   double(ds.px_hops[2] >> 24) * 1e-2 / double((ds.px_hops[2] & 0xffffffull) ? (ds.px_hops[2] & 0xffffffull) : 1));
    if (ds.prof[23])
      std::fprintf(stderr, "[vimg walk] cycles: refill+setup %llu  box %llu  leaf %llu  hand-over %llu   |  looks %llu  mean ring counts seen: finisher %.1f lambertian %.1f principled %.1f walk %.1f\n",
                   ds.walk_cyc[0], ds.walk_cyc[1], ds.walk_cyc[2], ds.walk_cyc[3], ds.prof[23], double(ds.prof[24]) / double(ds.prof[23]),
                   double(ds.prof[25]) / double(ds.prof[23]), double(ds.prof[26]) / double(ds.prof[23]), double(ds.prof[27]) / double(ds.prof[23]));
#ifdef VIMG_WALK_DIAG
    static const char* wd_names[12] = {"walk: refill+setup cyc", "walk: box loop cyc", "walk: leaf rounds cyc", "walk: retire cyc",
                                       "box trips", "box lanes", "leaf rounds", "leaf lanes", "leaf prim trips",
                                       "retire: cyc to get lock", "retire: cyc to unlock", "retire: lock takes"};
    for (int k = 0; k < 12; ++k) std::fprintf(stderr, "[vimg walk] %-24s %14llu\n", wd_names[k], ds.prof[16 + k]);
#endif
  }
#endif
#ifdef VIMG_PROFILE
  if (getenv("VIMG_HIP_DIAG") && ds.prof[PF_TOTAL]) {
    static const char* names[PF_COUNT] = {"total", "v_load+logic+hit_info", "v_light_sample", "v_bsdf_sample",
                                          "v_bsdf_eval_x2", "v_finish+regen", "v_store", "w_refill+setup",
                                          "w_box_loop", "w_leaf_loop", "w_retire", "v_batches", "v_lanes",
                                          "v_at_vertex", "w_rounds", "cyc_finisher", "cyc_lambertian",
                                          "cyc_principled", "cyc_other", "lanes_finisher", "lanes_lambertian",
                                          "lanes_principled", "lanes_other", "drain (after last fetch)",
                                          "longest wave"};
    for (int k = 0; k < PF_COUNT; ++k)
      std::fprintf(stderr, "[vimg prof] %-24s %14llu  %6.2f %%\n", names[k], ds.prof[k],
                   100.0 * double(ds.prof[k]) / double(ds.prof[PF_TOTAL]));
  }
#endif
  return VIMG_OK;
}

}  // namespace

extern "C" {

const char* vimg_hip_last_error(void) { return g_err.c_str(); }

int vimg_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(VIMG_E_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

int vimg_hip_init(int device_ordinal) {
  HIP_TRY(hipSetDevice(device_ordinal));
  if (!g_stream) HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  g_device = device_ordinal;
  return VIMG_OK;
}

void vimg_hip_options_default(VimgHipOptions* o) {
  if (!o) return;
  int32_t* f = reinterpret_cast<int32_t*>(o);
  for (size_t i = 1; i < sizeof(VimgHipOptions) / sizeof(int32_t); ++i) f[i] = VIMG_OPT_AUTO;
  o->struct_size = sizeof(VimgHipOptions);
}

int vimg_hip_scene_upload(const VimgScene* sc, VimgDeviceScene** out) {
  return vimg_hip_scene_upload_opts(sc, nullptr, out);
}

int vimg_hip_scene_upload_opts(const VimgScene* sc, const VimgHipOptions* opts, VimgDeviceScene** out) {
  if (!out) return fail(VIMG_E_INVALID, "null output pointer");
  *out = nullptr;
  if (g_device < 0) {
    int rc = vimg_hip_init(0);
    if (rc) return rc;
  }
  int rc = validate(sc);
  if (rc) return rc;

  auto* s = new VimgDeviceScene();
  auto bail = [&](int code) {
    vimg_hip_scene_free(s);
    return code;
  };
  DScene& d = s->d;

  // ---- camera: TLCam ctor (reference src/tl_camera.cpp:6-23) and the primary ray cone
  // (include/ray.h:44-48) are per-render constants, evaluated here with the expressions the
  // reference uses (tan is an unqualified call there: double)
  const VimgCamera& cam = sc->camera;
  std::memcpy(d.cam_to_world, cam.cam_to_world, sizeof(d.cam_to_world));
  {
    float theta = (cam.vfov_deg * std::numbers::pi) / 180.0;
    float ratio = static_cast<float>(cam.res_x) / cam.res_y;
    float img_height = 2.0f * (::tan(static_cast<double>(theta / 2.0f)));
    d.p_size0 = ratio * img_height;
    d.p_size1 = img_height;
    float vfov = (cam.vfov_deg * std::numbers::pi) / 180.f;
    // std::atan / std::tan of floats, evaluated in double and rounded once (DESIGN.md Numerics)
    float t = static_cast<float>(::tan(static_cast<double>(vfov / 2.f)));
    d.cone_spread = static_cast<float>(
        ::atan(static_cast<double>(2.f * t / static_cast<float>(static_cast<uint32_t>(cam.res_y)))));
  }
  d.aperture_radius = cam.aperture_radius;
  d.focal_dist = cam.focal_dist;
  d.res_x = cam.res_x;
  d.res_y = cam.res_y;

  // ---- BVH: only internal nodes get a record; they are renumbered breadth-first (root = 0) so
  // that the lowest indices are the top of the tree — the part staged into LDS.  Traversal order
  // depends on the tree, not on the numbering, so results are unchanged.
  const VimgBVH& b = sc->bvh;
  // A child reference packs "count << 25 | first leaf slot": 7 bits of count.  A leaf of more than
  // 127 primitives (no builder of this repository makes one - theirs stop at 8 - but a caller's
  // builder may) becomes a CHAIN of extra records: the first 127 primitives as the RIGHT child, the
  // rest as the left one, both with the leaf's own box.  With equal boxes the walk takes the right
  // child first (closest hit: `h2 > h1` is false; any hit: second sibling first), so the
  // primitives are still tested in obj_indices order, and a chunk the shortened ray no longer
  // reaches holds no hit the reference could have accepted (its entry distance exceeds maxT).
  // Images are identical; the event counts gain the chain's node visits.
  std::vector<DNode> extra;          // chain records, appended behind the tree's own
  uint32_t extra_depth = 0;
  size_t n_internal = 0;             // set below, before the first chain is made
  auto leaf_ref = [&](const VimgBVHNode& n, const float* bmin, const float* bmax, uint32_t& out) {
    if (uint64_t(n.first_index) + n.obj_count > (1u << 25)) return false;
    uint32_t first = n.first_index, count = n.obj_count, links = 0;
    if (count <= 127u) {
      out = (count << 25) | first;
      return true;
    }
    // build the chain back to front: the last link's left child is the (<= 127) remainder
    std::vector<std::pair<uint32_t, uint32_t>> chunks;   // (first, count) in test order
    while (count > 127u) {
      chunks.push_back({first, 127u});
      first += 127u, count -= 127u;
    }
    uint32_t rest = (count << 25) | first;
    for (size_t i = chunks.size(); i-- > 0;) {
      DNode dn{};
      dn.a = v4f{bmin[0], bmin[1], bmin[2], bmax[0]};
      dn.b = v4f{bmax[1], bmax[2], bmin[0], bmin[1]};
      dn.c = v4f{bmin[2], bmax[0], bmax[1], bmax[2]};
      dn.left_ref = rest;
      dn.right_ref = (chunks[i].second << 25) | chunks[i].first;
      extra.push_back(dn);
      rest = static_cast<uint32_t>(n_internal + extra.size() - 1);
      ++links;
    }
    extra_depth = std::max(extra_depth, links);
    out = rest;
    return true;
  };
  std::vector<uint32_t> order;   // internal nodes: new index -> old index
  std::vector<uint32_t> new_of(b.num_nodes, 0);
  if (b.nodes[0].obj_count == 0) order.push_back(0);
  for (size_t head = 0; head < order.size(); ++head) {
    const VimgBVHNode& n = b.nodes[order[head]];
    for (uint32_t c = n.first_index; c <= n.first_index + 1; ++c)
      if (b.nodes[c].obj_count == 0) {
        new_of[c] = static_cast<uint32_t>(order.size());
        order.push_back(c);
      }
  }
  n_internal = order.size();
  std::vector<DNode> nodes(order.size());
  for (size_t i = 0; i < order.size(); ++i) {
    const VimgBVHNode& n = b.nodes[order[i]];
    DNode dn{};
    uint32_t refs[2];
    const float* bb = b.bb_mins_maxes + (size_t(n.first_index) * 2 + 2) * 3;
    const float* lmin = bb, *rmin = bb + 3, *lmax = bb + 6, *rmax = bb + 9;
    for (int k = 0; k < 2; ++k) {
      const uint32_t c = n.first_index + k;
      if (b.nodes[c].obj_count == 0) {
        refs[k] = new_of[c];
      } else if (!leaf_ref(b.nodes[c], k == 0 ? lmin : rmin, k == 0 ? lmax : rmax, refs[k])) {
        return bail(fail(VIMG_E_UNSUPPORTED, "BVH with more than 2^25 primitives"));
      }
    }
    dn.left_ref = refs[0];
    dn.right_ref = refs[1];
    dn.a = v4f{lmin[0], lmin[1], lmin[2], lmax[0]};
    dn.b = v4f{lmax[1], lmax[2], rmin[0], rmin[1]};
    dn.c = v4f{rmin[2], rmax[0], rmax[1], rmax[2]};
    nodes[i] = dn;
  }
  if (b.nodes[0].obj_count == 0) {
    d.root_ref = 0;
  } else if (!leaf_ref(b.nodes[0], b.bb_mins_maxes + 0, b.bb_mins_maxes + 6, d.root_ref)) {
    return bail(fail(VIMG_E_UNSUPPORTED, "BVH with more than 2^25 primitives"));
  }
  nodes.insert(nodes.end(), extra.begin(), extra.end());
  if (nodes.size() >= (1u << 25)) return bail(fail(VIMG_E_UNSUPPORTED, "BVH has more than 2^25 internal nodes"));
  for (int a = 0; a < 3; ++a) {
    d.root_min[a] = b.bb_mins_maxes[0 * 3 + a];
    d.root_max[a] = b.bb_mins_maxes[2 * 3 + a];
  }
  d.num_nodes = static_cast<uint32_t>(nodes.size());
  if (b.max_depth + extra_depth + 2 > 96) return bail(fail(VIMG_E_INVALID, "BVH (with its leaf chains) deeper than the 94-level stack bound"));
  d.max_depth = b.max_depth + extra_depth;   // a chain link pushes one entry like any internal node

  // ---- per-triangle shading records and leaf slots
  std::vector<DTriShade> shade(sc->num_tris);
  std::vector<float> area_pdf(sc->num_tris);
  for (uint32_t t = 0; t < sc->num_tris; ++t) {
    const VimgMesh& m = sc->meshes[sc->tri_mesh[t]];
    DTriShade ts{};
    ts.mesh = sc->tri_mesh[t];
    ts.i0 = m.first_vertex + sc->tri_indices[t * 3 + 0];
    ts.i1 = m.first_vertex + sc->tri_indices[t * 3 + 1];
    ts.i2 = m.first_vertex + sc->tri_indices[t * 3 + 2];
    const uint32_t ids[3] = {ts.i0, ts.i1, ts.i2};
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) ts.p[k * 3 + a] = sc->vertices[size_t(ids[k]) * 3 + a];
    {
      // tri_normal and the area pdf with the reference's float expressions
      // (src/geometry/triangle.cpp:19-25,229-231; glm cross / normalize as in device_math.h)
      const float* v = ts.p;
      const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]};
      const float e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
      const float c12[3] = {e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0],
                            e1[0] * e2[1] - e2[0] * e1[1]};
      const float inv_len = 1.0f / std::sqrt(c12[0] * c12[0] + c12[1] * c12[1] + c12[2] * c12[2]);
      for (int a = 0; a < 3; ++a) ts.n[a] = c12[a] * inv_len;
      const float c21[3] = {e2[1] * e1[2] - e1[1] * e2[2], e2[2] * e1[0] - e1[2] * e2[0],
                            e2[0] * e1[1] - e1[0] * e2[1]};
      const float area = std::sqrt(c21[0] * c21[0] + c21[1] * c21[1] + c21[2] * c21[2]) / 2.0f;
      area_pdf[t] = 1.f / area;
    }
    shade[t] = ts;
  }
  std::vector<DLeafPrim> leaf(sc->num_prims);
  for (uint32_t j = 0; j < sc->num_prims; ++j) {
    const uint32_t prim = b.obj_indices[j];
    const VimgPrim& p = sc->prims[prim];
    DLeafPrim lp{};
    lp.prim = prim;
    {
      const uint32_t mat = p.type == VIMG_PRIM_TRIANGLE ? sc->meshes[sc->tri_mesh[p.index]].material
                                                        : sc->spheres[p.index].material;
      const uint32_t t = sc->materials[mat].type;
      lp.cls = t == VIMG_MAT_DIFFUSE_LIGHT ? 0u : t == VIMG_MAT_LAMBERTIAN ? 1u : t == VIMG_MAT_PRINCIPLED ? 2u : 3u;
    }
    if (p.type == VIMG_PRIM_TRIANGLE) {
      const float* v = shade[p.index].p;
      lp.a = v4f{v[0], v[1], v[2], v[3]};
      lp.b = v4f{v[4], v[5], v[6], v[7]};
      lp.c0 = v[8];
      // the degenerate-triangle reject of the reference (triangle.h:86-92) depends on the
      // vertices only: evaluate it once, with the same float expression
      float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]};
      float e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
      float cx = e2[1] * e1[2] - e1[1] * e2[2];
      float cy = e2[2] * e1[0] - e1[2] * e2[0];
      float cz = e2[0] * e1[1] - e1[0] * e2[1];
      float l2 = cx * cx + cy * cy + cz * cz;
      lp.kind = (l2 == 0.f) ? 2u : 0u;
    } else {
      const VimgSphere& sp = sc->spheres[p.index];
      lp.a = v4f{sp.center[0], sp.center[1], sp.center[2], sp.radius};
      lp.kind = 1u;
    }
    leaf[j] = lp;
  }

  // ---- emitters, baked (device_scene.h: DLight)
  std::vector<DLight> dlights(sc->num_lights);
  for (uint32_t i = 0; i < sc->num_lights; ++i) {
    DLight L{};
    const VimgLight& l = sc->lights[i];
    if (l.type == VIMG_LIGHT_BACKGROUND) {
      L.kind = 0u;
    } else {
      const VimgPrim& p = sc->prims[l.prim];
      L.index = p.index;
      uint32_t mat;
      if (p.type == VIMG_PRIM_TRIANGLE) {
        const VimgMesh& mesh = sc->meshes[sc->tri_mesh[p.index]];
        const DTriShade& ts = shade[p.index];
        L.kind = mesh.has_normals ? 2u : 1u;
        L.a = v4f{ts.p[0], ts.p[1], ts.p[2], ts.p[3]};
        L.b = v4f{ts.p[4], ts.p[5], ts.p[6], ts.p[7]};
        L.c = v4f{ts.p[8], ts.n[0], ts.n[1], ts.n[2]};
        L.d.w = area_pdf[p.index];
        mat = mesh.material;
      } else {
        const VimgSphere& sp = sc->spheres[p.index];
        L.kind = 3u;
        L.a = v4f{sp.center[0], sp.center[1], sp.center[2], sp.radius};
        mat = sp.material;
      }
      const VimgMaterial& m = sc->materials[mat];
      if (m.type == VIMG_MAT_DIFFUSE_LIGHT) L.d.x = m.emit[0], L.d.y = m.emit[1], L.d.z = m.emit[2];   // (Material::emitted of the others: 0)
    }
    dlights[i] = L;
  }

  // ---- material flags / kernel variant
  std::vector<uint32_t> mflags(sc->num_materials, 0);
  bool textured = (sc->background.type == VIMG_BG_ENVMAP);
  for (uint32_t i = 0; i < sc->num_materials; ++i) {
    const VimgMaterial& m = sc->materials[i];
    uint32_t f = 0;
    if (m.type == VIMG_MAT_PRINCIPLED) f |= MATF_NEEDS_FRAME;
    if (m.tex >= 0 && sc->textures[m.tex].type != VIMG_TEX_CONST) f |= MATF_NEEDS_UV;
    if (m.tex >= 0 && sc->textures[m.tex].type == VIMG_TEX_IMAGE) textured = true;
    if (m.mr_tex >= 0 || m.normal_map >= 0) {
      f |= MATF_NEEDS_UV;
      textured = true;
    }
    mflags[i] = f;
  }
  s->textured = textured;

#define UP(field, host, count)                                                     \
  do {                                                                             \
    const std::remove_cv_t<std::remove_pointer_t<decltype(host)>>* p_ = nullptr;   \
    int rc_ = upload(s, host, count, &p_);                                         \
    if (rc_) return bail(rc_);                                                     \
    d.field = (decltype(d.field))p_;                                               \
  } while (0)
  UP(nodes, nodes.data(), nodes.size());
  UP(leaf_prims, leaf.data(), leaf.size());
  s->num_leaf_prims = static_cast<uint32_t>(leaf.size());
  UP(prims, sc->prims, sc->num_prims);
  UP(tri_shade, shade.data(), shade.size());
  UP(tri_area_pdf, area_pdf.data(), area_pdf.size());
  UP(meshes, sc->meshes, sc->num_meshes);
  UP(normals, sc->normals, size_t(sc->num_vertices) * 3);
  UP(uvs, sc->uvs, sc->num_uvs * 2);
  UP(spheres, sc->spheres, sc->num_spheres);
  UP(materials, sc->materials, sc->num_materials);
  UP(material_flags, mflags.data(), mflags.size());
  UP(textures, sc->textures, sc->num_textures);
  UP(texels, sc->texels, sc->num_texels * 3);
  UP(rg_textures, sc->rg_textures, sc->num_rg_textures);
  UP(rg_texels, sc->rg_texels, sc->num_rg_texels * 2);
  UP(lights, sc->lights, sc->num_lights);
  UP(dlights, dlights.data(), dlights.size());
  UP(cdf_pool, sc->cdf_pool, sc->num_cdf);
#undef UP
  d.num_lights = sc->num_lights;
  d.background = sc->background;
  // Background::is_emissive (reference include/background.h:51-56,176)
  d.background_emissive = (sc->background.type == VIMG_BG_ENVMAP) ||
                          !(sc->background.col[0] == 0.f && sc->background.col[1] == 0.f &&
                            sc->background.col[2] == 0.f);

  // LANE register budget: scenes beyond the on-chip caches are latency-bound and want more waves
  // per SIMD; small scenes are VALU-bound and want the build that spills least
  s->waves_per_simd = (s->total_bytes > (32u << 20)) ? 3 : 2;
  vimg_hip_options_default(&s->opt);
  if (opts) {
    // accept shorter (older) structs: fields beyond the caller's struct_size stay AUTO
    const size_t n = std::min<size_t>(opts->struct_size, sizeof(VimgHipOptions));
    if (n >= sizeof(uint32_t)) std::memcpy(&s->opt, opts, n);
    s->opt.struct_size = sizeof(VimgHipOptions);
  }
  options_from_env(&s->opt);
  if (s->opt.scheduler != VIMG_OPT_AUTO && (s->opt.scheduler < VIMG_SCHED_LANE || s->opt.scheduler > VIMG_SCHED_CU))
    return bail(fail(VIMG_E_INVALID, "options: unknown scheduler"));
  if (s->opt.scheduler != VIMG_OPT_AUTO && s->opt.scheduler != VIMG_SCHED_LANE && s->opt.scheduler != VIMG_SCHED_CU &&
      !vimg_has_dev_schedulers())
    return bail(fail(VIMG_E_UNSUPPORTED, "options: the schedulers POOL, POOL4, POOL4G and STAGE are reference implementations "
                                         "of the development build (make dev), not part of this library"));
  s->too_wide = (cam.res_x > 65535 || cam.res_y > 65535);   // slots pack pixel coordinates in 16 bits
  hipDeviceProp_t prop{};
  if (hipGetDeviceProperties(&prop, g_device) != hipSuccess) return bail(fail(VIMG_E_DEVICE, "hipGetDeviceProperties failed"));
  s->num_cus = static_cast<uint32_t>(prop.multiProcessorCount);
  if (hipMalloc(reinterpret_cast<void**>(&s->d_stats), sizeof(DeviceStats)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&s->d_counter), 2 * sizeof(unsigned int)) != hipSuccess)
    return bail(fail(VIMG_E_DEVICE, "hipMalloc of scratch failed"));
  if (hipMemset(s->d_counter, 0, 2 * sizeof(unsigned int)) != hipSuccess) return bail(fail(VIMG_E_DEVICE, "hipMemset of scratch failed"));
  *out = s;
  return VIMG_OK;
}

int vimg_hip_scene_free(VimgDeviceScene* s) {
  if (!s) return VIMG_OK;
  for (void* p : s->allocs) (void)hipFree(p);
  if (s->d_stats) (void)hipFree(s->d_stats);
  if (s->d_counter) (void)hipFree(s->d_counter);
  if (s->d_frame) (void)hipFree(s->d_frame);
  if (s->d_pool_cold) (void)hipFree(s->d_pool_cold);
  if (s->d_stack_ovf) (void)hipFree(s->d_stack_ovf);
  if (s->d_pool_state) (void)hipFree(s->d_pool_state);
  for (void* q : {s->d_stage_ctl, s->d_stage_kargs, s->d_stage_rings, s->d_stage_pix_ring, s->d_stage_pix_state, s->d_stage_slots})
    if (q) (void)hipFree(q);
  delete s;
  return VIMG_OK;
}

const char* vimg_hip_launch_kernel(const VimgDeviceScene* s, const VimgRenderParams* p) {
  if (!s || !p) return "";
  static const char* names[2][2][2] = {
      {{"render_kernel<false,2>", "render_kernel<false,3>"}, {"render_kernel<true,2>", "render_kernel<true,3>"}},
      {{"render_pool_kernel<false,2>", "render_pool_kernel<false,3>"},
       {"render_pool_kernel<true,2>", "render_pool_kernel<true,3>"}}};
  static const char* deep_names[2][2] = {
      {"render_pool_kernel<false,2,deep>", "render_pool_kernel<false,3,deep>"},
      {"render_pool_kernel<true,2,deep>", "render_pool_kernel<true,3,deep>"}};
  static const char* stage_names[2][2] = {{"render_stage_kernel<false>", "render_stage_kernel<false,deep>"},
                                          {"render_stage_kernel<true>", "render_stage_kernel<true,deep>"}};
  if (p->tile_world == 0 || p->tile_rank >= p->tile_world) return "";
  const LaunchCfg c = make_launch(s, p, -1, -1);
  static const char* pool4_names[2][2] = {{"render_pool4_kernel<false>", "render_pool4_kernel<false,deep>"},
                                          {"render_pool4_kernel<true>", "render_pool4_kernel<true,deep>"}};   // (+ waves per SIMD, rays per lane)
  static const char* cu_names[2][2] = {{"render_cu_kernel<false>", "render_cu_kernel<false,deep>"},
                                       {"render_cu_kernel<true>", "render_cu_kernel<true,deep>"}};
  if (c.sched == VIMG_SCHED_CU) return cu_names[s->textured ? 1 : 0][c.deep ? 1 : 0];
  if (c.sched == VIMG_SCHED_STAGE) return stage_names[s->textured ? 1 : 0][c.deep ? 1 : 0];
  static const char* pool4g_names[2][2] = {{"render_pool4_kernel<false,group>", "render_pool4_kernel<false,deep,group>"},
                                           {"render_pool4_kernel<true,group>", "render_pool4_kernel<true,deep,group>"}};
  if (c.sched == VIMG_SCHED_POOL4 && c.group) return pool4g_names[s->textured ? 1 : 0][c.deep ? 1 : 0];
  if (c.sched == VIMG_SCHED_POOL4) return pool4_names[s->textured ? 1 : 0][c.deep ? 1 : 0];
  if (c.deep) return deep_names[s->textured ? 1 : 0][c.wps >= 3 ? 1 : 0];
  return names[c.pooled ? 1 : 0][s->textured ? 1 : 0][c.wps >= 3 ? 1 : 0];
}

const char* vimg_hip_scene_kernel(const VimgDeviceScene* s) {
  // the scheduler is chosen per launch: report the one of a whole frame
  const VimgRenderParams whole{VIMG_INTEGRATOR_MIS, 64, 1, 0, 1};
  return vimg_hip_launch_kernel(s, &whole);
}

int64_t vimg_hip_scene_bytes(const VimgDeviceScene* s) {
  return s ? static_cast<int64_t>(s->total_bytes) : 0;
}

int64_t vimg_hip_shard_pixels(const VimgDeviceScene* s, const VimgRenderParams* p) {
  int rc = check_params(s, p);
  if (rc) return rc;
  return int64_t(local_tiles(s, p)) * 64;
}

int vimg_hip_render_async(VimgDeviceScene* s, const VimgRenderParams* p, void* d_out, void* stream) {
  int rc = check_params(s, p);
  if (rc) return rc;
  if (!d_out) return fail(VIMG_E_INVALID, "null output pointer");
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : g_stream;
  return launch_render(s, p, static_cast<float*>(d_out), st, false, false, -1, -1);
}

int vimg_hip_check(VimgDeviceScene* s) {
  if (!s) return fail(VIMG_E_INVALID, "null scene");
  return check_kernel_error(s);
}

int vimg_hip_render(VimgDeviceScene* s, const VimgRenderParams* p, void* d_out, void* stream,
                    VimgRenderStats* stats) {
  int rc = check_params(s, p);
  if (rc) return rc;
  if (!d_out) return fail(VIMG_E_INVALID, "null output pointer");
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : g_stream;
  rc = launch_render(s, p, static_cast<float*>(d_out), st, stats != nullptr, stats != nullptr, -1, -1);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  if (int rc2 = check_kernel_error(s)) return rc2;
  if (stats) return fetch_stats(s, p, stats);
  return VIMG_OK;
}

// heatmap_img (reference src/integrators/heatmap.cpp:38-147) behind the same boundary: the
// integrator field of the parameters is not used, samples and the tile shard are.
int vimg_hip_render_heatmap(VimgDeviceScene* s, const VimgRenderParams* p, float factor, void* d_out,
                            void* stream) {
  int rc = check_params(s, p);
  if (rc) return rc;
  if (!d_out) return fail(VIMG_E_INVALID, "null output pointer");
  if (factor <= 0) factor = 20.f;   // heatmap.cpp:137-139
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : g_stream;
  LaunchCfg c = make_launch(s, p, -1, -1, false);
  if (c.args.num_local_tiles == 0) return VIMG_OK;
  if (c.lds_bytes > 48u * 1024u)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(heatmap_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(c.lds_bytes)));
  const uint32_t grid = (c.args.num_local_tiles * 64u + 255u) / 256u;
  hipLaunchKernelGGL(heatmap_kernel, dim3(grid), dim3(256), c.lds_bytes, st, s->d, c.args, factor,
                     static_cast<float*>(d_out));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return VIMG_OK;
}

int vimg_hip_render_to_host(VimgDeviceScene* s, const VimgRenderParams* p, float* out_host,
                            VimgRenderStats* stats) {
  int rc = check_params(s, p);
  if (rc) return rc;
  if (!out_host) return fail(VIMG_E_INVALID, "null output pointer");
  if (p->tile_world != 1) return fail(VIMG_E_INVALID, "render_to_host needs tile_world == 1");
  const size_t floats = size_t(s->d.res_x) * s->d.res_y * 3;
  if (s->frame_floats < floats) {
    if (s->d_frame) (void)hipFree(s->d_frame);
    s->d_frame = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_frame), floats * sizeof(float)));
    s->frame_floats = floats;
  }
  rc = vimg_hip_render(s, p, s->d_frame, nullptr, stats);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(out_host, s->d_frame, floats * sizeof(float), hipMemcpyDeviceToHost));
  return VIMG_OK;
}

int vimg_hip_trace_pixel(VimgDeviceScene* s, const VimgRenderParams* p, int x, int y,
                         float* out_host) {
  int rc = check_params(s, p);
  if (rc) return rc;
  if (!out_host || x < 0 || y < 0 || x >= s->d.res_x || y >= s->d.res_y)
    return fail(VIMG_E_INVALID, "trace_pixel: pixel out of range");
  if (s->frame_floats < 3) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_frame), 3 * sizeof(float)));
    s->frame_floats = 3;
  }
  rc = launch_render(s, p, s->d_frame, g_stream, false, false, x, y);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g_stream));
  HIP_TRY(hipMemcpy(out_host, s->d_frame, 3 * sizeof(float), hipMemcpyDeviceToHost));
  return VIMG_OK;
}

int vimg_hip_assemble_shards(const VimgDeviceScene* s, uint32_t world, int64_t shard_stride_pixels,
                             const void* d_shards, void* d_out, void* stream) {
  if (!s || !d_shards || !d_out || world == 0) return fail(VIMG_E_INVALID, "assemble: bad arguments");
  const uint32_t tx = tiles_of(s->d.res_x), ty = tiles_of(s->d.res_y);
  const uint64_t max_local = (uint64_t(tx) * ty + world - 1) / world;
  if (shard_stride_pixels < int64_t(max_local * 64))
    return fail(VIMG_E_INVALID, "assemble: shard stride smaller than the largest shard");
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : g_stream;
  const uint32_t threads = tx * ty * 64;
  hipLaunchKernelGGL(assemble_kernel, dim3((threads + 255) / 256), dim3(256), 0, st,
                     static_cast<const float*>(d_shards), static_cast<float*>(d_out),
                     uint32_t(s->d.res_x), uint32_t(s->d.res_y), tx, ty, world,
                     static_cast<long long>(shard_stride_pixels));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return VIMG_OK;
}

int vimg_hip_time_renders(VimgDeviceScene* s, const VimgRenderParams* p, void* d_out, int steps,
                          float* ms_per_launch) {
  int rc = check_params(s, p);
  if (rc) return rc;
  if (!d_out || steps <= 0 || !ms_per_launch) return fail(VIMG_E_INVALID, "time_renders: bad arguments");
  std::vector<hipEvent_t> ev(size_t(steps) * 2);
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  for (int i = 0; i < steps; ++i) {
    // the counter / queue resets are part of a launch's prologue; the events bracket the kernel only
    if (int rc2 = enqueue_render(s, p, static_cast<float*>(d_out), g_stream, false, false, -1, -1, ev[2 * i], ev[2 * i + 1]))
      return rc2;
  }
  HIP_TRY(hipStreamSynchronize(g_stream));
  if (int rc2 = check_kernel_error(s)) return rc2;
  for (int i = 0; i < steps; ++i) HIP_TRY(hipEventElapsedTime(&ms_per_launch[i], ev[2 * i], ev[2 * i + 1]));
  for (auto& e : ev) (void)hipEventDestroy(e);
  return VIMG_OK;
}

int vimg_hip_post_rgb8(const void* d_rgb, int w, int h, int tonemapper, void* d_rgb8, void* stream) {
  if (!d_rgb || !d_rgb8 || w <= 0 || h <= 0 || tonemapper < 0 || tonemapper > 3)
    return fail(VIMG_E_INVALID, "post_rgb8: bad arguments");
  if (g_device < 0) {
    int rc = vimg_hip_init(0);
    if (rc) return rc;
  }
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : g_stream;
  const size_t n = size_t(w) * h;
  static unsigned int* d_max = nullptr;
  if (!d_max) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_max), sizeof(unsigned int)));
  if (tonemapper == 2) {
    HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(unsigned int), st));
    hipLaunchKernelGGL(post_max_luminance_kernel, dim3(1024), dim3(256), 0, st,
                       static_cast<const float*>(d_rgb), n, d_max);
  }
  hipLaunchKernelGGL(post_rgb8_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0,
                     st, static_cast<const float*>(d_rgb), n, tonemapper, d_max,
                     static_cast<unsigned char*>(d_rgb8));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return VIMG_OK;
}

// ---- the pre-step of the path on the GPU; host buffers in and out: these
// replace the host library's OpenMP loops while a scene is being assembled, before the upload
namespace {
struct PreBuf {   // device scratch freed on every exit path
  void* p = nullptr;
  ~PreBuf() { if (p) (void)hipFree(p); }
  float* f32() { return static_cast<float*>(p); }
  uint8_t* u8() { return static_cast<uint8_t*>(p); }
};
int pre_ready() {
  if (g_device < 0) return vimg_hip_init(0);
  return VIMG_OK;
}
uint32_t pre_grid(size_t n) { return static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, 65536)); }
}  // namespace

uint64_t vimg_hip_mip_chain_texels(uint32_t w, uint32_t h, uint32_t* num_levels) {
  if (w == 0 || h == 0) {
    if (num_levels) *num_levels = 0;
    return 0;
  }
  // level count of the reference: min(ceil(log2(min(w, h))), 15), never fewer than level 0
  const int levels = std::max(1, std::min(static_cast<int>(std::ceil(std::log2(static_cast<float>(std::min(w, h))))),
                                          VIMG_MAX_MIP_LEVELS));
  uint64_t total = 0;
  uint32_t lw = w, lh = h;
  for (int l = 0; l < levels; ++l) {
    total += uint64_t(lw) * lh;
    lw = std::max(lw / 2u, 1u), lh = std::max(lh / 2u, 1u);
  }
  if (num_levels) *num_levels = static_cast<uint32_t>(levels);
  return total;
}

int vimg_hip_build_mip_chain(uint32_t w, uint32_t h, const float* level0, uint32_t wrap_u,
                             uint32_t wrap_v, float* out_levels) {
  if (!level0 || !out_levels || w == 0 || h == 0 || wrap_u > 2 || wrap_v > 2)
    return fail(VIMG_E_INVALID, "build_mip_chain: bad arguments");
  if (int rc = pre_ready()) return rc;
  uint32_t levels = 0;
  const uint64_t texels = vimg_hip_mip_chain_texels(w, h, &levels);
  PreBuf d;
  HIP_TRY(hipMalloc(&d.p, texels * 3 * sizeof(float)));
  float* base = d.f32();
  HIP_TRY(hipMemcpyAsync(base, level0, size_t(w) * h * 3 * sizeof(float), hipMemcpyHostToDevice, g_stream));
  uint64_t prev_off = 0;
  uint32_t pw = w, ph = h;
  for (uint32_t l = 1; l < levels; ++l) {
    const uint32_t nw = std::max(pw / 2u, 1u), nh = std::max(ph / 2u, 1u);
    const uint64_t next_off = prev_off + uint64_t(pw) * ph;
    hipLaunchKernelGGL(pre_mip_level_kernel, dim3((nw + 31) / 32, (nh + 7) / 8), dim3(256), 0, g_stream,
                       base + prev_off * 3, pw, ph, base + next_off * 3, nw, nh, wrap_u, wrap_v);
    prev_off = next_off;
    pw = nw, ph = nh;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out_levels, base, texels * 3 * sizeof(float), hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));
  return VIMG_OK;
}

int vimg_hip_build_env_cdfs(const float* img, uint32_t w, uint32_t h, float* row_cdf, float* col_cdfs) {
  if (!img || !row_cdf || !col_cdfs || w == 0 || h == 0)
    return fail(VIMG_E_INVALID, "build_env_cdfs: bad arguments");
  if (int rc = pre_ready()) return rc;
  // sin(pi * v) per row, in double as the reference evaluates it (sampling.h:180-181)
  std::vector<float> sin_elev(h);
  for (uint32_t y = 0; y < h; ++y) {
    float v = (static_cast<float>(y) + 0.5f) / static_cast<float>(h);
    sin_elev[y] = static_cast<float>(std::sin(3.141592653589793238462643383279502884 * v));
  }
  const size_t n = size_t(w) * h;
  PreBuf d_img, d_sin, d_lum, d_cdf, d_rowint, d_rowcdf, d_rowtot;
  HIP_TRY(hipMalloc(&d_img.p, n * 3 * sizeof(float)));
  HIP_TRY(hipMalloc(&d_sin.p, h * sizeof(float)));
  HIP_TRY(hipMalloc(&d_lum.p, n * sizeof(float)));
  HIP_TRY(hipMalloc(&d_cdf.p, size_t(h) * (w + 1) * sizeof(float)));
  HIP_TRY(hipMalloc(&d_rowint.p, h * sizeof(float)));
  HIP_TRY(hipMalloc(&d_rowcdf.p, (size_t(h) + 1) * sizeof(float)));
  HIP_TRY(hipMalloc(&d_rowtot.p, sizeof(float)));
  HIP_TRY(hipMemcpyAsync(d_img.p, img, n * 3 * sizeof(float), hipMemcpyHostToDevice, g_stream));
  HIP_TRY(hipMemcpyAsync(d_sin.p, sin_elev.data(), h * sizeof(float), hipMemcpyHostToDevice, g_stream));
  hipLaunchKernelGGL(pre_env_lum_kernel, dim3(pre_grid(n)), dim3(256), 0, g_stream, d_img.f32(), w, h,
                     d_sin.f32(), d_lum.f32());
  // one conditional distribution per image row, then the marginal over the row integrals
  hipLaunchKernelGGL(pre_cdf_scan_kernel, dim3(h), dim3(64), 0, g_stream, d_lum.f32(), h, w,
                     d_cdf.f32(), d_rowint.f32());
  hipLaunchKernelGGL(pre_cdf_normalise_kernel, dim3(pre_grid(size_t(h) * (w + 1))), dim3(256), 0, g_stream,
                     d_cdf.f32(), h, w, d_rowint.f32());
  hipLaunchKernelGGL(pre_cdf_scan_kernel, dim3(1), dim3(64), 0, g_stream, d_rowint.f32(), 1u, h,
                     d_rowcdf.f32(), d_rowtot.f32());
  hipLaunchKernelGGL(pre_cdf_normalise_kernel, dim3(pre_grid(size_t(h) + 1)), dim3(256), 0, g_stream,
                     d_rowcdf.f32(), 1u, h, d_rowtot.f32());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(col_cdfs, d_cdf.p, size_t(h) * (w + 1) * sizeof(float), hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipMemcpyAsync(row_cdf, d_rowcdf.p, (size_t(h) + 1) * sizeof(float), hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));
  return VIMG_OK;
}

int vimg_hip_lut8_to_float(const uint8_t* in, uint64_t n, const float* lut256, float* out) {
  if (!in || !lut256 || !out) return fail(VIMG_E_INVALID, "lut8_to_float: bad arguments");
  if (n == 0) return VIMG_OK;
  if (int rc = pre_ready()) return rc;
  PreBuf d_in, d_lut, d_out;
  HIP_TRY(hipMalloc(&d_in.p, n));
  HIP_TRY(hipMalloc(&d_lut.p, 256 * sizeof(float)));
  HIP_TRY(hipMalloc(&d_out.p, n * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(d_in.p, in, n, hipMemcpyHostToDevice, g_stream));
  HIP_TRY(hipMemcpyAsync(d_lut.p, lut256, 256 * sizeof(float), hipMemcpyHostToDevice, g_stream));
  hipLaunchKernelGGL(pre_lut8_kernel, dim3(pre_grid(n)), dim3(256), 0, g_stream, d_in.u8(), size_t(n),
                     d_lut.f32(), d_out.f32());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_out.p, n * sizeof(float), hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));
  return VIMG_OK;
}

int vimg_hip_rgb8_to_normal(const uint8_t* rgb8, uint64_t n_pixels, float scale, float* out) {
  if (!rgb8 || !out) return fail(VIMG_E_INVALID, "rgb8_to_normal: bad arguments");
  if (n_pixels == 0) return VIMG_OK;
  if (int rc = pre_ready()) return rc;
  PreBuf d_in, d_out;
  HIP_TRY(hipMalloc(&d_in.p, n_pixels * 3));
  HIP_TRY(hipMalloc(&d_out.p, n_pixels * 3 * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(d_in.p, rgb8, n_pixels * 3, hipMemcpyHostToDevice, g_stream));
  hipLaunchKernelGGL(pre_normal8_kernel, dim3(pre_grid(n_pixels)), dim3(256), 0, g_stream, d_in.u8(),
                     size_t(n_pixels), scale, d_out.f32());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_out.p, n_pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));
  return VIMG_OK;
}

// Unit-level probe (declared here, not in vimg_hip.h: it is a test hook, not part of the seam).
int vimg_hip_probe(VimgDeviceScene* s, int kind, int n, const float* in_host, float* out_host) {
  static const int n_in[9] = {0, 4, 6, 7, 12, 8, 4, 5, 1};
  static const int n_out[9] = {0, 8, 28, 1, 5, 7, 10, 4, 5};
  if (!s || kind < 1 || kind > 8 || n <= 0 || !in_host || !out_host)
    return fail(VIMG_E_INVALID, "probe: bad arguments");
  float *d_in = nullptr, *d_out = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_in), size_t(n) * n_in[kind] * sizeof(float)));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), size_t(n) * n_out[kind] * sizeof(float)));
  HIP_TRY(hipMemcpy(d_in, in_host, size_t(n) * n_in[kind] * sizeof(float), hipMemcpyHostToDevice));
  VimgRenderParams p{VIMG_INTEGRATOR_MIS, 1, 1, 0, 1};
  LaunchCfg c = make_launch(s, &p, -1, -1, false);
  const uint32_t grid = (uint32_t(n) + 255) / 256;
  if (s->textured)
    hipLaunchKernelGGL(probe_kernel<true>, dim3(grid), dim3(256), c.lds_bytes, g_stream, s->d, c.args,
                       kind, n, d_in, d_out);
  else
    hipLaunchKernelGGL(probe_kernel<false>, dim3(grid), dim3(256), c.lds_bytes, g_stream, s->d, c.args,
                       kind, n, d_in, d_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g_stream));
  HIP_TRY(hipMemcpy(out_host, d_out, size_t(n) * n_out[kind] * sizeof(float), hipMemcpyDeviceToHost));
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return VIMG_OK;
}

}  // extern "C"
