// Pooled scheduler, second build: the same pools of path slots, the same queues, the same segment
// hand-over as render_pool_kernel (see its header comment) - but the VERTEX stage is a real
// (non-inlined) function per batch class, the kernel is built for three and for four waves per SIMD,
// and (GRP builds) the four waves of a workgroup share one pool and one set of queues.
//
// Why calls (profiles/r1_flat, ISA of render_pool_kernel): with everything inlined the register
// allocator keeps the walk's persistent lane state, the batch's slot state and the shading
// temporaries alive together - 255 VGPRs + 336 B of scratch at two waves per SIMD, and a SIMD with
// two waves issues a vector instruction at best every other slot each (38 % issue measured).
// Compiled on their own the vertex stages need 74 (Lambertian) to 126 (Principled) registers.
// As calls they get a fresh register file: the walk's lane state sits in callee-saved registers,
// the wave-uniform scheduler state (queue heads and counts) travels through a 64-byte record in
// LDS, scene and launch parameters are read from one block in device memory through the constant
// address space (a callee has no kernel-argument pointer).  The pools shrink with the LDS share of
// a wave (the RNG record moves to the cold records in global memory to win some of it back).
//
// Why group pools: the rate of this scheduler is passes x lanes per pass in
// the walk and batches x slots per batch in the vertex stage, and both fills hang on the slots a
// wave can draw from.  Same arithmetic, same order in every build: bit-identical.
#pragma once
#include <type_traits>

#include "render_pool_kernel.h"

namespace vimg {

// block of scene + launch parameters in device memory (written by pool4_args_kernel before the launch)
struct Pool4KArgs {
  DScene g;
  RenderArgs A;
  float* out;
  DeviceStats* stats;
  unsigned int* work_counter;
};
typedef const __attribute__((address_space(4))) Pool4KArgs* Pool4KPtr;
VD Pool4KPtr pool4_kargs(uint32_t lo, uint32_t hi) {
  const unsigned long long a =
      static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lo)))) |
      (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(hi)))) << 32);
  return (Pool4KPtr)a;
}
static __global__ void pool4_args_kernel(const Pool4KArgs ka, Pool4KArgs* __restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = ka;
}

constexpr uint32_t P4_HOT = 3u;                      // hot records of a slot in LDS: SR_ORIGIN, SR_RAY, SR_SHADOW
// wave-uniform scheduler state, one record per wave in LDS: what the kernel's main loop and the
// vertex-stage calls both read and write
struct Pool4Wave {
  uint32_t qw_head, qw_count;
  uint32_t qv_head[4], qv_count[4];
  uint32_t pixels_left, skip_fin, idle_polls, nan_samples;
  uint32_t pad[2];
};
static_assert(sizeof(Pool4Wave) == 64, "Pool4Wave is one 64-byte record");
// GRP builds: the four waves of a workgroup share ONE pool of 4 x pool_slots slots and one set of
// queues (16-bit slot ids).  Every queue operation of a wave - taking rays, handing finished ones to
// the vertex queues, taking a batch, handing its slots back - runs under the group's lock; the
// record below holds what the per-wave record holds in the other builds.
struct Pool4Group {
  uint32_t lock, live, pixels_left, abort;   // live: slots that have not retired
  uint32_t qw_head, qw_count;
  uint32_t qv_head[4], qv_count[4];
  uint32_t pad[2];
};
static_assert(sizeof(Pool4Group) == 64, "Pool4Group is one 64-byte record");
constexpr uint32_t P4G_LDS_BYTES = 3u * 16u + 4u + 5u * 2u;   // hot records, primitive id, five 16-bit rings
VD uint32_t lds_aload(VIMG_LDS uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// Lock of the group's queue state: lane 0 spins (with a bound: a wave that cannot get it within
// seconds raises the launch's error word and the group's abort flag instead of hanging the GPU).
VD void grp_lock(VIMG_LDS Pool4Group* G, unsigned int* err_word) {
  if ((threadIdx.x & 63u) == 0u) {
    uint32_t spins = 0;
    for (;;) {
      uint32_t expect = 0u;
      if (__hip_atomic_compare_exchange_strong(&G->lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP))
        break;
      if (++spins > (1u << 22)) {
        atomicOr(err_word, 2u);
        __hip_atomic_store(&G->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
VD void grp_unlock(VIMG_LDS Pool4Group* G) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  // (a wave whose grp_lock gave up holds no lock: once the group is aborting nobody stores 0 over
  // another wave's lock; the frame is lost anyway and every loop leaves at its next abort test)
  if ((threadIdx.x & 63u) == 0u && lds_aload(&G->abort) == 0u)
    __hip_atomic_store(&G->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// where the shared tables of a group live behind its slot records (pool = first record, PS slots)
struct Pool4GLayout {
  VIMG_LDS uint32_t* q_prim;
  VIMG_LDS uint16_t* q_walk;
  VIMG_LDS uint16_t* q_vertex;   // four rings of capacity PS
  VIMG_LDS Pool4Group* G;
  VIMG_LDS uint16_t* bslots;     // [wave][64]: the slots of the batch a wave has taken
};
VD Pool4GLayout pool4g_layout(VIMG_LDS uint32_t* pool, uint32_t PS) {
  Pool4GLayout l;
  l.q_prim = pool + P4_HOT * 4u * PS;
  l.q_walk = reinterpret_cast<VIMG_LDS uint16_t*>(l.q_prim + PS);
  l.q_vertex = l.q_walk + PS;
  l.G = reinterpret_cast<VIMG_LDS Pool4Group*>(l.q_vertex + 4u * PS);   // PS is a multiple of 8: 16-byte aligned
  l.bslots = reinterpret_cast<VIMG_LDS uint16_t*>(l.G + 1);
  return l;
}
__host__ __device__ constexpr uint32_t pool4g_group_bytes(uint32_t slots_per_wave) {
  return P4G_LDS_BYTES * 4u * slots_per_wave + 64u + 4u * 64u * 2u;
}
// diagnostics of full-stats launches (VIMG_HIP_DIAG prints them): cycles in vertex calls by class
// (0 finisher, 1 Lambertian, 2 Principled, 3 other), in the walk stage (4) and idle (5); batches and
// slots per class
struct Pool4Diag {
  unsigned long long cyc[6], nbatch[4], nslots[4];
};
// Slot records of this build: the three records the WALK reads and writes stay in LDS (origin, path
// ray + flags, shadow ray / hit barycentrics); the RNG record joins the cold ones in global memory
// (one more 16-byte read and write per vertex batch and slot; 57 instead of 73 LDS bytes per slot
// = 28 % more slots, and the slot count is what this scheduler's rate hangs on: 64 slots 6.1,
// 80: 8.1, 102: 9.7 Grays/s on config 2 at four waves per SIMD).
constexpr uint32_t P4_LDS_BYTES = P4_HOT * 16u + 4u + 5u;
VD uint32_t pool4_wave_bytes(uint32_t slots) { return (P4_LDS_BYTES * slots + 15u) & ~15u; }
VD uint32_t uni(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }

// One vertex batch (<= 64 slots) of queue `cls`.  FIN: the finisher queue (cls 0); MT: material the
// shading is specialised for (-1 = any).  The body is render_pool_kernel's vertex stage.
// (WPS: the register budget of the kernel that calls it - a copy per budget, so that the three-wave
// build's callees get its 168 registers: 104 of them are caller-saved, and a callee that stays
// within those saves nothing - 8 to 29 saved registers per call instead of 32 to 48)
// GRP: the slots and queues are the workgroup's (Pool4Group); `n_in` slots of the batch wait in the
// wave's row of `bslots`.
template <bool TEX, bool FIN, int MTC, int WPS, bool GRP>
__device__ __noinline__ void pool4_vertex(uint32_t k_lo, uint32_t k_hi, VIMG_LDS uint32_t* pool,
                                          VIMG_LDS Pool4Wave* pw, uint32_t cls, uint32_t n_in) {
  const Pool4KPtr K = pool4_kargs(k_lo, k_hi);
  const DScene& g = *(const DScene*)&K->g;
  const RenderArgs& A = *(const RenderArgs*)&K->A;
  float* __restrict__ out = K->out;
  unsigned int* __restrict__ work_counter = K->work_counter;
  cls = uni(cls);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t W = static_cast<uint32_t>(g.res_x), H = static_cast<uint32_t>(g.res_y);
  const bool single = A.single_x >= 0;
  const uint32_t total_items = single ? 1u : A.num_local_tiles * 64u;
  const uint32_t n_seg = A.pool_segments, seg_len = A.pool_seg_len;
  const uint32_t total_claims = total_items * n_seg;
  constexpr uint32_t roulette_threshold = 5;
  const bool material_mode = (A.integrator == VIMG_INTEGRATOR_MATERIAL);
  const uint32_t P = GRP ? 4u * A.pool_slots : A.pool_slots;   // slots of the pool this batch belongs to
  // cold region of this wave (GRP: of this workgroup): [slot][4] main lines, then the accumulator
  // plane, then the cone plane (P is even, so every region starts on a 64-byte boundary)
  VIMG_GLOBAL v4u* cold = A.pool_cold + (GRP ? size_t(blockIdx.x) : size_t(blockIdx.x) * 4u + wave) *
                                            (size_t(pool4_cold_records(TEX)) * P);
  VIMG_GLOBAL v4u* cold_acc = cold + size_t(SC4_MAIN) * P;
  [[maybe_unused]] VIMG_GLOBAL v4u* cold_cone = cold_acc + P;
  auto crd = [&](uint32_t r, uint32_t slot) -> v4u { return cold[slot * SC4_MAIN + r]; };
  auto cwr = [&](uint32_t r, uint32_t slot, v4u v) { cold[slot * SC4_MAIN + r] = v; };
  using QId = std::conditional_t<GRP, uint16_t, uint8_t>;
  VIMG_LDS uint32_t* q_prim = pool + P4_HOT * 4u * P;
  VIMG_LDS QId* q_walk = reinterpret_cast<VIMG_LDS QId*>(q_prim + P);
  VIMG_LDS QId* q_vertex = q_walk + P;
  [[maybe_unused]] VIMG_LDS Pool4Group* G = reinterpret_cast<VIMG_LDS Pool4Group*>(q_vertex + 4u * P);   // (GRP layout)
  [[maybe_unused]] VIMG_LDS uint16_t* bslots = reinterpret_cast<VIMG_LDS uint16_t*>(G + 1) + wave * 64u;
  VIMG_LDS v4u* recs = reinterpret_cast<VIMG_LDS v4u*>(pool);
  auto rd = [&](uint32_t r, uint32_t slot) -> v4u { return recs[r * P + slot]; };
  auto wr = [&](uint32_t r, uint32_t slot, v4u v) { recs[r * P + slot] = v; };
  auto word = [&](uint32_t r, uint32_t k, uint32_t slot) -> VIMG_LDS uint32_t& {
    return pool[(r * P + slot) * 4u + k];
  };
  auto fu = [](float f) { return __float_as_uint(f); };
  auto uf = [](uint32_t u) { return __uint_as_float(u); };
  auto ring = [&](uint32_t i) { return i >= P ? i - P : i; };
  constexpr bool finisher_batch = FIN;

  // scheduler state of the wave
  uint32_t qw_head = GRP ? 0u : uni(pw->qw_head), qw_count = GRP ? 0u : uni(pw->qw_count);
  uint32_t qv_head0 = GRP ? 0u : uni(pw->qv_head[0]), qv_count0 = GRP ? 0u : uni(pw->qv_count[0]);
  bool pixels_left = (GRP ? uni(lds_aload(&G->pixels_left)) : uni(pw->pixels_left)) != 0u, skip_fin = uni(pw->skip_fin) != 0u;
  uint32_t idle_polls = uni(pw->idle_polls);
  uint32_t nan_here = 0;
  uint32_t n, slot;
  bool on;
  if constexpr (GRP) {
    n = uni(n_in);
    on = lane < n;
    slot = on ? bslots[lane] : 0u;
  } else {
    const uint32_t qv_count = uni(pw->qv_count[cls]), qv_head = uni(pw->qv_head[cls]);
    n = qv_count < 64u ? qv_count : 64u;
    on = lane < n;
    slot = on ? q_vertex[cls * P + ring(qv_head + lane)] : 0u;
    if (cls == 0u) {
      qv_head0 = ring(qv_head0 + n), qv_count0 -= n;
    } else if (lane == 0) {
      pw->qv_head[cls] = ring(qv_head + n), pw->qv_count[cls] = qv_count - n;
    }
  }

  const v4u r_ray = on ? rd(SR_RAY, slot) : v4u{0u, 0u, 0u, 0u};
  uint32_t flags = r_ray.w;
  const bool fresh = on && (flags & SF_FRESH);
  const bool have = on && !fresh;
  // slot state -> registers
  uint32_t px = 0, py = 0, smp = 0, item = 0, bounce = 0;
  Rng rng{0};
  f3 acc{0.f, 0.f, 0.f}, ray_o{0.f, 0.f, 0.f}, ray_d{0.f, 0.f, 1.f};
  f3 throughput{1.f, 1.f, 1.f}, result{0.f, 0.f, 0.f};
  RayCone cone{0.f, 0.f};
  float eta_scale = 1.f, prev_pdf = 0.f;
  bool primary = true, non_specular_bounce = false;
  v4u r_origin{0u, 0u, 0u, 0u}, r_shadow{0u, 0u, 0u, 0u}, r_nee{0u, 0u, 0u, 0u};
  uint32_t hit_prim = 0;
  if (have) {
    // the pixel accumulator is read and written by finisher batches only
    const v4u r_t = crd(SC_THROUGHPUT, slot), r_r = crd(SC_RESULT, slot);
    v4u r_a{0u, 0u, 0u, 0u};
    if (finisher_batch) r_a = cold_acc[slot];
    if (r_ray.w & SF_HAS_S) r_nee = crd(SC_NEE, slot);
    r_origin = rd(SR_ORIGIN, slot);
    r_shadow = rd(SR_SHADOW, slot);
    const v4u r_g = crd(SC4_RNG, slot);
    hit_prim = q_prim[slot];
    px = r_g.z & 0xffffu, py = r_g.z >> 16;
    smp = r_g.w;
    item = r_a.w;
    rng.s = uint64_t(r_g.x) | (uint64_t(r_g.y) << 32);
    acc = f3{uf(r_a.x), uf(r_a.y), uf(r_a.z)};
    ray_o = f3{uf(r_origin.x), uf(r_origin.y), uf(r_origin.z)};
    ray_d = f3{uf(r_ray.x), uf(r_ray.y), uf(r_ray.z)};
    throughput = f3{uf(r_t.x), uf(r_t.y), uf(r_t.z)};
    result = f3{uf(r_r.x), uf(r_r.y), uf(r_r.z)};
    eta_scale = uf(r_t.w);
    prev_pdf = uf(r_r.w);
    bounce = flags >> SF_BOUNCE_SHIFT;
    primary = (flags & SF_PRIMARY) != 0;
    non_specular_bounce = (flags & SF_NONSPEC) != 0;
    if constexpr (TEX) {
      const v4u r_c = cold_cone[slot];
      cone = RayCone{uf(r_c.x), uf(r_c.y)};
    }
  }

  bool finish = false, at_vertex = false;
  Hit hit;
  hit.p = f3{0.f, 0.f, 0.f};
  if (have) {
    // next-event estimation of the previous vertex (mis_integrator.cpp:64-78)
    if ((flags & SF_HAS_S) && !(flags & SF_OCCLUDED))
      result = result + f3{uf(r_nee.x), uf(r_nee.y), uf(r_nee.z)};
    if (!(flags & SF_HAS_R)) {
      finish = true;   // the BSDF sample failed there: return bounce_result (:86-88,:108-114)
    } else {
      const bool hit_any = (flags & SF_FOUND) != 0;
      if (hit_any) {
        HitRec hr;
        hr.e0 = uf(r_shadow.x), hr.e1 = uf(r_shadow.y), hr.e2 = uf(r_shadow.z);
        hr.inv_det = uf(r_shadow.w);
        hr.prim = hit_prim;
        hr.kind = (flags & SF_KIND_SPHERE) ? 1u : 0u;
        TravRay tr{ray_o, ray_d, 0.0001f, uf(r_origin.w)};
        make_hit_info<TEX>(g, hr, tr, hit);
      }
      if (material_mode) {
        // material_integrator (mat_integrator.cpp:16-23,79-81)
        if (!hit_any) {
          result = throughput * background_emit<TEX>(g, ray_d, cone);
          finish = true;
        } else {
          at_vertex = true;
        }
      } else if (A.integrator != VIMG_INTEGRATOR_MIS) {
        // shading_normal_integrator / geometric_normal_integrator
        if (hit_any) {
          f3 nn = (A.integrator == VIMG_INTEGRATOR_G_NORMAL) ? hit.ng : hit.ns;
          result = (nn + 1.0f) / 2.0f;
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
              float rr = static_cast<float>(pcg_next(rng)) / 4294967296.0f;
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
  }


  // ---- the next rays of a vertex
  bool has_s = false, has_r = false;
  f3 shadow_d{0.f, 0.f, 1.f}, nee_contrib{0.f, 0.f, 0.f};
  float shadow_max_t = 0.f;
  // (a finisher batch never holds a vertex to shade: the walk sends every hit on a non-emitter to
  // its material's class - class 3 for everything under the material integrator - so the finisher
  // build carries no shading code at all)
  if constexpr (!finisher_batch) {
  if (material_mode && at_vertex) {
    // mat_integrator.cpp:24-78: BSDF sampling only, throughput *= emitted + eval/pdf
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
        result = f3{0.f, 0.f, 0.f};
        finish = true;
      } else {
        ray_o = hit.p;
        ray_d = sc.wo;
        primary = false;
        has_r = true;
      }
    }
    at_vertex = false;
  }
  }
  // A batch of class 1 holds Lambertian vertices only and one of class 2 Principled ones only
  // (three-class sorting), so the shading code exists in a build per material with the type
  // dispatch folded away - the Lambertian build carries none of the Disney lobes' registers -
  // and a generic build for everything else.
  auto shade_vertex = [&](auto mt_tag) {
    constexpr int MT = decltype(mt_tag)::value;
    // mis_integrator.cpp:45-122.  Draw order: light pick + emitter sample, then sample_mat.
    const uint32_t mat_type = MT >= 0 ? uint32_t(MT) : g.materials[hit.mat].type;
    float hit_dist = 0.f, surface_spread_angle = 0.f;
    if constexpr (TEX) {
      hit_dist = length(ray_o - hit.p);
      surface_spread_angle =
          spread_angle_from_curvature(hit.curvature, cone.cone_width, ray_d, hit.ns);
    }
    f3 light_col{0.f, 0.f, 0.f};
    EmitterInfo li{f3{0.f, 0.f, 1.f}, 0.f, 0.f, 0.f};
    bool nee = false;
    if (mat_type != VIMG_MAT_DIELECTRIC) {   // !is_delta
      lights_sample<TEX>(g, hit.p, rng, light_col, li);
      nee = (li.pdf != 0.f);
    }

    const bool reg_before = non_specular_bounce;
    RayCone nee_cone = cone;
    Scatter sc = sample_mat<TEX, MT>(g, hit, ray_d, rng, reg_before);

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
    // both BSDF evaluations happen before either ray is traced: the evaluation towards the
    // light is pure, so doing it for a light that turns out occluded changes nothing; its
    // regularisation flag is the one from BEFORE this bounce
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool run = (k == 0) ? nee : sc.valid;
      if (run) {
        f3 f{0.f, 0.f, 0.f};
        float pdf = 0.f;
        const f3 wo = (k == 0) ? li.wi : sc.wo;
        const RayCone c = (k == 0) ? nee_cone : cone;
        const bool reg = (k == 0) ? reg_before : non_specular_bounce;
        eval_pdf_pair<TEX, MT>(g, hit, ray_d, wo, c, reg, f, pdf);
        if (k == 0) {
          if (pdf != 0 && !is_nan(pdf)) {
            float G = li.G;
            float mis_weight = balance_heuristic(li.pdf, pdf * G);
            nee_contrib = throughput * f * mis_weight * G * light_col / li.pdf;
          }
          // pdf == 0 / NaN: nothing is added, but the reference has traced its shadow ray by
          // then (mis_integrator.cpp:64): it is still traced and counted
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

    has_s = nee;
    has_r = sc.valid;
    shadow_d = li.wi;
    shadow_max_t = li.dist - 0.0001f;   // absolute epsilon of the reference (quirk Q15)
    ray_o = hit.p;
    ray_d = sc.wo;
    primary = false;
    if (!has_s && !has_r) finish = true;
  };
  if constexpr (!finisher_batch) {
    if (at_vertex) shade_vertex(std::integral_constant<int, MTC>{});
  }

  // ---- finished samples: accumulate, pixel write-back, next pixel, next camera ray
  bool need_pixel = fresh;
  bool retire = false;
  // a fresh slot may already hold a claim whose predecessor segment was not published yet
  bool have_claim = fresh && (flags & SF_PRIMARY);
  uint32_t claim = have_claim ? crd(SC4_RNG, slot).w : 0u;
  bool pending = false;
  if (!finisher_batch) {
    // a path that ended at this vertex (roulette, depth limit, no ray left) is accumulated by
    // the finisher stage: it travels there with neither ray set, which that stage reads as
    // "return bounce_result" (the !SF_HAS_R branch above)
    if (finish) has_s = false, has_r = false;
    skip_fin = false;
    idle_polls = 0;
  } else {
  if (finish) {
    if (is_nan(result.x) || is_nan(result.y) || is_nan(result.z)) nan_here = 1;
    acc = acc + result;
    smp += 1;
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
    } else if (n_seg > 1u && smp % seg_len == 0u) {
      // end of a segment: the pixel rests in its record until a slot draws its next segment
      // Every word of the record is read and written with agent-scope atomics (sc1: coherent
      // across the eight XCDs' L2s on their own), so publishing needs no L2 write-back and
      // picking up no L2 invalidate - with release / acquire fences at agent scope the cold
      // slot records would be flushed out of the L2 on every segment end.  Order: data words,
      // wait until they are acknowledged, then the tag.
      VIMG_GLOBAL uint32_t* st = reinterpret_cast<VIMG_GLOBAL uint32_t*>(A.pool_state + size_t(item) * 2u);
      state_store(st + 0, static_cast<uint32_t>(rng.s));
      state_store(st + 1, static_cast<uint32_t>(rng.s >> 32));
      state_store(st + 4, fu(acc.x));
      state_store(st + 5, fu(acc.y));
      state_store(st + 6, fu(acc.z));
      // the data words must have been performed before the tag is: wait for the wave's
      // outstanding vector stores (a workgroup-scope release fence compiles to a wait on the
      // LDS / scalar counter only), then store the tag
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt and lgkmcnt untouched
      state_store(st + 2, A.pool_epoch + smp / seg_len);
      need_pixel = true;
    }
  }
  // work fetch: repeated while some lane drew an off-image slot of a ragged tile
  while (__any(need_pixel && !pending)) {
    const bool want = need_pixel && !pending && !have_claim;
    const unsigned long long mask = __ballot(want);
    uint32_t base = 0;
    if (mask != 0ull && pixels_left) {
      const uint32_t cntp = __popcll(mask);
      const uint32_t leader = __ffsll(static_cast<long long>(mask)) - 1;
      if (lane == leader) base = atomicAdd(work_counter, cntp);
      base = __shfl(base, leader);
      if (base >= total_claims) pixels_left = false;
    }
    if (want) {
      claim = pixels_left ? base + lane_rank(mask, lane) : total_claims;
      have_claim = true;
    }
    if (need_pixel && !pending) {
      if (claim >= total_claims) {
        retire = true;
        need_pixel = false;
      } else {
        const uint32_t seg = claim / total_items;
        item = claim - seg * total_items;
        bool valid = true;
        if (single) {
          px = static_cast<uint32_t>(A.single_x), py = static_cast<uint32_t>(A.single_y);
        } else {
          const uint32_t tile = (item >> 6) * A.tile_world + A.tile_rank;
          const uint32_t within = item & 63u;
          const uint32_t tx = tile / A.tiles_y, ty = tile - tx * A.tiles_y;
          px = tx * 8 + (within & 7u);
          py = ty * 8 + (within >> 3);
          valid = (tx < A.tiles_x) && (px < W) && (py < H);
        }
        if (!valid) {
          have_claim = false;   // off the image in every segment: draw another item
        } else if (seg == 0u) {
          const uint64_t image_index = uint64_t(px) + uint64_t(H - 1 - py) * W;
          pcg_seed(rng, image_index);
          smp = 0;
          acc = f3{0.f, 0.f, 0.f};
          need_pixel = false;
        } else {
          VIMG_GLOBAL uint32_t* st =
              reinterpret_cast<VIMG_GLOBAL uint32_t*>(A.pool_state + size_t(item) * 2u);
          const uint32_t done = state_load(st + 2);
          if (done == A.pool_epoch + seg) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t r_lo = state_load(st + 0), r_hi = state_load(st + 1);
            rng.s = uint64_t(r_lo) | (uint64_t(r_hi) << 32);
            acc = f3{uf(state_load(st + 4)), uf(state_load(st + 5)), uf(state_load(st + 6))};
            smp = seg * seg_len;
            need_pixel = false;
          } else {
            pending = true;   // the previous segment of this pixel is still in flight somewhere
          }
        }
      }
    }
  }
  skip_fin = finisher_batch && (__ballot(pending) == __ballot(on));
  if (!skip_fin) idle_polls = 0;
  const bool regen = on && !retire && !pending && (finish || fresh);
  if (regen) {
    const f2 off = random_x_y_r2(px + py + smp);
    // right-to-left argument evaluation of the reference's call
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
    has_s = false;
    has_r = true;
  }
  }   // finisher_batch

  // ---- registers -> slot state, slot -> Q_walk
  const bool keep = on && !retire;
  if (keep) {
    // a slot that waits for its item's previous segment stays "fresh" and keeps the claim
    const uint32_t nf = pending ? (SF_FRESH | SF_PRIMARY)
                                : ((primary ? SF_PRIMARY : 0u) | (non_specular_bounce ? SF_NONSPEC : 0u)
                                   | (has_s ? SF_HAS_S : 0u) | (has_r ? SF_HAS_R : 0u)
                                   | (bounce << SF_BOUNCE_SHIFT));
    if (pending) has_s = false, has_r = false, smp = claim;
    wr(SR_ORIGIN, slot, v4u{fu(ray_o.x), fu(ray_o.y), fu(ray_o.z), fu(shadow_max_t)});
    wr(SR_RAY, slot, v4u{fu(ray_d.x), fu(ray_d.y), fu(ray_d.z), nf});
    wr(SR_SHADOW, slot, v4u{fu(shadow_d.x), fu(shadow_d.y), fu(shadow_d.z), 0u});
    cwr(SC_THROUGHPUT, slot, v4u{fu(throughput.x), fu(throughput.y), fu(throughput.z), fu(eta_scale)});
    cwr(SC_RESULT, slot, v4u{fu(result.x), fu(result.y), fu(result.z), fu(prev_pdf)});
    if (has_s) cwr(SC_NEE, slot, v4u{fu(nee_contrib.x), fu(nee_contrib.y), fu(nee_contrib.z), 0u});
    cwr(SC4_RNG, slot, v4u{static_cast<uint32_t>(rng.s), static_cast<uint32_t>(rng.s >> 32),
                         px | (py << 16), smp});
    if (finisher_batch) cold_acc[slot] = v4u{fu(acc.x), fu(acc.y), fu(acc.z), item};
    if constexpr (TEX) cold_cone[slot] = v4u{fu(cone.cone_width), fu(cone.spread_angle), 0u, 0u};
  }
  {
    const bool to_walk = keep && (has_s || has_r), to_fin = keep && !to_walk;
    const unsigned long long mask = __ballot(to_walk), mfin = __ballot(to_fin);
    if constexpr (GRP) {
      grp_lock(G, work_counter + 1);
      qw_head = uni(G->qw_head), qw_count = uni(G->qw_count);
      qv_head0 = uni(G->qv_head[0]), qv_count0 = uni(G->qv_count[0]);
    }
    if (to_walk) q_walk[ring(qw_head + qw_count + lane_rank(mask, lane))] = static_cast<QId>(slot);
    if (to_fin) q_vertex[ring(qv_head0 + qv_count0 + lane_rank(mfin, lane))] = static_cast<QId>(slot);
    qw_count += __popcll(mask);
    qv_count0 += __popcll(mfin);
    if constexpr (GRP) {
      if (lane == 0) {
        G->qw_count = qw_count, G->qv_count[0] = qv_count0;
        if (!pixels_left) G->pixels_left = 0u;
      }
      grp_unlock(G);
      const uint32_t n_retired = static_cast<uint32_t>(__popcll(__ballot(on && retire)));
      if (n_retired && lane == 0)
        __hip_atomic_fetch_sub(&G->live, n_retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  // scheduler state back to the wave's record
  if (lane == 0) {
    if constexpr (!GRP) {
      pw->qw_head = qw_head, pw->qw_count = qw_count;
      pw->qv_head[0] = qv_head0, pw->qv_count[0] = qv_count0;
      pw->pixels_left = pixels_left ? 1u : 0u;
    }
    pw->skip_fin = skip_fin ? 1u : 0u;
    pw->idle_polls = idle_polls;
  }
  if (__any(nan_here != 0u)) {
    const uint32_t c = static_cast<uint32_t>(__popcll(__ballot(nan_here != 0u)));
    if (lane == 0) pw->nan_samples += c;
  }
}

// NC: rays a lane walks at the same time (1 or 2).  With two, the box loop tests the node of each
// in one pass - twice the independent arithmetic and twice the loads in flight per trip, which is
// what a walk that waits for its node records (LDS on small scenes, L2 / HBM on large ones) lacks.
// GRP: one pool and one set of queues per workgroup instead of per wave (Pool4Group).
template <bool TEX, bool DEEP, int WPS, int NC, bool GRP>
__global__ void __launch_bounds__(256, WPS)
render_pool4_kernel(const Pool4KArgs* __restrict__ kargs) {
  static_assert(!GRP || NC == 1, "the group build walks one ray per lane");
  const uint32_t k_lo = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(kargs)),
                 k_hi = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(kargs) >> 32);
  const Pool4KPtr K = pool4_kargs(k_lo, k_hi);
  const DScene& g = *(const DScene*)&K->g;
  const RenderArgs& A = *(const RenderArgs*)&K->A;
  DeviceStats* __restrict__ stats = K->stats;
  unsigned int* __restrict__ work_counter = K->work_counter;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Lds L = stage_lds(g, A, (VIMG_LDS unsigned char*)lds_raw);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool full_stats = A.full_stats != 0;
  // event counters of the walk without a branch in the loops: += 1 or += 0 (wave-uniform)
  const uint32_t stat_inc = full_stats ? 1u : 0u;
  const uint32_t W = static_cast<uint32_t>(g.res_x), H = static_cast<uint32_t>(g.res_y);
  const bool single = A.single_x >= 0;
  const uint32_t total_items = single ? 1u : A.num_local_tiles * 64u;
  // work items of the counter: (segment, pixel) in segment-major order
  const uint32_t n_seg = A.pool_segments, seg_len = A.pool_seg_len;
  const uint32_t total_claims = total_items * n_seg;
  constexpr uint32_t roulette_threshold = 5;
  const bool material_mode = (A.integrator == VIMG_INTEGRATOR_MATERIAL);
  const uint32_t P = GRP ? 4u * A.pool_slots : A.pool_slots;   // slots of the pool this wave works on
  using QId = std::conditional_t<GRP, uint16_t, uint8_t>;
  // LDS carve-out of this wave (GRP: of the workgroup) behind the node planes and the four traversal stacks
  VIMG_LDS uint32_t* pool;
  VIMG_LDS QId* q_walk;
  VIMG_LDS v4f* lds_leaf;        // copy of leaf_prims (all of them) when A.lds_leaf != 0
  VIMG_LDS uint32_t* q_prim;     // primitive id of the hit, per slot
  VIMG_LDS QId* q_vertex;        // four rings of capacity P: class 0 finishers, 1 Lambertian (+rest), 2 Principled, 3 other
  VIMG_LDS Pool4Wave* pw;
  VIMG_LDS Pool4Diag* dg;
  [[maybe_unused]] VIMG_LDS Pool4Group* G = nullptr;
  [[maybe_unused]] VIMG_LDS uint16_t* bslots = nullptr;   // this wave's row
  VIMG_LDS uint32_t* stack0;   // this lane's stack of its first ray: entry k at stack0[k * 64]; second ray: + stack_entries * 64
  {
    const uint32_t node_bytes = (lds_node_bytes(A.lds_nodes) + 255u) & ~255u;
    const uint32_t stack_bytes = 4u * uint32_t(NC) * pool4_stack_rows_of(A.stack_entries, A.stack_lds) * 64u * 4u;   // [wave][ray of the lane][entry][lane]
    stack0 = reinterpret_cast<VIMG_LDS uint32_t*>((VIMG_LDS unsigned char*)lds_raw + node_bytes) +
             size_t(wave) * NC * pool4_stack_rows_of(A.stack_entries, A.stack_lds) * 64u + lane;
    VIMG_LDS uint32_t* base =
        reinterpret_cast<VIMG_LDS uint32_t*>((VIMG_LDS unsigned char*)lds_raw + node_bytes + stack_bytes);
    VIMG_LDS uint32_t* tail;   // behind the pools: per-wave records, leaf copy
    if constexpr (GRP) {
      pool = base;
      const Pool4GLayout lay = pool4g_layout(pool, P);
      q_prim = lay.q_prim, q_walk = lay.q_walk, q_vertex = lay.q_vertex, G = lay.G;
      bslots = lay.bslots + wave * 64u;
      tail = base + pool4g_group_bytes(A.pool_slots) / 4u;
    } else {
      const uint32_t per_wave = pool4_wave_bytes(P) / 4u;   // in dwords
      pool = base + wave * per_wave;
      q_prim = pool + P4_HOT * 4u * P;
      q_walk = reinterpret_cast<VIMG_LDS QId*>(q_prim + P);
      q_vertex = q_walk + P;
      tail = base + 4u * per_wave;
    }
    pw = reinterpret_cast<VIMG_LDS Pool4Wave*>(tail) + wave;
    dg = reinterpret_cast<VIMG_LDS Pool4Diag*>(tail + 4u * (sizeof(Pool4Wave) / 4u)) + wave;
    lds_leaf = reinterpret_cast<VIMG_LDS v4f*>(tail + 4u * ((sizeof(Pool4Wave) + sizeof(Pool4Diag)) / 4u));
    if (lane < sizeof(Pool4Diag) / 4u) reinterpret_cast<VIMG_LDS uint32_t*>(dg)[lane] = 0u;
    for (uint32_t i = threadIdx.x; i < A.lds_leaf * 3u; i += blockDim.x)
      lds_leaf[i] = reinterpret_cast<gptr<v4f>>(g.leaf_prims)[i];
    if (A.lds_leaf) __syncthreads();
  }
  const bool leaf_in_lds = A.lds_leaf != 0u;
  const uint32_t box_min = A.pool_boxmin;
  VIMG_LDS v4u* recs = reinterpret_cast<VIMG_LDS v4u*>(pool);
  auto rd = [&](uint32_t r, uint32_t slot) -> v4u { return recs[r * P + slot]; };
  auto wr = [&](uint32_t r, uint32_t slot, v4u v) { recs[r * P + slot] = v; };
  auto word = [&](uint32_t r, uint32_t k, uint32_t slot) -> VIMG_LDS uint32_t& {
    return pool[(r * P + slot) * 4u + k];
  };
  auto fu = [](float f) { return __float_as_uint(f); };
  auto uf = [](uint32_t u) { return __uint_as_float(u); };

  Counters cnt{0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t iter_wave = 0;

  // wave-uniform queue state (rings of capacity P)
  uint32_t qw_head = 0, qw_count = 0;
  uint32_t qv_head0 = 0, qv_head1 = 0, qv_head2 = 0, qv_head3 = 0;
  uint32_t qv_count0 = 0, qv_count1 = 0, qv_count2 = 0, qv_count3 = 0;
  auto ring = [&](uint32_t i) { return i >= P ? i - P : i; };

  // every slot starts "fresh": it needs a pixel
  if constexpr (GRP) {
    for (uint32_t s = threadIdx.x; s < P; s += 256) {
      word(SR_RAY, 3, s) = SF_FRESH;
      q_vertex[s] = static_cast<QId>(s);
    }
    if (threadIdx.x == 0) {
      G->lock = 0u, G->live = P, G->pixels_left = 1u, G->abort = 0u;
      G->qw_head = 0u, G->qw_count = 0u;
      for (int k = 0; k < 4; ++k) G->qv_head[k] = 0u, G->qv_count[k] = (k == 0) ? P : 0u;
    }
    if (lane == 0) pw->skip_fin = 0u, pw->idle_polls = 0u;
    __syncthreads();
  } else {
    for (uint32_t s = lane; s < P; s += 64) {
      word(SR_RAY, 3, s) = SF_FRESH;
      q_vertex[s] = static_cast<QId>(s);
    }
  }
  qv_count0 = P;
  bool skip_fin = false;
  uint32_t idle_polls = 0;
  [[maybe_unused]] uint32_t idle_wait = 0;   // GRP: looks at empty queues while other waves of the group hold the slots
  if (lane == 0) pw->pixels_left = 1u, pw->nan_samples = 0u;

  // ---- persistent walk registers of the lane: NC rays
  struct WalkCtx {
    uint32_t slot, phase, flags, cls, sp, cur;
    bool setup, any, found, exact;
    TravRay ray;
    f3 inv;
    TriRayConst rc;
    float dir_len2;
    HitRec rec;
  };
  WalkCtx C[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    C[k].slot = SLOT_IDLE, C[k].phase = 0, C[k].flags = 0, C[k].cls = 0, C[k].sp = 0, C[k].cur = REF_DONE;
    C[k].setup = C[k].any = C[k].found = C[k].exact = false;
    C[k].ray = TravRay{f3{0.f, 0.f, 0.f}, f3{0.f, 0.f, 1.f}, 0.0001f, VIMG_INF};
    C[k].inv = f3{1.f, 1.f, 1.f};
    C[k].rc = TriRayConst{0.f, 0.f, 1.f, 2};
    C[k].dir_len2 = 1.f;
    C[k].rec.prim = 0xffffffffu;
    C[k].rec.kind = 0;
    C[k].rec.e0 = C[k].rec.e1 = C[k].rec.e2 = C[k].rec.inv_det = 0.f;
  }
  const uint32_t kstride = pool4_stack_rows_of(A.stack_entries, A.stack_lds) * 64u;   // dwords between the stacks of a lane's two rays
  // Deep trees: only the first S entries of a lane's stack live in LDS (a ray of a 25-level tree
  // rarely holds more than a dozen), the rest in a per-wave region in global memory, so that the
  // LDS they would take goes to path slots.  Row S of the LDS stack takes the branch-free step's
  // writes and read-aheads of the entries above.
  const uint32_t S = A.stack_lds;
  const uint32_t ovf_stride = (A.stack_entries - S) * 64u;   // dwords between the overflow stacks of a lane's two rays
  VIMG_GLOBAL uint32_t* ovf0 = A.stack_ovf + (size_t(blockIdx.x) * 4u + wave) * NC * ovf_stride + lane;
  auto count_ctx = [&](auto pred) {
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < NC; ++k) n += static_cast<uint32_t>(__popcll(__ballot(pred(C[k]))));
    return n;
  };

  unsigned long long t_mark = full_stats ? __builtin_readcyclecounter() : 0ull;
#ifdef VIMG_WALK_DIAG   // measurement build only: where a walk round's cycles go (wave-uniform accumulators)
  unsigned long long wd_fill = 0, wd_box = 0, wd_leaf = 0, wd_ret = 0, wd_t = 0;
  unsigned long long wd_nbox = 0, wd_boxlanes = 0, wd_nleaf = 0, wd_leaflanes = 0, wd_primtrips = 0;
  unsigned long long wd_lockwait = 0, wd_lockhold = 0, wd_nlock = 0;   // group build, retire: until the lock is held / until it is released
#define WD_MARK() (wd_t = __builtin_readcyclecounter())
#define WD_ADD(acc) do { const unsigned long long n_ = __builtin_readcyclecounter(); acc += n_ - wd_t; wd_t = n_; } while (0)
#else
#define WD_MARK() ((void)0)
#define WD_ADD(acc) ((void)0)
#endif
  for (;;) {
    if constexpr (GRP) {
      if (uni(lds_aload(&G->abort)) != 0u) break;
    }
    if constexpr (GRP) {   // a look at the group's counters (decisions below are re-made under the lock)
      qw_count = uni(lds_aload(&G->qw_count));
      qv_count0 = uni(lds_aload(&G->qv_count[0])), qv_count1 = uni(lds_aload(&G->qv_count[1]));
      qv_count2 = uni(lds_aload(&G->qv_count[2])), qv_count3 = uni(lds_aload(&G->qv_count[3]));
    }
    const uint32_t n_walking = count_ctx([](const WalkCtx& c) { return c.slot != SLOT_IDLE; });
    const bool inflight = n_walking != 0u;
    // vertex batches are sorted by the material class of the hit (known from the primitive at the
    // end of the walk), so that a batch executes one material's code: a full batch of any class
    // runs at once; when the walkers have nothing left, the fullest class runs partially filled
    const uint32_t qv_elig0 = skip_fin ? 0u : qv_count0;
    const uint32_t qv_max01 = qv_elig0 > qv_count1 ? qv_elig0 : qv_count1;
    const uint32_t qv_max23 = qv_count2 > qv_count3 ? qv_count2 : qv_count3;
    const uint32_t qv_max = qv_max01 > qv_max23 ? qv_max01 : qv_max23;
    // a vertex batch runs when one is full, or when the walkers starve: no queued ray and
    // pool_starve or more idle lanes (the walk would go on half empty while slots wait here)
    const bool run_vertex = (qv_max >= A.pool_vbatch) ||
                            (qv_max > 0u && qw_count == 0u && 64u * NC - n_walking >= A.pool_starve * NC);
    if constexpr (GRP) {
      if (!run_vertex && qw_count == 0u && !inflight) {
        // nothing for this wave: done when every slot of the group has retired; else the other
        // waves hold the slots (in their lanes or in a batch), or the finisher queue holds only
        // slots that wait for a segment - look again shortly (both waits are bounded, as below)
        if (uni(lds_aload(&G->live)) == 0u || uni(lds_aload(&G->abort)) != 0u) break;
        const bool only_waiting = skip_fin && qv_count0 != 0u;
        skip_fin = false;
        if ((only_waiting && ++idle_polls > (1u << 20)) || ++idle_wait > (1u << 25)) {
          if (lane == 0) {
            atomicOr(work_counter + 1, 1u);
            __hip_atomic_store(&G->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(16);
        continue;
      }
      idle_wait = 0;
    }
    if (!GRP && !run_vertex && qw_count == 0u && !inflight) {
      if (!skip_fin || qv_count0 == 0u) break;   // every queue is empty: all done
      skip_fin = false;                          // only waiting slots are left: look at them again
      // Watchdog: the wait is for segments other waves are working on, i.e. for at most the time
      // of a few samples.  A wave that has looked a million times in a row (seconds) is stuck on
      // something that will not come; it raises the error word behind the work counter and
      // leaves, so that a scheduling bug ends as VIMG_E_DEVICE instead of a hung GPU.
      if (++idle_polls > (1u << 20)) {
        if (lane == 0) atomicOr(work_counter + 1, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    if (full_stats && lane == 0) iter_wave++;

    if (run_vertex) {
      // ================================================================== VERTEX stage (a call)
      uint32_t cls = (qv_elig0 == qv_max) ? 0u : (qv_count1 == qv_max ? 1u : (qv_count2 == qv_max ? 2u : 3u));
      uint32_t n_batch = qv_max < 64u ? qv_max : 64u;
      if constexpr (GRP) {
        // take the batch under the lock, from the queue that is fullest NOW (another wave may have
        // taken what the look above saw)
        grp_lock(G, work_counter + 1);
        const uint32_t c0 = skip_fin ? 0u : uni(G->qv_count[0]), c1 = uni(G->qv_count[1]), c2 = uni(G->qv_count[2]),
                       c3 = uni(G->qv_count[3]);
        const uint32_t m01 = c0 > c1 ? c0 : c1, m23 = c2 > c3 ? c2 : c3, mx = m01 > m23 ? m01 : m23;
        cls = (c0 == mx) ? 0u : (c1 == mx ? 1u : (c2 == mx ? 2u : 3u));
        n_batch = mx < 64u ? mx : 64u;
        if (n_batch != 0u) {
          const uint32_t head = uni(G->qv_head[cls]), have = uni(G->qv_count[cls]);
          if (lane < n_batch) bslots[lane] = q_vertex[cls * P + ring(head + lane)];
          if (lane == 0) G->qv_head[cls] = ring(head + n_batch), G->qv_count[cls] = have - n_batch;
        }
        grp_unlock(G);
        if (n_batch == 0u) continue;
      }
      if (full_stats) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0) dg->cyc[4] += now - t_mark, dg->nbatch[cls] += 1, dg->nslots[cls] += n_batch;
        t_mark = now;
      }
      if (lane == 0) {
        if constexpr (!GRP) {
          pw->qw_head = qw_head, pw->qw_count = qw_count;
          pw->qv_head[0] = qv_head0, pw->qv_head[1] = qv_head1, pw->qv_head[2] = qv_head2, pw->qv_head[3] = qv_head3;
          pw->qv_count[0] = qv_count0, pw->qv_count[1] = qv_count1, pw->qv_count[2] = qv_count2, pw->qv_count[3] = qv_count3;
        }
        pw->skip_fin = skip_fin ? 1u : 0u, pw->idle_polls = idle_polls;
      }
      if (cls == 0u)
        pool4_vertex<TEX, true, -1, WPS, GRP>(k_lo, k_hi, pool, pw, cls, n_batch);
      else if (cls == 1u && A.pool_classes == 3u)
        pool4_vertex<TEX, false, int(VIMG_MAT_LAMBERTIAN), WPS, GRP>(k_lo, k_hi, pool, pw, cls, n_batch);
      else if (cls == 2u && A.pool_classes == 3u)
        pool4_vertex<TEX, false, int(VIMG_MAT_PRINCIPLED), WPS, GRP>(k_lo, k_hi, pool, pw, cls, n_batch);
      else
        pool4_vertex<TEX, false, -1, WPS, GRP>(k_lo, k_hi, pool, pw, cls, n_batch);
      if constexpr (!GRP) {
        qw_head = uni(pw->qw_head), qw_count = uni(pw->qw_count);
        qv_head0 = uni(pw->qv_head[0]), qv_head1 = uni(pw->qv_head[1]), qv_head2 = uni(pw->qv_head[2]), qv_head3 = uni(pw->qv_head[3]);
        qv_count0 = uni(pw->qv_count[0]), qv_count1 = uni(pw->qv_count[1]), qv_count2 = uni(pw->qv_count[2]), qv_count3 = uni(pw->qv_count[3]);
      }
      skip_fin = uni(pw->skip_fin) != 0u, idle_polls = uni(pw->idle_polls);
      if (full_stats) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0) dg->cyc[cls] += now - t_mark;
        t_mark = now;
      }
    } else {
      // ================================================================== WALK stage
      skip_fin = false;
      idle_polls = 0;

      for (;;) {
        WD_MARK();
        // (1) idle rays of the lanes take queued slots
        {
          unsigned long long m[NC];
          uint32_t base[NC], n_idle = 0;
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            m[k] = __ballot(C[k].slot == SLOT_IDLE);
            base[k] = n_idle;
            n_idle += static_cast<uint32_t>(__popcll(m[k]));
          }
          bool locked = false;
          if constexpr (GRP) {
            qw_count = uni(lds_aload(&G->qw_count));
            if (n_idle != 0u && qw_count != 0u) {
              grp_lock(G, work_counter + 1);
              locked = true;
              qw_head = uni(G->qw_head), qw_count = uni(G->qw_count);
            }
          }
          const uint32_t take = (GRP && !locked) ? 0u : (n_idle < qw_count ? n_idle : qw_count);
          if (take) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
              const uint32_t r = base[k] + lane_rank(m[k], lane);
              if (C[k].slot == SLOT_IDLE && r < take) {
                C[k].slot = q_walk[ring(qw_head + r)];
                C[k].flags = word(SR_RAY, 3, C[k].slot);
                C[k].phase = (C[k].flags & SF_HAS_S) ? 0u : 1u;
                C[k].setup = true;
              }
            }
            qw_head = ring(qw_head + take);
            qw_count -= take;
          }
          if constexpr (GRP) {
            if (locked) {
              if (take && lane == 0) G->qw_head = qw_head, G->qw_count = qw_count;
              grp_unlock(G);
            }
          }
        }
        // (2) ray set-up (reference include/bvh.h:109-143): everything derived from the ray alone
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          WalkCtx& c = C[k];
          if (__any(c.setup)) {
            if (c.setup) {
              const v4u ro = rd(SR_ORIGIN, c.slot);
              c.ray.o = f3{uf(ro.x), uf(ro.y), uf(ro.z)};
              if (c.phase == 0) {
                const v4u rs = rd(SR_SHADOW, c.slot);
                c.ray.d = f3{uf(rs.x), uf(rs.y), uf(rs.z)};
                c.ray.max_t = uf(ro.w);
                c.any = true;
                cnt.shadow++;
              } else {
                const v4u rr = rd(SR_RAY, c.slot);
                c.ray.d = f3{uf(rr.x), uf(rr.y), uf(rr.z)};
                c.ray.max_t = VIMG_INF;
                c.any = false;
                cnt.closest++;
              }
              c.inv = f3{1.0f / c.ray.d.x, 1.0f / c.ray.d.y, 1.0f / c.ray.d.z};
              c.exact = (c.ray.d.x == 0.f) || (c.ray.d.y == 0.f) || (c.ray.d.z == 0.f);
              c.rc = tri_ray_const(c.ray.d);
              c.dir_len2 = dot(c.ray.d, c.ray.d);
              const float root =
                  slab(load3k(g.root_min), load3k(g.root_max), c.ray.o, c.inv, c.ray.min_t, c.ray.max_t);
              c.cur = is_inf(root) ? REF_DONE : g.root_ref;
              c.sp = 0;
              c.found = false;
              c.rec.prim = 0xffffffffu;
              c.setup = false;
            }
          }
        }

        WD_ADD(wd_fill);
        if (count_ctx([](const WalkCtx& c) { return c.slot != SLOT_IDLE; }) == 0u) break;
        // (3) walk until a quarter of the rays in flight have finished (or nothing is left to walk)
        // One step of a ray that stands at an internal node, its record in hand.  Branch-free: the
        // entry a pop would return is read before the box test (its latency hides behind the test);
        // the far child is written above the top of the stack whether it is kept or not (the slot
        // is free), and sp moves by select.
        auto box_step = [&](WalkCtx& c, int k, v4f na, v4f nb, v4f nc, v2u refs, auto exact_possible) {
          const uint32_t sp_below = c.sp != 0 ? c.sp - 1 : 0u;
          uint32_t popped = stack0[k * kstride + (DEEP ? (sp_below < S ? sp_below : S) : sp_below) * 64];
          cnt.internal += stat_inc;
          float h1, h2;
          if (decltype(exact_possible)::value && c.exact) {
            h1 = slab(f3{na.x, na.y, na.z}, f3{na.w, nb.x, nb.y}, c.ray.o, c.inv, c.ray.min_t, c.ray.max_t);
            h2 = slab(f3{nb.z, nb.w, nc.x}, f3{nc.y, nc.z, nc.w}, c.ray.o, c.inv, c.ray.min_t, c.ray.max_t);
          } else {
            h1 = slab_fast(f3{na.x, na.y, na.z}, f3{na.w, nb.x, nb.y}, c.ray.o, c.inv, c.ray.min_t, c.ray.max_t);
            h2 = slab_fast(f3{nb.z, nb.w, nc.x}, f3{nc.y, nc.z, nc.w}, c.ray.o, c.inv, c.ray.min_t, c.ray.max_t);
          }
          const bool in1 = !is_inf(h1), in2 = !is_inf(h2);
          const uint32_t c1 = refs.x, c2 = refs.y;
          const bool both = in1 && in2, any = in1 || in2;
          const bool first_is_near = c.any ? false : (h2 > h1);
          const uint32_t near_c = first_is_near ? c1 : c2;
          const uint32_t far_c = first_is_near ? c2 : c1;
          stack0[k * kstride + (DEEP ? (c.sp < S ? c.sp : S) : c.sp) * 64] = far_c;
          if constexpr (DEEP) {
            if (both && c.sp >= S) ovf0[k * ovf_stride + (c.sp - S) * 64] = far_c;
            if (!any && sp_below >= S) popped = ovf0[k * ovf_stride + (sp_below - S) * 64];   // (sp_below >= S > 0: sp != 0)
          }
          const uint32_t one_c = in1 ? c1 : c2;
          c.cur = both ? near_c : (any ? one_c : (c.sp != 0 ? popped : REF_DONE));
          c.sp = both ? c.sp + 1 : (any ? c.sp : sp_below);
        };
        // One primitive of a leaf against the ray, by select: a hit shortens the ray; an any-hit ray
        // stops at its first hit (returned), a closest-hit ray keeps the record of the last success.
        auto prim_step = [&](WalkCtx& c, v4f a, v4f b, v4f cc) -> bool {
          const float c0 = cc.x;
          const uint32_t lp_prim = __float_as_uint(cc.y), kind = __float_as_uint(cc.z),
                         lp_cls = __float_as_uint(cc.w);   // DLeafPrim: c0 | prim | kind | cls
          cnt.prim += stat_inc;
          bool hit = false;
          float t = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f, idet = 0.f;
          if (kind == 0) {
            hit = tri_test_flat(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c0}, c.ray, c.rc, t, e0, e1, e2, idet);
          } else if (kind == 1) {
            cnt.sphere += stat_inc;
            hit = sphere_test(f3{a.x, a.y, a.z}, a.w, c.ray, c.dir_len2, t);
          }
          c.ray.max_t = hit ? t : c.ray.max_t;
          c.found = c.found || hit;
          const bool keep_rec = hit && !c.any;
          c.rec.e0 = keep_rec ? e0 : c.rec.e0, c.rec.e1 = keep_rec ? e1 : c.rec.e1;
          c.rec.e2 = keep_rec ? e2 : c.rec.e2, c.rec.inv_det = keep_rec ? idet : c.rec.inv_det;
          c.rec.prim = keep_rec ? lp_prim : c.rec.prim;
          c.rec.kind = keep_rec ? kind : c.rec.kind;
          c.cls = keep_rec ? lp_cls : c.cls;
          return hit && c.any;
        };
        // a leaf is done: the next node comes off the stack (or the ray is finished)
        auto leaf_pop = [&](WalkCtx& c, int k, bool stop) {
          const uint32_t sp_below = c.sp != 0 ? c.sp - 1 : 0u;
          uint32_t popped = stack0[k * kstride + (DEEP ? (sp_below < S ? sp_below : S) : sp_below) * 64];
          if constexpr (DEEP) {
            if (!stop && sp_below >= S) popped = ovf0[k * ovf_stride + (sp_below - S) * 64];
          }
          c.cur = (stop || c.sp == 0) ? REF_DONE : popped;
          c.sp = sp_below;
        };
        bool exact_any = false;
#pragma unroll
        for (int k = 0; k < NC; ++k) exact_any = exact_any || (C[k].exact && C[k].slot != SLOT_IDLE);
        const bool exact_round = __any(exact_any);   // (the rays of the lanes do not change inside (3))

        for (;;) {
          // "while-while": the box loop until every ray stands at a leaf (or is done), then the
          // leaves.  The box loop comes in two builds: rays with a zero direction component need the
          // exact select form of the slab test (0 * inf); a round without such a ray runs the build
          // that has only the min/max form.  One pass of the loop steps every ray of the lane that
          // stands at an internal node.
          auto box_loop = [&](auto exact_possible) {
            for (;;) {
              bool act[NC];
              bool any_act = false;
#pragma unroll
              for (int k = 0; k < NC; ++k) {
                act[k] = C[k].cur != REF_DONE && ref_count(C[k].cur) == 0;
                any_act = any_act || act[k];
              }
              if (!__any(any_act)) break;
#ifdef VIMG_WALK_DIAG
              wd_nbox += 1, wd_boxlanes += __popcll(__ballot(act[0]));
#endif
#pragma unroll
              for (int k = 0; k < NC; ++k) {
                WalkCtx& c = C[k];
                // (the step runs under the lanes' execution mask: a lane whose ray does not stand at
                // an internal node issues nothing, so the VALU lane counters keep meaning work done)
                if (act[k]) {
                  v4f na, nb, nc;
                  v2u refs;
                  if (!DEEP || c.cur < L.n_nodes) {   // the build for trees that fit has every node in LDS
                    na = L.na[c.cur], nb = L.nb[c.cur], nc = L.nc[c.cur];
                    refs = L.nm[c.cur];
                  } else {
                    gptr<DNode> nd = g.nodes + c.cur;
                    na = nd->a, nb = nd->b, nc = nd->c;
                    refs = v2u{nd->left_ref, nd->right_ref};
                  }
                  box_step(c, k, na, nb, nc, refs, exact_possible);
                }
              }
              // deep trees: when only a few rays still descend, the ones that wait at a leaf go first
              // (the box loop of the config-5 stand-in ran with 27 % of its lanes busy)
              if constexpr (DEEP) {
                if (count_ctx([](const WalkCtx& c) { return c.cur != REF_DONE && ref_count(c.cur) == 0; }) < box_min * NC) break;
              }
            }
          };
          WD_MARK();
          if (exact_round)
            box_loop(std::true_type{});
          else
            box_loop(std::false_type{});
          WD_ADD(wd_box);
#ifdef VIMG_WALK_DIAG
          {
            const bool at_leaf = C[0].cur != REF_DONE && (!DEEP || ref_count(C[0].cur) != 0);
            const uint32_t cnt_l = at_leaf ? ref_count(C[0].cur) : 0u;
            uint32_t mx = cnt_l;
            for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
            wd_nleaf += 1, wd_leaflanes += __popcll(__ballot(at_leaf)), wd_primtrips += uni(mx);
          }
#endif

#pragma unroll
          for (int k = 0; k < NC; ++k) {
            WalkCtx& c = C[k];
            if (c.cur != REF_DONE && (!DEEP || ref_count(c.cur) != 0)) {
              const uint32_t first = ref_index(c.cur), count = ref_count(c.cur);
              cnt.leaf += stat_inc;
              bool stop = false;
              for (uint32_t i = 0; i < count && !stop; ++i) {
                gptr<DLeafPrim> lp = g.leaf_prims + (first + i);
                v4f a, b, cc;
                if (leaf_in_lds) {
                  const VIMG_LDS v4f* ll = lds_leaf + (first + i) * 3u;
                  a = ll[0], b = ll[1], cc = ll[2];
                } else {
                  a = lp->a, b = lp->b;
                  cc = reinterpret_cast<gptr<v4f>>(lp)[2];
                }
                stop = prim_step(c, a, b, cc);
              }
              leaf_pop(c, k, stop);
            }
          }

          WD_ADD(wd_leaf);
          const uint32_t n_fin = count_ctx([](const WalkCtx& c) { return c.slot != SLOT_IDLE && c.cur == REF_DONE; });
          const uint32_t n_act = count_ctx([](const WalkCtx& c) { return c.slot != SLOT_IDLE && c.cur != REF_DONE; });
          if (n_act == 0 || n_fin >= A.pool_refill * NC) break;
        }
        WD_MARK();
        // (4) retire finished rays: second ray of the item, or hand the slot to the vertex stage
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          WalkCtx& c = C[k];
          bool done_item = false;
          if (c.slot != SLOT_IDLE && c.cur == REF_DONE) {
            if (c.phase == 0) {
              if (c.found) c.flags |= SF_OCCLUDED;
              if (c.flags & SF_HAS_R) {
                c.phase = 1;
                c.setup = true;
              } else {
                done_item = true;
              }
            } else {
              if (c.found) {
                c.flags |= SF_FOUND | (c.rec.kind == 1 ? SF_KIND_SPHERE : 0u);
                wr(SR_SHADOW, c.slot, v4u{fu(c.rec.e0), fu(c.rec.e1), fu(c.rec.e2), fu(c.rec.inv_det)});
                q_prim[c.slot] = c.rec.prim;
                word(SR_ORIGIN, 3, c.slot) = fu(c.ray.max_t);
              }
              done_item = true;
            }
            if (done_item) word(SR_RAY, 3, c.slot) = c.flags;
          }
          // class of the batch this slot joins: 0 = its path ends (miss, no path ray, emitter hit under
          // mis, any hit under the normal integrators), else the material class of the vertex
          // (leaf record: 0 emitter, 1 Lambertian, 2 Principled, 3 other)
          uint32_t cls = 0;
          if (done_item && (c.flags & SF_FOUND) && A.integrator >= VIMG_INTEGRATOR_MATERIAL) {
            cls = c.cls;
            if (cls == 0 && material_mode) cls = 3;   // material_integrator shades emitters too
            if (cls != 0) {
              if (A.pool_classes == 1) cls = 1;
              else if (A.pool_classes == 2) cls = (cls == 2) ? 2u : 1u;
            }
          }
          const unsigned long long m0 = __ballot(done_item && cls == 0),
                                   m1 = __ballot(done_item && cls == 1),
                                   m2 = __ballot(done_item && cls == 2),
                                   m3 = __ballot(done_item && cls == 3);
          const bool any_done = (m0 | m1 | m2 | m3) != 0ull;
#ifdef VIMG_WALK_DIAG
          const unsigned long long wd_l0 = __builtin_readcyclecounter();
#endif
          if constexpr (GRP) {
            if (any_done) {
              grp_lock(G, work_counter + 1);
#ifdef VIMG_WALK_DIAG
              wd_lockwait += __builtin_readcyclecounter() - wd_l0, wd_nlock += 1;
#endif
              qv_head0 = uni(G->qv_head[0]), qv_head1 = uni(G->qv_head[1]), qv_head2 = uni(G->qv_head[2]), qv_head3 = uni(G->qv_head[3]);
              qv_count0 = uni(G->qv_count[0]), qv_count1 = uni(G->qv_count[1]), qv_count2 = uni(G->qv_count[2]), qv_count3 = uni(G->qv_count[3]);
            } else {
              qv_count0 = uni(lds_aload(&G->qv_count[0])), qv_count1 = uni(lds_aload(&G->qv_count[1]));
              qv_count2 = uni(lds_aload(&G->qv_count[2])), qv_count3 = uni(lds_aload(&G->qv_count[3]));
            }
          }
          if (done_item) {
            const QId id = static_cast<QId>(c.slot);
            if (cls == 0) q_vertex[ring(qv_head0 + qv_count0 + lane_rank(m0, lane))] = id;
            else if (cls == 1) q_vertex[P + ring(qv_head1 + qv_count1 + lane_rank(m1, lane))] = id;
            else if (cls == 2) q_vertex[2 * P + ring(qv_head2 + qv_count2 + lane_rank(m2, lane))] = id;
            else q_vertex[3 * P + ring(qv_head3 + qv_count3 + lane_rank(m3, lane))] = id;
            c.slot = SLOT_IDLE;
          }
          qv_count0 += __popcll(m0);
          qv_count1 += __popcll(m1);
          qv_count2 += __popcll(m2);
          qv_count3 += __popcll(m3);
          if constexpr (GRP) {
            if (any_done) {
              if (lane == 0) G->qv_count[0] = qv_count0, G->qv_count[1] = qv_count1, G->qv_count[2] = qv_count2, G->qv_count[3] = qv_count3;
              // while the lock is held: the lanes that have just come free take queued rays (the
              // refill at the head of the next round then finds nothing to do and takes no lock)
              const unsigned long long mi = __ballot(c.slot == SLOT_IDLE);
              qw_head = uni(G->qw_head), qw_count = uni(G->qw_count);
              const uint32_t n_idle = static_cast<uint32_t>(__popcll(mi));
              const uint32_t take = n_idle < qw_count ? n_idle : qw_count;
              if (take) {
                const uint32_t r = lane_rank(mi, lane);
                if (c.slot == SLOT_IDLE && r < take) {
                  c.slot = q_walk[ring(qw_head + r)];
                  c.flags = word(SR_RAY, 3, c.slot);
                  c.phase = (c.flags & SF_HAS_S) ? 0u : 1u;
                  c.setup = true;
                }
                qw_head = ring(qw_head + take), qw_count -= take;
                if (lane == 0) G->qw_head = qw_head, G->qw_count = qw_count;
              }
              grp_unlock(G);
#ifdef VIMG_WALK_DIAG
              wd_lockhold += __builtin_readcyclecounter() - wd_l0;
#endif
            }
          }
        }
        if constexpr (GRP) qw_count = uni(lds_aload(&G->qw_count));

        WD_ADD(wd_ret);
        if constexpr (GRP) {
          if (uni(lds_aload(&G->abort)) != 0u) break;   // the launch has been given up (watchdog): no wave keeps walking
        }
        // (5) leave when a full vertex batch waits, or when nothing is left to walk
        if (qv_count0 >= A.pool_vbatch || qv_count1 >= A.pool_vbatch || qv_count2 >= A.pool_vbatch ||
            qv_count3 >= A.pool_vbatch) {
          // (GRP: the full batch is the group's; a wave whose lanes are busy leaves it to one that is not)
          if (!GRP || count_ctx([](const WalkCtx& c) { return c.slot != SLOT_IDLE; }) <= A.pool_gbreak) break;
        }
        if (qw_count == 0u) {
          const uint32_t walking = count_ctx([](const WalkCtx& c) { return c.slot != SLOT_IDLE; });
          if (walking == 0u) break;
          if (64u * NC - walking >= A.pool_starve * NC && (qv_count0 | qv_count1 | qv_count2 | qv_count3) != 0u) break;
        }
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
                       c4 = wave_sum(cnt.prim), c6 = wave_sum(cnt.sphere), c9 = wave_sum(iter_wave);
    if (lane == 0) {
      atomicAdd(&stats->closest, c0);
      atomicAdd(&stats->shadow, c1);
      if (full_stats) {
        atomicAdd(&stats->internal, c2);
        atomicAdd(&stats->leaf, c3);
        atomicAdd(&stats->prim, c4);
        atomicAdd(&stats->sphere, c6);
        atomicAdd(&stats->iterations, c9);
      }
      const unsigned long long c5 = pw->nan_samples;
      if (c5) atomicAdd(&stats->nan_samples, c5);
      if (full_stats) {
        dg->cyc[4] += __builtin_readcyclecounter() - t_mark;
        for (int k = 0; k < 6; ++k) atomicAdd(&stats->prof[k], dg->cyc[k]);
        for (int k = 0; k < 4; ++k) atomicAdd(&stats->prof[6 + k], dg->nbatch[k]), atomicAdd(&stats->prof[11 + k], dg->nslots[k]);
        atomicAdd(&stats->prof[10], 1ull);
#ifdef VIMG_WALK_DIAG
        const unsigned long long wd[12] = {wd_fill, wd_box, wd_leaf, wd_ret, wd_nbox, wd_boxlanes, wd_nleaf, wd_leaflanes, wd_primtrips,
                                           wd_lockwait, wd_lockhold, wd_nlock};
        for (int k = 0; k < 12; ++k) atomicAdd(&stats->prof[16 + k], wd[k]);
#endif
      }
    }
  }
}

}  // namespace vimg
