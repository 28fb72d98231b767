// Staged scheduler of the render path: the per-path work of render_pool_kernel cut into STAGES that
// hand paths to each other through queues in global memory, all inside ONE persistent launch whose
// waves pick a stage per batch.
//
// Why (measured on render_pool_kernel, profiles/r1_flat): the megakernel keeps the walk's, the
// pool's and the shading's registers live together - 255 VGPRs + 336 B of scratch at two waves
// per SIMD - and the SIMDs issue vector instructions 38 % of the time.  Compiled on their own the
// stages need far less (shading of a Principled vertex 126 VGPRs, of a Lambertian one 74, the walk
// about 80) because nothing is live ACROSS stages when a path's state rests in memory between
// them.  So:
//   * path state lives in a global array of SLOTS, 12 records of 16 bytes per slot ([slot][record]:
//     the records of a slot share lines).  A slot is owned by exactly one lane at a time - the one
//     that popped its id from a queue - so the records need no atomics, only visibility: every
//     access is an L1-bypassing `sc1` buffer access (coherent across the eight XCDs), and a wave
//     waits for its stores (vmcnt(0)) before it publishes slot ids;
//   * five multi-producer / multi-consumer QUEUES of slot ids, each in 8 shards (a wave pushes to
//     the shard of its workgroup and pops there first): Q_walk (rays to be walked) and four vertex
//     queues by class - finishers (miss, emitter hit, path ended: MIS weight, pixel accumulation,
//     next camera ray), Lambertian, Principled, other materials.  A shard is a ring of tagged
//     entries (lap tag | slot id) with a reserved-tail and a claimed-head counter: producers
//     reserve with one atomic add per wave and store their entries; consumers claim with one CAS
//     per wave, never beyond the reserved tail, and read entries that are at worst being written;
//   * every wave loops { look at the queue counters, pick a stage, run one batch }.  One kernel,
//     one register budget: 128 VGPRs, four waves per SIMD, no scratch in the loops.  All stages of a
//     path run on whichever wave picks them up; co-residency is by construction (one launch), so no
//     stage can wait for a kernel that is not running;
//   * the WALK stage keeps the persistent while-while loop of the pooled kernel (lanes that finish
//     take the next ray) over a wave-private chunk of the queue staged in LDS;
//   * a PIXEL is bound to a slot for `seg_len` samples at a time.  Between segments its state
//     (RNG, accumulator, sample count) rests in a per-pixel record and its id in a FIFO of ready
//     pixels (tickets: the first W*H tickets are the pixels themselves in tile order, later ones
//     index a ring of pushed ids).  A slot that ends a segment pushes its pixel and takes the next
//     ticket, so all pixels advance at the same rate and the frame ends with the tail of one
//     segment, whatever the number of slots.  Slots retire when fewer pixels than slots are left.
// Every path still executes the reference's operations in the reference's order on its own RNG
// stream: images are bit-identical to render_kernel, render_pool_kernel and the oracle.
#pragma once
#include <type_traits>

#include "render_pool_kernel.h"

namespace vimg {

enum : uint32_t { GQ_FIN = 0, GQ_LAMB = 1, GQ_PRIN = 2, GQ_OTHER = 3, GQ_WALK = 4, GQ_COUNT = 5 };
constexpr uint32_t GQ_SHARDS = 8;
enum : uint32_t {
  GR_ORIGIN = 0,   // o.xyz | shadow max_t
  GR_RAY,          // d.xyz (camera / BSDF ray) | flags (SF_*)
  GR_SHADOW,       // shadow d.xyz | -
  GR_HIT0,         // walk result: e0 e1 e2 inv_det of the closest hit
  GR_HIT1,         // walk result: t | primitive id | SF_OCCLUDED / SF_FOUND / SF_KIND_SPHERE | -
  GR_RNG,          // rng lo | rng hi | px + (py << 16) | sample index
  GR_THR,          // throughput.xyz | eta_scale
  GR_RES,          // bounce_result.xyz | prev_pdf
  GR_NEE,          // unoccluded next-event contribution.xyz | -
  GR_ACC,          // accumulated pixel radiance.xyz | work item id
  GR_CONE,         // cone width | spread angle | - | -   (textured build only)
  GR_COUNT = 12
};
constexpr uint32_t GR_BYTES = GR_COUNT * 16u;
constexpr uint32_t GQ_ID_BITS = 20u;                 // slot ids < 2^20; the rest of an entry is the lap tag
constexpr uint32_t GQ_ID_MASK = (1u << GQ_ID_BITS) - 1u;
constexpr uint32_t STAGE_MAX_SLOTS = GQ_ID_MASK;     // (the host caps the pool well below)
constexpr uint32_t STAGE_WCHUNK_MAX = 256u;
constexpr uint32_t PIX_TAG_SHIFT = 27u;              // ready-pixel ring: work items < 2^27

struct StageWord {   // one counter per 128-byte line
  uint32_t v;
  uint32_t pad[31];
};
struct StageCtl {
  StageWord q_head[GQ_COUNT * GQ_SHARDS];   // entries claimed by consumers
  StageWord q_tail[GQ_COUNT * GQ_SHARDS];   // entries reserved by producers
  StageWord tk_head;                        // ready-pixel tickets claimed
  StageWord tk_tail;                        // ready-pixel ring entries reserved
  StageWord surplus;                        // pixels nobody has started yet (signed)
  StageWord retired;                        // slots that have left the pool
  StageWord error;                          // watchdog
};
struct StageArgs {
  VIMG_GLOBAL StageCtl* ctl;
  VIMG_GLOBAL uint32_t* rings;      // [GQ_COUNT * GQ_SHARDS][ring_cap]
  VIMG_GLOBAL uint32_t* pix_ring;   // [pix_cap]
  VIMG_GLOBAL v4u* pix_state;       // [items][2]: {rng lo, rng hi, samples done, -}{acc.xyz, -}
  VIMG_GLOBAL v4u* slots;           // [n_slots][GR_COUNT]
  uint32_t ring_cap, ring_shift;    // entries per shard (power of two >= n_slots)
  uint32_t pix_cap, pix_shift;      // entries of the ready-pixel ring (power of two >= items)
  uint32_t n_slots, seg_len, wchunk, walk_quota;
  uint32_t rings_bytes, pix_ring_bytes, pix_state_bytes, slots_bytes;
};

// Scene and launch parameters of the staged kernel live in ONE block in device memory (written by a
// one-thread set-up kernel in front of the launch).  The stage functions are real, non-inlined
// calls - one giant inlined body is exactly what defeats the register allocator in the megakernel -
// and a callee cannot see the kernel-argument segment (code object v5 hands it no kernarg pointer),
// so they receive the block's address, make it wave-uniform and read it through the constant
// address space: scalar loads into SGPRs, no copies of the scene on the stack.
struct StageKArgs {
  DScene g;
  RenderArgs A;
  StageArgs S;
  float* out;
  DeviceStats* stats;
};
typedef const __attribute__((address_space(4))) StageKArgs* StageKPtr;
VD StageKPtr stage_kargs(uint32_t lo, uint32_t hi) {
  const unsigned long long a =
      static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lo)))) |
      (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(hi)))) << 32);
  return (StageKPtr)a;
}
static __global__ void stage_args_kernel(const StageKArgs ka, StageKArgs* __restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = ka;
}

// ---- memory forms.  Slot records, pixel records and ring entries: raw buffer accesses with the
// sc1 cache policy (aux = 16): L1 is bypassed, stores are written through, and the buffer's byte
// limit turns a wild offset into a dropped access instead of a fault.
constexpr int AUX_SC1 = 16;
struct StageBufs {
  __amdgpu_buffer_rsrc_t slots, pix_state, rings, pix_ring;
};
VD StageBufs stage_bufs(const StageArgs& S) {
  StageBufs b;
  b.slots = __builtin_amdgcn_make_buffer_rsrc((void*)S.slots, 0, S.slots_bytes, 0x00020000);
  b.pix_state = __builtin_amdgcn_make_buffer_rsrc((void*)S.pix_state, 0, S.pix_state_bytes, 0x00020000);
  b.rings = __builtin_amdgcn_make_buffer_rsrc((void*)S.rings, 0, S.rings_bytes, 0x00020000);
  b.pix_ring = __builtin_amdgcn_make_buffer_rsrc((void*)S.pix_ring, 0, S.pix_ring_bytes, 0x00020000);
  return b;
}
VD v4u ld16(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1); }
VD void st16(__amdgpu_buffer_rsrc_t r, uint32_t off, v4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX_SC1); }
VD uint32_t ld4(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX_SC1); }
VD void st4(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, AUX_SC1); }
// all vector-memory operations of this wave have been performed (stores acknowledged)
VD void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
VD uint32_t ctl_load(VIMG_GLOBAL StageWord* w) { return __hip_atomic_load(&w->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
VD uint32_t ctl_add(VIMG_GLOBAL StageWord* w, uint32_t n) {
  return __hip_atomic_fetch_add(&w->v, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
VD uint32_t bcast(uint32_t v, uint32_t from_lane) {
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(from_lane)));
}
VD uint32_t first_lane_of(unsigned long long mask) { return static_cast<uint32_t>(__ffsll(static_cast<long long>(mask)) - 1); }

constexpr uint32_t STAGE_SPIN_LIMIT = 1u << 22;   // polls of one ring entry before the watchdog trips

// ---- queue of slot ids (one shard) ----------------------------------------------------------
VD uint32_t gq_index(uint32_t q, uint32_t sh) { return q * GQ_SHARDS + sh; }
VD uint32_t gq_tag(const StageArgs& S, uint32_t ticket) { return ((ticket >> S.ring_shift) & 0x7ffu) + 1u; }

// Lanes with `want` append `id` to queue q of shard sh.  The caller has drained its stores to the
// slots' records before (drain_vmem).
VD void gq_push(const StageArgs& S, const StageBufs& B, uint32_t q, uint32_t sh, bool want, uint32_t id,
                uint32_t lane) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull) return;
  const uint32_t leader = first_lane_of(mask);
  uint32_t base = 0;
  if (lane == leader) base = ctl_add(&S.ctl->q_tail[gq_index(q, sh)], static_cast<uint32_t>(__popcll(mask)));
  base = bcast(base, leader);
  if (want) {
    const uint32_t t = base + lane_rank(mask, lane);
    st4(B.rings, (gq_index(q, sh) * S.ring_cap + (t & (S.ring_cap - 1u))) * 4u, (gq_tag(S, t) << GQ_ID_BITS) | id);
  }
}

// Claims up to max_n entries of queue q, shard sh (one CAS by lane 0, never beyond the reserved
// tail).  Returns the number claimed (wave-uniform) and the first ticket.
VD uint32_t gq_claim(const StageArgs& S, uint32_t q, uint32_t sh, uint32_t max_n, uint32_t lane, uint32_t& first) {
  uint32_t n = 0, start = 0;
  if (lane == 0) {
    VIMG_GLOBAL StageWord* hw = &S.ctl->q_head[gq_index(q, sh)];
    uint32_t h = ctl_load(hw);
    const uint32_t t = ctl_load(&S.ctl->q_tail[gq_index(q, sh)]);
    for (int tries = 0; tries < 3; ++tries) {
      const int32_t avail = static_cast<int32_t>(t - h);
      if (avail <= 0) break;
      const uint32_t take = static_cast<uint32_t>(avail) < max_n ? static_cast<uint32_t>(avail) : max_n;
      uint32_t expect = h;
      if (__hip_atomic_compare_exchange_strong(&hw->v, &expect, h + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) {
        n = take, start = h;
        break;
      }
      h = expect;
    }
  }
  n = bcast(n, 0);
  first = bcast(start, 0);
  return n;
}
// The slot id behind ticket t (claimed by this wave): the entry is reserved, at worst its store is
// still on its way.  ~0u when the watchdog trips.
VD uint32_t gq_read(const StageArgs& S, const StageBufs& B, uint32_t q, uint32_t sh, uint32_t t) {
  const uint32_t off = (gq_index(q, sh) * S.ring_cap + (t & (S.ring_cap - 1u))) * 4u;
  const uint32_t tag = gq_tag(S, t);
  for (uint32_t i = 0; i < STAGE_SPIN_LIMIT; ++i) {
    const uint32_t e = ld4(B.rings, off);
    if ((e >> GQ_ID_BITS) == tag) return e & GQ_ID_MASK;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_fetch_or(&S.ctl->error.v, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return ~0u;
}

// per-wave event counts, in LDS so that nothing of them is live across stages
struct StageTotals {
  unsigned long long closest, shadow, internal, leaf, prim, sphere, nan_samples, trip_descend, trip_prim, batches;
  // diagnostics of full-stats launches (VIMG_HIP_DIAG prints them): cycles per stage (0-3 vertex
  // classes, 4 walk, 5 looking for work), batches and slots per stage, rays walked
  unsigned long long cyc[6], nbatch[5], nslots[5];
};
VD unsigned long long wave_sum64(uint32_t v) {
  unsigned long long s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  return s;
}

// ================================================================================ vertex stages
// One batch (<= 64 slots) of one class.  CLS 0 = finishers, 1 = Lambertian, 2 = Principled,
// 3 = any other material.  The body is the vertex stage of render_pool_kernel with the slot
// records in global memory and the pixel turnover through the ready-pixel FIFO.
template <bool TEX, uint32_t CLS>
__device__ __noinline__ void stage_vertex(uint32_t k_lo, uint32_t k_hi, VIMG_LDS StageTotals* tot, uint32_t sh,
                                          uint32_t n, uint32_t first_ticket) {
  const StageKPtr K = stage_kargs(k_lo, k_hi);
  const DScene& g = *(const DScene*)&K->g;
  const RenderArgs& A = *(const RenderArgs*)&K->A;
  const StageArgs& S = *(const StageArgs*)&K->S;
  float* __restrict__ out = K->out;
  const StageBufs B = stage_bufs(S);
  sh = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(sh)));
  n = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(n)));
  first_ticket = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(first_ticket)));
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t W = static_cast<uint32_t>(g.res_x), H = static_cast<uint32_t>(g.res_y);
  const bool single = A.single_x >= 0;
  const uint32_t total_items = single ? 1u : A.num_local_tiles * 64u;
  constexpr uint32_t roulette_threshold = 5;
  const bool material_mode = (A.integrator == VIMG_INTEGRATOR_MATERIAL);
  constexpr bool finisher_batch = (CLS == 0u);
  auto fu = [](float f) { return __float_as_uint(f); };
  auto uf = [](uint32_t u) { return __uint_as_float(u); };

  bool on = lane < n;
  uint32_t slot = 0;
  if (on) {
    slot = gq_read(S, B, CLS, sh, first_ticket + lane);
    if (slot == ~0u) on = false, slot = 0;
  }
  const uint32_t sb = slot * GR_BYTES;
  auto rd = [&](uint32_t r) -> v4u { return ld16(B.slots, sb + r * 16u); };
  auto wr = [&](uint32_t r, v4u v) { st16(B.slots, sb + r * 16u, v); };

  const v4u r_ray = on ? rd(GR_RAY) : v4u{0u, 0u, 0u, 0u};
  uint32_t flags = r_ray.w;
  const bool fresh = on && (flags & SF_FRESH);
  const bool have = on && !fresh;
  uint32_t px = 0, py = 0, smp = 0, item = 0, bounce = 0;
  Rng rng{0};
  f3 acc{0.f, 0.f, 0.f}, ray_o{0.f, 0.f, 0.f}, ray_d{0.f, 0.f, 1.f};
  f3 throughput{1.f, 1.f, 1.f}, result{0.f, 0.f, 0.f};
  RayCone cone{0.f, 0.f};
  float eta_scale = 1.f, prev_pdf = 0.f;
  bool primary = true, non_specular_bounce = false;
  v4u r_origin{0u, 0u, 0u, 0u}, r_hit0{0u, 0u, 0u, 0u}, r_hit1{0u, 0u, 0u, 0u}, r_nee{0u, 0u, 0u, 0u};
  if (have) {
    const v4u r_t = rd(GR_THR), r_r = rd(GR_RES);
    v4u r_a{0u, 0u, 0u, 0u};
    if (finisher_batch) r_a = rd(GR_ACC);
    if (flags & SF_HAS_S) r_nee = rd(GR_NEE);
    r_origin = rd(GR_ORIGIN);
    const v4u r_g = rd(GR_RNG);
    if (flags & (SF_HAS_S | SF_HAS_R)) {   // the slot comes from the walk: its results are current
      r_hit0 = rd(GR_HIT0);
      r_hit1 = rd(GR_HIT1);
      flags |= r_hit1.z & (SF_OCCLUDED | SF_FOUND | SF_KIND_SPHERE);
    }
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
      const v4u r_c = rd(GR_CONE);
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
        hr.e0 = uf(r_hit0.x), hr.e1 = uf(r_hit0.y), hr.e2 = uf(r_hit0.z);
        hr.inv_det = uf(r_hit0.w);
        hr.prim = r_hit1.y;
        hr.kind = (flags & SF_KIND_SPHERE) ? 1u : 0u;
        TravRay tr{ray_o, ray_d, 0.0001f, uf(r_hit1.x)};
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
  // A batch of class 1 holds Lambertian vertices only and one of class 2 Principled ones only,
  // so the shading code exists in a build per material with the type dispatch folded away.
  auto shade_vertex = [&](auto mt_tag) {
    constexpr int MT = decltype(mt_tag)::value;
    // mis_integrator.cpp:45-122.  Draw order: light pick + emitter sample, then sample_mat.
    const uint32_t mat_type = MT >= 0 ? uint32_t(MT) : g.materials[hit.mat].type;
    float hit_dist = 0.f, surface_spread_angle = 0.f;
    if constexpr (TEX) {
      hit_dist = length(ray_o - hit.p);
      surface_spread_angle = spread_angle_from_curvature(hit.curvature, cone.cone_width, ray_d, hit.ns);
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
    // both BSDF evaluations happen before either ray is traced: the evaluation towards the light
    // is pure; its regularisation flag is the one from BEFORE this bounce
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
          // pdf == 0 / NaN: nothing is added, but the reference has traced its shadow ray by then
          // (mis_integrator.cpp:64): it is still traced and counted
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
  if (at_vertex) {
    if constexpr (CLS == 1u)
      shade_vertex(std::integral_constant<int, int(VIMG_MAT_LAMBERTIAN)>{});
    else if constexpr (CLS == 2u)
      shade_vertex(std::integral_constant<int, int(VIMG_MAT_PRINCIPLED)>{});
    else
      shade_vertex(std::integral_constant<int, -1>{});
  }

  // ---- finished samples: accumulate, pixel write-back, pixel turnover, next camera ray
  bool retire = false;
  uint32_t nan_here = 0;
  if constexpr (!finisher_batch) {
    // a path that ended at this vertex (roulette, depth limit, no ray left) is accumulated by the
    // finisher stage: it travels there with neither ray set, which that stage reads as "return
    // bounce_result" (the !SF_HAS_R branch above)
    if (finish) has_s = false, has_r = false;
  } else {
    bool need_ticket = fresh;    // takes a ticket unconditionally (its pixel went back to the FIFO, or the slot is new)
    bool need_perm = false;      // its pixel is done: takes a ticket only if an unstarted pixel is left
    bool push_item = false;
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
        need_perm = true;
      } else if (smp % S.seg_len == 0u) {
        // end of a segment: the pixel rests in its record and its id in the FIFO of ready pixels
        st16(B.pix_state, item * 32u, v4u{static_cast<uint32_t>(rng.s), static_cast<uint32_t>(rng.s >> 32), smp, 0u});
        st16(B.pix_state, item * 32u + 16u, v4u{fu(acc.x), fu(acc.y), fu(acc.z), 0u});
        push_item = true;
        need_ticket = true;
      }
    }
    if (__any(push_item)) {
      drain_vmem();   // the records are in memory before the id can be seen
      const unsigned long long mask = __ballot(push_item);
      const uint32_t leader = first_lane_of(mask);
      uint32_t base = 0;
      if (lane == leader) base = ctl_add(&S.ctl->tk_tail, static_cast<uint32_t>(__popcll(mask)));
      base = bcast(base, leader);
      if (push_item) {
        const uint32_t i = base + lane_rank(mask, lane);
        const uint32_t tag = ((i >> S.pix_shift) & 15u) + 1u;
        st4(B.pix_ring, (i & (S.pix_cap - 1u)) * 4u, (tag << PIX_TAG_SHIFT) | item);
      }
    }
    // tickets: repeated while some lane drew an off-image item of a ragged tile (such an item
    // counts as a pixel that is done at once)
    while (__any(need_ticket || need_perm)) {
      const unsigned long long mp = __ballot(need_perm);
      if (mp != 0ull) {
        const uint32_t k = static_cast<uint32_t>(__popcll(mp));
        const uint32_t leader = first_lane_of(mp);
        uint32_t granted = 0;
        if (lane == leader) {
          const int32_t old = static_cast<int32_t>(
              __hip_atomic_fetch_sub(&S.ctl->surplus.v, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          granted = old <= 0 ? 0u : (static_cast<uint32_t>(old) < k ? static_cast<uint32_t>(old) : k);
          if (granted < k) ctl_add(&S.ctl->surplus, k - granted);
        }
        granted = bcast(granted, leader);
        if (need_perm) {
          if (lane_rank(mp, lane) < granted)
            need_ticket = true;
          else
            retire = true;
          need_perm = false;
        }
      }
      const unsigned long long mt = __ballot(need_ticket);
      if (mt != 0ull) {
        const uint32_t leader = first_lane_of(mt);
        uint32_t base = 0;
        if (lane == leader) base = ctl_add(&S.ctl->tk_head, static_cast<uint32_t>(__popcll(mt)));
        base = bcast(base, leader);
        if (need_ticket) {
          need_ticket = false;
          const uint32_t t = base + lane_rank(mt, lane);
          if (t < total_items) {
            // one of the first W*H tickets: the pixel itself, never started
            item = t;
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
              need_perm = true;   // off the image: done at once
            } else {
              const uint64_t image_index = uint64_t(px) + uint64_t(H - 1 - py) * W;
              pcg_seed(rng, image_index);
              smp = 0;
              acc = f3{0.f, 0.f, 0.f};
            }
          } else {
            // a pixel some slot has put back: its ring entry is reserved (every such ticket is
            // taken after a push), at worst still being written
            const uint32_t i = t - total_items;
            const uint32_t off = (i & (S.pix_cap - 1u)) * 4u;
            const uint32_t tag = ((i >> S.pix_shift) & 15u) + 1u;
            uint32_t e = 0;
            bool got = false;
            for (uint32_t spin = 0; spin < STAGE_SPIN_LIMIT; ++spin) {
              e = ld4(B.pix_ring, off);
              if ((e >> PIX_TAG_SHIFT) == tag) {
                got = true;
                break;
              }
              __builtin_amdgcn_s_sleep(2);
            }
            if (!got) {
              __hip_atomic_fetch_or(&S.ctl->error.v, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              retire = true;
            } else {
              item = e & ((1u << PIX_TAG_SHIFT) - 1u);
              const v4u p0 = ld16(B.pix_state, item * 32u), p1 = ld16(B.pix_state, item * 32u + 16u);
              rng.s = uint64_t(p0.x) | (uint64_t(p0.y) << 32);
              smp = p0.z;
              acc = f3{uf(p1.x), uf(p1.y), uf(p1.z)};
              if (single) {
                px = static_cast<uint32_t>(A.single_x), py = static_cast<uint32_t>(A.single_y);
              } else {
                const uint32_t tile = (item >> 6) * A.tile_world + A.tile_rank;
                const uint32_t within = item & 63u;
                const uint32_t tx = tile / A.tiles_y, ty = tile - tx * A.tiles_y;
                px = tx * 8 + (within & 7u);
                py = ty * 8 + (within >> 3);
              }
            }
          }
        }
      }
    }
    const bool regen = on && !retire && (finish || fresh);
    if (regen) {
      const f2 off = random_x_y_r2(px + py + smp);
      // right-to-left argument evaluation of the reference's call
      const float rand2 = rand_float(rng);
      const float rand1 = rand_float(rng);
      generate_ray(g, static_cast<float>(px) + off.x, static_cast<float>(py) + off.y, rand1, rand2, ray_o, ray_d);
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
  }

  // ---- registers -> slot records, slot -> next queue
  const bool keep = on && !retire;
  if (keep) {
    const uint32_t nf = (primary ? SF_PRIMARY : 0u) | (non_specular_bounce ? SF_NONSPEC : 0u) |
                        (has_s ? SF_HAS_S : 0u) | (has_r ? SF_HAS_R : 0u) | (bounce << SF_BOUNCE_SHIFT);
    wr(GR_ORIGIN, v4u{fu(ray_o.x), fu(ray_o.y), fu(ray_o.z), fu(shadow_max_t)});
    wr(GR_RAY, v4u{fu(ray_d.x), fu(ray_d.y), fu(ray_d.z), nf});
    if (has_s) wr(GR_SHADOW, v4u{fu(shadow_d.x), fu(shadow_d.y), fu(shadow_d.z), 0u});
    wr(GR_THR, v4u{fu(throughput.x), fu(throughput.y), fu(throughput.z), fu(eta_scale)});
    wr(GR_RES, v4u{fu(result.x), fu(result.y), fu(result.z), fu(prev_pdf)});
    if (has_s) wr(GR_NEE, v4u{fu(nee_contrib.x), fu(nee_contrib.y), fu(nee_contrib.z), 0u});
    wr(GR_RNG, v4u{static_cast<uint32_t>(rng.s), static_cast<uint32_t>(rng.s >> 32), px | (py << 16), smp});
    if constexpr (finisher_batch) wr(GR_ACC, v4u{fu(acc.x), fu(acc.y), fu(acc.z), item});
    if constexpr (TEX) wr(GR_CONE, v4u{fu(cone.cone_width), fu(cone.spread_angle), 0u, 0u});
  }
  drain_vmem();
  const bool to_walk = keep && (has_s || has_r), to_fin = keep && !to_walk;
  gq_push(S, B, GQ_WALK, sh, to_walk, slot, lane);
  gq_push(S, B, GQ_FIN, sh, to_fin, slot, lane);
  {
    const unsigned long long mr = __ballot(on && retire);
    if (mr != 0ull && lane == first_lane_of(mr)) ctl_add(&S.ctl->retired, static_cast<uint32_t>(__popcll(mr)));
  }
  if (__any(nan_here != 0u)) {
    const unsigned long long c = wave_sum64(nan_here);
    if (lane == 0) tot->nan_samples += c;
  }
}

// ================================================================================ walk stage
// A session of the walk: the wave stages up to `wchunk` slot ids of Q_walk in an LDS ring, its lanes
// take rays from it as they finish (shadow ray of a vertex first, then its path ray), results go
// to the slots' HIT records and the slot ids to per-class LDS lists that are flushed to the vertex
// queues in batches.  The session ends when Q_walk has nothing more for it or its quota is used up.
template <bool TEX, bool DEEP>
__device__ __noinline__ void stage_walk(uint32_t k_lo, uint32_t k_hi, VIMG_LDS unsigned char* lds_base,
                                        VIMG_LDS uint32_t* lq, const VIMG_LDS v4f* lds_leaf,
                                        VIMG_LDS StageTotals* tot, uint32_t sh0) {
  const StageKPtr K = stage_kargs(k_lo, k_hi);
  const DScene& g = *(const DScene*)&K->g;
  const RenderArgs& A = *(const RenderArgs*)&K->A;
  const StageArgs& S = *(const StageArgs*)&K->S;
  const StageBufs B = stage_bufs(S);
  const Lds L = lds_layout(A, lds_base);
  VIMG_LDS uint32_t* cq = lq + S.wchunk;
  sh0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(sh0)));
  const uint32_t lane = threadIdx.x & 63u;
  const bool full_stats = A.full_stats != 0;
  const uint32_t stat_inc = full_stats ? 1u : 0u;
  const bool material_mode = (A.integrator == VIMG_INTEGRATOR_MATERIAL);
  const bool leaf_in_lds = A.lds_leaf != 0u;
  const uint32_t box_min = A.pool_boxmin;
  const uint32_t CH = S.wchunk;   // capacity of lq and of each class list
  auto fu = [](float f) { return __float_as_uint(f); };
  auto uf = [](uint32_t u) { return __uint_as_float(u); };
  auto ringi = [&](uint32_t i) { return i >= CH ? i - CH : i; };

  Counters cnt{0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t lq_head = 0, lq_count = 0;
  uint32_t cq_count0 = 0, cq_count1 = 0, cq_count2 = 0, cq_count3 = 0;
  uint32_t taken = 0;          // slots this session has taken from Q_walk
  bool source_dry = false;     // Q_walk had nothing at the last look, or the quota is used up
  uint32_t sh = sh0;

  uint32_t w_slot = SLOT_IDLE, w_phase = 0, w_flags = 0, w_cls = 0;
  bool w_setup = false, w_any = false, w_found = false, w_exact = false;
  TravRay ray{f3{0.f, 0.f, 0.f}, f3{0.f, 0.f, 1.f}, 0.0001f, VIMG_INF};
  f3 w_inv{1.f, 1.f, 1.f}, w_path_d{0.f, 0.f, 1.f}, w_shadow_d{0.f, 0.f, 1.f};
  float w_shadow_max_t = 0.f;
  TriRayConst rc{0.f, 0.f, 1.f, 2};
  float w_dir_len2 = 1.f;
  uint32_t sp = 0, cur = REF_DONE;
  HitRec rec;
  rec.prim = 0xffffffffu;
  rec.kind = 0;
  rec.e0 = rec.e1 = rec.e2 = rec.inv_det = 0.f;

  // flush of one class list to its vertex queue (the HIT records of its slots are in memory)
  auto flush = [&](uint32_t cls, uint32_t& count) {
    if (count == 0u) return;
    drain_vmem();
    for (uint32_t base = 0; base < count; base += 64u) {
      const bool want = base + lane < count;
      const uint32_t id = want ? cq[cls * CH + base + lane] : 0u;
      gq_push(S, B, cls, sh0, want, id, lane);
    }
    count = 0;
  };

  for (;;) {
    // (0) top up the staged chunk when it runs low and lanes would otherwise idle
    {
      const uint32_t n_idle = static_cast<uint32_t>(__popcll(__ballot(w_slot == SLOT_IDLE)));
      if (!source_dry && lq_count < n_idle) {
        uint32_t room = CH - lq_count;
        if (taken + room > S.walk_quota) room = S.walk_quota > taken ? S.walk_quota - taken : 0u;
        uint32_t first = 0, n = 0;
        if (room != 0u) {
          n = gq_claim(S, GQ_WALK, sh, room, lane, first);
          if (n == 0u) {   // own shard empty: one look at the neighbours
            for (uint32_t k = 1; k < GQ_SHARDS && n == 0u; ++k) {
              const uint32_t s2 = (sh0 + k) & (GQ_SHARDS - 1u);
              n = gq_claim(S, GQ_WALK, s2, room, lane, first);
              if (n != 0u) sh = s2;
            }
          }
        }
        if (n == 0u) {
          source_dry = true;
        } else {
          for (uint32_t i = lane; i < n; i += 64u) {
            const uint32_t id = gq_read(S, B, GQ_WALK, sh, first + i);
            lq[ringi(lq_head + lq_count + i)] = id;
          }
          lq_count += n;
          taken += n;
          if (taken >= S.walk_quota) source_dry = true;
          sh = sh0;
        }
      }
    }
    // (1) idle lanes take staged slots
    {
      const bool idle = (w_slot == SLOT_IDLE);
      const unsigned long long mask = __ballot(idle);
      const uint32_t n_idle = static_cast<uint32_t>(__popcll(mask));
      const uint32_t take = n_idle < lq_count ? n_idle : lq_count;
      if (take) {
        const uint32_t r = lane_rank(mask, lane);
        if (idle && r < take) {
          const uint32_t id = lq[ringi(lq_head + r)];
          if (id != ~0u) {   // (~0u: an entry the watchdog gave up on)
            w_slot = id;
            const uint32_t sb = w_slot * GR_BYTES;
            const v4u ro = ld16(B.slots, sb + GR_ORIGIN * 16u);
            const v4u rr = ld16(B.slots, sb + GR_RAY * 16u);
            w_flags = rr.w & ~(SF_OCCLUDED | SF_FOUND | SF_KIND_SPHERE);
            ray.o = f3{uf(ro.x), uf(ro.y), uf(ro.z)};
            w_shadow_max_t = uf(ro.w);
            w_path_d = f3{uf(rr.x), uf(rr.y), uf(rr.z)};
            if (w_flags & SF_HAS_S) {
              const v4u rs = ld16(B.slots, sb + GR_SHADOW * 16u);
              w_shadow_d = f3{uf(rs.x), uf(rs.y), uf(rs.z)};
            }
            w_phase = (w_flags & SF_HAS_S) ? 0u : 1u;
            w_setup = true;
          }
        }
        lq_head = ringi(lq_head + take);
        lq_count -= take;
      }
    }
    // (2) ray set-up (reference include/bvh.h:109-143): everything derived from the ray alone
    if (__any(w_setup)) {
      if (w_setup) {
        if (w_phase == 0) {
          ray.d = w_shadow_d;
          ray.max_t = w_shadow_max_t;
          w_any = true;
          cnt.shadow++;
        } else {
          ray.d = w_path_d;
          ray.max_t = VIMG_INF;
          w_any = false;
          cnt.closest++;
        }
        w_inv = f3{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
        w_exact = (ray.d.x == 0.f) || (ray.d.y == 0.f) || (ray.d.z == 0.f);
        rc = tri_ray_const(ray.d);
        w_dir_len2 = dot(ray.d, ray.d);
        const float root = slab(load3k(g.root_min), load3k(g.root_max), ray.o, w_inv, ray.min_t, ray.max_t);
        cur = is_inf(root) ? REF_DONE : g.root_ref;
        sp = 0;
        w_found = false;
        rec.prim = 0xffffffffu;
        w_setup = false;
      }
    }
    if (!__any(w_slot != SLOT_IDLE)) {
      if (lq_count == 0u && source_dry) break;
      continue;
    }
    // (3) walk until a quarter of the wave has a finished ray (or nothing is left to walk)
    for (;;) {
      auto box_loop = [&](auto exact_possible) {
        while (cur != REF_DONE && ref_count(cur) == 0) {
          v4f na, nb, nc;
          v2u refs;
          if (!DEEP || cur < L.n_nodes) {   // the build for trees that fit has every node in LDS
            na = L.na[cur], nb = L.nb[cur], nc = L.nc[cur];
            refs = L.nm[cur];
          } else {
            gptr<DNode> nd = g.nodes + cur;
            na = nd->a, nb = nd->b, nc = nd->c;
            refs = v2u{nd->left_ref, nd->right_ref};
          }
          const uint32_t sp_below = sp != 0 ? sp - 1 : 0u;
          const uint32_t popped = L.stack[sp_below * 64];
          cnt.internal += stat_inc;
          float h1, h2;
          if (decltype(exact_possible)::value && w_exact) {
            h1 = slab(f3{na.x, na.y, na.z}, f3{na.w, nb.x, nb.y}, ray.o, w_inv, ray.min_t, ray.max_t);
            h2 = slab(f3{nb.z, nb.w, nc.x}, f3{nc.y, nc.z, nc.w}, ray.o, w_inv, ray.min_t, ray.max_t);
          } else {
            h1 = slab_fast(f3{na.x, na.y, na.z}, f3{na.w, nb.x, nb.y}, ray.o, w_inv, ray.min_t, ray.max_t);
            h2 = slab_fast(f3{nb.z, nb.w, nc.x}, f3{nc.y, nc.z, nc.w}, ray.o, w_inv, ray.min_t, ray.max_t);
          }
          const bool in1 = !is_inf(h1), in2 = !is_inf(h2);
          const uint32_t c1 = refs.x, c2 = refs.y;
          // branch-free step (render_pool_kernel.h): the entry a pop would return was read before
          // the box test; the far child is written above the top of the stack whether it is kept
          // or not, and sp moves by select
          const bool both = in1 && in2, any = in1 || in2;
          const bool first_is_near = w_any ? false : (h2 > h1);
          const uint32_t near_c = first_is_near ? c1 : c2;
          const uint32_t far_c = first_is_near ? c2 : c1;
          L.stack[sp * 64] = far_c;
          const uint32_t one_c = in1 ? c1 : c2;
          cur = both ? near_c : (any ? one_c : (sp != 0 ? popped : REF_DONE));
          sp = both ? sp + 1 : (any ? sp : sp_below);
          if constexpr (DEEP) {
            if (__popcll(__ballot(cur != REF_DONE && ref_count(cur) == 0)) < box_min) break;
          }
        }
      };
      if (__any(w_exact && w_slot != SLOT_IDLE))
        box_loop(std::true_type{});
      else
        box_loop(std::false_type{});
      if (cur != REF_DONE && (!DEEP || ref_count(cur) != 0)) {
        const uint32_t first = ref_index(cur), count = ref_count(cur);
        cnt.leaf += stat_inc;
        bool stop = false;
        for (uint32_t i = 0; i < count && !stop; ++i) {
          gptr<DLeafPrim> lp = g.leaf_prims + (first + i);
          v4f a, b, c;
          if (leaf_in_lds) {
            const VIMG_LDS v4f* ll = lds_leaf + (first + i) * 3u;
            a = ll[0], b = ll[1], c = ll[2];
          } else {
            a = lp->a, b = lp->b;
            c = reinterpret_cast<gptr<v4f>>(lp)[2];
          }
          const float c0 = c.x;
          const uint32_t lp_prim = __float_as_uint(c.y), kind = __float_as_uint(c.z),
                         lp_cls = __float_as_uint(c.w);   // DLeafPrim: c0 | prim | kind | cls
          cnt.prim += stat_inc;
          bool hit = false;
          float t = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f, idet = 0.f;
          if (kind == 0) {
            hit = tri_test_flat(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c0}, ray, rc, t, e0, e1, e2, idet);
          } else if (kind == 1) {
            cnt.sphere += stat_inc;
            hit = sphere_test(f3{a.x, a.y, a.z}, a.w, ray, w_dir_len2, t);
          }
          ray.max_t = hit ? t : ray.max_t;
          w_found = w_found || hit;
          stop = hit && w_any;
          const bool keep_rec = hit && !w_any;
          rec.e0 = keep_rec ? e0 : rec.e0, rec.e1 = keep_rec ? e1 : rec.e1;
          rec.e2 = keep_rec ? e2 : rec.e2, rec.inv_det = keep_rec ? idet : rec.inv_det;
          rec.prim = keep_rec ? lp_prim : rec.prim;
          rec.kind = keep_rec ? kind : rec.kind;
          w_cls = keep_rec ? lp_cls : w_cls;
        }
        const uint32_t sp_below = sp != 0 ? sp - 1 : 0u;
        const uint32_t popped = L.stack[sp_below * 64];
        cur = (stop || sp == 0) ? REF_DONE : popped;
        sp = sp_below;
      }
      const uint32_t n_fin = static_cast<uint32_t>(__popcll(__ballot(w_slot != SLOT_IDLE && cur == REF_DONE)));
      const uint32_t n_act = static_cast<uint32_t>(__popcll(__ballot(w_slot != SLOT_IDLE && cur != REF_DONE)));
      if (n_act == 0 || n_fin >= A.pool_refill) break;
    }
    // (4) retire finished rays: second ray of the item, or hand the slot to a vertex queue
    bool done_item = false;
    if (w_slot != SLOT_IDLE && cur == REF_DONE) {
      if (w_phase == 0) {
        if (w_found) w_flags |= SF_OCCLUDED;
        if (w_flags & SF_HAS_R) {
          w_phase = 1;
          w_setup = true;
        } else {
          done_item = true;
        }
      } else {
        if (w_found) w_flags |= SF_FOUND | (rec.kind == 1 ? SF_KIND_SPHERE : 0u);
        done_item = true;
      }
      if (done_item) {
        const uint32_t sb = w_slot * GR_BYTES;
        if (w_flags & SF_FOUND) st16(B.slots, sb + GR_HIT0 * 16u, v4u{fu(rec.e0), fu(rec.e1), fu(rec.e2), fu(rec.inv_det)});
        st16(B.slots, sb + GR_HIT1 * 16u,
             v4u{fu(ray.max_t), rec.prim, w_flags & (SF_OCCLUDED | SF_FOUND | SF_KIND_SPHERE), 0u});
      }
    }
    {
      // class of the vertex queue this slot joins: 0 = its path ends (miss, no path ray, emitter hit
      // under mis, any hit under the normal integrators), else the material class of the vertex
      uint32_t cls = 0;
      if (done_item && (w_flags & SF_FOUND) && A.integrator >= VIMG_INTEGRATOR_MATERIAL) {
        cls = w_cls;
        if (cls == 0 && material_mode) cls = 3;   // material_integrator shades emitters too
      }
      const unsigned long long m0 = __ballot(done_item && cls == 0), m1 = __ballot(done_item && cls == 1),
                               m2 = __ballot(done_item && cls == 2), m3 = __ballot(done_item && cls == 3);
      if (done_item) {
        if (cls == 0) cq[cq_count0 + lane_rank(m0, lane)] = w_slot;
        else if (cls == 1) cq[CH + cq_count1 + lane_rank(m1, lane)] = w_slot;
        else if (cls == 2) cq[2 * CH + cq_count2 + lane_rank(m2, lane)] = w_slot;
        else cq[3 * CH + cq_count3 + lane_rank(m3, lane)] = w_slot;
        w_slot = SLOT_IDLE;
      }
      cq_count0 += static_cast<uint32_t>(__popcll(m0));
      cq_count1 += static_cast<uint32_t>(__popcll(m1));
      cq_count2 += static_cast<uint32_t>(__popcll(m2));
      cq_count3 += static_cast<uint32_t>(__popcll(m3));
      // a list that could overflow at the next retire is flushed now; full batches go out early
      if (cq_count0 >= 64u) flush(0u, cq_count0);
      if (cq_count1 >= 64u) flush(1u, cq_count1);
      if (cq_count2 >= 64u) flush(2u, cq_count2);
      if (cq_count3 >= 64u) flush(3u, cq_count3);
    }
  }
  flush(0u, cq_count0);
  flush(1u, cq_count1);
  flush(2u, cq_count2);
  flush(3u, cq_count3);
  if (tot) {
    const unsigned long long c0 = wave_sum64(cnt.closest), c1 = wave_sum64(cnt.shadow);
    if (lane == 0) tot->closest += c0, tot->shadow += c1, tot->nslots[4] += taken;
    if (full_stats) {
      const unsigned long long c2 = wave_sum64(cnt.internal), c3 = wave_sum64(cnt.leaf), c4 = wave_sum64(cnt.prim),
                               c5 = wave_sum64(cnt.sphere);
      if (lane == 0) tot->internal += c2, tot->leaf += c3, tot->prim += c4, tot->sphere += c5;
    }
  }
}

// ================================================================================ the kernel
template <bool TEX, bool DEEP>
__global__ void __launch_bounds__(256, 4)
render_stage_kernel(const StageKArgs* __restrict__ kargs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t k_lo = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(kargs)),
                 k_hi = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(kargs) >> 32);
  const StageKPtr K = stage_kargs(k_lo, k_hi);
  const DScene& g = *(const DScene*)&K->g;
  const RenderArgs& A = *(const RenderArgs*)&K->A;
  const StageArgs& S = *(const StageArgs*)&K->S;
  DeviceStats* __restrict__ stats = K->stats;
  stage_lds(g, A, (VIMG_LDS unsigned char*)lds_raw);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const StageBufs B = stage_bufs(S);
  // wave-private LDS behind the node planes and the four traversal stacks: staged chunk of Q_walk,
  // four class lists, event totals; then (small scenes) the leaf records of the whole scene
  VIMG_LDS uint32_t* lq;
  VIMG_LDS StageTotals* tot;
  VIMG_LDS v4f* lds_leaf;
  {
    const uint32_t node_bytes = (lds_node_bytes(A.lds_nodes) + 255u) & ~255u;
    const uint32_t stack_bytes = 4u * A.stack_entries * 64u * 4u;
    const uint32_t per_wave = 5u * S.wchunk * 4u + 256u;   // bytes
    VIMG_LDS unsigned char* base = (VIMG_LDS unsigned char*)lds_raw + node_bytes + stack_bytes;
    lq = reinterpret_cast<VIMG_LDS uint32_t*>(base + wave * per_wave);
    tot = reinterpret_cast<VIMG_LDS StageTotals*>(lq + 5u * S.wchunk);
    lds_leaf = reinterpret_cast<VIMG_LDS v4f*>(base + 4u * per_wave);
    for (uint32_t i = threadIdx.x; i < A.lds_leaf * 3u; i += blockDim.x)
      lds_leaf[i] = reinterpret_cast<gptr<v4f>>(g.leaf_prims)[i];
    if (lane < sizeof(StageTotals) / 4u) reinterpret_cast<VIMG_LDS uint32_t*>(tot)[lane] = 0u;
    __syncthreads();
  }
  const uint32_t sh0 = blockIdx.x & (GQ_SHARDS - 1u);

  // ---- every slot starts "fresh" in the finisher queue: this wave's share of them
  {
    const uint32_t gw = blockIdx.x * 4u + wave, n_waves = gridDim.x * 4u;
    const uint32_t per = (S.n_slots + n_waves - 1u) / n_waves;
    const uint32_t lo = gw * per < S.n_slots ? gw * per : S.n_slots;
    const uint32_t hi = lo + per < S.n_slots ? lo + per : S.n_slots;
    for (uint32_t base = lo; base < hi; base += 64u) {
      const bool want = base + lane < hi;
      if (want) st16(B.slots, (base + lane) * GR_BYTES + GR_RAY * 16u, v4u{0u, 0u, 0u, SF_FRESH});
      drain_vmem();
      gq_push(S, B, GQ_FIN, sh0, want, base + lane, lane);
    }
  }

  uint32_t idle_polls = 0;
  const bool diag = stats && A.full_stats;
  unsigned long long t_mark = diag ? __builtin_readcyclecounter() : 0ull;
  auto lap = [&](uint32_t k) {
    if (diag) {
      const unsigned long long now = __builtin_readcyclecounter();
      if (lane == 0) tot->cyc[k] += now - t_mark;
      t_mark = now;
    }
  };
  for (;;) {
    // ---- look at the queues of this workgroup's shard (lanes 0..4: one queue each), then at all
    uint32_t q_pick = GQ_COUNT, sh_pick = sh0, avail_pick = 0;
    {
      uint32_t a = 0;
      if (lane < GQ_COUNT) {
        const uint32_t t = ctl_load(&S.ctl->q_tail[gq_index(lane, sh0)]), h = ctl_load(&S.ctl->q_head[gq_index(lane, sh0)]);
        a = static_cast<int32_t>(t - h) > 0 ? t - h : 0u;
      }
#pragma unroll
      for (uint32_t q = 0; q < GQ_COUNT; ++q) {
        const uint32_t aq = bcast(a, q);
        if (aq > avail_pick) avail_pick = aq, q_pick = q;
      }
    }
    if (q_pick == GQ_COUNT) {
      uint32_t a = 0;
      if (lane < GQ_COUNT * GQ_SHARDS) {
        const uint32_t t = ctl_load(&S.ctl->q_tail[lane]), h = ctl_load(&S.ctl->q_head[lane]);
        a = static_cast<int32_t>(t - h) > 0 ? t - h : 0u;
      }
      const unsigned long long nonempty = __ballot(a != 0u);
      if (nonempty != 0ull) {
        // the first non-empty (queue, shard) after this workgroup's own shard
        const uint32_t rot = sh0;
        uint32_t best = 64u;
        for (uint32_t k = 1; k <= GQ_SHARDS && best == 64u; ++k) {
          const uint32_t s2 = (rot + k) & (GQ_SHARDS - 1u);
          for (uint32_t q = 0; q < GQ_COUNT; ++q)
            if ((nonempty >> (q * GQ_SHARDS + s2)) & 1ull) { best = q * GQ_SHARDS + s2; break; }
        }
        if (best != 64u) q_pick = best / GQ_SHARDS, sh_pick = best % GQ_SHARDS, avail_pick = bcast(a, best);
      }
    }
    if (q_pick == GQ_COUNT) {
      // nothing queued anywhere: done when every slot has retired
      uint32_t done = 0;
      if (lane == 0) {
        const uint32_t r = ctl_load(&S.ctl->retired), e = ctl_load(&S.ctl->error);
        done = (r >= S.n_slots || e != 0u) ? 1u : 0u;
      }
      if (bcast(done, 0) != 0u) break;
      // Watchdog: paths are in flight on other waves, i.e. this lasts a stage's time.  A wave that
      // has looked a million times in a row is waiting for something that will not come: it
      // raises the error word and every wave leaves, so that a scheduling bug ends as
      // VIMG_E_DEVICE instead of a hung GPU.
      if (++idle_polls > (1u << 20)) {
        if (lane == 0) __hip_atomic_fetch_or(&S.ctl->error.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(16);
      continue;
    }
    idle_polls = 0;
    lap(5);
    if (q_pick == GQ_WALK) {
      stage_walk<TEX, DEEP>(k_lo, k_hi, (VIMG_LDS unsigned char*)lds_raw, lq, lds_leaf, stats ? tot : nullptr, sh_pick);
      lap(4);
      if (diag && lane == 0) tot->nbatch[4] += 1;
    } else {
      uint32_t first = 0;
      const uint32_t n = gq_claim(S, q_pick, sh_pick, 64u, lane, first);
      if (n != 0u) {
        if (q_pick == GQ_FIN) stage_vertex<TEX, 0u>(k_lo, k_hi, tot, sh_pick, n, first);
        else if (q_pick == GQ_LAMB) stage_vertex<TEX, 1u>(k_lo, k_hi, tot, sh_pick, n, first);
        else if (q_pick == GQ_PRIN) stage_vertex<TEX, 2u>(k_lo, k_hi, tot, sh_pick, n, first);
        else stage_vertex<TEX, 3u>(k_lo, k_hi, tot, sh_pick, n, first);
        lap(q_pick);
        if (diag && lane == 0) tot->nbatch[q_pick] += 1, tot->nslots[q_pick] += n;
      }
    }
    if (stats && lane == 0) tot->batches += 1;
  }

  // ---- flush event counts: one atomic per wave and counter
  if (stats && lane == 0) {
    atomicAdd(&stats->closest, tot->closest);
    atomicAdd(&stats->shadow, tot->shadow);
    if (A.full_stats) {
      atomicAdd(&stats->internal, tot->internal);
      atomicAdd(&stats->leaf, tot->leaf);
      atomicAdd(&stats->prim, tot->prim);
      atomicAdd(&stats->sphere, tot->sphere);
      atomicAdd(&stats->iterations, tot->batches);
      for (int k = 0; k < 6; ++k) atomicAdd(&stats->prof[k], tot->cyc[k]);
      for (int k = 0; k < 5; ++k) atomicAdd(&stats->prof[6 + k], tot->nbatch[k]), atomicAdd(&stats->prof[11 + k], tot->nslots[k]);
    }
    if (tot->nan_samples) atomicAdd(&stats->nan_samples, tot->nan_samples);
  }
}

}  // namespace vimg
