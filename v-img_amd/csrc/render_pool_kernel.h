// Pooled scheduler of the render path: every wave owns a POOL of about 200 path slots (three per
// lane) instead of binding one path to one lane.
//
// Why: in render_kernel a lane is idle whenever its own path does not need the phase the wave is
// in (31 % of lanes in the shadow walk, ~30 % in the vertex phase), and a path whose BVH walk is
// long holds its lane's next vertex back.  Here paths are decoupled from lanes:
//   * slot state is split by who touches it: the records the WALK reads and writes (origin, path
//     ray + flags, shadow ray / hit barycentrics), the RNG record and the primitive id live in
//     LDS, SoA [record][slot]; the records the vertex stage touches late (throughput, result, NEE
//     term, pixel accumulator, cone) live in a per-wave region of global memory, [slot][record]
//     (L2 / Infinity-Cache resident, read and written once per batch).  Keeping the cold 64 bytes
//     out of LDS takes a wave from 119 to 201 slots, and the measured gain per slot is steep
//     (72 slots 4.7, 104 slots 6.0, 119 slots 6.4, 201 slots 6.9 Grays/s on config 2);
//   * wave-private queues hold slot ids: Q_walk (rays ready to be walked) and four Q_vertex rings
//     (walks finished, by class).  The queues are used by one wave only, so pushing and popping is
//     __ballot/popcount arithmetic on wave-uniform counters — no atomics, no waiting;
//   * the VERTEX stage runs on batches of up to 64 slots of ONE class, known when the walk ends from
//     the flags and the material class baked into the leaf record: class 0 = "finishers" (miss,
//     emitter hit, failed BSDF sample, roulette death: NEE result, emitter / miss MIS weight, pixel
//     accumulation, next work item, next camera ray), classes 1-3 = path vertices by material (hit
//     record, roulette, light sample, BSDF sample, both BSDF evaluations).  A path that ends inside
//     a vertex batch is handed to the finisher queue, so every stage runs with its lanes full;
//   * the WALK stage is a persistent while-while loop whose lanes refill from Q_walk as soon as a
//     quarter of them is done (shadow ray first, then the path ray of the same vertex), so the box
//     loop and the primitive loop keep their lanes busy;
//   * a pixel is bound to a slot for one SEGMENT of its samples only (a sixteenth of them on
//     config 2): the work items of the global counter are (segment, pixel) pairs in segment-major
//     order, and between segments the pixel's state (RNG, accumulator) rests in a per-pixel record
//     in global memory, written and read word by word with agent-scope atomics (data words, then
//     the tag) by whichever slot draws the next segment.  A frame then ends with the ragged tail
//     of one segment instead of one whole pixel (config 2 with whole pixels: every wave's last
//     pixels ran 512 samples in a draining pool, a quarter of the frame at falling efficiency and
//     10 % spread between waves).  An item whose predecessor segment is still in flight is not
//     waited for: the slot keeps its claim and looks again at its next turn.
// Every path still executes exactly the reference's operations in the reference's order with its
// own RNG stream, so results are bit-identical to render_kernel and to the oracle.
#pragma once
#include "sched_common.h"

namespace vimg {

// stage timers of the -DVIMG_PROFILE build (tools/stage_profile.py); nothing in the product build
#ifdef VIMG_PROFILE
#define PROF_DECL unsigned long long prof_acc[PF_COUNT] = {}; unsigned long long prof_t = __builtin_readcyclecounter(); const unsigned long long prof_t0 = prof_t;
#define PROF_LAP(k) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[k] += now_ - prof_t; prof_t = now_; }
#define PROF_ADD(k, v) { prof_acc[k] += (v); }
#define PROF_NOW() __builtin_readcyclecounter()
#else
#define PROF_NOW() 0ull
#define PROF_DECL
#define PROF_LAP(k)
#define PROF_ADD(k, v)
#endif

// DEEP: the tree does not fit the LDS node cache (deep, memory-resident trees): the box loop hands
// over to the leaf loop as soon as fewer than pool_boxmin lanes still descend.  On trees that sit
// in LDS the test costs more than it brings (config 2: -3 %), hence a build without it.
template <bool TEX, int WPS, bool DEEP>
__global__ void __launch_bounds__(256, WPS)
render_pool_kernel(const DScene g, const RenderArgs A, float* __restrict__ out,
                   DeviceStats* __restrict__ stats, unsigned int* __restrict__ work_counter) {
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
  const uint32_t P = A.pool_slots;
  constexpr uint32_t NCOLD = TEX ? SC_COUNT : SC_COUNT - 1u;   // SC_CONE (last) only in the textured build
  // cold records of this wave, behind the regions of the waves before it: [slot][record], so the
  // records of a slot share one 64-byte line (one L2 request per lane and batch instead of four)
  VIMG_GLOBAL v4u* cold = A.pool_cold + (size_t(blockIdx.x) * 4u + wave) * (size_t(NCOLD) * P);
  auto crd = [&](uint32_t r, uint32_t slot) -> v4u { return cold[slot * NCOLD + r]; };
  auto cwr = [&](uint32_t r, uint32_t slot, v4u v) { cold[slot * NCOLD + r] = v; };

  // LDS carve-out of this wave behind the node planes and the four traversal stacks
  VIMG_LDS uint32_t* pool;
  VIMG_LDS uint8_t* q_walk;
  VIMG_LDS v4f* lds_leaf;        // copy of leaf_prims (all of them) when A.lds_leaf != 0
  VIMG_LDS uint32_t* q_prim;     // primitive id of the hit, per slot
  VIMG_LDS uint8_t* q_vertex;    // four rings of capacity P: class 0 finishers, 1 Lambertian (+rest), 2 Principled, 3 other
  {
    const uint32_t node_bytes = (lds_node_bytes(A.lds_nodes) + 255u) & ~255u;
    const uint32_t stack_bytes = 4u * A.stack_entries * 64u * 4u;
    const uint32_t per_wave = pool_wave_bytes(P) / 4u;   // in dwords
    VIMG_LDS uint32_t* base =
        reinterpret_cast<VIMG_LDS uint32_t*>((VIMG_LDS unsigned char*)lds_raw + node_bytes + stack_bytes);
    pool = base + wave * per_wave;
    q_prim = pool + SR_COUNT * 4u * P;
    q_walk = reinterpret_cast<VIMG_LDS uint8_t*>(q_prim + P);
    q_vertex = q_walk + P;
    // small scenes: the leaf records behind the four pools (an LDS read instead of an L1 hit per
    // primitive test; the walk is a chain of dependent loads at two waves per SIMD)
    lds_leaf = reinterpret_cast<VIMG_LDS v4f*>(base + 4u * per_wave);
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
  uint32_t nan_samples = 0, iter_wave = 0;
  PROF_DECL

  // wave-uniform queue state (rings of capacity P)
  uint32_t qw_head = 0, qw_count = 0;
  uint32_t qv_head0 = 0, qv_head1 = 0, qv_head2 = 0, qv_head3 = 0;
  uint32_t qv_count0 = 0, qv_count1 = 0, qv_count2 = 0, qv_count3 = 0;
  auto ring = [&](uint32_t i) { return i >= P ? i - P : i; };

  // every slot starts "fresh": it needs a pixel
  for (uint32_t s = lane; s < P; s += 64) {
    word(SR_RAY, 3, s) = SF_FRESH;
    q_vertex[s] = static_cast<uint8_t>(s);
  }
  qv_count0 = P;
  bool pixels_left = true;   // wave-uniform: the global counter still had work last time
  // wave-uniform: the last finisher batch held nothing but slots that wait for another slot's
  // segment; the finisher queue is then passed over until some other stage has run, so that a
  // wave that carries both the waiting slot and the slot it waits for keeps moving
  bool skip_fin = false;
  uint32_t idle_polls = 0;   // consecutive main-loop turns that found only waiting slots

  // ---- persistent walk registers of the lane
  uint32_t w_slot = SLOT_IDLE, w_phase = 0, w_flags = 0, w_cls = 0;
  bool w_setup = false, w_any = false, w_found = false, w_exact = false;
  TravRay ray{f3{0.f, 0.f, 0.f}, f3{0.f, 0.f, 1.f}, 0.0001f, VIMG_INF};
  f3 w_inv{1.f, 1.f, 1.f};
  TriRayConst rc{0.f, 0.f, 1.f, 2};
  float w_dir_len2 = 1.f;
  uint32_t sp = 0, cur = REF_DONE;
  HitRec rec;
  rec.prim = 0xffffffffu;
  rec.kind = 0;
  rec.e0 = rec.e1 = rec.e2 = rec.inv_det = 0.f;

  for (;;) {
    const uint32_t n_walking = __popcll(__ballot(w_slot != SLOT_IDLE));
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
                            (qv_max > 0u && qw_count == 0u && 64u - n_walking >= A.pool_starve);
    if (!run_vertex && qw_count == 0u && !inflight) {
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
      // ================================================================== VERTEX stage
      PROF_LAP(PF_W_RETIRE)
      [[maybe_unused]] const unsigned long long prof_v0 = PROF_NOW();
      const uint32_t cls = (qv_elig0 == qv_max) ? 0u : (qv_count1 == qv_max ? 1u : (qv_count2 == qv_max ? 2u : 3u));
      const uint32_t qv_count = cls == 0 ? qv_count0 : (cls == 1 ? qv_count1 : (cls == 2 ? qv_count2 : qv_count3));
      const uint32_t qv_head = cls == 0 ? qv_head0 : (cls == 1 ? qv_head1 : (cls == 2 ? qv_head2 : qv_head3));
      const bool finisher_batch = (cls == 0);
      const uint32_t n = qv_count < 64u ? qv_count : 64u;
      const bool on = lane < n;
      const uint32_t slot = on ? q_vertex[cls * P + ring(qv_head + lane)] : 0u;
      if (cls == 0) { qv_head0 = ring(qv_head0 + n); qv_count0 -= n; }
      else if (cls == 1) { qv_head1 = ring(qv_head1 + n); qv_count1 -= n; }
      else if (cls == 2) { qv_head2 = ring(qv_head2 + n); qv_count2 -= n; }
      else { qv_head3 = ring(qv_head3 + n); qv_count3 -= n; }

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
        if (finisher_batch) r_a = crd(SC_ACC, slot);
        if (r_ray.w & SF_HAS_S) r_nee = crd(SC_NEE, slot);
        r_origin = rd(SR_ORIGIN, slot);
        r_shadow = rd(SR_SHADOW, slot);
        const v4u r_g = rd(SR_RNG, slot);
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
          const v4u r_c = crd(SC_CONE, slot);
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

      PROF_LAP(PF_V_LOAD)
      PROF_ADD(PF_V_BATCHES, 1) PROF_ADD(PF_V_LANES, n) PROF_ADD(PF_V_ATVERTEX, __popcll(__ballot(at_vertex)))
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
        PROF_LAP(PF_V_LIGHT)
        const bool reg_before = non_specular_bounce;
        RayCone nee_cone = cone;
        Scatter sc = sample_mat<TEX, MT>(g, hit, ray_d, rng, reg_before);
        PROF_LAP(PF_V_SAMPLE)
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
        PROF_LAP(PF_V_EVAL)
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
        if (A.pool_classes == 3u && cls == 1u)
          shade_vertex(std::integral_constant<int, int(VIMG_MAT_LAMBERTIAN)>{});
        else if (A.pool_classes == 3u && cls == 2u)
          shade_vertex(std::integral_constant<int, int(VIMG_MAT_PRINCIPLED)>{});
        else
          shade_vertex(std::integral_constant<int, -1>{});
      }

      // ---- finished samples: accumulate, pixel write-back, next pixel, next camera ray
      bool need_pixel = fresh;
      bool retire = false;
      // a fresh slot may already hold a claim whose predecessor segment was not published yet
      bool have_claim = fresh && (flags & SF_PRIMARY);
      uint32_t claim = have_claim ? word(SR_RNG, 3, slot) : 0u;
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
        if (is_nan(result.x) || is_nan(result.y) || is_nan(result.z)) nan_samples++;
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
          if (base >= total_claims) {
            pixels_left = false;
#ifdef VIMG_PROFILE
            prof_acc[PF_DRAIN] = __builtin_readcyclecounter();   // time stamp: turned into a span at exit
#endif
          }
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

      PROF_LAP(PF_V_FINISH)
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
        wr(SR_RNG, slot, v4u{static_cast<uint32_t>(rng.s), static_cast<uint32_t>(rng.s >> 32),
                             px | (py << 16), smp});
        if (finisher_batch) cwr(SC_ACC, slot, v4u{fu(acc.x), fu(acc.y), fu(acc.z), item});
        if constexpr (TEX) cwr(SC_CONE, slot, v4u{fu(cone.cone_width), fu(cone.spread_angle), 0u, 0u});
      }
      {
        const bool to_walk = keep && (has_s || has_r), to_fin = keep && !to_walk;
        const unsigned long long mask = __ballot(to_walk), mfin = __ballot(to_fin);
        if (to_walk) q_walk[ring(qw_head + qw_count + lane_rank(mask, lane))] = static_cast<uint8_t>(slot);
        if (to_fin) q_vertex[ring(qv_head0 + qv_count0 + lane_rank(mfin, lane))] = static_cast<uint8_t>(slot);
        qw_count += __popcll(mask);
        qv_count0 += __popcll(mfin);
      }
      PROF_LAP(PF_V_STORE)
      PROF_ADD(PF_CLS_CYC0 + cls, PROF_NOW() - prof_v0) PROF_ADD(PF_CLS_LANES0 + cls, n)
    } else {
      // ================================================================== WALK stage
      skip_fin = false;
      idle_polls = 0;
      PROF_LAP(PF_W_RETIRE)
      for (;;) {
        PROF_ADD(PF_W_ROUNDS, 1)
        // (1) idle lanes take queued slots
        {
          const bool idle = (w_slot == SLOT_IDLE);
          const unsigned long long mask = __ballot(idle);
          const uint32_t n_idle = __popcll(mask);
          const uint32_t take = n_idle < qw_count ? n_idle : qw_count;
          if (take) {
            const uint32_t r = lane_rank(mask, lane);
            if (idle && r < take) {
              w_slot = q_walk[ring(qw_head + r)];
              w_flags = word(SR_RAY, 3, w_slot);
              w_phase = (w_flags & SF_HAS_S) ? 0u : 1u;
              w_setup = true;
            }
            qw_head = ring(qw_head + take);
            qw_count -= take;
          }
        }
        // (2) ray set-up (reference include/bvh.h:109-143): everything derived from the ray alone
        if (__any(w_setup)) {
          if (w_setup) {
            const v4u ro = rd(SR_ORIGIN, w_slot);
            ray.o = f3{uf(ro.x), uf(ro.y), uf(ro.z)};
            if (w_phase == 0) {
              const v4u rs = rd(SR_SHADOW, w_slot);
              ray.d = f3{uf(rs.x), uf(rs.y), uf(rs.z)};
              ray.max_t = uf(ro.w);
              w_any = true;
              cnt.shadow++;
            } else {
              const v4u rr = rd(SR_RAY, w_slot);
              ray.d = f3{uf(rr.x), uf(rr.y), uf(rr.z)};
              ray.max_t = VIMG_INF;
              w_any = false;
              cnt.closest++;
            }
            w_inv = f3{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
            w_exact = (ray.d.x == 0.f) || (ray.d.y == 0.f) || (ray.d.z == 0.f);
            rc = tri_ray_const(ray.d);
            w_dir_len2 = dot(ray.d, ray.d);
            const float root =
                slab(load3k(g.root_min), load3k(g.root_max), ray.o, w_inv, ray.min_t, ray.max_t);
            cur = is_inf(root) ? REF_DONE : g.root_ref;
            sp = 0;
            w_found = false;
            rec.prim = 0xffffffffu;
            w_setup = false;
          }
        }
        PROF_LAP(PF_W_REFILL)
        if (!__any(w_slot != SLOT_IDLE)) break;
        // (3) walk until a quarter of the wave has a finished ray (or nothing is left to walk)
        for (;;) {
          // the box loop in two builds: rays with a zero direction component need the exact
          // select form of the slab test (0 * inf); a round without such a ray runs the build
          // that has only the min/max form
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
#ifdef VIMG_PROFILE
            if (full_stats && first_active_lane()) cnt.trip_descend++;
#endif
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
            // branch-free step: the entry a pop would return was read before the box test (its
            // latency hides behind the test); the far child is written above the top of the stack
            // whether it is kept or not (the slot is free), and sp moves by select
            const bool both = in1 && in2, any = in1 || in2;
            const bool first_is_near = w_any ? false : (h2 > h1);
            const uint32_t near_c = first_is_near ? c1 : c2;
            const uint32_t far_c = first_is_near ? c2 : c1;
            L.stack[sp * 64] = far_c;
            const uint32_t one_c = in1 ? c1 : c2;
            cur = both ? near_c : (any ? one_c : (sp != 0 ? popped : REF_DONE));
            sp = both ? sp + 1 : (any ? sp : sp_below);
            // deep trees: when only a few lanes still descend, the lanes that wait at a leaf go
            // first (the box loop of the config-5 stand-in ran with 27 % of its lanes busy)
            if constexpr (DEEP) {
              if (__popcll(__ballot(cur != REF_DONE && ref_count(cur) == 0)) < box_min) break;
            }
          }
          };
          if (__any(w_exact && w_slot != SLOT_IDLE))
            box_loop(std::true_type{});
          else
            box_loop(std::false_type{});
          PROF_LAP(PF_W_BOX)
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
#ifdef VIMG_PROFILE
              if (full_stats && first_active_lane()) cnt.trip_prim++;
#endif
              bool hit = false;
              float t = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f, idet = 0.f;
              if (kind == 0) {
                hit = tri_test_flat(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c0}, ray, rc, t, e0,
                                    e1, e2, idet);
              } else if (kind == 1) {
                cnt.sphere += stat_inc;
                hit = sphere_test(f3{a.x, a.y, a.z}, a.w, ray, w_dir_len2, t);
              }
              // by select: a hit shortens the ray; an any-hit ray stops at its first hit, a
              // closest-hit ray keeps the record of the last success
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
          PROF_LAP(PF_W_LEAF)
          const uint32_t n_fin = __popcll(__ballot(w_slot != SLOT_IDLE && cur == REF_DONE));
          const uint32_t n_act = __popcll(__ballot(w_slot != SLOT_IDLE && cur != REF_DONE));
          if (n_act == 0 || n_fin >= A.pool_refill) break;
        }
        // (4) retire finished rays: second ray of the item, or hand the slot to the vertex stage
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
            if (w_found) {
              w_flags |= SF_FOUND | (rec.kind == 1 ? SF_KIND_SPHERE : 0u);
              wr(SR_SHADOW, w_slot, v4u{fu(rec.e0), fu(rec.e1), fu(rec.e2), fu(rec.inv_det)});
              q_prim[w_slot] = rec.prim;
              word(SR_ORIGIN, 3, w_slot) = fu(ray.max_t);
            }
            done_item = true;
          }
          if (done_item) word(SR_RAY, 3, w_slot) = w_flags;
        }
        {
          // class of the batch this slot joins: 0 = its path ends (miss, no path ray, emitter hit under
          // mis, any hit under the normal integrators), else the material class of the vertex
          // (leaf record: 0 emitter, 1 Lambertian, 2 Principled, 3 other)
          uint32_t cls = 0;
          if (done_item && (w_flags & SF_FOUND) && A.integrator >= VIMG_INTEGRATOR_MATERIAL) {
            cls = w_cls;
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
          if (done_item) {
            const uint8_t id = static_cast<uint8_t>(w_slot);
            if (cls == 0) q_vertex[ring(qv_head0 + qv_count0 + lane_rank(m0, lane))] = id;
            else if (cls == 1) q_vertex[P + ring(qv_head1 + qv_count1 + lane_rank(m1, lane))] = id;
            else if (cls == 2) q_vertex[2 * P + ring(qv_head2 + qv_count2 + lane_rank(m2, lane))] = id;
            else q_vertex[3 * P + ring(qv_head3 + qv_count3 + lane_rank(m3, lane))] = id;
            w_slot = SLOT_IDLE;
          }
          qv_count0 += __popcll(m0);
          qv_count1 += __popcll(m1);
          qv_count2 += __popcll(m2);
          qv_count3 += __popcll(m3);
        }
        PROF_LAP(PF_W_RETIRE)
        // (5) leave when a full vertex batch waits, or when nothing is left to walk
        if (qv_count0 >= A.pool_vbatch || qv_count1 >= A.pool_vbatch || qv_count2 >= A.pool_vbatch ||
            qv_count3 >= A.pool_vbatch)
          break;
        if (qw_count == 0u) {
          const uint32_t walking = __popcll(__ballot(w_slot != SLOT_IDLE));
          if (walking == 0u) break;
          if (64u - walking >= A.pool_starve && (qv_count0 | qv_count1 | qv_count2 | qv_count3) != 0u) break;
        }
      }
    }
  }

#ifdef VIMG_PROFILE
  if (stats && lane == 0) {
    const unsigned long long t_end = __builtin_readcyclecounter();
    prof_acc[PF_TOTAL] = t_end - prof_t0;
    prof_acc[PF_DRAIN] = prof_acc[PF_DRAIN] ? t_end - prof_acc[PF_DRAIN] : 0ull;   // after the last pixel fetch
    for (int k = 0; k < PF_COUNT; ++k)
      if (k != PF_MAXWAVE) atomicAdd(&stats->prof[k], prof_acc[k]);
    atomicMax(&stats->prof[PF_MAXWAVE], prof_acc[PF_TOTAL]);
  }
#endif
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
