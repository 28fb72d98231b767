// CU-wide scheduler: ONE pool of path slots per compute unit, waves with ROLES, queues without a lock.
//
// What round 2's default (render_pool4_kernel<..., group>) measured, and what this build changes:
//  * every queue operation of a wave ran under one workgroup lock: 19 % of the wave cycles of a
//    config-2 frame (760 cycles to get it, 2 300 to release, profiles/r2_final/walk_diag_*).
//    Here the five queues are multi-producer / multi-consumer TICKET RINGS in LDS: a producer
//    reserves with one returning ds_add on the ring's tail, writes its entries, then adds to the
//    ring's count; a consumer takes from the count (signed, over-draws are handed back), then
//    claims indices with one returning ds_add on the head and reads its entries (an entry whose
//    producer has reserved but not yet written reads EMPTY: the consumer re-reads).  No wave
//    ever waits for another wave's critical section.
//  * the vertex stage was a CALL out of the walk loop with the walk's lane state in callee-saved
//    registers: 0.7 TB of scratch traffic per frame and rays that stand still while their wave
//    shades.  Here a wave is a WALKER or a SHADER for as long as it holds state: walkers only walk
//    (and may run a batch when they hold no ray at all), shaders only run vertex batches; nothing
//    is live across a stage, every stage body is inlined, there is no call and no scratch.
//  * a vertex walked its shadow ray and THEN its path ray (two hops through the walk queue).  All
//    draws of a vertex precede both rays (src/integrators/mis_integrator.cpp:45-120), so both
//    are queued together and walked by whichever lanes are free; the second one to finish (an
//    atomic OR on the slot's flag word tells) hands the slot to the vertex queue of its class.
//  * the pool is the CU's (up to 16 waves share it), so batches are whatever 64 slots the CU has
//    waiting, and a thin shard's pixels all sit in LDS slots at once.
// Measured and NOT kept (DESIGN.md 4.4 has the whole list, profiles/r3_cu/sweeps_raw.txt the runs; all bit-identical):
//  * a second, urgent set of rings for slots whose pixel lags behind the pool's mean sample index:
//    the last pixel of an eighth of config 2 finished no earlier (19.2 against 18.4 ms at 64 spp - it
//    is bound by the service time of its hops, 9 Principled vertices per sample, not by queueing)
//    and the full frame paid 7 % for the second set of pops and ballots;
//  * both rays of a vertex as ONE ring entry walked by one lane (one pop and one hand-over instead of
//    two, no join) once many rays wait: config 2 323 against 316 ms, its eighth 141 against 122 ms,
//    stand-ins of configs 4 / 5 2.40 / 3.60 against 2.80 / 4.01 Grays/s - shadow rays queued
//    together are walked together;
//  * 12 waves per compute unit at 168 registers: 351 against 311 ms (config 2), 2.5 / 3.2 against
//    2.8 / 4.0 Grays/s (stand-ins 4 / 5): the fourth wave per SIMD hides more than 40 registers save;
//  * two-level node records for trees in global memory (a node's record followed by both children's,
//    192 B per fetch, a second step without a second round trip): stand-ins of configs 4 / 5 132.6 /
//    149.8 against 118.6 / 125.2 ms - three lines per lane and step load the CU's L1 path more than
//    the saved round trips relieve it;
//  * stacks for all sixteen waves (they fit beside 1 280 slots) and shading waves that walk while no
//    batch waits for them, taking rays only until a full batch does: config 2 297.5 against 296.9 ms
//    at any threshold of 32 to 512 waiting rays - the shading side has no idle time worth lending.
// Same device functions, same order of operations per path as every other scheduler: bit-identical.
#pragma once
#include <type_traits>

#include "sched_common.h"

namespace vimg {

// flag word of a slot (CR_DIR.w).  The shader writes it whole; walkers OR their result bits in.
enum : uint32_t {
  CF_PRIMARY = 1u, CF_NONSPEC = 2u, CF_HAS_S = 4u, CF_HAS_R = 8u, CF_OCCLUDED = 16u, CF_FOUND = 32u,
  CF_FRESH = 64u, CF_KIND_SPHERE = 128u, CF_DONE_S = 256u, CF_DONE_R = 512u,
  CF_CLS_SHIFT = 10u,     // two bits: material class of the hit primitive (leaf record)
  CF_DONE_V = 4096u,      // the vertex stage that queued the rays has written the slot's state
  CF_KILL_R = 8192u,      // the path ray was queued before the vertex stage found the path ended (NaN pdf): ignore its result
  CF_BOUNCE_SHIFT = 16u   // sixteen bits (Russian roulette ends paths long before)
};
// hot records of a slot in LDS, [record][slot]
enum : uint32_t {
  CR_ORG = 0,   // o.xyz | shadow max_t
  CR_DIR,       // d.xyz (camera / BSDF ray) | flags
  CR_SHD,       // shadow d.xyz | -
  CR_HIT,       // e0 e1 e2 inv_det of the path ray's hit (its walker writes it)
  CR_COUNT
};
constexpr uint32_t CU_EMPTY = 0xffffu;   // ring entry nobody has written yet
constexpr uint32_t CU_RAY_S = 0x8000u;   // walk-ring entry: the SHADOW ray of the slot (else its path ray)
// Rings: 0-3 vertex rings by class (0 finishers, 1 Lambertian, 2 Principled, 3 other), 4 the walk ring.
// (Tried and removed: a second, URGENT set of rings for slots whose pixel lags behind the pool's mean
// sample index, emptied first by every consumer.  On thin shards of config 2 the last pixel finished
// no earlier - 19.2 against 18.4 ms at 64 spp on an eighth of the frame: the pixels a thin shard ends
// with are bound by the service time of their hops, not by queueing - and the full frame paid 7 %
// for the second set of pops and ballots: 325 against 304 ms.)
constexpr uint32_t CQ_WALK = 4u, CQ_COUNT = 5u;
// LDS bytes per slot: hot records, {primitive id, t} of the hit, four vertex rings (capacity P),
// the walk ring (capacity 2 P: two rays per slot), time of the last hand-over (statistics launches)
constexpr uint32_t CU_LDS_BYTES = CR_COUNT * 16u + 8u + 4u * 2u + 2u * 2u + 4u;

// queue state of the workgroup, in LDS
struct CuCtl {
  int32_t avail[8];        // entries ready in ring q (0-3: one ds_read_b128; 4: the walk ring)
  uint32_t tail[CQ_COUNT], head[CQ_COUNT];   // monotonic ticket counters (index = counter mod capacity)
  uint32_t live;           // slots that have not retired
  uint32_t pixels_left;    // the global work counter still had items last time
  uint32_t abort;
  uint32_t t0_lo, t0_hi;   // s_memrealtime at the start of the workgroup (statistics launches)
  uint32_t pad[1];
};
static_assert(sizeof(CuCtl) == 96, "CuCtl layout");
// per wave, in LDS: what only statistics launches touch (event counts beyond the two ray counts,
// cycles by stage: 0-3 vertex batches by class, 4 walk, 5 idle; batches and slots by class)
struct CuWaveRec {
  unsigned long long cyc[6], nbatch[4], nslots[4];
  uint32_t internal, leaf, prim, sphere;
  unsigned long long box_pass, box_lanes, leaf_round, leaf_lanes, sessions, refills, refill_rays, pad;
  unsigned long long looks, q_sum[4];   // looks of the main loop and the ring counts they saw (finisher, Lambertian, Principled, walk)
  unsigned long long w_cyc[4];          // cycles of a walk session: refill + set-up, box loop, leaf rounds, hand-over
  unsigned long long wait_cyc[5], wait_n[5];   // cycles slots spent in the rings (vertex 0-3, walk) and how many
  unsigned long long ray_cyc, ray_n;           // cycles rays spent in a walking lane, from the pop to the hand-over
  unsigned long long pv_cyc[6], pv_n;          // Principled batches: cycles in state loads, hit record + path logic, light sample, BSDF sample, evaluations, stores + hand-over
};
static_assert(sizeof(CuWaveRec) == 416, "CuWaveRec layout");
__host__ __device__ constexpr uint32_t cu_pool_bytes(uint32_t slots, uint32_t waves) {
  return CU_LDS_BYTES * slots + uint32_t(sizeof(CuCtl)) + waves * uint32_t(sizeof(CuWaveRec));
}
// divisor d -> (magic, shift) with n / d == mulhi(n, magic) >> shift for every n < 2^31
// (Granlund & Montgomery; host side in vimg_hip.hip:make_launch)
VD uint32_t cu_mod(uint32_t n, uint32_t d, uint32_t magic, uint32_t shift) { return n - d * (__umulhi(n, magic) >> shift); }
VD int32_t lds_add_rtn(VIMG_LDS int32_t* p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
VD uint32_t lds_add_rtn(VIMG_LDS uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
VD uint32_t lds_load(VIMG_LDS uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
VD int32_t lds_load(VIMG_LDS int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
VD uint32_t cu_uni(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }


// Kernel arguments: ONE struct by value, read through a pointer to the kernel-argument segment that
// the optimiser cannot see through (cu_kargs: an empty asm on the address).  Every stage fetches
// what it needs of the scene and the launch with scalar loads when it starts; without this the
// compiler hoists the loads of ~70 pointers and parameters out of the persistent loop, runs out of
// scalar registers (122 spilled) and then out of vector registers (185 spilled, 532-632 bytes of
// scratch per lane for every shading stage; measured on the first build of this file).
struct CuKArgs {
  DScene g;
  RenderArgs A;
  float* out;
  DeviceStats* stats;
  unsigned int* work_counter;
};
typedef const __attribute__((address_space(4))) CuKArgs* CuKPtr;
VD CuKPtr cu_kargs() {
  unsigned long long p = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
  asm volatile("" : "+s"(p));
  return (CuKPtr)p;
}

// What every stage derives from the arguments: scene, launch parameters, the LDS carve-out behind
// the node planes (stacks of the walking waves, the pool, the rings, the group record, the per-wave
// records, the leaf copy), the workgroup's cold region, and the ring operations.
#define CU_STAGE_LOCALS(K)                                                                                              \
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];                                               \
  [[maybe_unused]] const DScene& g = *(const DScene*)&(K)->g;                                                           \
  [[maybe_unused]] const RenderArgs& A = *(const RenderArgs*)&(K)->A;                                                   \
  [[maybe_unused]] float* __restrict__ out = (K)->out;                                                                  \
  [[maybe_unused]] unsigned int* __restrict__ work_counter = (K)->work_counter;                                         \
  [[maybe_unused]] const uint32_t lane = threadIdx.x & 63, wave = cu_uni(threadIdx.x >> 6);                             \
  [[maybe_unused]] const bool full_stats = DIAG && A.full_stats != 0;   /* (the DIAG build serves statistics launches) */ \
  [[maybe_unused]] const uint32_t stat_inc = full_stats ? 1u : 0u;                                                      \
  [[maybe_unused]] const uint32_t W = static_cast<uint32_t>(g.res_x), H = static_cast<uint32_t>(g.res_y);               \
  [[maybe_unused]] const bool single = A.single_x >= 0;                                                                 \
  [[maybe_unused]] const uint32_t total_items = single ? 1u : A.num_local_tiles * 64u;                                  \
  [[maybe_unused]] const uint32_t n_seg = A.pool_segments, seg_len = A.pool_seg_len;                                    \
  [[maybe_unused]] const uint32_t total_claims = total_items * n_seg;                                                   \
  [[maybe_unused]] constexpr uint32_t roulette_threshold = 5;                                                           \
  [[maybe_unused]] const bool material_mode = (A.integrator == VIMG_INTEGRATOR_MATERIAL);                               \
  [[maybe_unused]] const uint32_t P = A.pool_slots;                                                                     \
  [[maybe_unused]] const bool can_walk = wave < A.cu_walkers;                                                           \
  const uint32_t stack_rows_ = pool4_stack_rows_of(A.stack_entries, A.stack_lds);                                       \
  const uint32_t node_bytes_ = (lds_node_bytes(A.lds_nodes) + 255u) & ~255u;                                            \
  [[maybe_unused]] VIMG_LDS uint32_t* stack0 =                                                                          \
      reinterpret_cast<VIMG_LDS uint32_t*>((VIMG_LDS unsigned char*)lds_raw + node_bytes_) +                            \
      size_t(can_walk ? wave : 0u) * stack_rows_ * 64u + lane;                                                          \
  VIMG_LDS unsigned char* base_ = (VIMG_LDS unsigned char*)lds_raw + node_bytes_ + A.cu_walkers * stack_rows_ * 256u;   \
  [[maybe_unused]] VIMG_LDS v4u* recs = reinterpret_cast<VIMG_LDS v4u*>(base_);                                         \
  [[maybe_unused]] VIMG_LDS v2u* hitx = reinterpret_cast<VIMG_LDS v2u*>(recs + CR_COUNT * P);                           \
  [[maybe_unused]] VIMG_LDS uint16_t* ring_v = reinterpret_cast<VIMG_LDS uint16_t*>(hitx + P);                          \
  [[maybe_unused]] VIMG_LDS uint16_t* ring_w = ring_v + 4u * P;                                                         \
  [[maybe_unused]] VIMG_LDS uint32_t* tq = reinterpret_cast<VIMG_LDS uint32_t*>(ring_w + 2u * P);                       \
  [[maybe_unused]] VIMG_LDS CuCtl* G = reinterpret_cast<VIMG_LDS CuCtl*>(tq + P);                                       \
  [[maybe_unused]] VIMG_LDS CuWaveRec* wrec = reinterpret_cast<VIMG_LDS CuWaveRec*>(G + 1) + wave;                      \
  [[maybe_unused]] VIMG_LDS v4f* lds_leaf = reinterpret_cast<VIMG_LDS v4f*>(reinterpret_cast<VIMG_LDS CuWaveRec*>(G + 1) + NW); \
  [[maybe_unused]] VIMG_LDS uint32_t* recw = reinterpret_cast<VIMG_LDS uint32_t*>(recs);                                \
  [[maybe_unused]] auto rd = [&](uint32_t r, uint32_t slot) -> v4u { return recs[r * P + slot]; };                      \
  [[maybe_unused]] auto wr = [&](uint32_t r, uint32_t slot, v4u v) { recs[r * P + slot] = v; };                         \
  [[maybe_unused]] auto flag_word = [&](uint32_t slot) -> VIMG_LDS uint32_t* { return recw + (CR_DIR * P + slot) * 4u + 3u; }; \
  [[maybe_unused]] auto fu = [](float f) { return __float_as_uint(f); };                                                \
  [[maybe_unused]] auto uf = [](uint32_t u) { return __uint_as_float(u); };                                             \
  [[maybe_unused]] auto mod_v = [&](uint32_t t) { return cu_mod(t, P, A.cu_magic_v, A.cu_shift_v); };                   \
  [[maybe_unused]] auto mod_w = [&](uint32_t t) { return cu_mod(t, 2u * P, A.cu_magic_w, A.cu_shift_w); };              \
  [[maybe_unused]] auto ring_at = [&](uint32_t q, uint32_t t) -> VIMG_LDS uint16_t* {                                   \
    return q >= CQ_WALK ? ring_w + mod_w(t) : ring_v + q * P + mod_v(t);                                                \
  };                                                                                                                    \
  [[maybe_unused]] const bool leaf_in_lds = A.lds_leaf != 0u;                                                           \
  [[maybe_unused]] const uint32_t box_min = A.pool_boxmin;                                                              \
  /* cold records of the workgroup's slots in global memory: [slot][4] main lines (throughput, result, NEE term, RNG),  \
     then the accumulator plane, then the cone plane (textured build) */                                                \
  [[maybe_unused]] VIMG_GLOBAL v4u* cold = A.pool_cold + size_t(blockIdx.x) * (size_t(pool4_cold_records(TEX)) * P);    \
  [[maybe_unused]] VIMG_GLOBAL v4u* cold_acc = cold + size_t(SC4_MAIN) * P;                                             \
  [[maybe_unused]] VIMG_GLOBAL v4u* cold_cone = cold_acc + P;                                                           \
  [[maybe_unused]] auto crd = [&](uint32_t r, uint32_t slot) -> v4u { return cold[slot * SC4_MAIN + r]; };              \
  [[maybe_unused]] auto cwr = [&](uint32_t r, uint32_t slot, v4u v) { cold[slot * SC4_MAIN + r] = v; };                 \
  /* error exit: the launch's error word (behind the work counter) and the group's abort flag */                        \
  [[maybe_unused]] auto raise = [&](uint32_t bit) {                                                                     \
    atomicOr(work_counter + 1, bit);                                                                                    \
    __hip_atomic_store(&G->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                                  \
  };                                                                                                                    \
  /* push: the lanes of `mask` append `entry`; the entries become visible to consumers with the add to the ring's      \
     count (LDS operations of one wave are performed in order) */                                                       \
  [[maybe_unused]] auto push = [&](uint32_t q, unsigned long long mask, uint32_t entry) {                               \
    const uint32_t n_ = static_cast<uint32_t>(__popcll(mask));                                                          \
    if (n_ == 0u) return;                                                                                               \
    uint32_t t_ = 0;                                                                                                    \
    if (lane == 0) {                                                                                                    \
      t_ = lds_add_rtn(&G->tail[q], n_);                                                                                \
      if (t_ > 0x7ff00000u) raise(8u); /* the modulo is exact below 2^31 pushes per ring and launch */                  \
    }                                                                                                                   \
    t_ = cu_uni(t_);                                                                                                    \
    if ((mask >> lane) & 1ull) *ring_at(q, t_ + lane_rank(mask, lane)) = static_cast<uint16_t>(entry);                  \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                                                              \
    if (lane == 0) lds_add_rtn(&G->avail[q], static_cast<int32_t>(n_));                                                 \
  };                                                                                                                    \
  /* A slot is complete when the vertex stage that queued its rays has written its state (DONE_V) and every ray it      \
     queued has been walked; `nw` is the flag word after the caller's own OR.  */                                       \
  [[maybe_unused]] auto slot_complete = [&](uint32_t nw) -> bool {                                                      \
    return (nw & CF_DONE_V) && (!(nw & CF_HAS_S) || (nw & CF_DONE_S)) && (!(nw & CF_HAS_R) || (nw & CF_DONE_R));        \
  };                                                                                                                    \
  /* the lanes whose OR completed their slot hand it to the vertex ring of its class: 0 = its path ends (miss, no path  \
     ray, emitter hit under mis, any hit under the normal integrators, a path ray whose vertex stage found the path    \
     ended), else the material class of the vertex (leaf record); lanes 0..3 reserve for the four rings in one add */  \
  [[maybe_unused]] auto hand_to_vertex = [&](bool complete, uint32_t nw, uint32_t slot_) {                              \
    uint32_t cls_ = 0;                                                                                                  \
    if (complete && (nw & CF_FOUND) && !(nw & CF_KILL_R) && A.integrator >= VIMG_INTEGRATOR_MATERIAL) {                 \
      cls_ = (nw >> CF_CLS_SHIFT) & 3u;                                                                                 \
      if (cls_ == 0 && material_mode) cls_ = 3; /* material_integrator shades emitters too */                           \
      if (cls_ != 0) {                                                                                                  \
        if (A.pool_classes == 1) cls_ = 1;                                                                              \
        else if (A.pool_classes == 2) cls_ = (cls_ == 2) ? 2u : 1u;                                                     \
      }                                                                                                                 \
    }                                                                                                                   \
    const unsigned long long m0_ = __ballot(complete && cls_ == 0), m1_ = __ballot(complete && cls_ == 1),              \
                             m2_ = __ballot(complete && cls_ == 2), m3_ = __ballot(complete && cls_ == 3);              \
    if ((m0_ | m1_ | m2_ | m3_) == 0ull) return;                                                                        \
    const uint32_t n_me_ = static_cast<uint32_t>(__popcll(lane == 0 ? m0_ : (lane == 1 ? m1_ : (lane == 2 ? m2_ : m3_)))); \
    uint32_t t_me_ = 0;                                                                                                 \
    if (lane < 4u && n_me_ != 0u) {                                                                                     \
      t_me_ = lds_add_rtn(&G->tail[lane], n_me_);                                                                       \
      if (t_me_ > 0x7ff00000u) raise(8u);                                                                               \
    }                                                                                                                   \
    const uint32_t t0_ = __shfl(t_me_, 0), t1_ = __shfl(t_me_, 1), t2_ = __shfl(t_me_, 2), t3_ = __shfl(t_me_, 3);      \
    if (complete) {                                                                                                     \
      const uint32_t tt_ = cls_ == 0 ? t0_ : (cls_ == 1 ? t1_ : (cls_ == 2 ? t2_ : t3_));                               \
      const unsigned long long mm_ = cls_ == 0 ? m0_ : (cls_ == 1 ? m1_ : (cls_ == 2 ? m2_ : m3_));                     \
      *ring_at(cls_, tt_ + lane_rank(mm_, lane)) = static_cast<uint16_t>(slot_);                                        \
      if (full_stats) tq[slot_] = static_cast<uint32_t>(__builtin_readcyclecounter());                                  \
    }                                                                                                                   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                                                              \
    if (lane < 4u && n_me_ != 0u) lds_add_rtn(&G->avail[lane], static_cast<int32_t>(n_me_));                            \
  };                                                                                                                    \
  /* the rays of the lanes of `ms_` (shadow rays) and `mr_` (path rays) into the walk ring, one reservation, shadow    \
     rays first (an occluded one is the shorter walk, and rays of a kind are then walked together) */                   \
  [[maybe_unused]] auto push_rays = [&](bool is_s, bool is_r, uint32_t slot_) {                                         \
    const unsigned long long ms_ = __ballot(is_s), mr_ = __ballot(is_r);                                                \
    const uint32_t n_s_ = static_cast<uint32_t>(__popcll(ms_)), n_r_ = static_cast<uint32_t>(__popcll(mr_));            \
    if (n_s_ + n_r_ == 0u) return;                                                                                      \
    uint32_t t_ = 0;                                                                                                    \
    if (lane == 0) {                                                                                                    \
      t_ = lds_add_rtn(&G->tail[CQ_WALK], n_s_ + n_r_);                                                                 \
      if (t_ > 0x7ff00000u) raise(8u);                                                                                  \
    }                                                                                                                   \
    t_ = cu_uni(t_);                                                                                                    \
    if (is_s) *ring_at(CQ_WALK, t_ + lane_rank(ms_, lane)) = static_cast<uint16_t>(slot_ | CU_RAY_S);                   \
    if (is_r) *ring_at(CQ_WALK, t_ + n_s_ + lane_rank(mr_, lane)) = static_cast<uint16_t>(slot_);                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                                                              \
    if (lane == 0) lds_add_rtn(&G->avail[CQ_WALK], static_cast<int32_t>(n_s_ + n_r_));                                  \
  };                                                                                                                    \
  /* pop: up to `want` entries (wave-uniform); the lanes of `takers` (at least `want` of them) receive them in rank    \
     order; returns the number taken.  An entry whose producer has reserved but not yet written reads EMPTY. */         \
  [[maybe_unused]] auto pop = [&](uint32_t q, uint32_t want, unsigned long long takers, uint32_t& entry) -> uint32_t {   \
    uint32_t got_ = 0, h_ = 0;                                                                                          \
    if (lane == 0) {                                                                                                    \
      VIMG_LDS int32_t* av_ = &G->avail[q];                                                                             \
      const int32_t old_ = lds_add_rtn(av_, -static_cast<int32_t>(want));                                               \
      const int32_t g2_ = old_ < 0 ? 0 : (old_ < static_cast<int32_t>(want) ? old_ : static_cast<int32_t>(want));       \
      if (g2_ < static_cast<int32_t>(want)) lds_add_rtn(av_, static_cast<int32_t>(want) - g2_);                         \
      if (g2_ > 0) h_ = lds_add_rtn(&G->head[q], static_cast<uint32_t>(g2_));                                           \
      got_ = static_cast<uint32_t>(g2_);                                                                                \
    }                                                                                                                   \
    got_ = cu_uni(got_), h_ = cu_uni(h_);                                                                               \
    if (got_ == 0u) return 0u;                                                                                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                                                              \
    bool stuck_ = false;                                                                                                \
    if ((takers >> lane) & 1ull) {                                                                                      \
      const uint32_t r_ = lane_rank(takers, lane);                                                                      \
      if (r_ < got_) {                                                                                                  \
        VIMG_LDS uint16_t* e_ = ring_at(q, h_ + r_);                                                                    \
        uint32_t v_ = __hip_atomic_load(e_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), spins_ = 0;                \
        while (v_ == CU_EMPTY && !stuck_) {                                                                             \
          __builtin_amdgcn_s_sleep(1);                                                                                  \
          v_ = __hip_atomic_load(e_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                                   \
          if (++spins_ > (1u << 22)) stuck_ = true;                                                                     \
        }                                                                                                               \
        __hip_atomic_store(e_, static_cast<uint16_t>(CU_EMPTY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        \
        entry = v_;                                                                                                     \
      }                                                                                                                 \
    }                                                                                                                   \
    if (__any(stuck_)) {                                                                                                \
      if (lane == 0) raise(4u);                                                                                         \
      return 0u;                                                                                                        \
    }                                                                                                                   \
    return got_;                                                                                                        \
  }

// ======================================================================== one vertex batch
// (the body is render_pool4_kernel's vertex stage; FIN: the finisher queue, MTC: material the shading
// is specialised for, -1 = any).  `n` slots; lane i < n holds its slot id in `slot`.
template <bool TEX, int NW, bool DIAG, int EARLY, bool FIN, int MTC>
VD void cu_vertex(uint32_t n, uint32_t slot, bool& all_pending, uint32_t& n_nan, uint32_t& n_dead) {
  const CuKPtr K = cu_kargs();
  CU_STAGE_LOCALS(K);
  constexpr bool finisher_batch = FIN;
  // A batch of at most 32 slots of a material whose two BSDF evaluations are worth it runs SPLIT: lanes
  // 32-63 shadow lanes 0-31 (same slot, same loads, same draws, same arithmetic) up to the two
  // evaluations of a vertex - towards the light and along the sampled direction, independent of each
  // other - of which each half of the wave then does ONE, in the same instructions; the owner lane takes
  // the other result from its mirror lane.  A wave that has its SIMD to itself issues one instruction per
  // four cycles however few lanes are on, so on a frame short of pixels (where the last pixels' chains
  // of Principled vertices are the frame time) this takes a fifth off the longest stage.
  constexpr bool CAN_SPLIT = !FIN && MTC != int(VIMG_MAT_LAMBERTIAN);
  const bool split = CAN_SPLIT && n <= 32u && !(A.cu_flex & 16u);
  const bool mirror = split && lane >= 32u;
  if (split) slot = static_cast<uint32_t>(__shfl(static_cast<int>(slot), static_cast<int>(lane & 31u)));
  const bool on = (split ? (lane & 31u) : lane) < n;
  if (!on) slot = 0u;
  const v4u r_ray = on ? rd(CR_DIR, slot) : v4u{0u, 0u, 0u, 0u};
  uint32_t flags = r_ray.w;
  const bool fresh = on && (flags & CF_FRESH);
  const bool have = on && !fresh;
  uint32_t px = 0, py = 0, smp = 0, item = 0, bounce = 0;
  Rng rng{0};
  f3 acc{0.f, 0.f, 0.f}, ray_o{0.f, 0.f, 0.f}, ray_d{0.f, 0.f, 1.f};
  f3 throughput{1.f, 1.f, 1.f}, result{0.f, 0.f, 0.f};
  RayCone cone{0.f, 0.f};
  float eta_scale = 1.f, prev_pdf = 0.f;
  bool primary = true, non_specular_bounce = false;
  v4u r_origin{0u, 0u, 0u, 0u}, r_hit{0u, 0u, 0u, 0u}, r_nee{0u, 0u, 0u, 0u};
  v2u r_hx{0u, 0u};
  constexpr bool PV = DIAG && MTC == int(VIMG_MAT_PRINCIPLED);   // statistics launches: where a Principled batch's cycles go
  [[maybe_unused]] unsigned long long pv_t = PV ? __builtin_readcyclecounter() : 0ull;
  [[maybe_unused]] auto pv_lap = [&](int k) {
    if constexpr (PV) {
      if (full_stats) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0) wrec->pv_cyc[k] += now - pv_t;
        pv_t = now;
      }
    }
  };
  uint32_t hops = 0;   // statistics launches: vertex-stage visits of the slot's pixel so far (kept in CR_SHD.w)
  if (full_stats && have) hops = recw[(CR_SHD * P + slot) * 4u + 3u] + 1u;
  if (have) {
    const v4u r_t = crd(SC_THROUGHPUT, slot), r_r = crd(SC_RESULT, slot);
    v4u r_a{0u, 0u, 0u, 0u};
    if (finisher_batch) r_a = cold_acc[slot];
    if (r_ray.w & CF_HAS_S) r_nee = crd(SC_NEE, slot);
    r_origin = rd(CR_ORG, slot);
    r_hit = rd(CR_HIT, slot);
    r_hx = hitx[slot];
    const v4u r_g = crd(SC4_RNG, slot);
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
    bounce = flags >> CF_BOUNCE_SHIFT;
    primary = (flags & CF_PRIMARY) != 0;
    non_specular_bounce = (flags & CF_NONSPEC) != 0;
    if constexpr (TEX) {
      const v4u r_c = cold_cone[slot];
      cone = RayCone{uf(r_c.x), uf(r_c.y)};
    }
  }

  pv_lap(0);
  bool finish = false, at_vertex = false;
  Hit hit;
  hit.p = f3{0.f, 0.f, 0.f};
  if (have) {
    // next-event estimation of the previous vertex (mis_integrator.cpp:64-78)
    if ((flags & CF_HAS_S) && !(flags & CF_OCCLUDED))
      result = result + f3{uf(r_nee.x), uf(r_nee.y), uf(r_nee.z)};
    if (!(flags & CF_HAS_R) || (flags & CF_KILL_R)) {
      finish = true;   // the BSDF sample failed there: return bounce_result (:86-88,:108-114)
    } else {
      const bool hit_any = (flags & CF_FOUND) != 0;
      if (hit_any) {
        HitRec hr;
        hr.e0 = uf(r_hit.x), hr.e1 = uf(r_hit.y), hr.e2 = uf(r_hit.z);
        hr.inv_det = uf(r_hit.w);
        hr.prim = r_hx.x;
        hr.kind = (flags & CF_KIND_SPHERE) ? 1u : 0u;
        TravRay tr{ray_o, ray_d, 0.0001f, uf(r_hx.y)};
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
              float Gt = geometric_term(ray_o, hit.p, hit.ng);
              float mis_weight = balance_heuristic(prev_pdf * Gt, light_pdf);
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

  pv_lap(1);
  // A class batch does not change the path's running result after this point (under the material integrator
  // it is ASSIGNED below and stored again): it leaves the registers here, not at the end of the stage.
  if constexpr (!finisher_batch) {
    if (have && !mirror) {
      VIMG_GLOBAL uint32_t* rw = reinterpret_cast<VIMG_GLOBAL uint32_t*>(cold + (slot * SC4_MAIN + SC_RESULT));
      rw[0] = fu(result.x), rw[1] = fu(result.y), rw[2] = fu(result.z);
    }
  }
  // ---- the next rays of a vertex
  bool has_s = false, has_r = false;
  bool early = false, pushed_r = false;   // EARLY launches: this lane queues its rays before the end of the stage / has queued its path ray
  bool rec_written = false;               // this lane's rays are in the slot's record already (a vertex of the mis integrator)
  f3 shadow_d{0.f, 0.f, 1.f}, nee_contrib{0.f, 0.f, 0.f};
  float shadow_max_t = 0.f;
  // (a finisher batch never holds a vertex to shade: the walk sends every hit on a non-emitter to
  // its material's class - class 3 for everything under the material integrator)
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
    // mis_integrator.cpp:45-122 in three phases.  The DRAW order is the reference's: light pick +
    // emitter sample (always six draws), then sample_mat - but the BSDF sample is COMPUTED first,
    // from the state six draws on (pcg_skip6), and the light sample after it from the state as it
    // was: same numbers, and the path ray - the one whose walk the next vertex waits for - is known
    // a light sample earlier.  Between the phases - at wave level, every lane there - the EARLY
    // build of a launch queues each ray as soon as it is known, so that both are being walked
    // while this wave still samples the light, evaluates the BSDF twice and stores the slot's
    // state.  On a frame short of pixels the last pixels' chains of (vertex stage, walk) hops are
    // the frame time, and the walk then runs beside most of the vertex stage instead of behind it.
    // A slot is complete when its rays AND its vertex stage have finished (CF_DONE_V, the last OR decides).
    constexpr int MT = MTC;
    const bool early_on = EARLY == 2 ? (A.cu_flex & 32u) != 0u : (EARLY == 1);   // (a build of its own: as a runtime branch it costs the whole frame 1.2 %)
    uint32_t mat_type = 0u;
    float hit_dist = 0.f, surface_spread_angle = 0.f;
    f3 light_col{0.f, 0.f, 0.f};
    EmitterInfo li{f3{0.f, 0.f, 1.f}, 0.f, 0.f, 0.f};
    bool nee = false, reg_before = false;
    RayCone nee_cone = cone;
    Scatter sc = no_scatter();
    Rng rng_l{0};   // the stream where the light sample draws
    if (at_vertex) {
      mat_type = MT >= 0 ? uint32_t(MT) : g.materials[hit.mat].type;
      if constexpr (TEX) {
        hit_dist = length(ray_o - hit.p);
        surface_spread_angle = spread_angle_from_curvature(hit.curvature, cone.cone_width, ray_d, hit.ns);
      }
      reg_before = non_specular_bounce;
      rng_l = rng;
      if (mat_type != VIMG_MAT_DIELECTRIC) pcg_skip6(rng);   // (!is_delta: the light sample's six draws)
      sc = sample_mat<TEX, MT>(g, hit, ray_d, rng, reg_before);
      if constexpr (TEX) nee_cone = propagate_reflect_cone(cone, surface_spread_angle * 2.f, hit_dist);
      if (sc.valid) {
        if (!sc.is_specular) non_specular_bounce = true;
        if (sc.eta != 0.f) {
          eta_scale /= (sc.eta * sc.eta);
          if constexpr (TEX) cone = propagate_refract_cone(cone, ray_d, surface_spread_angle, sc.eta, sc.wo);
        } else {
          if constexpr (TEX) cone = nee_cone;
        }
      }
    }
    early = early_on && at_vertex && !mirror;
    pushed_r = early && sc.valid;
    // The rays go into the slot's record as soon as they are known - origin, path ray, below the shadow
    // ray - whenever they are queued: the evaluations take their directions from there, and ten
    // registers are free while they run.  EARLY launches: with the flag word the walkers will OR into
    // (nobody else touches it yet), before the first ray is queued; else the word follows at the end.
    rec_written = at_vertex && !mirror;
    if (rec_written) {
      wr(CR_ORG, slot, v4u{fu(hit.p.x), fu(hit.p.y), fu(hit.p.z), 0u});
      wr(CR_DIR, slot, v4u{fu(sc.wo.x), fu(sc.wo.y), fu(sc.wo.z),
                           (sc.valid ? CF_HAS_R : 0u) | (non_specular_bounce ? CF_NONSPEC : 0u) | (bounce << CF_BOUNCE_SHIFT)});
      if (full_stats && early) tq[slot] = static_cast<uint32_t>(__builtin_readcyclecounter());
    }
    if (early_on) push_rays(false, pushed_r, slot);
    pv_lap(3);
    if (at_vertex && mat_type != VIMG_MAT_DIELECTRIC) {
      lights_sample<TEX>(g, hit.p, rng_l, light_col, li);
      nee = (li.pdf != 0.f);
    }
    if (rec_written && nee) {
      // the shadow ray and its reach (the reference's absolute epsilon, quirk Q15) beside a record whose
      // other words a walker may be reading: one dword
      recw[(CR_ORG * P + slot) * 4u + 3u] = fu(li.dist - 0.0001f);
      wr(CR_SHD, slot, v4u{fu(li.wi.x), fu(li.wi.y), fu(li.wi.z), hops});
      if (early) __hip_atomic_fetch_or(flag_word(slot), CF_HAS_S, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (rec_written && full_stats) {
      recw[(CR_SHD * P + slot) * 4u + 3u] = hops;
    }
    if (early_on) push_rays(early && nee, false, slot);
    pv_lap(2);
    if (at_vertex) {
      // both BSDF evaluations happen before either ray's result is looked at: the evaluation towards
      // the light is pure, so doing it for a light that turns out occluded changes nothing; its
      // regularisation flag is the one from BEFORE this bounce
      f3 f_l{0.f, 0.f, 0.f}, f_s{0.f, 0.f, 0.f};   // towards the light / along the sampled direction
      float pdf_l = 0.f, pdf_s = 0.f;
#pragma unroll 1
      for (int pass = 0; pass < (split ? 1 : 2); ++pass) {
        const bool second = split ? mirror : (pass == 1);   // which of the two this lane evaluates in this pass
        const bool run = second ? sc.valid : nee;
        f3 f{0.f, 0.f, 0.f};
        float pdf = 0.f;
        if (run) {
          const v4u w4 = rd(second ? CR_DIR : CR_SHD, slot);   // (sc.wo / li.wi, from the slot's record)
          const f3 wo{uf(w4.x), uf(w4.y), uf(w4.z)};
          const RayCone c = second ? cone : nee_cone;
          const bool reg = second ? non_specular_bounce : reg_before;
          eval_pdf_pair<TEX, MT>(g, hit, ray_d, wo, c, reg, f, pdf);
        }
        if (second) f_s = f, pdf_s = pdf;
        else f_l = f, pdf_l = pdf;
      }
      pv_lap(4);
      if (split) {   // the owner lane takes the sampled direction's evaluation from its mirror lane
        const float sx = __shfl_xor(f_s.x, 32), sy = __shfl_xor(f_s.y, 32), sz = __shfl_xor(f_s.z, 32), sp = __shfl_xor(pdf_s, 32);
        if (!mirror) f_s = f3{sx, sy, sz}, pdf_s = sp;
      }
      if (nee) {
        if (pdf_l != 0 && !is_nan(pdf_l)) {
          float Gt = li.G;
          float mis_weight = balance_heuristic(li.pdf, pdf_l * Gt);
          nee_contrib = throughput * f_l * mis_weight * Gt * light_col / li.pdf;
        }
        // pdf == 0 / NaN: nothing is added, but the reference has traced its shadow ray by
        // then (mis_integrator.cpp:64): it is still traced and counted
      }
      if (sc.valid) {
        if (is_nan(pdf_s)) {
          sc.valid = false;   // NaN pdf terminates the path (mis_integrator.cpp:108-114)
        } else {
          throughput = throughput * (f_s / pdf_s);
          prev_pdf = pdf_s;
        }
      }
      has_s = nee;
      has_r = sc.valid;
      primary = false;
      if (!has_s && !has_r) finish = true;
    }
  }

  // ---- finished samples: accumulate, pixel write-back, next pixel, next camera ray
  bool need_pixel = fresh;
  bool retire = false;
  bool have_claim = fresh && (flags & CF_PRIMARY);   // a fresh slot may hold a claim whose predecessor segment was not published yet
  uint32_t claim = have_claim ? crd(SC4_RNG, slot).w : 0u;
  bool pending = false;
  if constexpr (!finisher_batch) {
    // a path that ended at this vertex is accumulated by the finisher stage: it travels there
    // with neither ray set, which that stage reads as "return bounce_result"
    if (finish) has_s = false, has_r = false;
    all_pending = false;
  } else {
    bool pixels_left = cu_uni(lds_load(&G->pixels_left)) != 0u;
    bool is_nan_sample = false;
    if (finish) {
      is_nan_sample = is_nan(result.x) || is_nan(result.y) || is_nan(result.z);
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
        if (full_stats) {   // when pixels finish, in 10 ns ticks since the workgroup started: sum, count, latest; and their hops
          const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - (static_cast<unsigned long long>(G->t0_lo) | (static_cast<unsigned long long>(G->t0_hi) << 32));
          DeviceStats* __restrict__ stp = K->stats;
          if (stp) {
            atomicAdd(&stp->px_done[0], dt), atomicAdd(&stp->px_done[1], 1ull), atomicMax(&stp->px_done[2], dt);
            // (time << 24 | hops: the maximum is the pixel that finished last, with its hop count)
            atomicMax(&stp->px_hops[2], (dt << 24) | static_cast<unsigned long long>(hops & 0xffffffu));
            atomicAdd(&stp->px_hops[0], static_cast<unsigned long long>(hops)), atomicMax(&stp->px_hops[1], static_cast<unsigned long long>(hops));
          }
          hops = 0;
        }
      } else if (n_seg > 1u && smp % seg_len == 0u) {
        // end of a segment: the pixel rests in its record until a slot draws its next segment
        // (words written and read with agent-scope relaxed atomics: data, wait, then the tag)
        VIMG_GLOBAL uint32_t* st = reinterpret_cast<VIMG_GLOBAL uint32_t*>(A.pool_state + size_t(item) * 2u);
        state_store(st + 0, static_cast<uint32_t>(rng.s));
        state_store(st + 1, static_cast<uint32_t>(rng.s >> 32));
        state_store(st + 4, fu(acc.x));
        state_store(st + 5, fu(acc.y));
        state_store(st + 6, fu(acc.z));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
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
          if (lane == 0) __hip_atomic_store(&G->pixels_left, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
            VIMG_GLOBAL uint32_t* st = reinterpret_cast<VIMG_GLOBAL uint32_t*>(A.pool_state + size_t(item) * 2u);
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
    n_nan += static_cast<uint32_t>(__popcll(__ballot(is_nan_sample)));
    all_pending = (__ballot(pending) == __ballot(on));
    const bool regen = on && !retire && !pending && (finish || fresh);
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

  // ---- registers -> slot state, slot -> walk ring (both rays of the vertex at once) or finisher ring
  const bool keep = on && !retire && !mirror;
  if (keep) {
    if (!early) {
      // a slot that waits for its item's previous segment stays "fresh" and keeps the claim
      const uint32_t nf = pending ? (CF_FRESH | CF_PRIMARY)
                                  : ((primary ? CF_PRIMARY : 0u) | (non_specular_bounce ? CF_NONSPEC : 0u) |
                                     (has_s ? CF_HAS_S : 0u) | (has_r ? CF_HAS_R : 0u) | CF_DONE_V | (bounce << CF_BOUNCE_SHIFT));
      if (pending) has_s = false, has_r = false, smp = claim;
      if (rec_written) {
        *flag_word(slot) = nf;
      } else {
        wr(CR_ORG, slot, v4u{fu(ray_o.x), fu(ray_o.y), fu(ray_o.z), fu(shadow_max_t)});
        wr(CR_DIR, slot, v4u{fu(ray_d.x), fu(ray_d.y), fu(ray_d.z), nf});
        if (has_s) wr(CR_SHD, slot, v4u{fu(shadow_d.x), fu(shadow_d.y), fu(shadow_d.z), hops});
        else if (full_stats) recw[(CR_SHD * P + slot) * 4u + 3u] = hops;
      }
    }
    cwr(SC_THROUGHPUT, slot, v4u{fu(throughput.x), fu(throughput.y), fu(throughput.z), fu(eta_scale)});
    if constexpr (finisher_batch) {
      cwr(SC_RESULT, slot, v4u{fu(result.x), fu(result.y), fu(result.z), fu(prev_pdf)});
      cwr(SC4_RNG, slot, v4u{static_cast<uint32_t>(rng.s), static_cast<uint32_t>(rng.s >> 32), px | (py << 16), smp});
    } else {
      // (the result went out above; the pixel and the sample count of the record are the finisher's)
      VIMG_GLOBAL uint32_t* rw = reinterpret_cast<VIMG_GLOBAL uint32_t*>(cold + (slot * SC4_MAIN + SC_RESULT));
      rw[3] = fu(prev_pdf);
      if (material_mode) rw[0] = fu(result.x), rw[1] = fu(result.y), rw[2] = fu(result.z);
      *reinterpret_cast<VIMG_GLOBAL v2u*>(cold + (slot * SC4_MAIN + SC4_RNG)) = v2u{static_cast<uint32_t>(rng.s), static_cast<uint32_t>(rng.s >> 32)};
    }
    if (has_s) cwr(SC_NEE, slot, v4u{fu(nee_contrib.x), fu(nee_contrib.y), fu(nee_contrib.z), 0u});
    if (finisher_batch) cold_acc[slot] = v4u{fu(acc.x), fu(acc.y), fu(acc.z), item};
    if constexpr (TEX) cold_cone[slot] = v4u{fu(cone.cone_width), fu(cone.spread_angle), 0u, 0u};
  }
  if (full_stats && keep && !early) tq[slot] = static_cast<uint32_t>(__builtin_readcyclecounter());
  // the next stage of a slot may run on another wave of the CU: its cold records must have left
  // this wave before the slot id does (same L1: performed = visible)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  {
    // (lanes that have not queued their rays yet) one reservation for both kinds of ray
    push_rays(keep && !early && has_s, keep && !early && has_r, slot);
    // a path that ended here goes to the finisher ring; so do slots that wait for a segment
    push(0u, __ballot(keep && !early && !has_s && !has_r), slot);
    const uint32_t n_retired = static_cast<uint32_t>(__popcll(__ballot(on && retire)));
    if (n_retired && lane == 0) __hip_atomic_fetch_sub(&G->live, n_retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if constexpr (!FIN) {
    if (EARLY == 2 ? (cu_uni(A.cu_flex) & 32u) != 0u : (EARLY == 1)) {
      // lanes whose rays are already out: the state is written; if both rays have been walked meanwhile
      // this OR completes the slot and this wave hands it on.  A path ray queued before the evaluation found
      // the path ended (NaN pdf) is walked for nothing: its result is marked dead and the ray is not counted
      // (the reference never traced it).
      const bool mine = keep && early, dead = mine && pushed_r && !has_r;
      uint32_t nw = 0;
      bool complete = false;
      if (mine) {
        const uint32_t orv = CF_DONE_V | (dead ? CF_KILL_R : 0u);
        nw = __hip_atomic_fetch_or(flag_word(slot), orv, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) | orv;
        complete = slot_complete(nw);
      }
      hand_to_vertex(complete, nw, slot);
      n_dead += static_cast<uint32_t>(__popcll(__ballot(dead)));
    }
  }
  pv_lap(5);
  if constexpr (PV) {
    if (full_stats && lane == 0) wrec->pv_n += 1;
  }
}

// ======================================================================== one walk session
// Refill idle lanes from the walk ring, step the rays, hand finished ones over; ends when the wave
// holds no ray and the ring is empty.
template <bool TEX, bool DEEP, int NW, bool DIAG>
VD void cu_walk(uint32_t& n_closest, uint32_t& n_shadow) {
  const CuKPtr K = cu_kargs();
  CU_STAGE_LOCALS(K);
  const Lds L = lds_layout(A, (VIMG_LDS unsigned char*)lds_raw);
  const uint32_t S = A.stack_lds;
  VIMG_GLOBAL uint32_t* ovf0 = A.stack_ovf + (size_t(blockIdx.x) * A.cu_walkers + (can_walk ? wave : 0u)) * ((A.stack_entries - S) * 64u) + lane;
  // walk state of the lane: nothing of it lives outside a session
  uint32_t w_slot = SLOT_IDLE, w_type = 0, w_cls = 0, sp = 0, cur = REF_DONE;
  bool w_setup = false, w_any = false, w_found = false, w_exact = false;
  TravRay ray{f3{0.f, 0.f, 0.f}, f3{0.f, 0.f, 1.f}, 0.0001f, VIMG_INF};
  f3 w_inv{1.f, 1.f, 1.f};
  TriRayConst rc{0.f, 0.f, 1.f, 2};
  float w_dir_len2 = 1.f;
  HitRec rec;
  rec.prim = 0xffffffffu, rec.kind = 0;
  rec.e0 = rec.e1 = rec.e2 = rec.inv_det = 0.f;
  uint32_t c_internal = 0, c_leaf = 0, c_prim = 0, c_sphere = 0;   // per-lane event counts of statistics launches
  uint32_t d_box = 0, d_boxl = 0, d_leaf = 0, d_leafl = 0, d_ref = 0, d_refr = 0;   // wave-uniform: passes and lanes (statistics launches)
  uint32_t t_in = 0;   // statistics launches: when this lane took its ray
  unsigned long long d_waitc = 0, d_rayc = 0;   // ... per-lane sums (one atomic per wave and session, not one per ray: the
  uint32_t d_waitn = 0, d_rayn = 0;             //     statistics must not be what they measure)
  unsigned long long wt = full_stats ? __builtin_readcyclecounter() : 0ull, wc0 = 0, wc1 = 0, wc2 = 0, wc3 = 0;
  auto wlap = [&](unsigned long long& acc) {
    if (full_stats) {
      const unsigned long long now = __builtin_readcyclecounter();
      acc += now - wt;
      wt = now;
    }
  };
  for (;;) {
    // (1) idle lanes take queued rays
    {
      const unsigned long long m_idle = __ballot(w_slot == SLOT_IDLE);
      const uint32_t n_idle = static_cast<uint32_t>(__popcll(m_idle));
      if (n_idle >= A.pool_refill || n_idle == 64u) {
        const int32_t aw = static_cast<int32_t>(cu_uni(static_cast<uint32_t>(lds_load(&G->avail[CQ_WALK]))));
        if (aw > 0) {
          const uint32_t want = n_idle < static_cast<uint32_t>(aw) ? n_idle : static_cast<uint32_t>(aw);
          uint32_t e = 0;
          const uint32_t got = pop(CQ_WALK, want, m_idle, e);
          d_ref += stat_inc, d_refr += full_stats ? got : 0u;
          if (w_slot == SLOT_IDLE && lane_rank(m_idle, lane) < got) {
            if (full_stats) {
              t_in = static_cast<uint32_t>(__builtin_readcyclecounter());
              d_waitc += static_cast<unsigned long long>(t_in - tq[e & 0x7fffu]), d_waitn += 1u;
            }
            w_slot = e & 0x7fffu;
            w_type = (e & CU_RAY_S) ? 0u : 1u;
            w_setup = true;
          }
        }
      }
    }
    // (2) ray set-up (reference include/bvh.h:109-143): everything derived from the ray alone
    if (__any(w_setup)) {
      n_shadow += static_cast<uint32_t>(__popcll(__ballot(w_setup && w_type == 0u)));
      n_closest += static_cast<uint32_t>(__popcll(__ballot(w_setup && w_type != 0u)));
      if (w_setup) {
        const v4u ro = rd(CR_ORG, w_slot);
        ray.o = f3{uf(ro.x), uf(ro.y), uf(ro.z)};
        if (w_type == 0u) {
          const v4u rs = rd(CR_SHD, w_slot);
          ray.d = f3{uf(rs.x), uf(rs.y), uf(rs.z)};
          ray.max_t = uf(ro.w);
          w_any = true;
        } else {
          const v4u rr = rd(CR_DIR, w_slot);
          ray.d = f3{uf(rr.x), uf(rr.y), uf(rr.z)};
          ray.max_t = VIMG_INF;
          w_any = false;
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
    wlap(wc0);
    if (__ballot(w_slot != SLOT_IDLE) == 0ull) break;

    // (3) walk until pool_refill rays have finished (or nothing is left to walk): "while-while",
    // the box loop until every ray stands at a leaf (or is done), then the leaves
    const bool exact_round = __any(w_exact && w_slot != SLOT_IDLE);
    for (;;) {
      auto box_loop = [&](auto exact_possible) {
        for (;;) {
          const bool act = cur != REF_DONE && ref_count(cur) == 0;
          if (!__any(act)) break;
          if (full_stats) d_box += 1u, d_boxl += static_cast<uint32_t>(__popcll(__ballot(act)));
          if (act) {
            v4f na, nb, nc;
            v2u refs;
            if (!DEEP || cur < L.n_nodes) {
              na = L.na[cur], nb = L.nb[cur], nc = L.nc[cur];
              refs = L.nm[cur];
            } else {
              gptr<DNode> nd = g.nodes + cur;
              na = nd->a, nb = nd->b, nc = nd->c;
              refs = v2u{nd->left_ref, nd->right_ref};
            }
            // one step, branch-free: the entry a pop would return is read before the box test;
            // the far child is written above the top of the stack whether it is kept or not
            const uint32_t sp_below = sp != 0 ? sp - 1 : 0u;
            uint32_t popped = stack0[(DEEP ? (sp_below < S ? sp_below : S) : sp_below) * 64];
            c_internal += stat_inc;
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
            const bool both = in1 && in2, any = in1 || in2;
            const bool first_is_near = w_any ? false : (h2 > h1);
            const uint32_t near_c = first_is_near ? c1 : c2;
            const uint32_t far_c = first_is_near ? c2 : c1;
            stack0[(DEEP ? (sp < S ? sp : S) : sp) * 64] = far_c;
            if constexpr (DEEP) {
              if (both && sp >= S) ovf0[(sp - S) * 64] = far_c;
              if (!any && sp_below >= S) popped = ovf0[(sp_below - S) * 64];
            }
            const uint32_t one_c = in1 ? c1 : c2;
            cur = both ? near_c : (any ? one_c : (sp != 0 ? popped : REF_DONE));
            sp = both ? sp + 1 : (any ? sp : sp_below);
          }
          if constexpr (DEEP) {
            if (static_cast<uint32_t>(__popcll(__ballot(cur != REF_DONE && ref_count(cur) == 0))) < box_min) break;
          }
        }
      };
      if (exact_round)
        box_loop(std::true_type{});
      else
        box_loop(std::false_type{});
      wlap(wc1);

      if (full_stats) d_leaf += 1u, d_leafl += static_cast<uint32_t>(__popcll(__ballot(cur != REF_DONE && (!DEEP || ref_count(cur) != 0))));
      if (cur != REF_DONE && (!DEEP || ref_count(cur) != 0)) {
        const uint32_t first = ref_index(cur), count = ref_count(cur);
        c_leaf += stat_inc;
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
          const float c0 = cc.x;
          const uint32_t lp_prim = __float_as_uint(cc.y), kind = __float_as_uint(cc.z), lp_cls = __float_as_uint(cc.w);
          c_prim += stat_inc;
          bool hit = false;
          float t = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f, idet = 0.f;
          if (kind == 0) {
            hit = tri_test_flat(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c0}, ray, rc, t, e0, e1, e2, idet);
          } else if (kind == 1) {
            c_sphere += stat_inc;
            hit = sphere_test(f3{a.x, a.y, a.z}, a.w, ray, w_dir_len2, t);
          }
          ray.max_t = hit ? t : ray.max_t;
          w_found = w_found || hit;
          const bool keep_rec = hit && !w_any;
          rec.e0 = keep_rec ? e0 : rec.e0, rec.e1 = keep_rec ? e1 : rec.e1;
          rec.e2 = keep_rec ? e2 : rec.e2, rec.inv_det = keep_rec ? idet : rec.inv_det;
          rec.prim = keep_rec ? lp_prim : rec.prim;
          rec.kind = keep_rec ? kind : rec.kind;
          w_cls = keep_rec ? lp_cls : w_cls;
          stop = hit && w_any;
        }
        // the leaf is done: the next node comes off the stack (or the ray is finished)
        const uint32_t sp_below = sp != 0 ? sp - 1 : 0u;
        uint32_t popped = stack0[(DEEP ? (sp_below < S ? sp_below : S) : sp_below) * 64];
        if constexpr (DEEP) {
          if (!stop && sp_below >= S) popped = ovf0[(sp_below - S) * 64];
        }
        cur = (stop || sp == 0) ? REF_DONE : popped;
        sp = sp_below;
      }
      wlap(wc2);
      const uint32_t n_fin = static_cast<uint32_t>(__popcll(__ballot(w_slot != SLOT_IDLE && cur == REF_DONE)));
      const uint32_t n_act = static_cast<uint32_t>(__popcll(__ballot(w_slot != SLOT_IDLE && cur != REF_DONE)));
      if (n_act == 0 || n_fin >= A.pool_refill) break;
    }

    // (4) finished rays: result into the slot, OR into its flag word; whoever completes the slot - a
    // ray, or the vertex stage that queued it when that one finishes last - hands it to the vertex
    // ring of its class
    {
      const bool done = w_slot != SLOT_IDLE && cur == REF_DONE;
      bool complete = false;
      uint32_t nw = 0;
      if (done) {
        uint32_t orv;
        if (w_type == 0u) {
          orv = CF_DONE_S | (w_found ? CF_OCCLUDED : 0u);
        } else {
          orv = CF_DONE_R;
          if (w_found) {
            orv |= CF_FOUND | (rec.kind == 1 ? CF_KIND_SPHERE : 0u) | (w_cls << CF_CLS_SHIFT);
            wr(CR_HIT, w_slot, v4u{fu(rec.e0), fu(rec.e1), fu(rec.e2), fu(rec.inv_det)});
            hitx[w_slot] = v2u{rec.prim, fu(ray.max_t)};
          }
        }
        nw = __hip_atomic_fetch_or(flag_word(w_slot), orv, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) | orv;
        complete = slot_complete(nw);
      }
      if (__any(done)) {
        hand_to_vertex(complete, nw, w_slot);
        if (full_stats && done) d_rayc += static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_readcyclecounter()) - t_in), d_rayn += 1u;
        if (done) w_slot = SLOT_IDLE;
      }
    }
    wlap(wc3);
    if (cu_uni(lds_load(&G->abort)) != 0u) break;
  }
  if (full_stats) {
    if (lane == 0) wrec->w_cyc[0] += wc0, wrec->w_cyc[1] += wc1, wrec->w_cyc[2] += wc2, wrec->w_cyc[3] += wc3;
    auto wsum = [](unsigned long long v) {
      for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
      return v;
    };
    const unsigned long long s_int = wsum(c_internal), s_leaf = wsum(c_leaf), s_prim = wsum(c_prim), s_sph = wsum(c_sphere);
    const unsigned long long s_wc = wsum(d_waitc), s_wn = wsum(d_waitn), s_rc = wsum(d_rayc), s_rn = wsum(d_rayn);
    if (lane == 0) {
      wrec->internal += static_cast<uint32_t>(s_int), wrec->leaf += static_cast<uint32_t>(s_leaf);
      wrec->prim += static_cast<uint32_t>(s_prim), wrec->sphere += static_cast<uint32_t>(s_sph);
      wrec->wait_cyc[4] += s_wc, wrec->wait_n[4] += s_wn, wrec->ray_cyc += s_rc, wrec->ray_n += s_rn;
    }
    if (lane == 0) {
      wrec->box_pass += d_box, wrec->box_lanes += d_boxl, wrec->leaf_round += d_leaf, wrec->leaf_lanes += d_leafl;
      wrec->sessions += 1, wrec->refills += d_ref, wrec->refill_rays += d_refr;
    }
  }
}

// NW: waves of the workgroup (16: one workgroup is the whole CU at four waves per SIMD, 128 registers);
// WPS: waves per SIMD the register budget leaves room for.
// DIAG: the build of statistics launches (full_stats): event counts beyond the two ray counts, cycles by
// stage, ring occupancy and waiting times, passes and lanes of the walk loops, when pixels finish.  The
// build without them is what a frame is timed on - as runtime branches they cost it 4 % (304 -> 316 ms).
// EARLY: 1 = vertex stages queue their rays as soon as they are known (cu_flex bit 5, by policy on frames
// short of pixels and on trees in global memory), 0 = at the end of the stage, 2 = by the bit at run time
// (the statistics build).
template <bool TEX, bool DEEP, int NW, int WPS, bool DIAG, int EARLY>
__global__ void __launch_bounds__(NW * 64, WPS)
render_cu_kernel(const CuKArgs ka) {
  {
    const CuKPtr K = cu_kargs();
    CU_STAGE_LOCALS(K);
    stage_lds(g, A, (VIMG_LDS unsigned char*)lds_raw);
    for (uint32_t i = lane; i < sizeof(CuWaveRec) / 4u; i += 64u) reinterpret_cast<VIMG_LDS uint32_t*>(wrec)[i] = 0u;
    for (uint32_t i = threadIdx.x; i < A.lds_leaf * 3u; i += blockDim.x)
      lds_leaf[i] = reinterpret_cast<gptr<v4f>>(g.leaf_prims)[i];
    // every slot starts "fresh" in the finisher queue: it needs a pixel
    for (uint32_t s = threadIdx.x; s < P; s += blockDim.x) {
      recw[(CR_DIR * P + s) * 4u + 3u] = CF_FRESH;
      ring_v[s] = static_cast<uint16_t>(s);
      ring_v[P + s] = CU_EMPTY, ring_v[2u * P + s] = CU_EMPTY, ring_v[3u * P + s] = CU_EMPTY;
      ring_w[s] = CU_EMPTY, ring_w[P + s] = CU_EMPTY;
      tq[s] = static_cast<uint32_t>(__builtin_readcyclecounter());
    }
    if (threadIdx.x == 0) {
      for (uint32_t q = 0; q < 8u; ++q) G->avail[q] = (q == 0u) ? static_cast<int32_t>(P) : 0;
      G->live = P, G->pixels_left = 1u, G->abort = 0u;
      for (uint32_t q = 0; q < CQ_COUNT; ++q) G->tail[q] = (q == 0u) ? P : 0u, G->head[q] = 0u;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      G->t0_lo = static_cast<uint32_t>(t0), G->t0_hi = static_cast<uint32_t>(t0 >> 32);
    }
    __syncthreads();
  }
  // wave-uniform event counts (scalar registers): the two ray counts every stats launch reports
  uint32_t n_closest = 0, n_shadow = 0, n_nan = 0, iter_wave = 0;
  uint32_t n_dead = 0;   // path rays queued early and walked for nothing (their vertex found the path ended): not counted, as the reference never traced them
  unsigned long long t_mark = __builtin_readcyclecounter();
  bool skip_fin = false;   // the last finisher batch of this wave held only slots that wait for another slot's segment
  uint32_t polls = 0;      // looks that found nothing to do
  unsigned long long idle_since = 0;
  for (;;) {
    const CuKPtr K = cu_kargs();
    CU_STAGE_LOCALS(K);
    auto lap = [&](uint32_t k) {
      if (full_stats) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0) wrec->cyc[k] += now - t_mark;
        t_mark = now;
      }
    };
    if (cu_uni(lds_load(&G->abort)) != 0u) break;
    const v4u av4 = *reinterpret_cast<VIMG_LDS v4u*>(&G->avail[0]);
    const int32_t a0 = skip_fin ? 0 : static_cast<int32_t>(cu_uni(av4.x)), a1 = static_cast<int32_t>(cu_uni(av4.y)),
                  a2 = static_cast<int32_t>(cu_uni(av4.z)), a3 = static_cast<int32_t>(cu_uni(av4.w));
    const int32_t aw = static_cast<int32_t>(cu_uni(static_cast<uint32_t>(lds_load(&G->avail[CQ_WALK]))));
    const int32_t m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3, qmax = m01 > m23 ? m01 : m23;
    if (full_stats && lane == 0) {
      wrec->looks += 1;
      wrec->q_sum[0] += static_cast<uint32_t>(a0 > 0 ? a0 : 0), wrec->q_sum[1] += static_cast<uint32_t>(a1 > 0 ? a1 : 0);
      wrec->q_sum[2] += static_cast<uint32_t>(a2 > 0 ? a2 : 0), wrec->q_sum[3] += static_cast<uint32_t>(aw > 0 ? aw : 0);
    }
    // rays gather in the waves that already walk (they refill on their own); a wave that holds none
    // joins when cu_join rays wait, or when fewer have waited through cu_patience of its looks
    if (can_walk && aw > 0 && (aw >= static_cast<int32_t>(A.cu_join) || polls >= A.cu_patience)) {
      // ---------------------------------------------------------------- WALK
      lap(5);
      skip_fin = false, polls = 0, idle_since = 0;
      if (full_stats) iter_wave++;
      if (A.cu_flex & 4u) __builtin_amdgcn_s_setprio(1);
      cu_walk<TEX, DEEP, NW, DIAG>(n_closest, n_shadow);
      if (A.cu_flex & 4u) __builtin_amdgcn_s_setprio(0);
      lap(4);
      continue;
    }
    // a vertex batch: a full one at once; a partial one when the walkers are about to run dry (few
    // rays queued) and it is worth a wave's while (pool_starve slots, or this wave has looked in vain
    // a few times).  Walking waves shade only when they hold no ray (here) and A.cu_flex allows it.
    const bool may_shade = !can_walk || (A.cu_flex & 1u);
    const bool run = may_shade && qmax > 0 &&
                     (qmax >= static_cast<int32_t>(A.pool_vbatch) ||
                      (aw < static_cast<int32_t>(A.cu_lowwater) && (qmax >= static_cast<int32_t>(A.pool_starve) || polls >= A.cu_patience)));
    if (run) {
      // ---------------------------------------------------------------- VERTEX batch
      const uint32_t cls = (a0 == qmax) ? 0u : (a1 == qmax ? 1u : (a2 == qmax ? 2u : 3u));
      const uint32_t want = qmax < 64 ? static_cast<uint32_t>(qmax) : 64u;
      uint32_t e = 0;
      const uint32_t n = pop(cls, want, ~0ull, e);
      if (n == 0u) continue;   // another wave took them
      lap(5);
      if (full_stats) iter_wave++;
      if (full_stats && lane == 0) wrec->nbatch[cls] += 1, wrec->nslots[cls] += n;
      if (full_stats) {
        unsigned long long w = lane < n ? static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_readcyclecounter()) - tq[e]) : 0ull;
        for (int o = 32; o; o >>= 1) w += __shfl_xor(w, o);
        if (lane == 0) wrec->wait_cyc[cls] += w, wrec->wait_n[cls] += n;
      }
      bool all_pending = false;
      const bool by_class = A.pool_classes == 3u;
      if (A.cu_flex & 2u) __builtin_amdgcn_s_setprio(1);
      if (cls == 0u)
        cu_vertex<TEX, NW, DIAG, EARLY, true, -1>(n, e, all_pending, n_nan, n_dead);
      else if (cls == 1u && by_class)
        cu_vertex<TEX, NW, DIAG, EARLY, false, int(VIMG_MAT_LAMBERTIAN)>(n, e, all_pending, n_nan, n_dead);
      else if (cls == 2u && by_class)
        cu_vertex<TEX, NW, DIAG, EARLY, false, int(VIMG_MAT_PRINCIPLED)>(n, e, all_pending, n_nan, n_dead);
      else
        cu_vertex<TEX, NW, DIAG, EARLY, false, -1>(n, e, all_pending, n_nan, n_dead);
      if (A.cu_flex & 2u) __builtin_amdgcn_s_setprio(0);
      skip_fin = all_pending;
      if (!all_pending) polls = 0, idle_since = 0;   // (a batch of waiting slots only is not progress: the watchdog keeps its time)
      lap(cls);
      continue;
    }
    // ------------------------------------------------------------------ nothing for this wave now
    if (cu_uni(lds_load(&G->live)) == 0u) break;   // every slot of the CU has retired
    skip_fin = false;
    polls += 1;
    // Watchdog by wall clock (s_memrealtime counts at 100 MHz): a wave that has found nothing to do
    // for ten seconds while slots are live is waiting for something that will not come; it raises
    // the launch's error word so that a scheduling bug ends as VIMG_E_DEVICE instead of a hung GPU.
    // The one legitimate long wait is for a pixel's previous SEGMENT in another workgroup's hands, and a
    // segment's time grows with its samples: 50 ms for each of them on top (hundreds of their paths; the
    // limit comes from the host, in units of 2^20 ticks).
    if ((polls & 255u) == 0u) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (idle_since == 0) idle_since = now;
      else if (static_cast<uint32_t>((now - idle_since) >> 20) > A.cu_watchdog) {
        if (lane == 0) raise(1u);
        break;
      }
    }
    // (s_sleep takes an immediate)
    if (A.cu_sleep >= 32u) __builtin_amdgcn_s_sleep(32);
    else if (A.cu_sleep >= 16u) __builtin_amdgcn_s_sleep(16);
    else if (A.cu_sleep >= 8u) __builtin_amdgcn_s_sleep(8);
    else if (A.cu_sleep >= 4u) __builtin_amdgcn_s_sleep(4);
    else __builtin_amdgcn_s_sleep(1);
  }

  // ---- flush event counts: one atomic per wave and counter
  {
    const CuKPtr K = cu_kargs();
    CU_STAGE_LOCALS(K);
    DeviceStats* __restrict__ stats = K->stats;
    if (full_stats && lane == 0) wrec->cyc[5] += __builtin_readcyclecounter() - t_mark;
    if (stats && lane == 0) {
      atomicAdd(&stats->closest, static_cast<unsigned long long>(n_closest) - static_cast<unsigned long long>(n_dead));   // (modulo 2^64: the sum over the waves is what counts)
      atomicAdd(&stats->shadow, static_cast<unsigned long long>(n_shadow));
      if (n_nan) atomicAdd(&stats->nan_samples, static_cast<unsigned long long>(n_nan));
      if (full_stats) {
        // (32-bit per-wave event counts: a wave sees 2^32 node visits only beyond ~10^13 per frame)
        atomicAdd(&stats->internal, static_cast<unsigned long long>(wrec->internal));
        atomicAdd(&stats->leaf, static_cast<unsigned long long>(wrec->leaf));
        atomicAdd(&stats->prim, static_cast<unsigned long long>(wrec->prim));
        atomicAdd(&stats->sphere, static_cast<unsigned long long>(wrec->sphere));
        atomicAdd(&stats->iterations, static_cast<unsigned long long>(iter_wave));
        for (int k = 0; k < 6; ++k) atomicAdd(&stats->prof[k], wrec->cyc[k]);
        for (int k = 0; k < 4; ++k) atomicAdd(&stats->prof[6 + k], wrec->nbatch[k]), atomicAdd(&stats->prof[11 + k], wrec->nslots[k]);
        atomicAdd(&stats->prof[10], 1ull);
        const unsigned long long wd[7] = {wrec->box_pass, wrec->box_lanes, wrec->leaf_round, wrec->leaf_lanes, wrec->sessions,
                                          wrec->refills, wrec->refill_rays};
        for (int k = 0; k < 7; ++k) atomicAdd(&stats->prof[16 + k], wd[k]);
        atomicAdd(&stats->prof[23], wrec->looks);
        for (int k = 0; k < 4; ++k) atomicAdd(&stats->prof[24 + k], wrec->q_sum[k]);
        for (int k = 0; k < 4; ++k) atomicAdd(&stats->walk_cyc[k], wrec->w_cyc[k]);
        for (int k = 0; k < 5; ++k) atomicAdd(&stats->wait_cyc[k], wrec->wait_cyc[k]), atomicAdd(&stats->wait_n[k], wrec->wait_n[k]);
        atomicAdd(&stats->ray_cyc[0], wrec->ray_cyc), atomicAdd(&stats->ray_cyc[1], wrec->ray_n);
        for (int k = 0; k < 6; ++k) atomicAdd(&stats->pv_cyc[k], wrec->pv_cyc[k]);
        atomicAdd(&stats->pv_cyc[6], wrec->pv_n);
      }
    }
  }
}

}  // namespace vimg
