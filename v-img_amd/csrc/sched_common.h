// What the pooled schedulers share: the slot records, the flag bits, the per-pixel hand-over words.
#pragma once
#include "render_kernels.h"

namespace vimg {

// Slot state: 16-byte records stored [record][slot], so that one ds_read_b128 / ds_write_b128 (LDS)
// or one global dwordx4 access moves a whole record.
enum : uint32_t {   // hot records, LDS
  SR_ORIGIN = 0,   // o.xyz | shadow max_t            (after the walk .w = t of the hit)
  SR_RAY,          // d.xyz (camera / BSDF ray) | flags
  SR_SHADOW,       // shadow d.xyz | -               (after the walk: e0 e1 e2 inv_det of the hit)
  SR_RNG,          // rng lo | rng hi | px + (py << 16) | sample index   (first thing a vertex needs)
  SR_COUNT
};
enum : uint32_t {   // cold records, global memory
  SC_THROUGHPUT = 0,   // throughput.xyz | eta_scale
  SC_RESULT,           // bounce_result.xyz | prev_pdf
  SC_NEE,              // unoccluded next-event contribution.xyz | -
  SC_ACC,              // accumulated pixel radiance.xyz | work item id
  SC_CONE,             // cone width | spread angle | - | -   (textured build only)
  SC_COUNT
};
// LDS bytes per slot and wave: the hot records, the primitive id plane and five queue rings of
// one-byte slot ids (a wave has at most 256 slots)
constexpr uint32_t POOL_LDS_BYTES = SR_COUNT * 16u + 4u + 5u;
VD uint32_t pool_wave_bytes(uint32_t slots) { return (POOL_LDS_BYTES * slots + 15u) & ~15u; }
enum : uint32_t {
  SF_PRIMARY = 1u, SF_NONSPEC = 2u, SF_HAS_S = 4u, SF_HAS_R = 8u, SF_OCCLUDED = 16u,
  SF_FOUND = 32u, SF_FRESH = 64u, SF_KIND_SPHERE = 128u, SF_BOUNCE_SHIFT = 8u
};
constexpr uint32_t SLOT_IDLE = 0xffffffffu;

// words of a pixel's between-segments record: agent-scope relaxed atomics (global_load/store sc1)
VD void state_store(VIMG_GLOBAL uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
VD uint32_t state_load(VIMG_GLOBAL uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
VD uint32_t lane_rank(unsigned long long mask, uint32_t lane) {
  return __popcll(mask & ((1ull << lane) - 1ull));
}

// ---- cold records of render_pool4_kernel / render_cu_kernel: the four every vertex batch reads and
// writes are ONE aligned 64-byte line, [slot][4] (throughput, result, NEE term, RNG); the pixel
// accumulator (finisher batches only) and the ray cone (textured build only) live in planes of
// their own behind them, so that a vertex batch moves one line per slot and not two
constexpr uint32_t SC4_RNG = 3u;                     // rng lo | rng hi | px + (py << 16) | sample index  (replaces SC_ACC's place)
constexpr uint32_t SC4_MAIN = 4u;                    // SC_THROUGHPUT, SC_RESULT, SC_NEE, SC4_RNG
// 16-byte records per slot in a cold region: main line + accumulator + cone
__host__ __device__ constexpr uint32_t pool4_cold_records(bool tex) { return SC4_MAIN + 1u + (tex ? 1u : 0u); }
// rows of a lane's LDS stack: all entries, or the first stack_lds and one more that takes the
// writes of the entries kept in global memory
__host__ __device__ constexpr uint32_t pool4_stack_rows_of(uint32_t entries, uint32_t in_lds) { return in_lds < entries ? in_lds + 1u : entries; }

}  // namespace vimg
