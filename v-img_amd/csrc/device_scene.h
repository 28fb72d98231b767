// Device-resident scene layout (gfx950).  Built once by vimg_hip_scene_upload from the VimgScene
// tables; every render reads only these arrays.
//
// Layout rules (DESIGN.md "Data layout in HBM"):
//  * everything a lane fetches in one step is one 16-byte-aligned record read with dwordx4 loads;
//    records that are always consumed together share a 64-byte line (BVH node = its own header +
//    both child boxes; leaf primitive = everything the intersection test needs);
//  * records are stored in the order traversal consumes them (leaf primitives in obj_indices
//    order, so a leaf's primitives are consecutive);
//  * the index chasing of the reference (obj_indices -> prims -> mesh -> indices -> vertices,
//    include/bvh.h:152, include/geometry/triangle.h:75-79) is resolved at upload time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vimg_scene.h"

namespace vimg {

// Explicit address spaces: pointers that live inside a by-value kernel argument are otherwise
// generic and every access becomes a flat_* instruction (which also ties up the LDS counter).
#define VIMG_GLOBAL __attribute__((address_space(1)))
#define VIMG_LDS __attribute__((address_space(3)))
template <typename T>
using gptr = const VIMG_GLOBAL T*;
// builtin vector types (unlike HIP's float4 class) can be loaded from any address space
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// One INTERNAL BVH node, 64 B: the boxes of both children exactly as BB_mins_maxes[2c+2 .. 2c+5]
// holds them (reference include/bvh.h:171-188) and a reference to each child.  A child
// reference packs "primitive count << 25 | index": count == 0 -> index of another DNode,
// count > 0 -> a leaf whose primitives are leaf_prims[index .. index+count).  Leaves therefore
// need no node record and no load of their own.
struct __attribute__((aligned(16))) DNode {
  v4f a;         // Lmin.x Lmin.y Lmin.z Lmax.x
  v4f b;         // Lmax.y Lmax.z Rmin.x Rmin.y
  v4f c;         // Rmin.z Rmax.x Rmax.y Rmax.z
  uint32_t left_ref, right_ref, pad0, pad1;
};
static_assert(sizeof(DNode) == 64, "DNode must be one 64-byte line");

// One leaf slot (position j of obj_indices), 48 B: all an intersection test reads.
//  kind 0 triangle : v = p0 p1 p2          kind 2 degenerate triangle (never hit,
//  kind 1 sphere   : v = centre, radius         include/geometry/triangle.h:86-92)
struct __attribute__((aligned(16))) DLeafPrim {
  v4f a, b;
  float c0;
  uint32_t prim;   // index into the reference's list_objects order
  uint32_t kind;
  uint32_t cls;    // material class of the primitive: 0 emitter, 1 Lambertian, 2 Principled, 3 other
};
static_assert(sizeof(DLeafPrim) == 48, "DLeafPrim must be 48 bytes");

// Per-triangle shading record, 64 B (positions again so hit_info needs no second indirection).
// `n` is normalize(cross(p1-p0, p2-p0)) — a function of the vertices only, evaluated once at
// upload with the float expression the reference evaluates per hit (src/geometry/triangle.cpp:25).
struct __attribute__((aligned(16))) DTriShade {
  float p[9];
  uint32_t mesh;
  uint32_t i0, i1, i2;   // global vertex ids (normals / uv lookup)
  float n[3];
};
static_assert(sizeof(DTriShade) == 64, "DTriShade must be 64 bytes");

// One emitter of GroupOfEmitters::list_of_emitters, 80 B: everything its sample() reads, in one fetch
// instead of the chase light -> primitive -> triangle / sphere -> mesh -> material (five dependent loads
// in front of every next-event estimate).
//  kind 0 background        kind 1 triangle without vertex normals: a b c = p0 p1 p2 n (the baked face normal)
//  kind 3 sphere: a = centre, radius      kind 2 triangle with vertex normals: sampled through the tables (index)
//  d = the material's emission (zero when it is no DiffuseLight), w = the triangle's area pdf
struct __attribute__((aligned(16))) DLight {
  v4f a, b, c, d;
  uint32_t kind, index, pad0, pad1;
};
static_assert(sizeof(DLight) == 80, "DLight must be 80 bytes");

struct DScene {
  // camera (TLCam members, reference src/tl_camera.cpp:6-23) — derived values computed on the host
  float cam_to_world[16];
  float p_size0, p_size1;
  float cone_spread;        // raycone_for_primary_ray(...) is a per-render constant
  float aperture_radius, focal_dist;
  int32_t res_x, res_y;

  // BVH
  float root_min[3], root_max[3];
  uint32_t num_nodes, max_depth;   // num_nodes = internal nodes
  uint32_t root_ref;               // packed reference of the root (a leaf for 1-node trees)
  gptr<DNode> nodes;
  gptr<DLeafPrim> leaf_prims;

  // primitives / shading data
  gptr<VimgPrim> prims;
  gptr<DTriShade> tri_shade;
  gptr<float> tri_area_pdf;   // 1 / (|cross(e2, e1)| / 2) per triangle (triangle.cpp:229-231,246)
  gptr<VimgMesh> meshes;
  gptr<float> normals;
  gptr<float> uvs;
  gptr<VimgSphere> spheres;
  gptr<VimgMaterial> materials;
  gptr<uint32_t> material_flags;   // MATF_* per material
  gptr<VimgTexture> textures;
  gptr<float> texels;
  gptr<VimgTextureRG> rg_textures;
  gptr<float> rg_texels;
  gptr<VimgLight> lights;
  gptr<DLight> dlights;       // the same list, baked
  uint32_t num_lights;
  gptr<float> cdf_pool;
  VimgBackground background;
  uint32_t background_emissive;
};

enum : uint32_t {
  MATF_NEEDS_UV = 1u,      // colour texture is not constant (checker / image) or RG / normal map
  MATF_NEEDS_FRAME = 2u,   // Principled: reads HitInfo::n_frame
};

struct RenderArgs {
  uint32_t integrator, samples, depth;
  uint32_t tile_rank, tile_world;
  uint32_t tiles_x, tiles_y;        // ceil(W/8), ceil(H/8)
  uint32_t num_local_tiles;
  uint32_t full_stats;
  uint32_t stack_entries;           // per-lane LDS stack depth (max_depth + 2)
  uint32_t lds_nodes;               // number of top-of-tree nodes staged into LDS
  uint32_t stack_lds;               // pool4, deep trees: stack entries kept in LDS (== stack_entries: all of them);
                                    // the LDS stack then has stack_lds + 1 rows (the last one takes the writes above)
  VIMG_GLOBAL uint32_t* stack_ovf;  // pool4: the entries beyond, [wave of the grid][entry - stack_lds][lane] (scene-owned scratch)
  VIMG_GLOBAL v4u* pool_cold;       // pooled kernel: cold slot records, [wave][slot][record] (scene-owned scratch)
  uint32_t lds_leaf;                // pooled kernel: number of leaf records copied to LDS (all or 0)
  VIMG_GLOBAL v4u* pool_state;      // pooled kernel: per work item {rng lo, rng hi, epoch + segments done, -}{acc.xyz, -}
  uint32_t pool_segments;           // pooled kernel: segments a pixel's samples are split into (>= 1)
  uint32_t pool_seg_len;            // pooled kernel: samples per segment
  uint32_t pool_epoch;              // pooled kernel: tag base of this launch (stale records of earlier launches never match)
  uint32_t pool_slots;              // pooled kernel: path slots per wave (0 = lane-bound kernel)
  uint32_t pool_refill;             // pooled kernel: finished rays that trigger a refill pass
  uint32_t pool_vbatch;             // pooled kernel: queued slots of one class that start a vertex batch
  uint32_t pool_boxmin;             // pooled kernel: leave the box loop when fewer lanes than this still descend (0 = never)
  uint32_t pool_starve;             // pooled kernel: idle walk lanes (with no ray queued) that force a partial vertex batch
  uint32_t pool_gbreak;             // pool4 group build: a wave leaves the walk for a full batch only with this many rays or fewer in its lanes
  uint32_t pool_classes;            // pooled kernel: vertex queues: 1 = one, 2 = Principled apart, 3 = + Lambertian apart
  uint32_t cu_walkers;              // CU scheduler: waves [0, cu_walkers) own traversal stacks and walk; the others only shade
  uint32_t cu_flex;                 // CU scheduler: bit 0 = a walking wave that holds no ray may run a vertex batch
  uint32_t cu_lowwater;             // CU scheduler: partial vertex batches only while fewer rays than this wait in the walk ring
  uint32_t cu_patience;             // CU scheduler: looks in vain after which a wave takes a partial batch of any size
  uint32_t cu_join;                 // CU scheduler: queued rays at which a walking wave that holds none starts to walk (fewer: after cu_patience looks)
  uint32_t cu_sleep;                // CU scheduler: s_sleep argument of a wave that found nothing to do
  uint32_t cu_watchdog;             // CU scheduler: a wave idle for longer than this (2^20 ticks of the 100 MHz clock) while slots are live gives up
  uint32_t cu_magic_v, cu_shift_v;  // CU scheduler: n / pool_slots == mulhi(n, magic) >> shift (n < 2^31)
  uint32_t cu_magic_w, cu_shift_w;  // ... n / (2 * pool_slots)
  int32_t single_x, single_y;       // trace_pixel mode when >= 0
};

struct DeviceStats {
  unsigned long long closest, shadow, internal, leaf, prim, sphere, nan_samples;
  unsigned long long trip_descend, trip_prim, iterations;   // wave-level loop trips (diagnostic)
  // -DVIMG_PROFILE builds only (make prof): s_memtime cycles of wave 0.. summed over waves, per
  // stage of render_pool_kernel, and lanes switched on per vertex batch
  unsigned long long prof[28];
  unsigned long long wait_cyc[5], wait_n[5];   // render_cu_kernel, statistics launches: cycles slots waited in the rings (vertex 0-3, walk), and how many
  unsigned long long px_done[3];    // render_cu_kernel, statistics launches: when pixels finished (10 ns ticks since their workgroup started): sum, count, latest
  unsigned long long px_hops[3];    // ... vertex-stage visits per pixel (whole segments only): sum, largest, and (finish time << 24 | hops) of the last pixel
  unsigned long long ray_cyc[2];    // ... cycles rays spent in a walking lane (pop to hand-over): sum, count
  unsigned long long pv_cyc[7];     // ... Principled batches: cycles in state loads, hit record + path logic, light sample, BSDF sample, evaluations, stores + hand-over; batches
  unsigned long long walk_cyc[4];   // render_cu_kernel, statistics launches: cycles of the walk sessions in refill + set-up, box loop, leaf rounds, hand-over
};
enum : int { PF_TOTAL = 0, PF_V_LOAD, PF_V_LIGHT, PF_V_SAMPLE, PF_V_EVAL, PF_V_FINISH, PF_V_STORE,
             PF_W_REFILL, PF_W_BOX, PF_W_LEAF, PF_W_RETIRE, PF_V_BATCHES, PF_V_LANES, PF_V_ATVERTEX,
             PF_W_ROUNDS, PF_CLS_CYC0, PF_CLS_CYC1, PF_CLS_CYC2, PF_CLS_CYC3, PF_CLS_LANES0, PF_CLS_LANES1,
             PF_CLS_LANES2, PF_CLS_LANES3, PF_DRAIN, PF_MAXWAVE, PF_COUNT };

}  // namespace vimg
