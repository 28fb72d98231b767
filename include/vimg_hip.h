/*
 * vimg_hip.h — C ABI of libvimg_hip.so, the MI355X (gfx950) implementation of v-img's hot path.
 *
 * The reference has no FFI; its narrowest seam is the template call
 *   std::vector<glm::vec3> scene_integrator(render_data, bvh, prims, lights, integrator)
 *   (reference include/integrators.h:36-153, called from src/main.cpp:219-248)
 * and its single-pixel twin trace_pixel (include/integrators.h:181-220, src/main.cpp:257-298).
 * The entry points below replace exactly those two calls; INTEGRATION.md shows the binding a
 * maintainer adds on the reference side.  Plain pointers and sizes only.
 *
 * Conventions: every function returns 0 on success and a negative VIMG_E_* code on failure;
 * vimg_hip_last_error() returns the message for the calling thread's last failure.  Nothing
 * throws across the ABI.  One device context per process (one process per GPU).
 */
#ifndef VIMG_HIP_H
#define VIMG_HIP_H

#include "vimg_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  VIMG_OK = 0,
  VIMG_E_INVALID = -1,     /* bad argument / inconsistent scene tables */
  VIMG_E_DEVICE = -2,      /* HIP runtime error (no device, allocation, launch) */
  VIMG_E_UNSUPPORTED = -3  /* feature outside what the path reproduces */
};

typedef struct VimgDeviceScene VimgDeviceScene;   /* opaque: device-resident scene */

/* Selects the GPU for this process (hipSetDevice) and creates the render stream. */
int vimg_hip_init(int device_ordinal);

/* Number of visible HIP devices, or a negative error code. */
int vimg_hip_device_count(void);

/* How a scene's frames are scheduled on the GPU.  Every field: VIMG_OPT_AUTO (-1) = the library's
 * policy (stated per field); the schedulers execute the same per-path arithmetic and give
 * the same bits.  The reference has no counterpart (its scheduler is the tile loop of
 * include/integrators.h:57-101); these are the knobs of OUR replacement of that loop, at the
 * boundary instead of in the environment.  (For tools/ only, VIMG_HIP_* environment variables
 * still override single fields at upload: scheduler VIMG_HIP_SCHED=cu|lane (pool|stage|pool4|pool4g in the
 * development build), the others as named in vimg_hip.hip:options_from_env.) */
#define VIMG_OPT_AUTO (-1)
enum {
  VIMG_SCHED_LANE = 1,   /* render_kernel: one path per lane, persistent waves */
  /* 2-5: the schedulers of rounds 1 and 2, reference implementations that only the development build of the
   * library contains (make dev, v-img_amd/lib/dev/libvimg_hip.so); the product library answers VIMG_E_UNSUPPORTED */
  VIMG_SCHED_POOL = 2,   /* render_pool_kernel: ~240 path slots per wave in LDS, walk + vertex stages in one wave */
  VIMG_SCHED_STAGE = 3,  /* render_stage_kernel: path state in HBM slots, stages coupled by global queues, 4 waves/SIMD */
  VIMG_SCHED_POOL4 = 4,  /* render_pool4_kernel: the pooled scheduler with the vertex stage as calls */
  VIMG_SCHED_POOL4G = 5, /* the same with ONE pool and one set of queues per workgroup (four waves share them under a lock in LDS) */
  VIMG_SCHED_CU = 6      /* render_cu_kernel: one pool per compute unit, walking and shading waves, lock-free rings in LDS, both rays of a vertex walked at once */
};
typedef struct VimgHipOptions {
  uint32_t struct_size;       /* sizeof(VimgHipOptions): lets the library accept older callers */
  int32_t scheduler;          /* AUTO (vimg_hip.hip:make_launch_cu): CU for every launch; LANE for frames wider than 65 535 pixels */
  int32_t waves_per_simd;     /* register budget: LANE / POOL 2 or 3 (AUTO: LANE 3 for scenes > 32 MiB else 2; POOL 2), POOL4 / POOL4G 3 or 4 (AUTO: 4 for full frames on trees that fit in LDS, else 3) */
  int32_t lds_budget_kb;      /* LDS per workgroup for BVH top + stacks.  AUTO: 40 (LANE), stacks + 4.5 (POOL / STAGE) */
  int32_t pool_slots;         /* CU: path slots per compute unit.  AUTO: what the CU's LDS holds, <= 1280 on trees in LDS, pixels / 2.7 on launches of 1 to 2.7 pools' worth of pixels, never more than pixels per CU + 8.  (POOL..: per wave, <= 256) */
  int32_t pool_segments;      /* POOL: segments a pixel's samples are cut into.  AUTO: ~56 / pool generations, <= 16 (POOL4G at four waves: ~176 / generations, <= 64) */
  int32_t pool_refill;        /* walk: finished rays of a wave that trigger hand-over and refill.  AUTO 16; CU on trees in global memory 2 */
  int32_t pool_vbatch;        /* POOL: queued slots of one class that start a vertex batch.  AUTO 64 */
  int32_t pool_classes;       /* POOL: vertex queues by material, 1..3.  AUTO 3; per-wave POOL4 on trees beyond LDS: 1 */
  int32_t pool_starve;        /* CU: smallest partial vertex batch a wave takes at once.  AUTO 16.  (POOL..: idle walk lanes that force a partial batch, AUTO 24 / 32) */
  int32_t pool_boxmin;        /* POOL / STAGE, deep trees: leave the box loop below this many descending lanes.  AUTO 16 */
  int32_t lds_leaf;           /* POOL / STAGE: 0 = never copy the leaf records to LDS.  AUTO: when they fit 4 KiB */
  int32_t stage_slots;        /* STAGE: path slots in flight.  AUTO: 2 x resident lanes, <= pixels of the launch */
  int32_t stage_seg_len;      /* STAGE: samples a pixel stays bound to a slot.  AUTO 4 */
  int32_t stage_wchunk;       /* STAGE: slot ids a walking wave stages in LDS, 128..256.  AUTO 128 */
  int32_t stage_walk_quota;   /* STAGE: rays a wave walks before it looks at the queues again.  AUTO 2048 */
  int32_t pool4_rays;         /* reserved (two rays per lane in the walk measured slower; 1 is what runs) */
  int32_t lds_stack;          /* POOL4, trees beyond LDS: entries of a lane's traversal stack kept in LDS, the rest in global memory.  AUTO 32 */
  int32_t pool_gbreak;        /* POOL4G: a wave leaves the walk for a full vertex batch only with this many rays or fewer in its lanes.  AUTO 32 */
  int32_t cu_waves;           /* CU: reserved; one 16-wave workgroup is a whole compute unit (12 waves at 168 registers measured slower) */
  int32_t cu_walkers;         /* CU: waves of the 16 that walk (the rest only shade).  AUTO: 9 on trees in LDS, 10 on trees in global memory, all 16 when every pixel of the launch owns a slot (tree in LDS) */
  int32_t cu_flex;            /* CU: bit 0: a walking wave that holds no ray may run a vertex batch; bit 4 (16): no split batches (the two BSDF evaluations of a vertex on the two halves of the wave when a batch has <= 32 slots); bit 5 (32): EARLY rays - a vertex stage queues its shadow ray right after the light sample and its path ray right after the BSDF sample and finishes (evaluations, stores) beside their walks; bits 1, 2: shading / walking at wave priority 1 (measurements).  AUTO 1, + 32 on launches of fewer than three pools' worth of pixels and on trees in global memory */
  int32_t cu_lowwater;        /* CU: partial vertex batches run only while fewer rays than this wait in the walk ring.  AUTO 64 */
  int32_t cu_patience;        /* CU: looks in vain after which a wave takes a partial batch of any size.  AUTO 4 */
  int32_t cu_join;            /* CU: queued rays at which a walking wave that holds no ray starts to walk (fewer: after cu_patience looks).  AUTO 1 */
  int32_t cu_sleep;           /* CU: s_sleep argument (64 cycles each) of a wave that found nothing to do.  AUTO 4 */
} VimgHipOptions;
/* Fills every field with VIMG_OPT_AUTO (and struct_size). */
void vimg_hip_options_default(VimgHipOptions* opts);

/* Validates the tables of `scene` (index ranges, BVH child ranges, stack bound), bakes them
 * into the device layout described in DESIGN.md and copies them to HBM.  The host arrays may
 * be freed afterwards.  Replaces nothing in the reference (its scene is already in RAM); it is
 * the "load once" half of the seam so that the timed render starts with inputs resident.
 * `opts` may be NULL (= all AUTO); vimg_hip_scene_upload(scene, out) is that case. */
int vimg_hip_scene_upload(const VimgScene* scene, VimgDeviceScene** out);
int vimg_hip_scene_upload_opts(const VimgScene* scene, const VimgHipOptions* opts, VimgDeviceScene** out);
int vimg_hip_scene_free(VimgDeviceScene* scene);

/* Number of float triples a shard's compact framebuffer holds
 * (= 64 * number of 8x8 tiles owned by tile_rank). */
int64_t vimg_hip_shard_pixels(const VimgDeviceScene* scene, const VimgRenderParams* params);

/* scene_integrator (reference include/integrators.h:36-153) with any of the four integrators
 * of integrator_func: s_normal, g_normal, material (src/integrators/mat_integrator.cpp), mis.
 *  d_out_rgb : DEVICE pointer.
 *      tile_world == 1: W*H float triples, linear radiance, index x + (H-1-y)*W, i.e. exactly
 *                       the reference's image_accumulated vector (include/integrators.h:113,137).
 *      tile_world  > 1: the shard's compact buffer, vimg_hip_shard_pixels() triples, tile-major
 *                       ([local_tile][ty*8+tx]); assemble with vimg_hip_assemble_shards().
 *  stream    : a hipStream_t cast to void*, or NULL for the library's own stream.
 *  stats     : optional HOST pointer, filled when the call returns.
 * The call enqueues the kernels and waits for them (the reference call is blocking too).
 * Limit of one launch: the rings of a compute unit count the rays it queues in 32 bits - about 2^31 per
 * launch, i.e. the whole 1800x800 frame of disney_spheres up to ~60 000 samples per pixel (the reference's
 * scenes ask for 512-2048); beyond it the launch ends with VIMG_E_DEVICE and says so. */
int vimg_hip_render(VimgDeviceScene* scene, const VimgRenderParams* params, void* d_out_rgb,
                    void* stream, VimgRenderStats* stats);

/* Same as vimg_hip_render but only enqueues (no host wait, no stats); used by bench.py to time
 * back-to-back launches with HIP events on `stream`.  A scene renders one frame at a time: its
 * work counter and the scheduler's scratch (path-slot records, per-pixel records) are owned by the
 * scene, so launches on the same scene must be ordered on one stream.  A kernel-side failure (the
 * scheduler's watchdog) is reported by the next blocking call on the scene or by vimg_hip_check, not by this one. */
int vimg_hip_render_async(VimgDeviceScene* scene, const VimgRenderParams* params, void* d_out_rgb,
                          void* stream);

/* The error word of the scene's launches since it was last read: VIMG_OK, or VIMG_E_DEVICE when a
 * kernel's watchdog gave a frame up (the frame is then incomplete).  Blocking calls read it
 * themselves; after vimg_hip_render_async the caller synchronises the stream and then asks here,
 * before it uses, gathers or times the frame.  Reading clears the word. */
int vimg_hip_check(VimgDeviceScene* scene);

/* Convenience: tile_world must be 1; renders into an internal device buffer and copies the
 * W*H*3 floats to out_rgb_host (what a reference maintainer would call from main.cpp). */
int vimg_hip_render_to_host(VimgDeviceScene* scene, const VimgRenderParams* params,
                            float* out_rgb_host, VimgRenderStats* stats);

/* trace_pixel (reference include/integrators.h:181-220): one pixel (x, y), all samples;
 * writes 3 floats to out_rgb_host. */
int vimg_hip_trace_pixel(VimgDeviceScene* scene, const VimgRenderParams* params, int x, int y,
                         float* out_rgb_host);

/* heatmap_img (reference src/integrators/heatmap.cpp:38-147, called from src/main.cpp:255): the
 * BVH traversal-cost picture, turbo(colour) of cost / factor (factor <= 0 -> 20).  Uses
 * params->samples and the tile shard; the output layout is vimg_hip_render's.  d_out_rgb: device
 * pointer.  Blocking. */
int vimg_hip_render_heatmap(VimgDeviceScene* scene, const VimgRenderParams* params, float factor,
                            void* d_out_rgb, void* stream);

/* De-interleaves `world` gathered compact shard buffers (concatenated in rank order, each
 * padded to `shard_stride_pixels` triples) into the reference image layout.  d_shards and
 * d_out_rgb are DEVICE pointers. */
int vimg_hip_assemble_shards(const VimgDeviceScene* scene, uint32_t world,
                             int64_t shard_stride_pixels, const void* d_shards, void* d_out_rgb,
                             void* stream);

/* Times `steps` back-to-back renders with hipEvents recorded on the launch stream.
 * ms_per_launch[i] (host, `steps` floats) = duration of launch i.  Used for roofline.achieved. */
int vimg_hip_time_renders(VimgDeviceScene* scene, const VimgRenderParams* params, void* d_out_rgb,
                          int steps, float* ms_per_launch);

/* Post chain of reference src/main.cpp:304-356 on the GPU: tonemapper 0 clamp (simple_clamp),
 * 1 AgX (src/tonemap/agx.cpp), 2 Reinhard on the image's largest luminance
 * (src/tonemap/reinhard.cpp), 3 ACES (src/tonemap/aces.cpp); then sRGB_gamma_correction
 * (include/color_utils.h:45-68) and the 8-bit quantisation with NaN -> magenta.
 * d_rgb: DEVICE, w*h float triples; d_rgb8: DEVICE, w*h byte triples.  Saves the 12 B/pixel
 * download when only the picture is wanted. */
int vimg_hip_post_rgb8(const void* d_rgb, int w, int h, int tonemapper, void* d_rgb8,
                       void* stream);

/* Bytes of HBM the uploaded scene occupies. */
int64_t vimg_hip_scene_bytes(const VimgDeviceScene* scene);

/* ---- the pre-step of the path on the GPU --------------------------------
 * Host buffers in and out: these replace the OpenMP loops of the reference's scene set-up
 * (src/image_texture.cpp:60-130,257-275; include/rng/sampling.h:113-135,168-197) while a scene
 * is assembled; their results are byte-identical to libvimg_host's.  vimg_hip_build_mip_chain and
 * vimg_hip_build_env_cdfs have the signatures vimg_host_set_precompute() takes. */

/* Texels of all levels of the reference's chain for a w x h image (level 0 included) and the
 * level count min(ceil(log2(min(w, h))), 15). */
uint64_t vimg_hip_mip_chain_texels(uint32_t w, uint32_t h, uint32_t* num_levels);
/* out_levels: 3 floats per texel, level 0 first, each level (max(w >> l, 1) x max(h >> l, 1))
 * behind the one before; wrap modes VIMG_WRAP_*. */
int vimg_hip_build_mip_chain(uint32_t w, uint32_t h, const float* level0_rgb, uint32_t wrap_u,
                             uint32_t wrap_v, float* out_levels);
/* Env-map importance tables from the w x h lat-long image: row_cdf[h + 1] (marginal over rows)
 * and col_cdfs[h][w + 1] (one conditional per row), as ArraySampling2D builds them. */
int vimg_hip_build_env_cdfs(const float* img_rgb, uint32_t w, uint32_t h, float* row_cdf,
                            float* col_cdfs);
/* out[i] = lut256[in[i]]: convert_sRGB_to_linear on 8-bit data with the caller's table
 * (vimg_host_srgb8_lut evaluates the reference's expression for the 256 inputs). */
int vimg_hip_lut8_to_float(const uint8_t* in, uint64_t n, const float* lut256, float* out);
/* convert_RGB_to_normal: normalize((rgb / 127.5 - 1) * (scale, scale, 1)) per pixel. */
int vimg_hip_rgb8_to_normal(const uint8_t* rgb8, uint64_t n_pixels, float scale, float* out_xyz);

/* ---- GPU BVH builders --------------------------------------------------
 * A linear BVH (Morton order, Karras' radix tree, bottom-up boxes; one primitive per leaf) in the
 * reference's layout (include/bvh.h:22-57).  Not the reference's SAH builders (those stay on the
 * host, vimg_host_build_bvh): for geometry that changes between frames.  The tree AND its layout
 * (breadth-first numbering, sibling-pair boxes, obj_indices, depth) are made by kernels; the host
 * copies the arrays in and out.  Host buffers:
 *   bounds6     : n x {min.xyz, max.xyz} of the primitives, in list_objects order
 *   nodes       : capacity 2n - 1;   bb : capacity (2 (2n - 1) + 3) float triples;
 *   obj_indices : n entries.
 * Has the signature vimg_host_build_bvh_with() takes.  The same input gives the same arrays. */
int vimg_hip_build_lbvh(uint32_t n, const float* bounds6, uint32_t* num_nodes, uint32_t* max_depth,
                        VimgBVHNode* nodes, float* bb, uint32_t* obj_indices);
/* The quality builder: PLOC (parallel locally-ordered clustering over the Morton order, search
 * radius 12), leaves ended by the reference builders' surface-area heuristic (up to 8 primitives,
 * include/bvh.h:17-20), and the top of the tree - the 16 384 subtrees of largest area - rebuilt
 * top-down by binned SAH (the quantity src/bvh/sweep_bvh.cpp:7-49 sweeps), all in kernels: 519 K
 * triangles in 4.5 ms of kernels + 2.7 ms of layout and download; the renderer runs within 3 % of its
 * rate on the host's sweep-SAH tree.  Same buffers, same layout, same hook as
 * vimg_hip_build_lbvh. */
int vimg_hip_build_ploc(uint32_t n, const float* bounds6, uint32_t* num_nodes, uint32_t* max_depth,
                        VimgBVHNode* nodes, float* bb, uint32_t* obj_indices);

/* Name of the render kernel a whole frame of this scene is launched with (textured or not,
 * register budget, scheduler) - what a rocprofv3 kernel trace will show - and the one a launch
 * with these parameters gets (the scheduler is chosen per launch: thin shards may differ). */
const char* vimg_hip_scene_kernel(const VimgDeviceScene* scene);
const char* vimg_hip_launch_kernel(const VimgDeviceScene* scene, const VimgRenderParams* params);

const char* vimg_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
