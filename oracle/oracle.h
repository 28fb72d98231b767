/*
 * oracle.h — TEST INFRASTRUCTURE.  CPU restatement of v-img's per-pixel path-tracing hot path.
 *
 * This library is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped render path (v-img_amd/csrc, behind
 * include/vimg_hip.h) neither includes, links nor calls anything in this directory.
 *
 * Every function in oracle.cpp cites the reference file:line it restates.  It reads the same
 * flattened VimgScene (include/vimg_scene.h) the GPU library uploads.
 *
 * PINNING: the reference cannot be built in this image (glm, fastgltf,
 * nlohmann 3.11 are not vendored; building against written stand-ins is not allowed), and it has
 * no tests.  The oracle is pinned by the fixtures the reference does hold — the analytic
 * sphere-light scenes with their converged *-ref.png images — plus the canonical PCG32 vector and
 * the values SURVEY.md Appendix B recorded from the reference's own code.
 */
#ifndef VIMG_ORACLE_H
#define VIMG_ORACLE_H

#include "vimg_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* scene_integrator (reference include/integrators.h:36-153).  out_rgb: W*H*3 floats in the
 * reference layout (index x + (H-1-y)*W); with tile_world > 1 only the shard's pixels are
 * written.  num_threads <= 0 -> hardware_concurrency, as the reference. Returns threads used. */
int oracle_render(const VimgScene* scene, const VimgRenderParams* params, int num_threads,
                  float* out_rgb, VimgRenderStats* stats);

/* heatmap_img (reference src/integrators/heatmap.cpp:38-147; BVH::hit<float>, include/bvh.h:83-225):
 * turbo colour map of the per-pixel BVH traversal cost.  out_counts (optional): W*H truncated
 * averages before the colour map.  Same layout and sharding as oracle_render. */
int oracle_heatmap(const VimgScene* scene, const VimgRenderParams* params, float factor,
                   int num_threads, float* out_rgb, float* out_counts);

/* trace_pixel (reference include/integrators.h:181-220) */
int oracle_trace_pixel(const VimgScene* scene, const VimgRenderParams* params, int x, int y,
                       float* out_rgb3);

/* ---- known-answer entry points (unit level) ---- */
void oracle_pcg32_srandom(uint64_t state_inc[2], uint64_t initstate, uint64_t initseq);
uint32_t oracle_pcg32_random(uint64_t state_inc[2]);
float oracle_rand_float(uint64_t state_inc[2]);
void oracle_random_x_y_r2(uint32_t n, float out_xy[2]);

/* probe kinds: per item `in` -> `out` floats (layouts in oracle.cpp, mirrored by the HIP probe) */
enum {
  ORACLE_PROBE_CAMERA_RAY = 1,   /* in 4: x y rand1 rand2            out 8: o3 d3 cone_width spread */
  ORACLE_PROBE_CLOSEST_HIT = 2,  /* in 6: o3 d3                      out 28: see oracle.cpp */
  ORACLE_PROBE_OCCLUDED = 3,     /* in 7: o3 d3 maxT                 out 1 */
  ORACLE_PROBE_BSDF_EVAL = 4,    /* in 12: o3 d3 wo3 cone2 regularize  out 5: hit f3 pdf */
  ORACLE_PROBE_BSDF_SAMPLE = 5,  /* in 8: o3 d3 seed regularize      out 7: hit has wo3 eta spec */
  ORACLE_PROBE_LIGHT_SAMPLE = 6, /* in 4: p3 seed                    out 10: Le3 wi3 pdf dist G */
  ORACLE_PROBE_BACKGROUND = 7    /* in 5: d3 cone2                   out 4: emit3 pdf */
};
int oracle_probe(const VimgScene* scene, int kind, int n, const float* in, float* out);

/* Post chain of the reference (tonemapper 0 clamp, 1 AgX, 2 Reinhard, 3 ACES; then sRGB OETF and
 * 8-bit quantisation with NaN -> magenta): reference src/main.cpp:304-356. */
int oracle_post_rgb8(const float* rgb, int w, int h, int tonemapper, uint8_t* out_rgb8);

/* 1 when built with -DORACLE_LIBM_FLOAT (the reference's own float libm calls) */
int oracle_uses_float_libm(void);

#ifdef __cplusplus
}
#endif
#endif
