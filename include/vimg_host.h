/*
 * vimg_host.h — C ABI of libvimg_host.so: the host side the north star keeps on the CPU
 * (scene loading, SAH BVH build, flattening, tonemap + PNG).  It produces the VimgScene that
 * vimg_hip_scene_upload() consumes.  None of this is on the timed path.
 *
 * Reference counterparts: set_scene_from_json (src/scene_loading/json_scene.cpp:395-442),
 * setup_for_bvh (src/main.cpp:26-36), BVH::build_sweep_bvh (src/bvh/sweep_bvh.cpp:218-292),
 * BVH::build_bin_bvh (src/bvh/bin_bvh.cpp:194-235), ImageTexture mip build
 * (src/image_texture.cpp:60-130), ArraySampling2D (include/rng/sampling.h:159-197).
 */
#ifndef VIMG_HOST_H
#define VIMG_HOST_H

#include "vimg_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct VimgHostScene VimgHostScene;   /* owns every array a VimgScene view points to */

enum { VIMG_BVH_BINNED = 0, VIMG_BVH_SWEEP = 1 };   /* the reference's -b flag, src/main.cpp:76 */

/* ---- loading the reference's JSON scene format ---- */
int vimg_host_scene_from_json_file(const char* path, VimgHostScene** out);
int vimg_host_scene_from_json_text(const char* text, VimgHostScene** out);

/* ---- programmatic construction (for content the JSON format cannot express: meshes with
 *      normals / uv sets, image textures, normal maps, env maps, thin lens) ---- */
VimgHostScene* vimg_host_scene_new(void);
void vimg_host_scene_free(VimgHostScene* s);

void vimg_host_set_camera_lookat(VimgHostScene* s, const float from[3], const float at[3],
                                 const float up[3], float vfov_deg, int res_x, int res_y,
                                 float aperture_radius, float focal_dist);
void vimg_host_set_render_defaults(VimgHostScene* s, uint32_t integrator, uint32_t samples,
                                   uint32_t depth);

int vimg_host_add_texture_const(VimgHostScene* s, const float rgb[3]);
int vimg_host_add_texture_checker(VimgHostScene* s, uint32_t w, uint32_t h, const float a[3],
                                  const float b[3]);
/* level-0 image (w*h rgb floats, row 0 = top); the mip chain is built here with the reference's
 * 8-tap down-sampling filter. */
int vimg_host_add_texture_image(VimgHostScene* s, uint32_t w, uint32_t h, const float* rgb,
                                uint32_t wrap_u, uint32_t wrap_v);
int vimg_host_add_texture_rg(VimgHostScene* s, uint32_t w, uint32_t h, const float* rg,
                             uint32_t wrap_u, uint32_t wrap_v);

int vimg_host_add_material(VimgHostScene* s, const VimgMaterial* m);

/* uv_sets: n_uv_sets pointers to num_vertices float2 each (may be NULL when n_uv_sets == 0).
 * Registers the triangles as lights (in the reference's reverse order,
 * src/geometry/mesh_loading.cpp:96-103) when the material is emissive. */
int vimg_host_add_mesh(VimgHostScene* s, uint32_t num_vertices, const float* vertices,
                       const float* normals, uint32_t n_uv_sets, const float* const* uv_sets,
                       uint32_t num_tris, const uint32_t* indices, uint32_t material,
                       uint32_t color_tex_uv, uint32_t normal_tex_uv, uint32_t mr_tex_uv);
/* create_quad_mesh (src/geometry/mesh_loading.cpp:67-85) with a column-major 4x4 transform */
int vimg_host_add_quad(VimgHostScene* s, const float xform[16], uint32_t material);
int vimg_host_add_sphere(VimgHostScene* s, const float center[3], float radius,
                         uint32_t material);

void vimg_host_set_background_const(VimgHostScene* s, const float rgb[3], int add_to_lights);
/* env_tex must be an IMAGE texture; builds the 2-D sampling CDFs from its level 0. */
int vimg_host_set_background_envmap(VimgHostScene* s, int env_tex, const float world_to_env[16],
                                    const float env_to_world[16], float radiance_scale);

/* ---- precompute of image textures and env-map tables ----
 * add_texture_image builds the mip chain and set_background_envmap the sampling CDFs with the
 * host loops (OpenMP, as the reference).  A caller that has libvimg_hip loaded installs its GPU
 * builders here (vimg_hip_build_mip_chain / vimg_hip_build_env_cdfs have these signatures; both
 * return 0 on success - on failure the host loops run).  NULL restores the host loops. */
typedef int (*vimg_mip_builder_fn)(uint32_t w, uint32_t h, const float* level0_rgb,
                                   uint32_t wrap_u, uint32_t wrap_v, float* out_levels);
typedef int (*vimg_env_cdf_builder_fn)(const float* img_rgb, uint32_t w, uint32_t h,
                                       float* row_cdf, float* col_cdfs);
void vimg_host_set_precompute(vimg_mip_builder_fn mip, vimg_env_cdf_builder_fn cdf);
/* 8-bit image conversions of the reference's texture loaders (src/image_texture.cpp:257-275):
 * sRGB -> linear (table of the 256 inputs, and applied), RGB -> unit normal. */
void vimg_host_srgb8_lut(float lut[256]);
void vimg_host_srgb8_to_linear(const uint8_t* in, uint64_t n, float* out);
void vimg_host_rgb8_to_normal(const uint8_t* rgb8, uint64_t n_pixels, float scale, float* out_xyz);

/* ---- finalisation ---- */
/* Primitive AABBs/centres as setup_for_bvh, then the chosen builder (sweep: max 8 prims per
 * leaf as src/main.cpp:200; binned: 16 bins as src/main.cpp:41). */
int vimg_host_build_bvh(VimgHostScene* s, int bvh_type);

/* The same with a caller-supplied builder (libvimg_hip's vimg_hip_build_lbvh has this signature):
 * it receives the primitive bounds the host builders use and fills the reference's BVH arrays. */
typedef int (*vimg_bvh_builder_fn)(uint32_t n, const float* bounds6, uint32_t* num_nodes,
                                   uint32_t* max_depth, VimgBVHNode* nodes, float* bb,
                                   uint32_t* obj_indices);
int vimg_host_build_bvh_with(VimgHostScene* s, vimg_bvh_builder_fn builder);

/* View valid until the scene is modified or freed.  NULL before vimg_host_build_bvh. */
const VimgScene* vimg_host_scene_view(const VimgHostScene* s);
void vimg_host_default_params(const VimgHostScene* s, VimgRenderParams* out);

/* ---- post (reference src/main.cpp:304-372): tonemap 0 clamp,1 AgX,2 Reinhard,3 ACES; sRGB
 *      OETF; 8-bit quantise (NaN -> magenta); PNG ---- */
int vimg_host_tonemap_to_rgb8(const float* rgb, int w, int h, int tonemapper, uint8_t* out_rgb8);
int vimg_host_write_png(const char* path, const uint8_t* rgb8, int w, int h);

const char* vimg_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
