/*
 * vimg_scene.h — the flattened, pointer-free ("POD") scene that crosses the drop-in boundary.
 *
 * The reference (atom501/v-img) keeps its scene as an object graph with private members
 * (TLCam include/tl_camera.h:11-19, Sphere include/geometry/sphere.h:28-31, Principled
 * include/material/principled.h:25-40, GroupOfEmitters include/geometry/emitters.h:29-31 ...),
 * so nothing outside it can walk the graph.  The seam the hot path is entered through is
 *   scene_integrator(render_data, bvh, list_objects, lights, mis_integrator)
 *   (reference src/main.cpp:245-246, include/integrators.h:36-39).
 * This header is the C restatement of exactly the data that call consumes, as plain arrays:
 * every table below cites the reference type it flattens.  Host code (v-img_amd/host) fills it,
 * the HIP library (include/vimg_hip.h) uploads it, the test oracle (oracle/) reads the same
 * bytes.  All arrays are caller-owned host memory; nothing here contains a pointer into
 * reference objects, a torch type or a C++ type.
 */
#ifndef VIMG_SCENE_H
#define VIMG_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIMG_NO_UV 255u          /* MeshConsts::no_uv, reference include/geometry/mesh.h:10-12 */
#define VIMG_MAX_UV_SETS 4
#define VIMG_MAX_MIP_LEVELS 15   /* max_mipmap_level, reference src/image_texture.cpp:73 */

/* integrator_func, reference include/integrators.h:24 (heatmap is not on the path) */
enum { VIMG_INTEGRATOR_S_NORMAL = 0, VIMG_INTEGRATOR_G_NORMAL = 1, VIMG_INTEGRATOR_MATERIAL = 2,
       VIMG_INTEGRATOR_MIS = 3 };

/* Surface subclasses, reference include/geometry/{triangle,sphere}.h */
enum { VIMG_PRIM_TRIANGLE = 0, VIMG_PRIM_SPHERE = 1 };

/* Material subclasses, reference include/material/{lambertian,dielectric,diffuse_light,principled}.h */
enum { VIMG_MAT_LAMBERTIAN = 0, VIMG_MAT_DIELECTRIC = 1, VIMG_MAT_DIFFUSE_LIGHT = 2,
       VIMG_MAT_PRINCIPLED = 3 };

/* TextureRGB subclasses, reference include/texture/texture_RGB.h:45-149 */
enum { VIMG_TEX_CONST = 0, VIMG_TEX_CHECKER = 1, VIMG_TEX_IMAGE = 2 };

/* TextureWrappingMode, reference include/texture/texture_common.h:7 */
enum { VIMG_WRAP_CLAMP = 0, VIMG_WRAP_MIRROR = 1, VIMG_WRAP_REPEAT = 2 };

/* Emitter kinds in GroupOfEmitters order, reference include/geometry/emitters.h:9-26 */
enum { VIMG_LIGHT_PRIM = 0, VIMG_LIGHT_BACKGROUND = 1 };

/* Background subclasses, reference include/background.h:25-179 */
enum { VIMG_BG_CONST = 0, VIMG_BG_ENVMAP = 1 };

/* TLCam ctor arguments, reference src/tl_camera.cpp:6-23.  cam_to_world is glm column-major. */
typedef struct VimgCamera {
  float cam_to_world[16];
  float vfov_deg;
  int32_t res_x, res_y;
  float aperture_radius;
  float focal_dist;
} VimgCamera;

/* One entry of list_objects (std::vector<std::unique_ptr<Surface>>, reference src/main.cpp:48).
 * type TRIANGLE: index = global triangle id (row of tri_* arrays); SPHERE: index = sphere id. */
typedef struct VimgPrim {
  uint32_t type;
  uint32_t index;
} VimgPrim;

/* struct Mesh, reference include/geometry/mesh.h:14-57.  Vertex attributes of all meshes are
 * concatenated; a mesh owns vertices [first_vertex, first_vertex+num_vertices). */
typedef struct VimgMesh {
  uint32_t first_vertex;
  uint32_t num_vertices;
  uint32_t has_normals;                      /* normals.size() > 0 */
  uint32_t num_uv_sets;                      /* texcoords.size() */
  uint32_t uv_offset[VIMG_MAX_UV_SETS];      /* start (in float2 units) of set k in uvs[] */
  uint32_t color_tex_uv;                     /* VIMG_NO_UV when absent */
  uint32_t normal_tex_uv;
  uint32_t metallic_roughness_tex_uv;
  uint32_t material;                         /* index into materials[] */
} VimgMesh;

/* class Sphere, reference include/geometry/sphere.h:27-31 */
typedef struct VimgSphere {
  float center[3];
  float radius;
  uint32_t material;
} VimgSphere;

/* All Material subclasses in one record.  Field meaning per type:
 *  LAMBERTIAN   : tex                                  (src/material/lambertian.cpp)
 *  DIELECTRIC   : ior                                  (src/material/dielectric.cpp)
 *  DIFFUSE_LIGHT: emit                                 (include/material/diffuse_light.h)
 *  PRINCIPLED   : tex, mr_tex, metallic/roughness factor and the ten Disney scalars
 *                                                      (include/material/principled.h:25-40)
 * normal_map is Material::normal_map (include/material/material.h:27), -1 = nullptr. */
typedef struct VimgMaterial {
  uint32_t type;
  int32_t tex;          /* index into textures[], -1 = none */
  int32_t mr_tex;       /* index into rg_textures[], -1 = nullptr */
  int32_t normal_map;   /* index into textures[] (an IMAGE texture), -1 = nullptr */
  float emit[3];
  float ior;
  float metallic_factor, roughness_factor;   /* metallic_roughness_factor */
  float specular_transmission, subsurface, specular, specular_tint, anisotropic, sheen,
      sheen_tint, clearcoat, clearcoat_gloss, eta;
} VimgMaterial;

/* ConstColor / Checkerboard / ImageTexture, reference include/texture/texture_RGB.h:45-149.
 * IMAGE: mip level l has max(width>>l,1) x max(height>>l,1) texels (rgb float triples, row 0 =
 * top) starting at texels[level_offset[l]*3]. */
typedef struct VimgTexture {
  uint32_t type;
  float col_a[3];       /* ConstColor::albedo or Checkerboard::col_a */
  float col_b[3];
  uint32_t width, height;   /* checker cell counts or image size */
  uint32_t num_levels;
  uint32_t wrap_u, wrap_v;
  uint64_t level_offset[VIMG_MAX_MIP_LEVELS];   /* in texels */
} VimgTexture;

/* class TextureRG, reference include/texture/texture_RG.h:11-57 (rg float pairs) */
typedef struct VimgTextureRG {
  uint32_t width, height;
  uint32_t wrap_u, wrap_v;
  uint64_t offset;      /* in texels, into rg_texels[] */
} VimgTextureRG;

/* One entry of GroupOfEmitters::list_of_emitters */
typedef struct VimgLight {
  uint32_t type;        /* VIMG_LIGHT_PRIM / VIMG_LIGHT_BACKGROUND */
  uint32_t prim;        /* index into prims[] for VIMG_LIGHT_PRIM */
} VimgLight;

/* ConstBackground / EnvMap, reference include/background.h:25-179.  ENVMAP: env_tex is an IMAGE
 * texture; row_cdf has (H+1) floats, col_cdf has H*(W+1) floats (ArraySampling2D,
 * include/rng/sampling.h:159-223), both offsets are in floats into cdf_pool[]. */
typedef struct VimgBackground {
  uint32_t type;
  float col[3];
  int32_t env_tex;
  float world_to_env[16];
  float env_to_world[16];
  float radiance_scale;
  uint64_t row_cdf_offset;
  uint64_t col_cdf_offset;
} VimgBackground;

/* struct BVHNode, reference include/bvh.h:22-28 */
typedef struct VimgBVHNode {
  uint32_t first_index;
  uint32_t obj_count;
} VimgBVHNode;

/* class BVH public fields, reference include/bvh.h:53-57.  bb_mins_maxes has
 * (2*num_nodes+3) float triples laid out exactly like BB_mins_maxes: [0]=root min, [2]=root max,
 * children (c, c+1) of a node at [2c+2..2c+5] = {Lmin, Rmin, Lmax, Rmax}. */
typedef struct VimgBVH {
  uint32_t num_nodes;
  uint32_t max_depth;
  const VimgBVHNode* nodes;
  const float* bb_mins_maxes;
  const uint32_t* obj_indices;     /* num_prims entries */
} VimgBVH;

typedef struct VimgScene {
  VimgCamera camera;
  VimgBackground background;

  uint32_t num_prims;      const VimgPrim* prims;
  uint32_t num_tris;       const uint32_t* tri_indices;   /* 3 per tri, mesh-local vertex ids */
                           const uint32_t* tri_mesh;      /* mesh id per tri */
  uint32_t num_meshes;     const VimgMesh* meshes;
  uint32_t num_vertices;   const float* vertices;         /* xyz */
                           const float* normals;          /* xyz, same indexing; rows of meshes
                                                             without normals are unused */
  uint64_t num_uvs;        const float* uvs;              /* float2 pool */
  uint32_t num_spheres;    const VimgSphere* spheres;
  uint32_t num_materials;  const VimgMaterial* materials;
  uint32_t num_textures;   const VimgTexture* textures;
  uint64_t num_texels;     const float* texels;           /* rgb pool */
  uint32_t num_rg_textures; const VimgTextureRG* rg_textures;
  uint64_t num_rg_texels;  const float* rg_texels;        /* rg pool */
  uint32_t num_lights;     const VimgLight* lights;
  uint64_t num_cdf;        const float* cdf_pool;
  VimgBVH bvh;
} VimgScene;

/* integrator_data minus the object pointers, reference include/integrators.h:26-34, plus the
 * tile shard this call renders (tile_world = 1 renders everything).  The image is cut into the
 * reference's 8x8 tiles in its x-major work_list order (include/integrators.h:57-65); shard r of
 * n owns tiles t with t % n == r. */
typedef struct VimgRenderParams {
  uint32_t integrator;
  uint32_t samples;
  uint32_t depth;
  uint32_t tile_rank;
  uint32_t tile_world;
} VimgRenderParams;

/* Event counts of one render, defined on the reference's call sites: a "ray" is one
 * bvh.hit<optional<HitInfo>> (src/integrators/mis_integrator.cpp:34,121) or one bvh.occlude
 * (:64).  Used for Mrays/s and for the algorithmic-bytes figure. */
typedef struct VimgRenderStats {
  uint64_t paths;
  uint64_t closest_rays;
  uint64_t shadow_rays;
  uint64_t internal_visits;
  uint64_t leaf_visits;
  uint64_t prim_tests;       /* triangle + sphere intersection tests */
  uint64_t sphere_tests;     /* the sphere share of prim_tests */
  uint64_t nan_samples;
} VimgRenderStats;

#ifdef __cplusplus
}
#endif
#endif
