// Owner of the arrays behind a VimgScene view (host side, not on the timed path).
#pragma once
#include <string>
#include <vector>

#include "../../include/vimg_host.h"
#include "hmath.hpp"

struct PrimBounds {
  hm::V3 bmin, bmax;
};

struct HostBVH {
  std::vector<VimgBVHNode> nodes;
  std::vector<float> bb;            // (2*nodes+3) triples
  std::vector<uint32_t> obj_indices;
  uint32_t max_depth = 0;
};

struct VimgHostScene {
  VimgCamera camera{};
  VimgBackground background{};
  VimgRenderParams defaults{VIMG_INTEGRATOR_S_NORMAL, 30, 30, 0, 1};

  std::vector<VimgPrim> prims;
  std::vector<uint32_t> tri_indices, tri_mesh;
  std::vector<VimgMesh> meshes;
  std::vector<float> vertices, normals, uvs;
  std::vector<VimgSphere> spheres;
  std::vector<VimgMaterial> materials;
  std::vector<VimgTexture> textures;
  std::vector<float> texels;
  std::vector<VimgTextureRG> rg_textures;
  std::vector<float> rg_texels;
  std::vector<VimgLight> lights;
  std::vector<float> cdf_pool;
  HostBVH bvh;
  bool bvh_built = false;

  VimgScene view{};
  void refresh_view();
};

// bvh_build.cpp
void prim_bounds(const VimgHostScene& s, std::vector<PrimBounds>& bounds,
                 std::vector<hm::V3>& centers);
HostBVH build_sweep_bvh(const std::vector<PrimBounds>& bboxes, const std::vector<hm::V3>& centers,
                        uint32_t max_node_prims);
HostBVH build_bin_bvh(const std::vector<PrimBounds>& bboxes, const std::vector<hm::V3>& centers,
                      size_t num_bins);

// texture_build.cpp
bool build_mip_chain(uint32_t w, uint32_t h, const float* level0, uint32_t wrap_u, uint32_t wrap_v,
                     VimgTexture& tex, std::vector<float>& texel_pool);
bool build_env_cdfs(const float* level0, uint32_t w, uint32_t h, std::vector<float>& pool,
                    uint64_t& row_off, uint64_t& col_off);

void host_set_error(const std::string& msg);
