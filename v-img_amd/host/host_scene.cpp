// libvimg_host.so: scene loading + flattening (host side kept on the CPU by the north star).
//   JSON format     : reference src/scene_loading/json_scene.cpp:67-442
//   quads / lights  : reference src/geometry/mesh_loading.cpp:67-104
#include "host_scene.hpp"

#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "json_mini.hpp"

using hm::M4;
using hm::V3;

static thread_local std::string g_err;
void host_set_error(const std::string& msg) { g_err = msg; }
extern "C" const char* vimg_host_last_error(void) { return g_err.c_str(); }

void VimgHostScene::refresh_view() {
  view = VimgScene{};
  view.camera = camera;
  view.background = background;
  view.num_prims = static_cast<uint32_t>(prims.size());
  view.prims = prims.data();
  view.num_tris = static_cast<uint32_t>(tri_mesh.size());
  view.tri_indices = tri_indices.data();
  view.tri_mesh = tri_mesh.data();
  view.num_meshes = static_cast<uint32_t>(meshes.size());
  view.meshes = meshes.data();
  view.num_vertices = static_cast<uint32_t>(vertices.size() / 3);
  view.vertices = vertices.data();
  view.normals = normals.data();
  view.num_uvs = uvs.size() / 2;
  view.uvs = uvs.data();
  view.num_spheres = static_cast<uint32_t>(spheres.size());
  view.spheres = spheres.data();
  view.num_materials = static_cast<uint32_t>(materials.size());
  view.materials = materials.data();
  view.num_textures = static_cast<uint32_t>(textures.size());
  view.textures = textures.data();
  view.num_texels = texels.size() / 3;
  view.texels = texels.data();
  view.num_rg_textures = static_cast<uint32_t>(rg_textures.size());
  view.rg_textures = rg_textures.data();
  view.num_rg_texels = rg_texels.size() / 2;
  view.rg_texels = rg_texels.data();
  view.num_lights = static_cast<uint32_t>(lights.size());
  view.lights = lights.data();
  view.num_cdf = cdf_pool.size();
  view.cdf_pool = cdf_pool.data();
  view.bvh.num_nodes = static_cast<uint32_t>(bvh.nodes.size());
  view.bvh.max_depth = bvh.max_depth;
  view.bvh.nodes = bvh.nodes.data();
  view.bvh.bb_mins_maxes = bvh.bb.data();
  view.bvh.obj_indices = bvh.obj_indices.data();
}

namespace {

void mat_to_array(const M4& m, float out[16]) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) out[c * 4 + r] = m[c][r];
}
M4 array_to_mat(const float in[16]) {
  M4 m;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) m[c][r] = in[c * 4 + r];
  return m;
}

bool material_is_emissive(const VimgHostScene& s, uint32_t mat) {
  return s.materials[mat].type == VIMG_MAT_DIFFUSE_LIGHT;
}

int add_mesh_impl(VimgHostScene& s, uint32_t nv, const float* verts, const float* norms,
                  uint32_t n_uv, const float* const* uv_sets, uint32_t nt, const uint32_t* idx,
                  uint32_t material, uint32_t color_uv, uint32_t normal_uv, uint32_t mr_uv) {
  if (material >= s.materials.size()) {
    host_set_error("add_mesh: material index out of range");
    return -1;
  }
  if (n_uv > VIMG_MAX_UV_SETS) {
    host_set_error("add_mesh: too many uv sets");
    return -1;
  }
  for (uint32_t i = 0; i < nt * 3; ++i)
    if (idx[i] >= nv) {
      host_set_error("add_mesh: vertex index out of range");
      return -1;
    }
  auto uv_ok = [&](uint32_t u) { return u == VIMG_NO_UV || u < n_uv; };
  if (!uv_ok(color_uv) || !uv_ok(normal_uv) || !uv_ok(mr_uv)) {
    host_set_error("add_mesh: uv set selector out of range");
    return -1;
  }
  VimgMesh m{};
  m.first_vertex = static_cast<uint32_t>(s.vertices.size() / 3);
  m.num_vertices = nv;
  m.has_normals = norms ? 1u : 0u;
  m.num_uv_sets = n_uv;
  m.color_tex_uv = color_uv;
  m.normal_tex_uv = normal_uv;
  m.metallic_roughness_tex_uv = mr_uv;
  m.material = material;
  s.vertices.insert(s.vertices.end(), verts, verts + static_cast<size_t>(nv) * 3);
  if (norms)
    s.normals.insert(s.normals.end(), norms, norms + static_cast<size_t>(nv) * 3);
  else
    s.normals.insert(s.normals.end(), static_cast<size_t>(nv) * 3, 0.f);
  for (uint32_t k = 0; k < n_uv; ++k) {
    m.uv_offset[k] = static_cast<uint32_t>(s.uvs.size() / 2);
    s.uvs.insert(s.uvs.end(), uv_sets[k], uv_sets[k] + static_cast<size_t>(nv) * 2);
  }
  const uint32_t mesh_id = static_cast<uint32_t>(s.meshes.size());
  s.meshes.push_back(m);

  // add_tri_list_to_scene: surfaces in order, emissive triangles registered last-to-first
  const uint32_t first_tri = static_cast<uint32_t>(s.tri_mesh.size());
  const uint32_t first_prim = static_cast<uint32_t>(s.prims.size());
  for (uint32_t t = 0; t < nt; ++t) {
    s.tri_indices.push_back(idx[t * 3]);
    s.tri_indices.push_back(idx[t * 3 + 1]);
    s.tri_indices.push_back(idx[t * 3 + 2]);
    s.tri_mesh.push_back(mesh_id);
    s.prims.push_back(VimgPrim{VIMG_PRIM_TRIANGLE, first_tri + t});
  }
  if (material_is_emissive(s, material))
    for (uint32_t t = nt; t > 0; --t)
      s.lights.push_back(VimgLight{VIMG_LIGHT_PRIM, first_prim + t - 1});
  s.bvh_built = false;
  return static_cast<int>(mesh_id);
}

int add_quad_impl(VimgHostScene& s, const M4& xform, uint32_t material) {
  V3 v[4] = {{-1, -1, 0}, {-1, 1, 0}, {1, 1, 0}, {1, -1, 0}};
  float verts[12];
  for (int i = 0; i < 4; ++i) {
    V3 p = hm::xform_point(xform, v[i]);
    verts[i * 3] = p.x;
    verts[i * 3 + 1] = p.y;
    verts[i * 3 + 2] = p.z;
  }
  const float uv[8] = {0, 0, 0, 1, 1, 1, 1, 0};
  const float* uv_sets[1] = {uv};
  const uint32_t idx[6] = {0, 2, 1, 2, 0, 3};
  // the legacy Mesh ctor: one uv set, color_tex_uv = 0 (include/geometry/mesh.h:33-41)
  return add_mesh_impl(s, 4, verts, nullptr, 1, uv_sets, 2, idx, material, 0, VIMG_NO_UV,
                       VIMG_NO_UV);
}

// load_from_obj, reference src/geometry/mesh_loading.cpp:21-65: positions only ("v" lines,
// transformed by the surface transform with the perspective divide) and the triangles of every
// face ("f" lines: vertex index before the first '/', 1-based or negative = relative to the end).
// Triangulation is tinyobjloader's (reference include/tiny_obj_loader.h:1488-1583, real_t = float):
// a quad is split along its SHORTER diagonal, measured on the untransformed file positions -
// [0,1,2],[0,2,3] only when |v2-v0|^2 < |v3-v1|^2, otherwise (ties included, i.e. every square and
// rectangle) [0,1,3],[1,2,3].  Polygons with more than four vertices go through an ear-clipping
// loop there; they are refused here with an explicit error rather than triangulated differently.
// Normals and texture coordinates of the file are ignored, as in the reference; the mesh goes
// through the legacy Mesh ctor with no uv set (include/geometry/mesh.h:33-41).
int add_obj_mesh(VimgHostScene& s, const std::string& path, const M4& xform, uint32_t material) {
  std::ifstream f(path);
  if (!f) {
    g_err = "Tinyobj failed to load the mesh: cannot open " + path;
    return -1;
  }
  std::vector<float> verts, raw;   // transformed positions; positions as the file has them
  std::vector<uint32_t> idx;
  std::vector<long> face;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ls(line);
    std::string tag;
    if (!(ls >> tag)) continue;
    if (tag == "v") {
      V3 p{0, 0, 0};
      ls >> p.x >> p.y >> p.z;
      V3 q = hm::xform_point(xform, p);
      verts.insert(verts.end(), {q.x, q.y, q.z});
      raw.insert(raw.end(), {p.x, p.y, p.z});
    } else if (tag == "f") {
      face.clear();
      std::string tok;
      const long nv = static_cast<long>(verts.size() / 3);
      while (ls >> tok) {
        long vi = std::strtol(tok.c_str(), nullptr, 10);   // stops at the first '/'
        if (vi < 0) vi = nv + vi; else vi -= 1;
        if (vi < 0 || vi >= nv) {
          g_err = "obj face refers to a vertex that is not defined yet: " + path;
          return -1;
        }
        face.push_back(vi);
      }
      auto tri = [&](int a, int b, int c) {
        idx.insert(idx.end(), {static_cast<uint32_t>(face[a]), static_cast<uint32_t>(face[b]),
                               static_cast<uint32_t>(face[c])});
      };
      if (face.size() < 3) continue;   // "Degenerated face": skipped by tinyobj too
      if (face.size() == 3) {
        tri(0, 1, 2);
      } else if (face.size() == 4) {
        const float* v0 = &raw[size_t(face[0]) * 3];
        const float* v1 = &raw[size_t(face[1]) * 3];
        const float* v2 = &raw[size_t(face[2]) * 3];
        const float* v3 = &raw[size_t(face[3]) * 3];
        const float e02x = v2[0] - v0[0], e02y = v2[1] - v0[1], e02z = v2[2] - v0[2];
        const float e13x = v3[0] - v1[0], e13y = v3[1] - v1[1], e13z = v3[2] - v1[2];
        const float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
        const float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
        if (sqr02 < sqr13) {
          tri(0, 1, 2);
          tri(0, 2, 3);
        } else {
          tri(0, 1, 3);
          tri(1, 2, 3);
        }
      } else {
        g_err = "obj face with more than 4 vertices (tinyobj ear-clips these; not reproduced): " + path;
        return -1;
      }
    }
  }
  if (idx.empty()) {
    g_err = "obj file has no faces: " + path;
    return -1;
  }
  return add_mesh_impl(s, static_cast<uint32_t>(verts.size() / 3), verts.data(), nullptr, 0, nullptr,
                       static_cast<uint32_t>(idx.size() / 3), idx.data(), material, VIMG_NO_UV,
                       VIMG_NO_UV, VIMG_NO_UV);
}

V3 json_vec3(const jmini::Value& v) {
  return V3{v.at(size_t{0}).as_float(), v.at(size_t{1}).as_float(), v.at(size_t{2}).as_float()};
}

// get_transform, reference src/scene_loading/json_scene.cpp:67-121 (each op pre-multiplies)
M4 json_transform(const jmini::Value& surf) {
  M4 xform = hm::identity();
  if (!surf.contains("transform")) return xform;
  for (const jmini::Value& e : surf.at("transform").arr) {
    if (e.contains("scale")) {
      const jmini::Value& sc = e.at("scale");
      V3 s3 = sc.is_array() ? json_vec3(sc) : V3{sc.as_float(), sc.as_float(), sc.as_float()};
      xform = hm::mul(hm::scale(s3), xform);
    } else if (e.contains("rotate")) {
      const jmini::Value& q = e.at("rotate");
      xform = hm::mul(hm::quat_to_mat4(q.at(size_t{0}).as_float(), q.at(size_t{1}).as_float(),
                                       q.at(size_t{2}).as_float(), q.at(size_t{3}).as_float()),
                      xform);
    } else if (e.contains("translate")) {
      xform = hm::mul(hm::translate(json_vec3(e.at("translate"))), xform);
    } else if (e.contains("x") || e.contains("y") || e.contains("z") || e.contains("o")) {
      // the reference reads at most ONE of x / y / z here (else-if chain) and never "o"
      V3 x{1, 0, 0}, y{0, 1, 0}, z{0, 0, 1}, o{0, 0, 0};
      if (e.contains("x"))
        x = json_vec3(e.at("x"));
      else if (e.contains("y"))
        y = json_vec3(e.at("y"));
      else if (e.contains("z"))
        z = json_vec3(e.at("z"));
      M4 m{{{x.x, x.y, x.z, 0.f}, {y.x, y.y, y.z, 0.f}, {z.x, z.y, z.z, 0.f},
            {o.x, o.y, o.z, 1.f}}};
      xform = hm::mul(m, xform);
    }
  }
  return xform;
}

int add_const_texture(VimgHostScene& s, V3 c) {
  VimgTexture t{};
  t.type = VIMG_TEX_CONST;
  t.col_a[0] = c.x;
  t.col_a[1] = c.y;
  t.col_a[2] = c.z;
  s.textures.push_back(t);
  return static_cast<int>(s.textures.size() - 1);
}

// json_to_texture, reference src/scene_loading/json_scene.cpp:233-271
int json_texture(VimgHostScene& s, const jmini::Value& mat) {
  if (!mat.contains("texture")) return add_const_texture(s, json_vec3(mat.at("albedo")));
  const jmini::Value& td = mat.at("texture");
  const std::string& type = td.at("type").as_string();
  if (type == "constant") return add_const_texture(s, json_vec3(td.at("albedo")));
  if (type == "checkered") {
    VimgTexture t{};
    t.type = VIMG_TEX_CHECKER;
    t.width = td.at("width").as_u32();
    t.height = td.at("height").as_u32();
    V3 a = json_vec3(td.at("col1")), b = json_vec3(td.at("col2"));
    t.col_a[0] = a.x, t.col_a[1] = a.y, t.col_a[2] = a.z;
    t.col_b[0] = b.x, t.col_b[1] = b.y, t.col_b[2] = b.z;
    s.textures.push_back(t);
    return static_cast<int>(s.textures.size() - 1);
  }
  throw std::runtime_error("unknown texture type " + type);
}

VimgMaterial blank_material(uint32_t type) {
  VimgMaterial m{};
  m.type = type;
  m.tex = -1;
  m.mr_tex = -1;
  m.normal_map = -1;
  return m;
}

void load_json(VimgHostScene& s, const std::string& text, const std::string& scene_dir) {
  jmini::Value root = jmini::Parser(text).parse();

  // ---- set_integrator_data, reference json_scene.cpp:155-231
  if (!root.contains("camera")) throw std::runtime_error("Camera settings not given");
  const jmini::Value& cam = root.at("camera");
  int res_x = 500, res_y = 500;
  if (cam.contains("resolution")) {
    res_x = static_cast<int>(cam.at("resolution").at(size_t{0}).as_float());
    res_y = static_cast<int>(cam.at("resolution").at(size_t{1}).as_float());
  }
  if (!cam.contains("transform")) throw std::runtime_error("Camera transform not given");
  const jmini::Value& ct = cam.at("transform");
  V3 from{0, 0, 0}, at{0, 0, 0}, up{0, 1, 0};
  if (ct.contains("from")) from = json_vec3(ct.at("from"));
  if (ct.contains("at")) at = json_vec3(ct.at("at"));
  if (ct.contains("up")) up = json_vec3(ct.at("up"));
  mat_to_array(hm::cam_to_world(from, at, up), s.camera.cam_to_world);
  s.camera.vfov_deg = cam.value_f("vfov", 40.0f);
  s.camera.res_x = res_x;
  s.camera.res_y = res_y;
  s.camera.aperture_radius = 0.f;  // JSON scenes: TLCam(cam_xform, res, vfov, 0.f, 1.f)
  s.camera.focal_dist = 1.f;

  uint32_t samples = 30, depth = 30;
  if (root.contains("sampler")) {
    samples = root.at("sampler").value_u32("samples", samples);
    depth = root.at("sampler").value_u32("depth", depth);
  }
  // "background" is parsed and ignored by the reference: always black, never a light
  s.background = VimgBackground{};
  s.background.type = VIMG_BG_CONST;
  s.background.env_tex = -1;

  uint32_t func = VIMG_INTEGRATOR_S_NORMAL;
  if (root.contains("integrator")) {
    const std::string& t = root.at("integrator").at("type").as_string();
    if (t == "s_normal") func = VIMG_INTEGRATOR_S_NORMAL;
    else if (t == "g_normal") func = VIMG_INTEGRATOR_G_NORMAL;
    else if (t == "material") func = VIMG_INTEGRATOR_MATERIAL;
    else if (t == "mis") func = VIMG_INTEGRATOR_MIS;
  }
  s.defaults = VimgRenderParams{func, samples, depth, 0, 1};

  // ---- set_list_of_materials, reference json_scene.cpp:273-331
  std::unordered_map<std::string, uint32_t> name_to_mat;
  if (!root.contains("materials")) throw std::runtime_error("Material loading failed");
  for (const jmini::Value& md : root.at("materials").arr) {
    const std::string& type = md.at("type").as_string();
    VimgMaterial m;
    if (type == "lambertian") {
      m = blank_material(VIMG_MAT_LAMBERTIAN);
      m.tex = json_texture(s, md);
    } else if (type == "diffuse_light") {
      m = blank_material(VIMG_MAT_DIFFUSE_LIGHT);
      V3 e{0.5f, 0.5f, 0.5f};
      if (md.contains("albedo")) e = json_vec3(md.at("albedo"));
      m.emit[0] = e.x, m.emit[1] = e.y, m.emit[2] = e.z;
    } else if (type == "dielectric") {
      m = blank_material(VIMG_MAT_DIELECTRIC);
      m.ior = md.contains("ior") ? md.at("ior").as_float() : 1.5f;
    } else if (type == "principled") {
      m = blank_material(VIMG_MAT_PRINCIPLED);
      m.tex = add_const_texture(s, json_vec3(md.at("base_color")));
      m.roughness_factor = md.value_f("roughness", 0.5f);
      m.anisotropic = md.value_f("anisotropic", 0.f);
      m.eta = md.value_f("eta", 1.5f);
      m.subsurface = md.value_f("subsurface", 0.f);
      m.metallic_factor = md.value_f("metallic", 0.f);
      m.specular_transmission = md.value_f("spec_trans", 0.f);
      m.specular = md.value_f("specular", 0.5f);
      m.specular_tint = md.value_f("spec_tint", 0.f);
      m.sheen = md.value_f("sheen", 0.f);
      m.sheen_tint = md.value_f("sheen_tint", 0.5f);
      m.clearcoat = md.value_f("clearcoat", 0.f);
      m.clearcoat_gloss = md.value_f("clearcoat_gloss", 1.f);
    } else {
      throw std::runtime_error("Unknown material " + type);
    }
    s.materials.push_back(m);
    name_to_mat[md.at("name").as_string()] = static_cast<uint32_t>(s.materials.size() - 1);
  }

  // ---- set_list_of_objects, reference json_scene.cpp:333-393
  if (!root.contains("surfaces")) throw std::runtime_error("Json file does not contain surfaces");
  for (const jmini::Value& sd : root.at("surfaces").arr) {
    M4 xform = json_transform(sd);
    auto it = name_to_mat.find(sd.at("mat_name").as_string());
    if (it == name_to_mat.end()) throw std::runtime_error("unknown mat_name");
    const uint32_t mat = it->second;
    const std::string& type = sd.at("type").as_string();
    if (type == "quad") {
      if (add_quad_impl(s, xform, mat) < 0) throw std::runtime_error(g_err);
    } else if (type == "sphere") {
      V3 c = json_vec3(sd.at("center"));
      VimgSphere sp{{c.x, c.y, c.z}, sd.value_f("radius", 1.0f), mat};
      s.spheres.push_back(sp);
      s.prims.push_back(
          VimgPrim{VIMG_PRIM_SPHERE, static_cast<uint32_t>(s.spheres.size() - 1)});
      if (material_is_emissive(s, mat))
        s.lights.push_back(
            VimgLight{VIMG_LIGHT_PRIM, static_cast<uint32_t>(s.prims.size() - 1)});
    } else if (type == "mesh") {
      // json_scene.cpp:366-385: the .obj path is relative to the scene file's directory
      const std::string rel = sd.at("filename").as_string();
      const std::string path = (!rel.empty() && rel[0] == '/') ? rel : scene_dir + rel;
      if (add_obj_mesh(s, path, xform, mat) < 0) throw std::runtime_error(g_err);
    } else {
      throw std::runtime_error("Unknown surface " + type);
    }
  }
}

int from_text(const std::string& text, VimgHostScene** out, const std::string& scene_dir = "") {
  if (!out) {
    host_set_error("null output pointer");
    return -1;
  }
  auto* s = new VimgHostScene();
  try {
    load_json(*s, text, scene_dir);
  } catch (const std::exception& e) {
    host_set_error(e.what());
    delete s;
    return -1;
  }
  *out = s;
  return 0;
}

}  // namespace

extern "C" {

int vimg_host_scene_from_json_file(const char* path, VimgHostScene** out) {
  std::ifstream f(path, std::ios::in | std::ios::binary);
  if (!f) {
    host_set_error(std::string("Json scene file does not exists: ") + (path ? path : "(null)"));
    return -1;
  }
  std::stringstream ss;
  ss << f.rdbuf();
  std::string dir(path);
  const size_t slash = dir.find_last_of('/');
  dir = (slash == std::string::npos) ? std::string() : dir.substr(0, slash + 1);
  return from_text(ss.str(), out, dir);
}

int vimg_host_scene_from_json_text(const char* text, VimgHostScene** out) {
  if (!text) {
    host_set_error("null text");
    return -1;
  }
  return from_text(text, out);
}

VimgHostScene* vimg_host_scene_new(void) {
  auto* s = new VimgHostScene();
  s->background.type = VIMG_BG_CONST;
  s->background.env_tex = -1;
  mat_to_array(hm::identity(), s->camera.cam_to_world);
  s->camera.vfov_deg = 40.f;
  s->camera.res_x = s->camera.res_y = 500;
  s->camera.focal_dist = 1.f;
  return s;
}

void vimg_host_scene_free(VimgHostScene* s) { delete s; }

void vimg_host_set_camera_lookat(VimgHostScene* s, const float from[3], const float at[3],
                                 const float up[3], float vfov_deg, int res_x, int res_y,
                                 float aperture_radius, float focal_dist) {
  mat_to_array(hm::cam_to_world(V3{from[0], from[1], from[2]}, V3{at[0], at[1], at[2]},
                                V3{up[0], up[1], up[2]}),
               s->camera.cam_to_world);
  s->camera.vfov_deg = vfov_deg;
  s->camera.res_x = res_x;
  s->camera.res_y = res_y;
  s->camera.aperture_radius = aperture_radius;
  s->camera.focal_dist = focal_dist;
}

void vimg_host_set_render_defaults(VimgHostScene* s, uint32_t integrator, uint32_t samples,
                                   uint32_t depth) {
  s->defaults = VimgRenderParams{integrator, samples, depth, 0, 1};
}

int vimg_host_add_texture_const(VimgHostScene* s, const float rgb[3]) {
  return add_const_texture(*s, V3{rgb[0], rgb[1], rgb[2]});
}

int vimg_host_add_texture_checker(VimgHostScene* s, uint32_t w, uint32_t h, const float a[3],
                                  const float b[3]) {
  VimgTexture t{};
  t.type = VIMG_TEX_CHECKER;
  t.width = w;
  t.height = h;
  std::memcpy(t.col_a, a, 12);
  std::memcpy(t.col_b, b, 12);
  s->textures.push_back(t);
  return static_cast<int>(s->textures.size() - 1);
}

int vimg_host_add_texture_image(VimgHostScene* s, uint32_t w, uint32_t h, const float* rgb,
                                uint32_t wrap_u, uint32_t wrap_v) {
  if (w == 0 || h == 0 || !rgb || wrap_u > 2 || wrap_v > 2) {
    host_set_error("add_texture_image: bad arguments");
    return -1;
  }
  VimgTexture t{};
  if (!build_mip_chain(w, h, rgb, wrap_u, wrap_v, t, s->texels)) return -1;
  s->textures.push_back(t);
  return static_cast<int>(s->textures.size() - 1);
}

int vimg_host_add_texture_rg(VimgHostScene* s, uint32_t w, uint32_t h, const float* rg,
                             uint32_t wrap_u, uint32_t wrap_v) {
  if (w == 0 || h == 0 || !rg || wrap_u > 2 || wrap_v > 2) {
    host_set_error("add_texture_rg: bad arguments");
    return -1;
  }
  VimgTextureRG t{w, h, wrap_u, wrap_v, s->rg_texels.size() / 2};
  s->rg_texels.insert(s->rg_texels.end(), rg, rg + static_cast<size_t>(w) * h * 2);
  s->rg_textures.push_back(t);
  return static_cast<int>(s->rg_textures.size() - 1);
}

int vimg_host_add_material(VimgHostScene* s, const VimgMaterial* m) {
  if (!m || m->type > VIMG_MAT_PRINCIPLED) {
    host_set_error("add_material: bad material");
    return -1;
  }
  auto tex_ok = [&](int32_t t) { return t >= -1 && t < static_cast<int32_t>(s->textures.size()); };
  if (!tex_ok(m->tex) || !tex_ok(m->normal_map) ||
      m->mr_tex < -1 || m->mr_tex >= static_cast<int32_t>(s->rg_textures.size())) {
    host_set_error("add_material: texture index out of range");
    return -1;
  }
  if ((m->type == VIMG_MAT_LAMBERTIAN || m->type == VIMG_MAT_PRINCIPLED) && m->tex < 0) {
    host_set_error("add_material: lambertian/principled need a colour texture");
    return -1;
  }
  if (m->normal_map >= 0 && s->textures[m->normal_map].type != VIMG_TEX_IMAGE) {
    host_set_error("add_material: normal map must be an image texture");
    return -1;
  }
  s->materials.push_back(*m);
  return static_cast<int>(s->materials.size() - 1);
}

int vimg_host_add_mesh(VimgHostScene* s, uint32_t num_vertices, const float* vertices,
                       const float* normals, uint32_t n_uv_sets, const float* const* uv_sets,
                       uint32_t num_tris, const uint32_t* indices, uint32_t material,
                       uint32_t color_tex_uv, uint32_t normal_tex_uv, uint32_t mr_tex_uv) {
  return add_mesh_impl(*s, num_vertices, vertices, normals, n_uv_sets, uv_sets, num_tris, indices,
                       material, color_tex_uv, normal_tex_uv, mr_tex_uv);
}

int vimg_host_add_quad(VimgHostScene* s, const float xform[16], uint32_t material) {
  if (material >= s->materials.size()) {
    host_set_error("add_quad: material index out of range");
    return -1;
  }
  return add_quad_impl(*s, array_to_mat(xform), material);
}

int vimg_host_add_sphere(VimgHostScene* s, const float center[3], float radius,
                         uint32_t material) {
  if (material >= s->materials.size()) {
    host_set_error("add_sphere: material index out of range");
    return -1;
  }
  s->spheres.push_back(VimgSphere{{center[0], center[1], center[2]}, radius, material});
  s->prims.push_back(VimgPrim{VIMG_PRIM_SPHERE, static_cast<uint32_t>(s->spheres.size() - 1)});
  if (material_is_emissive(*s, material))
    s->lights.push_back(VimgLight{VIMG_LIGHT_PRIM, static_cast<uint32_t>(s->prims.size() - 1)});
  s->bvh_built = false;
  return static_cast<int>(s->spheres.size() - 1);
}

void vimg_host_set_background_const(VimgHostScene* s, const float rgb[3], int add_to_lights) {
  s->background = VimgBackground{};
  s->background.type = VIMG_BG_CONST;
  s->background.env_tex = -1;
  std::memcpy(s->background.col, rgb, 12);
  if (add_to_lights) s->lights.push_back(VimgLight{VIMG_LIGHT_BACKGROUND, 0});
}

int vimg_host_set_background_envmap(VimgHostScene* s, int env_tex, const float world_to_env[16],
                                    const float env_to_world[16], float radiance_scale) {
  if (env_tex < 0 || env_tex >= static_cast<int>(s->textures.size()) ||
      s->textures[env_tex].type != VIMG_TEX_IMAGE) {
    host_set_error("set_background_envmap: env_tex must be an image texture");
    return -1;
  }
  VimgBackground bg{};
  bg.type = VIMG_BG_ENVMAP;
  bg.env_tex = env_tex;
  std::memcpy(bg.world_to_env, world_to_env, 64);
  std::memcpy(bg.env_to_world, env_to_world, 64);
  bg.radiance_scale = radiance_scale;
  const VimgTexture& t = s->textures[env_tex];
  if (!build_env_cdfs(s->texels.data() + t.level_offset[0] * 3, t.width, t.height, s->cdf_pool,
                      bg.row_cdf_offset, bg.col_cdf_offset))
    return -1;
  s->background = bg;
  s->lights.push_back(VimgLight{VIMG_LIGHT_BACKGROUND, 0});
  return 0;
}

int vimg_host_build_bvh(VimgHostScene* s, int bvh_type) {
  if (s->prims.empty()) {
    host_set_error("build_bvh: scene has no surfaces");
    return -1;
  }
  std::vector<PrimBounds> bounds;
  std::vector<V3> centers;
  prim_bounds(*s, bounds, centers);
  if (bvh_type == VIMG_BVH_SWEEP)
    s->bvh = build_sweep_bvh(bounds, centers, 8);   // src/main.cpp:200
  else
    s->bvh = build_bin_bvh(bounds, centers, 16);    // NUM_BINS, src/main.cpp:41
  s->bvh_built = true;
  s->refresh_view();
  return 0;
}

int vimg_host_build_bvh_with(VimgHostScene* s, vimg_bvh_builder_fn builder) {
  if (s->prims.empty() || !builder) {
    host_set_error("build_bvh_with: scene has no surfaces or no builder given");
    return -1;
  }
  std::vector<PrimBounds> bounds;
  std::vector<V3> centers;
  prim_bounds(*s, bounds, centers);
  const uint32_t n = static_cast<uint32_t>(bounds.size());
  std::vector<float> b6(size_t(n) * 6);
  for (uint32_t i = 0; i < n; ++i) {
    b6[i * 6 + 0] = bounds[i].bmin.x, b6[i * 6 + 1] = bounds[i].bmin.y, b6[i * 6 + 2] = bounds[i].bmin.z;
    b6[i * 6 + 3] = bounds[i].bmax.x, b6[i * 6 + 4] = bounds[i].bmax.y, b6[i * 6 + 5] = bounds[i].bmax.z;
  }
  HostBVH bvh;
  bvh.nodes.resize(size_t(n) * 2 - 1);
  bvh.bb.resize((bvh.nodes.size() * 2 + 3) * 3);
  bvh.obj_indices.resize(n);
  uint32_t num_nodes = 0, depth = 0;
  if (builder(n, b6.data(), &num_nodes, &depth, bvh.nodes.data(), bvh.bb.data(), bvh.obj_indices.data()) != 0 ||
      num_nodes == 0 || num_nodes > bvh.nodes.size()) {
    host_set_error("build_bvh_with: the builder failed");
    return -1;
  }
  bvh.nodes.resize(num_nodes);
  bvh.bb.resize((size_t(num_nodes) * 2 + 3) * 3);
  bvh.max_depth = depth;
  s->bvh = std::move(bvh);
  s->bvh_built = true;
  s->refresh_view();
  return 0;
}

const VimgScene* vimg_host_scene_view(const VimgHostScene* s) {
  if (!s || !s->bvh_built) {
    host_set_error("scene_view: call vimg_host_build_bvh first");
    return nullptr;
  }
  return &s->view;
}

void vimg_host_default_params(const VimgHostScene* s, VimgRenderParams* out) { *out = s->defaults; }

}  // extern "C"
