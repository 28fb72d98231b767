// vimg-amd: the C++ host program around the GPU hot path — the counterpart of the reference's
// main (src/main.cpp:38-377) with its flags: -f scene.json, -t threads (ignored: the render runs
// on the GPU), -c tonemapper 0-3 (clamp, AgX, Reinhard, ACES; default AgX as main.cpp:97-114),
// -d "x y" single-pixel trace, -b 0 binned / 1 sweep BVH (default 0 as main.cpp:183-187),
// -m factor heatmap mode (BVH traversal cost, main.cpp:62-65,98-100,250-256), plus
// -s spp override and -o output path.  Scene loading, the SAH BVH build and PNG writing happen
// here on the host (libvimg_host); the render and the post chain go through the C ABI of
// libvimg_hip.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "vimg_hip.h"
#include "vimg_host.h"

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  std::string scene_path, out_path = "v_img_amd.png";
  int tonemapper = 0, bvh_type = VIMG_BVH_BINNED, px = -1, py = -1;   // clamp, as src/main.cpp:46
  long spp_override = -1;
  float heatmap_max = -1.f;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "-f") scene_path = next();
    else if (a == "-t") next();
    else if (a == "-c") {
      // an out-of-range value is ignored and the default kept (src/main.cpp:107-112)
      const int v = std::atoi(next());
      if (v >= 0 && v < 4) tonemapper = v;
    }
    else if (a == "-b") bvh_type = std::atoi(next()) == 1 ? VIMG_BVH_SWEEP : VIMG_BVH_BINNED;
    else if (a == "-s") spp_override = std::atol(next());
    else if (a == "-m") heatmap_max = static_cast<float>(std::atof(next()));
    else if (a == "-o") out_path = next();
    else if (a == "-d") {
      if (std::sscanf(next(), "%d %d", &px, &py) != 2) {
        std::fprintf(stderr, "-d needs \"x y\"\n");
        return 2;
      }
    } else {
      std::fprintf(stderr, "usage: vimg-amd -f scene.json [-c 0..3] [-b 0|1] [-m factor] [-s spp] [-d \"x y\"] [-o out.png]\n");
      return 2;
    }
  }
  if (scene_path.empty()) {
    std::fprintf(stderr, "No input file given\n");
    return 2;
  }
  double t0 = now_s();
  VimgHostScene* hs = nullptr;
  if (vimg_host_scene_from_json_file(scene_path.c_str(), &hs) != 0) {
    std::fprintf(stderr, "scene loading failed: %s\n", vimg_host_last_error());
    return 1;
  }
  double t1 = now_s();
  if (vimg_host_build_bvh(hs, bvh_type) != 0) {
    std::fprintf(stderr, "BVH build failed: %s\n", vimg_host_last_error());
    return 1;
  }
  const VimgScene* view = vimg_host_scene_view(hs);
  double t2 = now_s();
  std::printf("Number of lights loaded %u\nNumber of Surfaces loaded %u\nBVH max depth %u\n",
              view->num_lights, view->num_prims, view->bvh.max_depth);
  std::printf("scene loading %.3f s, BVH construction (%s) %.3f s\n", t1 - t0,
              bvh_type == VIMG_BVH_SWEEP ? "sweep" : "binned", t2 - t1);

  VimgRenderParams params;
  vimg_host_default_params(hs, &params);
  if (spp_override > 0) params.samples = static_cast<uint32_t>(spp_override);
  // main forces 4 spp and the clamp tonemapper for the normal integrators (src/main.cpp:220-237)
  // ... and for the heatmap (src/main.cpp:250-254)
  const bool heatmap = heatmap_max >= 0.f && px < 0;
  if (heatmap || params.integrator == VIMG_INTEGRATOR_S_NORMAL ||
      params.integrator == VIMG_INTEGRATOR_G_NORMAL) {
    params.samples = 4;   // unconditionally, -s or not: the reference overwrites the sample count here
    tonemapper = 0;
  }
  const int W = view->camera.res_x, H = view->camera.res_y;
  std::printf("Image resolution %dx%d, samples per pixel %u, max ray depth %u\n", W, H,
              params.samples, params.depth);

  VimgDeviceScene* dev = nullptr;
  if (vimg_hip_init(0) != VIMG_OK || vimg_hip_scene_upload(view, &dev) != VIMG_OK) {
    std::fprintf(stderr, "GPU set-up failed: %s\n", vimg_hip_last_error());
    return 1;
  }
  if (px >= 0) {
    float rgb[3];
    if (vimg_hip_trace_pixel(dev, &params, px, py, rgb) != VIMG_OK) {
      std::fprintf(stderr, "trace_pixel failed: %s\n", vimg_hip_last_error());
      return 1;
    }
    std::printf("Value of pixel in linear space is (%.9g, %.9g, %.9g)\n", rgb[0], rgb[1], rgb[2]);
    return 0;
  }
  float* d_rgb = nullptr;
  unsigned char* d_rgb8 = nullptr;
  const size_t n = size_t(W) * H;
  if (hipMalloc(reinterpret_cast<void**>(&d_rgb), n * 3 * sizeof(float)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&d_rgb8), n * 3) != hipSuccess) {
    std::fprintf(stderr, "hipMalloc failed\n");
    return 1;
  }
  VimgRenderStats st;
  double t3 = now_s();
  if (heatmap) {
    std::printf("Creating Heatmap for ray intersection\nHeatmap factor set to %g\n",
                heatmap_max <= 0 ? 20.0 : double(heatmap_max));
    if (vimg_hip_render_heatmap(dev, &params, heatmap_max, d_rgb, nullptr) != VIMG_OK) {
      std::fprintf(stderr, "heatmap failed: %s\n", vimg_hip_last_error());
      return 1;
    }
    std::printf("image rendering %.3f s\n", now_s() - t3);
  } else {
    if (vimg_hip_render(dev, &params, d_rgb, nullptr, &st) != VIMG_OK) {
      std::fprintf(stderr, "render failed: %s\n", vimg_hip_last_error());
      return 1;
    }
    double t4 = now_s();
    const double rays = double(st.closest_rays + st.shadow_rays);
    std::printf("image rendering %.3f s: %.1f Mrays/s (%.4f rays per camera path), %llu NaN samples\n",
                t4 - t3, rays / (t4 - t3) / 1e6, rays / double(st.paths),
                static_cast<unsigned long long>(st.nan_samples));
  }
  if (vimg_hip_post_rgb8(d_rgb, W, H, tonemapper, d_rgb8, nullptr) != VIMG_OK) {
    std::fprintf(stderr, "post failed: %s\n", vimg_hip_last_error());
    return 1;
  }
  std::vector<uint8_t> rgb8(n * 3);
  if (hipMemcpy(rgb8.data(), d_rgb8, n * 3, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (vimg_host_write_png(out_path.c_str(), rgb8.data(), W, H) != 0) {
    std::fprintf(stderr, "PNG write failed: %s\n", vimg_host_last_error());
    return 1;
  }
  std::printf("output image written to %s\n", out_path.c_str());
  (void)hipFree(d_rgb);
  (void)hipFree(d_rgb8);
  vimg_hip_scene_free(dev);
  vimg_host_scene_free(hs);
  return 0;
}
