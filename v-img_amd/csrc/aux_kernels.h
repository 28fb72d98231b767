// Kernels beside the render path that only the ABI translation unit launches: shard assembly and
// the unit-level probes of the parity tests.
#pragma once
#include "render_kernels.h"

namespace vimg {

// ================================================================================ shard assembly
// De-interleave gathered compact shard buffers into the reference image layout.
__global__ void assemble_kernel(const float* __restrict__ shards, float* __restrict__ out,
                                uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t tiles_y,
                                uint32_t world, long long shard_stride_pixels) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t tile = gid >> 6, within = gid & 63u;
  if (tile >= tiles_x * tiles_y) return;
  const uint32_t tx = tile / tiles_y, ty = tile - tx * tiles_y;
  const uint32_t px = tx * 8 + (within & 7u), py = ty * 8 + (within >> 3);
  if (px >= W || py >= H) return;
  const uint32_t rank = tile % world, local_tile = tile / world;
  const float* src = shards + (size_t(rank) * shard_stride_pixels + size_t(local_tile) * 64 + within) * 3;
  float* dst = out + (size_t(px) + size_t(H - 1 - py) * W) * 3;
  dst[0] = src[0];
  dst[1] = src[1];
  dst[2] = src[2];
}

// ================================================================================ probe kernel
// Unit-level entry points used by the parity tests; layouts mirror the checker's probe API.
enum {
  PROBE_CAMERA_RAY = 1, PROBE_CLOSEST_HIT = 2, PROBE_OCCLUDED = 3, PROBE_BSDF_EVAL = 4,
  PROBE_BSDF_SAMPLE = 5, PROBE_LIGHT_SAMPLE = 6, PROBE_BACKGROUND = 7, PROBE_SINCOS = 8
};
template <bool TEX>
__global__ void __launch_bounds__(256)
probe_kernel(const DScene g, const RenderArgs A, int kind, int n, const float* __restrict__ in,
             float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Lds L = stage_lds(g, A, (VIMG_LDS unsigned char*)lds_raw);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Counters cnt{0, 0, 0, 0, 0, 0, 0, 0};
  auto trace = [&](const float* p, Hit& h, TravRay& tr) {
    tr = TravRay{f3{p[0], p[1], p[2]}, f3{p[3], p[4], p[5]}, 0.0001f, VIMG_INF};
    HitRec rec;
    bool hit = traverse<false>(g, L, tr, rec, cnt, false);
    if (hit) {
      // the probe wants every field: request the full record regardless of material flags
      make_hit_info<true>(g, rec, tr, h);
    }
    return hit;
  };
  switch (kind) {
    case PROBE_CAMERA_RAY: {
      const float* p = in + 4 * i;
      float* o = out + 8 * i;
      f3 ro, rd;
      generate_ray(g, p[0], p[1], p[2], p[3], ro, rd);
      o[0] = ro.x, o[1] = ro.y, o[2] = ro.z, o[3] = rd.x, o[4] = rd.y, o[5] = rd.z;
      o[6] = 0.f, o[7] = g.cone_spread;
      break;
    }
    case PROBE_CLOSEST_HIT: {
      float* o = out + 28 * i;
      for (int k = 0; k < 28; ++k) o[k] = 0.f;
      Hit h;
      TravRay tr;
      if (trace(in + 6 * i, h, tr)) {
        o[0] = 1.f, o[1] = tr.max_t, o[2] = static_cast<float>(h.prim), o[3] = static_cast<float>(h.mat);
        o[4] = h.p.x, o[5] = h.p.y, o[6] = h.p.z, o[7] = h.ns.x, o[8] = h.ns.y, o[9] = h.ns.z;
        o[10] = h.ng.x, o[11] = h.ng.y, o[12] = h.ng.z, o[13] = h.uv.x, o[14] = h.uv.y;
        o[15] = h.mr_uv.x, o[16] = h.mr_uv.y, o[17] = h.tu.x, o[18] = h.tu.y, o[19] = h.tu.z;
        o[20] = h.tv.x, o[21] = h.tv.y, o[22] = h.tv.z, o[23] = h.prim_area, o[24] = h.tex_area;
        o[25] = h.curvature;
      }
      break;
    }
    case PROBE_OCCLUDED: {
      const float* p = in + 7 * i;
      TravRay tr{f3{p[0], p[1], p[2]}, f3{p[3], p[4], p[5]}, 0.0001f, p[6]};
      HitRec rec;
      out[i] = traverse<true>(g, L, tr, rec, cnt, false) ? 1.f : 0.f;
      break;
    }
    case PROBE_BSDF_EVAL: {
      const float* p = in + 12 * i;
      float* o = out + 5 * i;
      for (int k = 0; k < 5; ++k) o[k] = 0.f;
      Hit h;
      TravRay tr;
      if (trace(p, h, tr)) {
        f3 f;
        float pdf;
        eval_pdf_pair<TEX>(g, h, tr.d, f3{p[6], p[7], p[8]}, RayCone{p[9], p[10]}, p[11] != 0.f, f,
                           pdf);
        o[0] = 1.f, o[1] = f.x, o[2] = f.y, o[3] = f.z, o[4] = pdf;
      }
      break;
    }
    case PROBE_BSDF_SAMPLE: {
      const float* p = in + 8 * i;
      float* o = out + 7 * i;
      for (int k = 0; k < 7; ++k) o[k] = 0.f;
      Hit h;
      TravRay tr;
      if (trace(p, h, tr)) {
        Rng rng;
        pcg_seed(rng, static_cast<uint64_t>(p[6]));
        Scatter sc = sample_mat<TEX>(g, h, tr.d, rng, p[7] != 0.f);
        o[0] = 1.f;
        o[1] = sc.valid ? 1.f : 0.f;
        if (sc.valid) {
          o[2] = sc.wo.x, o[3] = sc.wo.y, o[4] = sc.wo.z, o[5] = sc.eta;
          o[6] = sc.is_specular ? 1.f : 0.f;
        }
      }
      break;
    }
    case PROBE_LIGHT_SAMPLE: {
      const float* p = in + 4 * i;
      float* o = out + 10 * i;
      for (int k = 0; k < 10; ++k) o[k] = 0.f;
      if (g.num_lights == 0) break;
      Rng rng;
      pcg_seed(rng, static_cast<uint64_t>(p[3]));
      f3 le;
      EmitterInfo li;
      lights_sample<TEX>(g, f3{p[0], p[1], p[2]}, rng, le, li);
      o[0] = le.x, o[1] = le.y, o[2] = le.z, o[3] = li.wi.x, o[4] = li.wi.y, o[5] = li.wi.z;
      o[6] = li.pdf, o[7] = li.dist, o[8] = li.G;
      break;
    }
    case PROBE_BACKGROUND: {
      const float* p = in + 5 * i;
      float* o = out + 4 * i;
      f3 d{p[0], p[1], p[2]};
      f3 e = background_emit<TEX>(g, d, RayCone{p[3], p[4]});
      o[0] = e.x, o[1] = e.y, o[2] = e.z, o[3] = background_pdf<TEX>(g, d);
      break;
    }
    case PROBE_SINCOS: {
      // D_sincos (device_math.h) against the two calls it replaces: the floats the renderer uses, and
      // whether the doubles behind them are the same bits
      const float x = in[i];
      float* o = out + 5 * i;
      const double c0 = ::cos(static_cast<double>(x)), s0 = ::sin(static_cast<double>(x));
      double s1, c1;
      D_sincos(x, s1, c1);
      o[0] = static_cast<float>(c0), o[1] = static_cast<float>(s0), o[2] = static_cast<float>(c1), o[3] = static_cast<float>(s1);
      o[4] = (__double_as_longlong(c0) == __double_as_longlong(c1) && __double_as_longlong(s0) == __double_as_longlong(s1)) ? 1.f : 0.f;
      break;
    }
    default:
      break;
  }
}

}  // namespace vimg
