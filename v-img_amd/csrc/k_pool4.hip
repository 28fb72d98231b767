// render_pool4_kernel: pools per wave / per workgroup with the vertex stage as calls (round 2's default).
// Since render_cu_kernel no launch policy picks it: like the other retired schedulers it is built
// into the development library only (make dev), where the GPU tests cross-check it.
#include "kernel_tus.h"
#ifdef VIMG_DEV_SCHEDULERS
#include "render_pool4_kernel.h"

namespace vimg {
namespace {
template <int WPS, bool GRP>
Pool4Kernel pool4_build(bool tex, bool deep) {
  if (tex) return deep ? render_pool4_kernel<true, true, WPS, 1, GRP> : render_pool4_kernel<true, false, WPS, 1, GRP>;
  return deep ? render_pool4_kernel<false, true, WPS, 1, GRP> : render_pool4_kernel<false, false, WPS, 1, GRP>;
}
}  // namespace
// (two rays per lane - NC = 2, both stepped in one pass of the box loop - measured slower and is not
// built: config 2 9.3 against 11.7 Grays/s at 64 spp, 228 B of scratch)
Pool4Kernel vimg_pool4_kernel(bool tex, bool deep, int wps, bool group) {
  if (group) return wps >= 4 ? pool4_build<4, true>(tex, deep) : pool4_build<3, true>(tex, deep);
  return wps >= 4 ? pool4_build<4, false>(tex, deep) : pool4_build<3, false>(tex, deep);
}
}  // namespace vimg
#else
namespace vimg {
Pool4Kernel vimg_pool4_kernel(bool, bool, int, bool) { return nullptr; }
}  // namespace vimg
#endif
