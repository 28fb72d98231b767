// The two schedulers no launch policy picks: round 1's pooled kernel (everything inlined, 255
// registers) and the staged kernel (path state in HBM, stages coupled by global queues).  They are
// reference implementations - the GPU tests cross-check the shipped schedulers against them - and
// live in the development build only (make dev; tests select it through VIMG_HIP_LIB).
#include "kernel_tus.h"
#ifdef VIMG_DEV_SCHEDULERS
#include "render_pool_kernel.h"
#include "render_stage_kernel.h"
#endif

namespace vimg {
#ifdef VIMG_DEV_SCHEDULERS
RenderKernel vimg_pool_kernel(bool tex, int wps, bool deep) {
  if (deep) {
    if (tex) return wps >= 3 ? render_pool_kernel<true, 3, true> : render_pool_kernel<true, 2, true>;
    return wps >= 3 ? render_pool_kernel<false, 3, true> : render_pool_kernel<false, 2, true>;
  }
  if (tex) return wps >= 3 ? render_pool_kernel<true, 3, false> : render_pool_kernel<true, 2, false>;
  return wps >= 3 ? render_pool_kernel<false, 3, false> : render_pool_kernel<false, 2, false>;
}
StageKernel vimg_stage_kernel(bool tex, bool deep) {
  if (tex) return deep ? render_stage_kernel<true, true> : render_stage_kernel<true, false>;
  return deep ? render_stage_kernel<false, true> : render_stage_kernel<false, false>;
}
bool vimg_has_dev_schedulers() { return true; }
#else
RenderKernel vimg_pool_kernel(bool, int, bool) { return nullptr; }
StageKernel vimg_stage_kernel(bool, bool) { return nullptr; }
bool vimg_has_dev_schedulers() { return false; }
#endif
}  // namespace vimg
