// render_kernel<TEX, WPS>: one lane owns one pixel (trace_pixel, probes' LDS layout, the by-name scheduler "lane")
#include "kernel_tus.h"
#include "render_kernels.h"

namespace vimg {
RenderKernel vimg_lane_kernel(bool tex, int wps) {
  if (tex) return wps >= 3 ? render_kernel<true, 3> : render_kernel<true, 2>;
  return wps >= 3 ? render_kernel<false, 3> : render_kernel<false, 2>;
}
}  // namespace vimg
