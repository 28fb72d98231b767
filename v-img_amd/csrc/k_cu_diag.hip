// render_cu_kernel, the build of statistics launches (DIAG: event counts, cycles by stage, ring occupancy,
// waiting times, walk passes): a translation unit of its own so that it compiles beside the timed build
#include "kernel_tus.h"
#include "render_cu_kernel.h"

namespace vimg {
CuKernel vimg_cu_kernel_diag(bool tex, bool deep, int) {
  if (tex) return deep ? render_cu_kernel<true, true, 16, 4, true, 2> : render_cu_kernel<true, false, 16, 4, true, 2>;
  return deep ? render_cu_kernel<false, true, 16, 4, true, 2> : render_cu_kernel<false, false, 16, 4, true, 2>;
}
}  // namespace vimg
