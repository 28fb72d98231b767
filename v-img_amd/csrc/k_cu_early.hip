// render_cu_kernel, the build whose vertex stages queue their rays as soon as they are known (EARLY): frames
// short of pixels and trees in global memory (vimg_hip.hip:make_launch_cu)
#include "kernel_tus.h"
#include "render_cu_kernel.h"

namespace vimg {
CuKernel vimg_cu_kernel_early(bool tex, bool deep, int) {
  if (tex) return deep ? render_cu_kernel<true, true, 16, 4, false, 1> : render_cu_kernel<true, false, 16, 4, false, 1>;
  return deep ? render_cu_kernel<false, true, 16, 4, false, 1> : render_cu_kernel<false, false, 16, 4, false, 1>;
}
}  // namespace vimg
