// render_cu_kernel: one pool per compute unit, walking and shading waves, lock-free rings
#include "kernel_tus.h"
#include "render_cu_kernel.h"

namespace vimg {
namespace {
template <int NW>
CuKernel cu_build(bool tex, bool deep) {
  if (tex) return deep ? render_cu_kernel<true, true, NW, 4> : render_cu_kernel<true, false, NW, 4>;
  return deep ? render_cu_kernel<false, true, NW, 4> : render_cu_kernel<false, false, NW, 4>;
}
}  // namespace
CuKernel vimg_cu_kernel(bool tex, bool deep, int) { return cu_build<16>(tex, deep); }
}  // namespace vimg
