// render_cu_kernel: one pool per compute unit, walking and shading waves, lock-free rings - the build whole
// frames are timed on (vertex stages queue their rays at their end)
#include "kernel_tus.h"
#include "render_cu_kernel.h"

namespace vimg {
// 16 waves per compute unit at 128 registers.  (12 waves at 168 registers, measured: config 2 351 against
// 311 ms, halves / quarters of it 207 / 163 against 177 / 149 ms, stand-ins of configs 3-5 7.2 / 2.5 / 3.2
// against 7.9 / 2.8 / 4.0 Grays/s - the fourth wave per SIMD hides more latency than 40 registers save.)
CuKernel vimg_cu_kernel(bool tex, bool deep, int) {
  if (tex) return deep ? render_cu_kernel<true, true, 16, 4, false, 0> : render_cu_kernel<true, false, 16, 4, false, 0>;
  return deep ? render_cu_kernel<false, true, 16, 4, false, 0> : render_cu_kernel<false, false, 16, 4, false, 0>;
}
}  // namespace vimg
