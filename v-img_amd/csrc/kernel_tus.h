// The render kernels are instantiated in translation units of their own (k_*.hip: compiled in
// parallel, and the two retired schedulers only into the development build); the ABI unit gets
// them through these getters.
#pragma once
#include "device_scene.h"

namespace vimg {

struct StageKArgs;
struct Pool4KArgs;
struct CuKArgs;
using RenderKernel = void (*)(const DScene, const RenderArgs, float*, DeviceStats*, unsigned int*);
using StageKernel = void (*)(const StageKArgs*);
using Pool4Kernel = void (*)(const Pool4KArgs*);
using CuKernel = void (*)(const CuKArgs);

RenderKernel vimg_lane_kernel(bool tex, int wps);                          // render_kernel<TEX, WPS>
Pool4Kernel vimg_pool4_kernel(bool tex, bool deep, int wps, bool group);    // render_pool4_kernel<TEX, DEEP, WPS, 1, GRP>
CuKernel vimg_cu_kernel(bool tex, bool deep, int nw);         // render_cu_kernel<TEX, DEEP, 16, 4, false, 0>
CuKernel vimg_cu_kernel_early(bool tex, bool deep, int nw);   // ... <..., false, 1>: rays queued as soon as they are known
CuKernel vimg_cu_kernel_diag(bool tex, bool deep, int nw);    // ... <..., true, 2>: statistics launches                   // render_cu_kernel<TEX, DEEP, NW, 4>
// development build (make dev, -DVIMG_DEV_SCHEDULERS): round 1's pooled kernel and the staged kernel,
// kept as cross-checks of the schedulers that ship; nullptr in the product library
RenderKernel vimg_pool_kernel(bool tex, int wps, bool deep);                // render_pool_kernel<TEX, WPS, DEEP>
StageKernel vimg_stage_kernel(bool tex, bool deep);                         // render_stage_kernel<TEX, DEEP>
bool vimg_has_dev_schedulers();

}  // namespace vimg
