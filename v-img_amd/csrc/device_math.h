// Device-side arithmetic for the gfx950 path tracer.
//
// Numerics contract (DESIGN.md "Numerics"): the translation unit is compiled with
// -ffp-contract=off and without fast-math, so +,-,*,/ and sqrt round as IEEE binary32/binary64
// (hipcc keeps correctly rounded fp32 divide/sqrt and fp32 denormals).  A fused multiply-add is
// emitted only where the reference writes std::fma (include/geometry/triangle.h:11-21,
// include/material/disney_helpers/disney_common.h:40).  min/max follow std::min/std::max and
// glm::min/glm::max — "(b<a)?b:a" / "(a<b)?b:a" — NOT v_min/v_max, whose NaN handling differs.
// Transcendentals the reference calls in float (std::cos(float) ...) are evaluated with the
// double-precision OCML function and rounded to float; those it calls in double (unqualified
// cos(float) resolves to ::cos(double) under libstdc++) are evaluated in double.  MI355X runs
// FP64 vector math at half the FP32 rate, so keeping the reference's double sub-expressions is
// affordable here where it would not be on a consumer GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vimg {

#define VD __device__ __forceinline__

constexpr double kPi = 3.141592653589793238462643383279502884;      // std::numbers::pi
constexpr double kInvPi = 0.318309886183790671537767526745028724;   // std::numbers::inv_pi
constexpr float kInvPiF = 0.318309886183790671537767526745028724f;
#define VIMG_INF (__builtin_huge_valf())

struct f2 {
  float x, y;
};
struct f3 {
  float x, y, z;
};

VD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
VD f3 splat3(float s) { return f3{s, s, s}; }
VD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
VD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
VD f3 operator/(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }
VD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
VD f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
VD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
VD f3 operator+(f3 a, float s) { return f3{a.x + s, a.y + s, a.z + s}; }
VD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }

VD f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
VD f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
VD f2 operator*(f2 a, float s) { return f2{a.x * s, a.y * s}; }
VD f2 operator*(float s, f2 a) { return f2{s * a.x, s * a.y}; }
VD f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
VD f2 operator-(f2 a) { return f2{-a.x, -a.y}; }

VD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VD float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
VD float length2(f3 v) { return dot(v, v); }
VD float length(f3 v) { return __builtin_sqrtf(dot(v, v)); }
VD f3 normalize(f3 v) { return v * (1.0f / __builtin_sqrtf(dot(v, v))); }
VD f2 normalize(f2 v) { return v * (1.0f / __builtin_sqrtf(dot(v, v))); }
VD f3 cross(f3 x, f3 y) {
  return f3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
VD float sel_min(float a, float b) { return (b < a) ? b : a; }   // std::min / glm::min
VD float sel_max(float a, float b) { return (a < b) ? b : a; }   // std::max / glm::max
VD float clampf(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
VD int clampi(int v, int lo, int hi) { return v < lo ? lo : (hi < v ? hi : v); }
VD f3 mix3(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }
VD f2 mix2(f2 x, f2 y, float a) { return x * (1.0f - a) + y * a; }
VD bool is_nan(float x) { return x != x; }
VD bool is_inf(float x) { return __builtin_fabsf(x) == VIMG_INF; }
VD float absf(float x) { return __builtin_fabsf(x); }
VD float sqrt_f(float x) { return __builtin_sqrtf(x); }

// float-overload transcendentals of the reference: double evaluation, one rounding to float
VD float F_cos(float x) { return static_cast<float>(::cos(static_cast<double>(x))); }
VD float F_sin(float x) { return static_cast<float>(::sin(static_cast<double>(x))); }
// sin and cos of the same angle: one argument reduction and one pair of polynomials instead of two (the
// compiler does not merge the two calls; ocml's sin, cos and sincos share the reduction and the kernel, so
// the results are the same doubles - checked for every float in [-2 pi, 1e6] once, and on 2^24 arguments by tests/test_gpu_parity.py)
VD void D_sincos(float x, double& s, double& c) { ::sincos(static_cast<double>(x), &s, &c); }
VD float F_acos(float x) { return static_cast<float>(::acos(static_cast<double>(x))); }
VD float F_atan2(float y, float x) {
  return static_cast<float>(::atan2(static_cast<double>(y), static_cast<double>(x)));
}
VD float F_log(float x) { return static_cast<float>(::log(static_cast<double>(x))); }
VD float F_log2(float x) { return static_cast<float>(::log2(static_cast<double>(x))); }

// column-major mat4 * vec4 with glm's association: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
VD void mat_mul4(const float* m, float vx, float vy, float vz, float vw, float out[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float add0 = m[r] * vx + m[4 + r] * vy;
    float add1 = m[8 + r] * vz + m[12 + r] * vw;
    out[r] = add0 + add1;
  }
}
VD f3 mat_dir(const float* m, f3 d) {
  float r[4];
  mat_mul4(m, d.x, d.y, d.z, 0.0f, r);
  return f3{r[0], r[1], r[2]};
}

}  // namespace vimg
