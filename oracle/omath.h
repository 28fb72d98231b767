// TEST INFRASTRUCTURE (see oracle.h).  Vector arithmetic and the transcendental policy of the
// CPU restatement.
//
// Vector ops: the reference does them through glm 1.0.1 (CMakeLists.txt:67-71; NOT vendored, so
// its algorithm is restated from glm's published definitions):
//   dot(a,b)      = a.x*b.x + a.y*b.y + a.z*b.z           (detail/func_geometric.inl compute_dot)
//   length(v)     = sqrt(dot(v,v));  length2(v) = dot(v,v) (gtx/norm.inl)
//   normalize(v)  = v * (1 / sqrt(dot(v,v)))               (inversesqrt)
//   cross(x,y)    = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
//   mix(x,y,a)    = x*(1-a) + y*a
//   min/max       = (b<a)?b:a / (a<b)?b:a ; clamp = min(max(x,lo),hi)
//   mat4*vec4     = (m0*v0 + m1*v1) + (m2*v2 + m3*v3)      (detail/type_mat4x4.inl)
//   vec / scalar  = component-wise division (not multiplication by a reciprocal)
// The library is compiled with -ffp-contract=off: a fused multiply-add appears only where the
// reference writes std::fma.
//
// Transcendentals: the reference calls the float overloads (std::cos(float) = cosf ...) in some
// places and, through unqualified calls that only see ::cos(double), the double functions in
// others (probe: with <cmath> only, decltype(cos(1.0f)) is double under g++ 11 / libstdc++).
// Double calls are restated as double calls.  Float calls go through F_* below:
//   default build : (float)fn((double)x) — the correctly rounded float in all but ~1e-8 of
//                   arguments, and reproducible bit for bit on the GPU (which evaluates the same
//                   double function and rounds), so the HIP path can be compared exactly;
//   -DORACLE_LIBM_FLOAT : the reference's literal float libm call.  tests/ checks the two builds
//                   against each other (unit values within 1 ulp, images within noise).
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

namespace om {

#if defined(ORACLE_LIBM_FLOAT) && ORACLE_LIBM_FLOAT
inline float F_cos(float x) { return ::cosf(x); }
inline float F_sin(float x) { return ::sinf(x); }
inline float F_tan(float x) { return ::tanf(x); }
inline float F_acos(float x) { return ::acosf(x); }
inline float F_atan(float x) { return ::atanf(x); }
inline float F_atan2(float y, float x) { return ::atan2f(y, x); }
inline float F_log(float x) { return ::logf(x); }
inline float F_log2(float x) { return ::log2f(x); }
#else
inline float F_cos(float x) { return static_cast<float>(::cos(static_cast<double>(x))); }
inline float F_sin(float x) { return static_cast<float>(::sin(static_cast<double>(x))); }
inline float F_tan(float x) { return static_cast<float>(::tan(static_cast<double>(x))); }
inline float F_acos(float x) { return static_cast<float>(::acos(static_cast<double>(x))); }
inline float F_atan(float x) { return static_cast<float>(::atan(static_cast<double>(x))); }
inline float F_atan2(float y, float x) {
  return static_cast<float>(::atan2(static_cast<double>(y), static_cast<double>(x)));
}
inline float F_log(float x) { return static_cast<float>(::log(static_cast<double>(x))); }
inline float F_log2(float x) { return static_cast<float>(::log2(static_cast<double>(x))); }
#endif

constexpr double kPi = 3.141592653589793238462643383279502884;      // std::numbers::pi
constexpr double kInvPi = 0.318309886183790671537767526745028724;   // std::numbers::inv_pi
constexpr float kInvPiF = 0.318309886183790671537767526745028724f;  // inv_pi_v<float>
constexpr float kInf = std::numeric_limits<float>::infinity();

struct vec2 {
  float x, y;
};
struct vec3 {
  float x, y, z;
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct vec4 {
  float x, y, z, w;
};

inline vec3 v3(float s) { return {s, s, s}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3 operator-(vec3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3& operator+=(vec3& a, vec3 b) { return a = a + b; }
inline vec3& operator*=(vec3& a, vec3 b) { return a = a * b; }
inline vec3& operator/=(vec3& a, float s) { return a = a / s; }
inline bool operator==(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
inline vec2 operator*(float s, vec2 a) { return {s * a.x, s * a.y}; }
inline vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }
inline vec2 operator-(vec2 a) { return {-a.x, -a.y}; }

inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
inline float length2(vec3 v) { return dot(v, v); }
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline vec2 normalize(vec2 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline vec3 cross(vec3 x, vec3 y) {
  return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline float fmin_(float a, float b) { return (b < a) ? b : a; }   // std::min / glm::min
inline float fmax_(float a, float b) { return (a < b) ? b : a; }   // std::max / glm::max
inline float clampf(float v, float lo, float hi) {                  // std::clamp
  return (v < lo) ? lo : ((hi < v) ? hi : v);
}
inline vec3 vmin(vec3 a, vec3 b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline vec3 vmax(vec3 a, vec3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline vec3 vabs(vec3 a) { return {std::abs(a.x), std::abs(a.y), std::abs(a.z)}; }
inline vec3 mix(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }
inline vec2 mix(vec2 x, vec2 y, float a) { return x * (1.0f - a) + y * a; }

// column-major 4x4 (glm::mat4) stored as float[16]
inline vec4 mat_mul(const float* m, vec4 v) {
  vec4 c0{m[0], m[1], m[2], m[3]}, c1{m[4], m[5], m[6], m[7]}, c2{m[8], m[9], m[10], m[11]},
      c3{m[12], m[13], m[14], m[15]};
  vec4 add0 = c0 * v.x + c1 * v.y;
  vec4 add1 = c2 * v.z + c3 * v.w;
  return add0 + add1;
}

}  // namespace om
