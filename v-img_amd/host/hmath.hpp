// Host-side float vector/matrix helpers.  The reference does this arithmetic through glm 1.0.1
// (CMakeLists.txt:67-71, not vendored); the operation orders below restate glm's published
// definitions (dot = x*x' + y*y' + z*z' left to right, normalize = v * (1/sqrt(dot)),
// mat*vec = (m0*v0 + m1*v1) + (m2*v2 + m3*v3)) so that flattened scenes carry the same bits the
// reference's loaders would produce.
#pragma once
#include <cmath>
#include <cstdint>

namespace hm {

struct V3 {
  float x, y, z;
  float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V4 {
  float x, y, z, w;
  float& operator[](int i) { return (&x)[i]; }
  float operator[](int i) const { return (&x)[i]; }
};
struct M4 {
  V4 c[4];  // columns
  V4& operator[](int i) { return c[i]; }
  const V4& operator[](int i) const { return c[i]; }
};

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline V3 vmin(V3 a, V3 b) {
  return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z};
}
inline V3 vmax(V3 a, V3 b) {
  return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z};
}

inline M4 identity() {
  return M4{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
inline V4 mul(const M4& m, V4 v) {
  V4 mul0 = m[0] * v.x, mul1 = m[1] * v.y, add0 = mul0 + mul1;
  V4 mul2 = m[2] * v.z, mul3 = m[3] * v.w, add1 = mul2 + mul3;
  return add0 + add1;
}
inline M4 mul(const M4& a, const M4& b) {
  M4 r;
  for (int j = 0; j < 4; ++j) r[j] = a[0] * b[j].x + a[1] * b[j].y + a[2] * b[j].z + a[3] * b[j].w;
  return r;
}
inline M4 scale(V3 s) {
  M4 m = identity();
  m[0].x = s.x;
  m[1].y = s.y;
  m[2].z = s.z;
  return m;
}
inline M4 translate(V3 t) {
  M4 m = identity();
  m[3] = {t.x, t.y, t.z, 1.f};
  return m;
}
// glm::toMat4(quat) with quat fields (x, y, z, w)
inline M4 quat_to_mat4(float qx, float qy, float qz, float qw) {
  float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz;
  float qxz = qx * qz, qxy = qx * qy, qyz = qy * qz;
  float qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;
  M4 m = identity();
  m[0] = {1.f - 2.f * (qyy + qzz), 2.f * (qxy + qwz), 2.f * (qxz - qwy), 0.f};
  m[1] = {2.f * (qxy - qwz), 1.f - 2.f * (qxx + qzz), 2.f * (qyz + qwx), 0.f};
  m[2] = {2.f * (qxz + qwy), 2.f * (qyz - qwx), 1.f - 2.f * (qxx + qyy), 0.f};
  return m;
}
// camToWorld, reference src/tl_camera.cpp:55-61
inline M4 cam_to_world(V3 from, V3 at, V3 up) {
  V3 z = normalize(from - at);
  V3 x = normalize(cross(up, z));
  V3 y = normalize(cross(z, x));
  return M4{{{x.x, x.y, x.z, 0.f}, {y.x, y.y, y.z, 0.f}, {z.x, z.y, z.z, 0.f},
             {from.x, from.y, from.z, 1.f}}};
}
// point transform with perspective divide as the loaders do (result /= result.w)
inline V3 xform_point(const M4& m, V3 p) {
  V4 r = mul(m, V4{p.x, p.y, p.z, 1.f});
  return {r.x / r.w, r.y / r.w, r.z / r.w};
}

}  // namespace hm
