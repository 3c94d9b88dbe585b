// oracle_math.hpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Scalar double-precision restatement of the geometry the reference's hot path uses.  The reference
// gets these operations from Eigen3 (un-vendored header library, >= 3.1.0 per CMakeLists.txt:39; the
// prebuilt libraries in the reference tree were made on Ubuntu 18.04 whose package is Eigen 3.3.4), so
// the Eigen algorithms are restated here from their published definitions and every function names the
// reference call site it serves.  Nothing in the product (libppo_ba.so) includes this file.
#pragma once
#include <cmath>
#include <cstring>
#include <utility>

namespace orc {

struct V3 {
  double v[3];
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};
struct M3 {
  double m[3][3];
};
struct Quat {  // Eigen::Quaterniond coefficient order x,y,z,w
  double x, y, z, w;
};

inline V3 mk(double a, double b, double c) { return V3{{a, b, c}}; }
inline V3 add(const V3& a, const V3& b) { return mk(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline V3 sub(const V3& a, const V3& b) { return mk(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline V3 scl(const V3& a, double s) { return mk(a[0] * s, a[1] * s, a[2] * s); }
inline double dot(const V3& a, const V3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline V3 cross(const V3& a, const V3& b) {
  return mk(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
inline V3 mul(const M3& A, const V3& x) {
  V3 r;
  for (int i = 0; i < 3; ++i) r[i] = A.m[i][0] * x[0] + A.m[i][1] * x[1] + A.m[i][2] * x[2];
  return r;
}
inline M3 mul(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
  return C;
}
inline M3 transpose(const M3& A) {
  M3 T;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T.m[i][j] = A.m[j][i];
  return T;
}
inline M3 identity3() { return M3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }

// Thirdparty/g2o/g2o/types/se3_ops.hpp:27-38
inline M3 skew(const V3& v) { return M3{{{0, -v[2], v[1]}, {v[2], 0, -v[0]}, {-v[1], v[0], 0}}}; }

// ---- Eigen::Quaternion pieces (used via se3quat.h:58-64,110-134,331-336) -------------------------
// Quaternion -> rotation matrix, Eigen QuaternionBase::toRotationMatrix.
inline M3 quat_to_matrix(const Quat& q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 R;
  R.m[0][0] = 1 - (tyy + tzz); R.m[0][1] = txy - twz;       R.m[0][2] = txz + twy;
  R.m[1][0] = txy + twz;       R.m[1][1] = 1 - (txx + tzz); R.m[1][2] = tyz - twx;
  R.m[2][0] = txz - twy;       R.m[2][1] = tyz + twx;       R.m[2][2] = 1 - (txx + tyy);
  return R;
}
// Rotation matrix -> quaternion, Eigen quaternionbase_assign_impl<Matrix3> (Shoemake's method).
inline Quat quat_from_matrix(const M3& R) {
  Quat q;
  double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R.m[2][1] - R.m[1][2]) * t;
    q.y = (R.m[0][2] - R.m[2][0]) * t;
    q.z = (R.m[1][0] - R.m[0][1]) * t;
  } else {
    int i = 0;
    if (R.m[1][1] > R.m[0][0]) i = 1;
    if (R.m[2][2] > R.m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double c[3];
    t = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R.m[k][j] - R.m[j][k]) * t;
    c[j] = (R.m[j][i] + R.m[i][j]) * t;
    c[k] = (R.m[k][i] + R.m[i][k]) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}
// Hamilton product a*b, Eigen quat_product.
inline Quat quat_mul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
// q * v, Eigen QuaternionBase::_transformVector: v + 2w(u x v) + 2 u x (u x v).
inline V3 quat_rotate(const Quat& q, const V3& v) {
  const V3 u = mk(q.x, q.y, q.z);
  V3 uv = cross(u, v);
  uv = add(uv, uv);
  return add(add(v, scl(uv, q.w)), cross(u, uv));
}
inline Quat quat_conj(const Quat& q) { return Quat{-q.x, -q.y, -q.z, q.w}; }
// Eigen::AngleAxisd -> quaternion
inline Quat quat_from_angle_axis(double angle, const V3& axis) {
  const double ha = 0.5 * angle, s = std::sin(ha);
  return Quat{s * axis[0], s * axis[1], s * axis[2], std::cos(ha)};
}
// Eigen::AngleAxisd::toRotationMatrix (used by AngleAxisd * Vector3d, G2O_Plane3D.h:100-101)
inline M3 angle_axis_to_matrix(double angle, const V3& ax) {
  const double s = std::sin(angle), c = std::cos(angle);
  const V3 sin_axis = scl(ax, s);
  const V3 cos1_axis = scl(ax, 1.0 - c);
  M3 R;
  double tmp;
  tmp = cos1_axis[0] * ax[1]; R.m[0][1] = tmp - sin_axis[2]; R.m[1][0] = tmp + sin_axis[2];
  tmp = cos1_axis[0] * ax[2]; R.m[0][2] = tmp + sin_axis[1]; R.m[2][0] = tmp - sin_axis[1];
  tmp = cos1_axis[1] * ax[2]; R.m[1][2] = tmp - sin_axis[0]; R.m[2][1] = tmp + sin_axis[0];
  R.m[0][0] = cos1_axis[0] * ax[0] + c;
  R.m[1][1] = cos1_axis[1] * ax[1] + c;
  R.m[2][2] = cos1_axis[2] * ax[2] + c;
  return R;
}

// ---- g2o::SE3Quat (Thirdparty/g2o/g2o/types/se3quat.h) -------------------------------------------
struct SE3 {
  Quat r;
  V3 t;
};
inline SE3 se3_identity() { return SE3{Quat{0, 0, 0, 1}, mk(0, 0, 0)}; }
// se3quat.h:331-336
inline void se3_normalize_rotation(SE3& T) {
  if (T.r.w < 0) { T.r.x = -T.r.x; T.r.y = -T.r.y; T.r.z = -T.r.z; T.r.w = -T.r.w; }
  const double n = std::sqrt(T.r.x * T.r.x + T.r.y * T.r.y + T.r.z * T.r.z + T.r.w * T.r.w);
  T.r.x /= n; T.r.y /= n; T.r.z /= n; T.r.w /= n;
}
// se3quat.h:62-64 (ctor normalises)
inline SE3 se3_make(const Quat& q, const V3& t) {
  SE3 T{q, t};
  se3_normalize_rotation(T);
  return T;
}
// se3quat.h:110-116
inline SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 r = a;
  r.t = add(r.t, quat_rotate(a.r, b.t));
  r.r = quat_mul(a.r, b.r);
  se3_normalize_rotation(r);
  return r;
}
// se3quat.h:125-127, 268-271
inline V3 se3_map(const SE3& T, const V3& p) { return add(quat_rotate(T.r, p), T.t); }
// se3quat.h:129-134
inline SE3 se3_inverse(const SE3& T) {
  SE3 r;
  r.r = quat_conj(T.r);
  r.t = quat_rotate(r.r, scl(T.t, -1.0));
  return r;
}
// se3quat.h:274-308 (small-angle branch kept exactly, including R = I + W + W^2)
inline SE3 se3_exp(const double upd[6]) {
  const V3 omega = mk(upd[0], upd[1], upd[2]);
  const V3 upsilon = mk(upd[3], upd[4], upd[5]);
  const double theta = norm(omega);
  const M3 Om = skew(omega);
  M3 R, V;
  const M3 I = identity3();
  if (theta < 0.00001) {
    const M3 Om2 = mul(Om, Om);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R.m[i][j] = I.m[i][j] + Om.m[i][j] + Om2.m[i][j];
    V = R;
  } else {
    const M3 Om2 = mul(Om, Om);
    const double a = std::sin(theta) / theta;
    const double b = (1 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        R.m[i][j] = I.m[i][j] + a * Om.m[i][j] + b * Om2.m[i][j];
        V.m[i][j] = I.m[i][j] + b * Om.m[i][j] + c * Om2.m[i][j];
      }
  }
  return se3_make(quat_from_matrix(R), mul(V, upsilon));
}
// VertexSE3Expmap::oplusImpl, types_six_dof_expmap.h:73-91 (fix_rotation/fix_translation are false)
inline SE3 pose_oplus(const SE3& est, const double upd[6]) { return se3_mul(se3_exp(upd), est); }

// ---- g2o::Plane3D (include/G2O_Plane3D.h:22-140) --------------------------------------------------
struct Plane {
  double c[4];
};
// G2O_Plane3D.h:120-125
inline void plane_normalize(Plane& p) {
  const double n = std::sqrt(p.c[0] * p.c[0] + p.c[1] * p.c[1] + p.c[2] * p.c[2]);
  const double inv = 1. / n;
  for (int i = 0; i < 4; ++i) p.c[i] = p.c[i] * inv;
  if (p.c[3] < 0.0)
    for (int i = 0; i < 4; ++i) p.c[i] = -p.c[i];
}
// G2O_Plane3D.h:45-48
inline Plane plane_from_vector(const double v[4]) {
  Plane p;
  std::memcpy(p.c, v, sizeof(p.c));
  plane_normalize(p);
  return p;
}
// src/Converter.cc:172-181 (sign flip BEFORE normalisation; values arrive as float32 there)
inline Plane plane_from_cv(const double v[4]) {
  double t[4] = {v[0], v[1], v[2], v[3]};
  if (t[3] < 0.0)
    for (int i = 0; i < 4; ++i) t[i] = -t[i];
  return plane_from_vector(t);
}
inline V3 plane_normal(const Plane& p) { return mk(p.c[0], p.c[1], p.c[2]); }
inline double plane_distance(const Plane& p) { return -p.c[3]; }         // :58-60
inline double azimuth(const V3& v) { return std::atan2(v[1], v[0]); }    // :50-52
inline double elevation(const V3& v) { return std::atan2(v[2], std::sqrt(v[0] * v[0] + v[1] * v[1])); }  // :54-56
// G2O_Plane3D.h:66-72  (AngleAxis * AngleAxis is a quaternion product in Eigen)
inline M3 plane_rotation(const V3& v) {
  const double az = azimuth(v), el = elevation(v);
  const Quat qa = quat_from_angle_axis(az, mk(0, 0, 1));
  const Quat qe = quat_from_angle_axis(-el, mk(0, 1, 0));
  return quat_to_matrix(quat_mul(qa, qe));
}
// G2O_Plane3D.h:74-87
inline Plane plane_oplus(const Plane& p, const double v[3]) {
  const double az = v[0], el = v[1];
  const double s = std::sin(el), c = std::cos(el);
  const V3 n = mk(c * std::cos(az), c * std::sin(az), s);
  const M3 R = plane_rotation(plane_normal(p));
  const double d = plane_distance(p) + v[2];
  const V3 rn = mul(R, n);
  Plane q;
  q.c[0] = rn[0]; q.c[1] = rn[1]; q.c[2] = rn[2]; q.c[3] = -d;
  plane_normalize(q);
  return q;
}
// G2O_Plane3D.h:89-95   this->ominus(plane)
inline void plane_ominus(const Plane& self, const Plane& other, double e[3]) {
  const M3 R = transpose(plane_rotation(plane_normal(self)));
  const V3 n = mul(R, plane_normal(other));
  e[0] = azimuth(n);
  e[1] = elevation(n);
  e[2] = plane_distance(self) - plane_distance(other);
}
// G2O_Plane3D.h:97-106
inline void plane_ominus_ver(const Plane& self, const Plane& other, double e[2]) {
  const V3 v = cross(plane_normal(self), plane_normal(other));
  const double vn = norm(v);
  const M3 ver = angle_axis_to_matrix(M_PI / 2, mk(v[0] / vn, v[1] / vn, v[2] / vn));  // Eigen: v / v.norm() divides
  const V3 b = mul(ver, plane_normal(self));
  const M3 R = transpose(plane_rotation(b));
  const V3 n = mul(R, plane_normal(other));
  e[0] = azimuth(n);
  e[1] = elevation(n);
}
// G2O_Plane3D.h:108-117
inline void plane_ominus_par(const Plane& self, const Plane& other, double e[2]) {
  V3 nor = plane_normal(self);
  if (dot(plane_normal(other), nor) < 0) nor = scl(nor, -1.0);
  const M3 R = transpose(plane_rotation(nor));
  const V3 n = mul(R, plane_normal(other));
  e[0] = azimuth(n);
  e[1] = elevation(n);
}
// operator*(Isometry3D, Plane3D), G2O_Plane3D.h:131-140; Isometry from SE3Quat se3quat.h:341-346
inline Plane plane_transform(const SE3& T, const Plane& p) {
  const M3 R = quat_to_matrix(T.r);
  const V3 n2 = mul(R, plane_normal(p));
  double v2[4] = {n2[0], n2[1], n2[2], p.c[3] - dot(T.t, n2)};
  if (v2[3] < 0.0)
    for (int i = 0; i < 4; ++i) v2[i] = -v2[i];
  return plane_from_vector(v2);
}

// ---- g2o::cuboid (include/g2o_cuboid.h:30-257, src/g2o_cuboid.cc:6-67) ---------------------------
struct Cuboid {
  SE3 pose;  // object -> world
  V3 scale;  // half extents
};
// src/g2o_cuboid.cc:6-36
inline SE3 exptwist_norollpitch(const double upd[6]) {
  const V3 omega = mk(upd[0], upd[1], upd[2]);
  const V3 upsilon = mk(upd[3], upd[4], upd[5]);
  const double theta = norm(omega);
  const M3 Om = skew(omega);
  const double cz = std::cos(omega[2]), sz = std::sin(omega[2]);
  const M3 R = M3{{{cz, -sz, 0}, {sz, cz, 0}, {0, 0, 1}}};
  M3 V;
  if (theta < 0.00001) {
    V = R;
  } else {
    const M3 Om2 = mul(Om, Om);
    const M3 I = identity3();
    const double b = (1 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) V.m[i][j] = I.m[i][j] + b * Om.m[i][j] + c * Om2.m[i][j];
  }
  return se3_make(quat_from_matrix(R), mul(V, upsilon));
}
// VertexCuboid::oplusImpl, src/g2o_cuboid.cc:39-67.  flags: bit0 fixrollpitch, bit1 fixheight, bit2 fixrotation
inline Cuboid cuboid_oplus(const Cuboid& est, const double upd[9], unsigned flags, const double fixedscale[3]) {
  Cuboid nc;
  nc.pose = se3_identity();
  nc.scale = mk(0, 0, 0);
  if (flags & 4u) {  // whether_fixrotation: the fresh cuboid keeps identity rotation, as in the reference
    nc.pose.t = add(est.pose.t, mk(upd[3], upd[4], upd[5]));
  } else if (flags & 1u) {
    double u2[6] = {0, 0, upd[2], upd[3], upd[4], upd[5]};
    nc.pose = se3_mul(est.pose, exptwist_norollpitch(u2));
  } else {
    nc.pose = se3_mul(est.pose, se3_exp(upd));
  }
  if (flags & 2u) nc.pose.t = mk(nc.pose.t[0], est.pose.t[1], nc.pose.t[2]);
  if (fixedscale && fixedscale[0] > 0)
    nc.scale = mk(fixedscale[0], fixedscale[1], fixedscale[2]);
  else
    nc.scale = add(est.scale, mk(upd[6], upd[7], upd[8]));
  return nc;
}
// SE3Quat::log, Thirdparty/g2o/g2o/types/se3quat.h:229-266 (deltaR: se3_ops.hpp:40-47)
inline void se3_log(const SE3& T, double out[6]) {
  const M3 R = quat_to_matrix(T.r);
  const double d = 0.5 * (R.m[0][0] + R.m[1][1] + R.m[2][2] - 1);
  const V3 dR = mk(R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]);
  V3 omega;
  M3 Vinv;
  const M3 I = identity3();
  if (d > 0.99999) {
    omega = scl(dR, 0.5);
    const M3 Om = skew(omega);
    const M3 Om2 = mul(Om, Om);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Vinv.m[i][j] = I.m[i][j] - 0.5 * Om.m[i][j] + (1. / 12.) * Om2.m[i][j];
  } else {
    const double theta = std::acos(d);
    omega = scl(dR, theta / (2 * std::sqrt(1 - d * d)));
    const M3 Om = skew(omega);
    const M3 Om2 = mul(Om, Om);
    const double k = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Vinv.m[i][j] = I.m[i][j] - 0.5 * Om.m[i][j] + k * Om2.m[i][j];
  }
  const V3 ups = mul(Vinv, T.t);
  for (int i = 0; i < 3; ++i) { out[i] = omega[i]; out[i + 3] = ups[i]; }
}
// cuboid::rotate_cuboid, include/g2o_cuboid.h:112-123: the same box with another face called "front"
inline Cuboid cuboid_rotate(const Cuboid& c, double yaw_angle) {
  Cuboid r;
  const SE3 rot = se3_make(Quat{0, 0, std::sin(yaw_angle * 0.5), std::cos(yaw_angle * 0.5)}, mk(0, 0, 0));
  r.pose = se3_mul(c.pose, rot);
  r.scale = c.scale;
  if ((yaw_angle == M_PI / 2.0) || (yaw_angle == -M_PI / 2.0) || (yaw_angle == 3 * M_PI / 2.0)) std::swap(r.scale[0], r.scale[1]);
  return r;
}
// cuboid::cube_log_error, include/g2o_cuboid.h:73-80
inline void cuboid_log_error(const Cuboid& self, const Cuboid& other, double e[9]) {
  const SE3 diff = se3_mul(se3_inverse(other.pose), self.pose);
  se3_log(diff, e);
  for (int i = 0; i < 3; ++i) e[6 + i] = self.scale[i] - other.scale[i];
}
// cuboid::min_log_error, include/g2o_cuboid.h:83-109: the smallest of the four errors against the measured box turned by
// -90, 0, 90, 180 degrees about its z axis (Eigen's minCoeff keeps the FIRST minimum)
inline void cuboid_min_log_error(const Cuboid& self, const Cuboid& other, double e[9]) {
  const double angles[4] = {-1, 0, 1, 2};
  double best = 0;
  for (int i = 0; i < 4; ++i) {
    double ei[9];
    cuboid_log_error(self, cuboid_rotate(other, angles[i] * M_PI / 2.0), ei);
    double n2 = 0;
    for (int k = 0; k < 9; ++k) n2 += ei[k] * ei[k];
    const double n = std::sqrt(n2);
    if (i == 0 || n < best) {
      best = n;
      for (int k = 0; k < 9; ++k) e[k] = ei[k];
    }
  }
}
// EdgeSE3Cuboid::computeError, include/g2o_cuboid.h:332-341: the measured cuboid (camera frame) taken to the world with the
// inverse of the world->camera pose, compared with the cuboid vertex
inline void cuboid3d_error(const SE3& Tcw, const Cuboid& global_cube, const Cuboid& meas_local, double e[9]) {
  Cuboid esti;
  esti.pose = se3_mul(se3_inverse(Tcw), meas_local.pose);
  esti.scale = meas_local.scale;
  cuboid_min_log_error(global_cube, esti, e);
}
// corner sign table, include/g2o_cuboid.h:202-204
static const double kCornerSign[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {1, -1, -1, 1, 1, -1, -1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
// compute3D_BoxCorner + projectOntoImage, include/g2o_cuboid.h:173-215 (homogeneous w is exactly 1)
inline void cuboid_project_corners(const Cuboid& c, const SE3& Tcw, const double K[9], double uv[8][2]) {
  const M3 Rwo = quat_to_matrix(c.pose.r);
  const M3 Rcw = quat_to_matrix(Tcw.r);
  for (int k = 0; k < 8; ++k) {
    V3 pw, pc;
    for (int i = 0; i < 3; ++i) {
      double acc = (Rwo.m[i][0] * c.scale[0]) * kCornerSign[0][k];
      acc += (Rwo.m[i][1] * c.scale[1]) * kCornerSign[1][k];
      acc += (Rwo.m[i][2] * c.scale[2]) * kCornerSign[2][k];
      acc += c.pose.t[i];
      pw[i] = acc;
    }
    for (int i = 0; i < 3; ++i) pc[i] = Rcw.m[i][0] * pw[0] + Rcw.m[i][1] * pw[1] + Rcw.m[i][2] * pw[2] + Tcw.t[i];
    double h[3];
    for (int i = 0; i < 3; ++i) h[i] = K[3 * i + 0] * pc[0] + K[3 * i + 1] * pc[1] + K[3 * i + 2] * pc[2];
    uv[k][0] = h[0] / h[2];
    uv[k][1] = h[1] / h[2];
  }
}
// projectOntoImageRect + projectOntoImageBbox, include/g2o_cuboid.h:218-234
inline void cuboid_project_bbox(const Cuboid& c, const SE3& Tcw, const double K[9], double bbox[4]) {
  double uv[8][2];
  cuboid_project_corners(c, Tcw, K, uv);
  double lo[2] = {uv[0][0], uv[0][1]}, hi[2] = {uv[0][0], uv[0][1]};
  for (int k = 1; k < 8; ++k)
    for (int a = 0; a < 2; ++a) {
      if (uv[k][a] < lo[a]) lo[a] = uv[k][a];
      if (uv[k][a] > hi[a]) hi[a] = uv[k][a];
    }
  bbox[0] = (hi[0] + lo[0]) / 2;
  bbox[1] = (hi[1] + lo[1]) / 2;
  bbox[2] = hi[0] - lo[0];
  bbox[3] = hi[1] - lo[1];
}
// point_boundary_error, include/g2o_cuboid.h:237-255 (point_scale = 1)
inline V3 cuboid_point_boundary_error(const Cuboid& c, const V3& p, double max_outside_margin_ratio) {
  const V3 l = se3_map(se3_inverse(c.pose), p);
  V3 e;
  for (int i = 0; i < 3; ++i) {
    const double a = 1.0 * std::fabs(l[i]);
    if (a < c.scale[i])
      e[i] = 0;
    else if (a < (max_outside_margin_ratio + 1) * c.scale[i])
      e[i] = a - c.scale[i];
    else
      e[i] = max_outside_margin_ratio * c.scale[i];
  }
  return e;
}

// ---- RobustKernelHuber::robustify, Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:78-91 ----------
inline void huber_robustify(double e, double delta, double rho[3]) {
  const double dsqr = (double)(float)(delta * delta);  // setDelta, :65-69; the member is `float dsqr` (robust_kernel_impl.h:84)
  if (e <= dsqr) {
    rho[0] = e; rho[1] = 1.; rho[2] = 0.;
  } else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

}  // namespace orc
