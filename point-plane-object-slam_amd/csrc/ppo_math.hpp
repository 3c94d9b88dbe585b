// ppo_math.hpp -- FP64 geometry used by the HIP kernels of libppo_ba.so.
//
// Every function is __host__ __device__ so that tests can exercise the exact device formulas on the CPU
// (csrc/hostcheck.cpp); the product path only ever calls them from kernels.
// Semantics follow the reference (paths relative to the reference checkout, G2O = Thirdparty/g2o/g2o):
//   SE3 exp / compose / normalise    G2O/types/se3quat.h:110-134,274-336
//   pose update                      G2O/types/types_six_dof_expmap.h:73-91
//   plane algebra                    include/G2O_Plane3D.h:45-140
//   cuboid update / projection       src/g2o_cuboid.cc:6-67, include/g2o_cuboid.h:173-255
//   Huber                            G2O/core/robust_kernel_impl.cpp:78-91
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PPO_HD __host__ __device__ __forceinline__
#else
#define PPO_HD inline
#endif

namespace ppo {

struct Vec3 {
  double x, y, z;
};
struct Mat3 {  // row-major
  double a[9];
};
struct Quat {
  double x, y, z, w;
};
struct Pose {  // rigid transform as unit quaternion (w >= 0) + translation
  Quat q;
  Vec3 t;
};
struct PlaneC {  // (n, d) with |n| = 1 and d >= 0
  double nx, ny, nz, d;
};
struct Cub {
  Pose T;  // object -> world
  Vec3 s;  // half extents
};

// x^3 from two-term products by FMA: within one ulp of the exact cube (the last line rounds twice, so correct rounding is not guaranteed; it equals
// the correctly rounded cube on every one of the 50 000 arguments tests/test_lm_control.py draws).  The reference takes the Levenberg-Marquardt decision
// on the host with alpha = 1 - pow(2 rho - 1, 3) (levenberg.cpp:131); glibc's pow is within 0.52 ulp of the exact value, i.e. equal to this
// except for about one argument in a thousand, where it is the neighbouring double (tests/test_lm_control.py).  k_decide (ppo_kernels.hpp).
PPO_HD double cube_rounded_once(double x) {
  const double c = x * x * x;
  const double m = c < 0 ? -c : c;
  if (!(m < 1.0e300) || !(m > 1.0e-280)) return c;  // (overflow / underflow / NaN: nothing to refine)
  const double hi = x * x, lo = __builtin_fma(x, x, -hi);
  const double p = hi * x, pe = __builtin_fma(hi, x, -p);
  return p + (pe + lo * x);
}
PPO_HD Vec3 v3(double x, double y, double z) { return Vec3{x, y, z}; }
PPO_HD Vec3 operator+(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PPO_HD Vec3 operator-(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PPO_HD Vec3 operator*(Vec3 a, double s) { return Vec3{a.x * s, a.y * s, a.z * s}; }
PPO_HD double dot3(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PPO_HD Vec3 cross3(Vec3 a, Vec3 b) { return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PPO_HD Vec3 matvec(const Mat3& M, Vec3 v) {
  return Vec3{M.a[0] * v.x + M.a[1] * v.y + M.a[2] * v.z, M.a[3] * v.x + M.a[4] * v.y + M.a[5] * v.z,
              M.a[6] * v.x + M.a[7] * v.y + M.a[8] * v.z};
}
PPO_HD Vec3 matTvec(const Mat3& M, Vec3 v) {
  return Vec3{M.a[0] * v.x + M.a[3] * v.y + M.a[6] * v.z, M.a[1] * v.x + M.a[4] * v.y + M.a[7] * v.z,
              M.a[2] * v.x + M.a[5] * v.y + M.a[8] * v.z};
}

// unit quaternion -> rotation matrix (the expansion Eigen uses, so products round alike)
PPO_HD Mat3 rotmat(Quat q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  Mat3 R;
  R.a[0] = 1 - (tyy + tzz); R.a[1] = txy - twz;       R.a[2] = txz + twy;
  R.a[3] = txy + twz;       R.a[4] = 1 - (txx + tzz); R.a[5] = tyz - twx;
  R.a[6] = txz - twy;       R.a[7] = tyz + twx;       R.a[8] = 1 - (txx + tyy);
  return R;
}
// rotation(-like) matrix -> quaternion, largest-component branch selection (Shoemake)
PPO_HD Quat quat_of(const Mat3& R) {
  Quat q;
  const double tr = R.a[0] + R.a[4] + R.a[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    q.w = 0.5 * s;
    s = 0.5 / s;
    q.x = (R.a[7] - R.a[5]) * s;
    q.y = (R.a[2] - R.a[6]) * s;
    q.z = (R.a[3] - R.a[1]) * s;
  } else if (R.a[0] >= R.a[4] && R.a[0] >= R.a[8]) {  // i = 0
    double s = sqrt(R.a[0] - R.a[4] - R.a[8] + 1.0);
    q.x = 0.5 * s;
    s = 0.5 / s;
    q.w = (R.a[7] - R.a[5]) * s;
    q.y = (R.a[3] + R.a[1]) * s;
    q.z = (R.a[6] + R.a[2]) * s;
  } else if (R.a[4] > R.a[0] && R.a[4] >= R.a[8]) {  // i = 1
    double s = sqrt(R.a[4] - R.a[8] - R.a[0] + 1.0);
    q.y = 0.5 * s;
    s = 0.5 / s;
    q.w = (R.a[2] - R.a[6]) * s;
    q.z = (R.a[7] + R.a[5]) * s;
    q.x = (R.a[1] + R.a[3]) * s;
  } else {  // i = 2
    double s = sqrt(R.a[8] - R.a[0] - R.a[4] + 1.0);
    q.z = 0.5 * s;
    s = 0.5 / s;
    q.w = (R.a[3] - R.a[1]) * s;
    q.x = (R.a[6] + R.a[2]) * s;
    q.y = (R.a[5] + R.a[7]) * s;
  }
  return q;
}
PPO_HD Quat qmul(Quat a, Quat b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
PPO_HD Vec3 qrot(Quat q, Vec3 v) {  // v + 2w (u x v) + 2 u x (u x v)
  const Vec3 u = v3(q.x, q.y, q.z);
  Vec3 uv = cross3(u, v);
  uv = uv + uv;
  return v + uv * q.w + cross3(u, uv);
}
PPO_HD Quat canon(Quat q) {  // g2o normalizeRotation: w >= 0, unit norm
  if (q.w < 0) q = Quat{-q.x, -q.y, -q.z, -q.w};
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / n, q.y / n, q.z / n, q.w / n};
}
PPO_HD Pose compose(const Pose& A, const Pose& B) {  // A * B
  Pose r;
  r.t = A.t + qrot(A.q, B.t);
  r.q = canon(qmul(A.q, B.q));
  return r;
}
PPO_HD Pose inverse(const Pose& A) {
  Pose r;
  r.q = Quat{-A.q.x, -A.q.y, -A.q.z, A.q.w};
  r.t = qrot(r.q, A.t * -1.0);
  return r;
}
PPO_HD Vec3 apply(const Pose& A, Vec3 p) { return qrot(A.q, p) + A.t; }

// W = [w]_x and W^2 in closed form
PPO_HD void skew_sq(Vec3 w, Mat3& W, Mat3& W2) {
  W = Mat3{{0, -w.z, w.y, w.z, 0, -w.x, -w.y, w.x, 0}};
  W2 = Mat3{{-(w.y * w.y) - w.z * w.z, w.x * w.y, w.x * w.z,
             w.x * w.y, -(w.z * w.z) - w.x * w.x, w.y * w.z,
             w.x * w.z, w.y * w.z, -(w.x * w.x) - w.y * w.y}};
}
// SE3 exponential with g2o's branches: below 1e-5 rad R = V = I + W + W^2 (not I + W + W^2/2)
PPO_HD Pose exp_se3(const double* u) {
  const Vec3 om = v3(u[0], u[1], u[2]), up = v3(u[3], u[4], u[5]);
  const double th = sqrt(dot3(om, om));
  Mat3 W, W2, R, V;
  skew_sq(om, W, W2);
  // g2o forms Omega*Omega by a matrix product; the closed form above has the same terms but may round
  // differently in the last bit -- irrelevant at |omega|^2 ~ 1e-18 next to 1.
  double a = 1.0, b = 1.0, c = 1.0;
  const bool small = th < 0.00001;
  if (!small) {
    a = sin(th) / th;
    b = (1 - cos(th)) / (th * th);
    c = (th - sin(th)) / (th * th * th);
  }
  for (int i = 0; i < 9; ++i) {
    const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
    R.a[i] = id + a * W.a[i] + b * W2.a[i];
    V.a[i] = small ? R.a[i] : id + b * W.a[i] + c * W2.a[i];
  }
  Pose P;
  P.q = canon(quat_of(R));
  P.t = matvec(V, up);
  return P;
}
PPO_HD Pose pose_boxplus(const Pose& T, const double* u) { return compose(exp_se3(u), T); }

// SE3 logarithm with g2o's branches (SE3Quat::log, Thirdparty/g2o/g2o/types/se3quat.h:229-266): [omega, upsilon]
PPO_HD void log_se3(const Pose& T, double* out) {
  const Mat3 R = rotmat(T.q);
  const double d = 0.5 * (R.a[0] + R.a[4] + R.a[8] - 1);
  const Vec3 dR = v3(R.a[7] - R.a[5], R.a[2] - R.a[6], R.a[3] - R.a[1]);
  Vec3 om;
  double k;
  if (d > 0.99999) {
    om = dR * 0.5;
    k = 1. / 12.;
  } else {
    const double th = acos(d);
    om = dR * (th / (2 * sqrt(1 - d * d)));
    k = (1 - th / (2 * tan(th / 2))) / (th * th);
  }
  Mat3 W, W2;
  skew_sq(om, W, W2);
  Mat3 Vi;
  for (int i = 0; i < 9; ++i) Vi.a[i] = ((i == 0 || i == 4 || i == 8) ? 1.0 : 0.0) - 0.5 * W.a[i] + k * W2.a[i];
  const Vec3 up = matvec(Vi, T.t);
  out[0] = om.x; out[1] = om.y; out[2] = om.z; out[3] = up.x; out[4] = up.y; out[5] = up.z;
}

// ---- planes ------------------------------------------------------------------------------------------
PPO_HD PlaneC plane_canon(double nx, double ny, double nz, double d) {
  const double inv = 1. / sqrt(nx * nx + ny * ny + nz * nz);
  nx *= inv; ny *= inv; nz *= inv; d *= inv;
  if (d < 0.0) { nx = -nx; ny = -ny; nz = -nz; d = -d; }
  return PlaneC{nx, ny, nz, d};
}
PPO_HD double az_of(Vec3 v) { return atan2(v.y, v.x); }
PPO_HD double el_of(Vec3 v) { return atan2(v.z, sqrt(v.x * v.x + v.y * v.y)); }
// sin / cos of a numeric-differentiation step: for |x| < 1e-8 the correctly rounded results are x and 1 exactly
// (x^3/6 and x^2/2 lie below half an ulp), so the library calls are skipped on that path
PPO_HD void sincos_step(double x, double& s, double& c) {
  if (fabs(x) < 1e-8) { s = x; c = 1.0; }
  else { s = sin(x); c = cos(x); }
}
// Rz(azimuth) * Ry(-elevation) of a direction n.  The reference builds it as AngleAxis(az, z) * AngleAxis(-el, y) from
// az = atan2(y, x), el = atan2(z, |xy|) (include/G2O_Plane3D.h:66-72); the same matrix written with the sines and cosines
// those angles have by construction -- cos az = x/r, sin az = y/r, cos el = r/|n|, sin el = z/|n| -- needs no inverse
// trigonometry and no half-angle quaternion (2 atan2 + 4 sin/cos of ~150 FP64 instructions each per call).
PPO_HD Mat3 plane_frame(Vec3 n) {
  const double r2 = n.x * n.x + n.y * n.y;
  const double r = sqrt(r2), nn = sqrt(r2 + n.z * n.z);
  // r == 0 (a normal exactly along z): atan2(+-0, +0) = +-0 but atan2(+-0, -0) = +-pi -- the sign of a ZERO x decides the chart,
  // and the reference does produce (-0, -0, 1) whenever Plane3D::normalize flips (0, 0, -1, d < 0).  Pinned on the reference
  // binary (tests/test_orbslam_pin.py, plane_oplus case 37).
  double ca = __builtin_signbit(n.x) ? -1.0 : 1.0, sa = __builtin_signbit(n.y) ? -0.0 : 0.0;
  // (`!(r == 0)`, not `r > 0`: a NaN direction -- ominus_ver of two parallel normals, G2O_Plane3D.h:97-106 -- must come out as NaN,
  //  as atan2 / AngleAxis give it in the reference, not be swallowed into an identity frame)
  if (!(r == 0.0)) { const double ir = 1.0 / r; ca = n.x * ir; sa = n.y * ir; }
  double ce = 1.0, se = 0.0;
  if (!(nn == 0.0)) { const double in = 1.0 / nn; ce = r * in; se = n.z * in; }
  Mat3 R;
  R.a[0] = ca * ce; R.a[1] = -sa; R.a[2] = -(ca * se);
  R.a[3] = sa * ce; R.a[4] = ca;  R.a[5] = -(sa * se);
  R.a[6] = se;      R.a[7] = 0.0; R.a[8] = ce;
  return R;
}
PPO_HD PlaneC plane_boxplus(const PlaneC& p, const double* u) {
  double s0, c0, s, c;
  sincos_step(u[0], s0, c0);
  sincos_step(u[1], s, c);
  const Vec3 n = v3(c * c0, c * s0, s);
  const Vec3 rn = matvec(plane_frame(v3(p.nx, p.ny, p.nz)), n);
  const double dist = -p.d + u[2];
  return plane_canon(rn.x, rn.y, rn.z, -dist);
}
PPO_HD PlaneC plane_into_camera(const Pose& Tcw, const PlaneC& p) {
  const Vec3 n2 = matvec(rotmat(Tcw.q), v3(p.nx, p.ny, p.nz));
  double d2 = p.d - dot3(Tcw.t, n2);
  if (d2 < 0.0) return plane_canon(-n2.x, -n2.y, -n2.z, -d2);
  return plane_canon(n2.x, n2.y, n2.z, d2);
}
// kind 0: Plane3D::ominus (3), kind 1: ominus_ver (2), kind 2: ominus_par (2).  a = transformed estimate, m = measurement
PPO_HD void plane_residual(const PlaneC& a, const PlaneC& m, int kind, double* e) {
  const Vec3 na = v3(a.nx, a.ny, a.nz), nm = v3(m.nx, m.ny, m.nz);
  Vec3 ref = na;
  if (kind == 1) {
    Vec3 ax = cross3(na, nm);
    const double an = sqrt(dot3(ax, ax));
    ax = v3(ax.x / an, ax.y / an, ax.z / an);
    // rotate na by +90 degrees about ax (Rodrigues with the constants Eigen would compute for M_PI/2)
    const double s = 1.0, c = 6.123233995736766e-17;  // sin, cos of the double M_PI / 2
    const Vec3 sa = ax * s, ca = ax * (1.0 - c);
    Mat3 R;
    double t;
    t = ca.x * ax.y; R.a[1] = t - sa.z; R.a[3] = t + sa.z;
    t = ca.x * ax.z; R.a[2] = t + sa.y; R.a[6] = t - sa.y;
    t = ca.y * ax.z; R.a[5] = t - sa.x; R.a[7] = t + sa.x;
    R.a[0] = ca.x * ax.x + c; R.a[4] = ca.y * ax.y + c; R.a[8] = ca.z * ax.z + c;
    ref = matvec(R, na);
  } else if (kind == 2) {
    if (dot3(nm, na) < 0) ref = na * -1.0;
  }
  const Vec3 n = matTvec(plane_frame(ref), nm);
  e[0] = az_of(n);
  e[1] = el_of(n);
  if (kind == 0) e[2] = (-a.d) - (-m.d);
}

// ---- cuboids -----------------------------------------------------------------------------------------
PPO_HD Pose exp_yaw_only(const double* u) {  // src/g2o_cuboid.cc:6-36
  const Vec3 om = v3(u[0], u[1], u[2]), up = v3(u[3], u[4], u[5]);
  const double th = sqrt(dot3(om, om));
  double cz, sz;
  sincos_step(om.z, sz, cz);
  const Mat3 R = Mat3{{cz, -sz, 0, sz, cz, 0, 0, 0, 1}};
  Mat3 V = R;
  if (!(th < 0.00001)) {
    Mat3 W, W2;
    skew_sq(om, W, W2);
    const double b = (1 - cos(th)) / (th * th), c = (th - sin(th)) / (th * th * th);
    for (int i = 0; i < 9; ++i) V.a[i] = ((i == 0 || i == 4 || i == 8) ? 1.0 : 0.0) + b * W.a[i] + c * W2.a[i];
  }
  Pose P;
  P.q = canon(quat_of(R));
  P.t = matvec(V, up);
  return P;
}
PPO_HD Cub cuboid_boxplus(const Cub& c, const double* u, unsigned flags, double fsx, double fsy, double fsz) {
  Cub n;
  if (flags & 4u) {
    n.T.q = Quat{0, 0, 0, 1};
    n.T.t = c.T.t + v3(u[3], u[4], u[5]);
  } else if (flags & 1u) {
    const double u2[6] = {0, 0, u[2], u[3], u[4], u[5]};
    n.T = compose(c.T, exp_yaw_only(u2));
  } else {
    n.T = compose(c.T, exp_se3(u));
  }
  if (flags & 2u) n.T.t.y = c.T.t.y;
  if (fsx > 0) n.s = v3(fsx, fsy, fsz);
  else n.s = c.s + v3(u[6], u[7], u[8]);
  return n;
}
// corner k of the body frame: signs (x,y,z) of include/g2o_cuboid.h:202-204
PPO_HD void corner_sign(int k, double& sx, double& sy, double& sz) {
  sx = ((k & 3) == 0 || (k & 3) == 1) ? 1.0 : -1.0;
  sy = ((k & 3) == 0 || (k & 3) == 3) ? 1.0 : -1.0;
  sz = (k < 4) ? -1.0 : 1.0;
}
PPO_HD void cuboid_corner_px(const Cub& c, const Mat3& Rwo, const Mat3& Rcw, Vec3 tcw, const double* K, int k, double& u, double& v) {
  double sx, sy, sz;
  corner_sign(k, sx, sy, sz);
  Vec3 pw;
  pw.x = (Rwo.a[0] * c.s.x) * sx + (Rwo.a[1] * c.s.y) * sy + (Rwo.a[2] * c.s.z) * sz + c.T.t.x;
  pw.y = (Rwo.a[3] * c.s.x) * sx + (Rwo.a[4] * c.s.y) * sy + (Rwo.a[5] * c.s.z) * sz + c.T.t.y;
  pw.z = (Rwo.a[6] * c.s.x) * sx + (Rwo.a[7] * c.s.y) * sy + (Rwo.a[8] * c.s.z) * sz + c.T.t.z;
  const Vec3 pc = matvec(Rcw, pw) + tcw;
  const double hx = K[0] * pc.x + K[1] * pc.y + K[2] * pc.z;
  const double hy = K[3] * pc.x + K[4] * pc.y + K[5] * pc.z;
  const double hz = K[6] * pc.x + K[7] * pc.y + K[8] * pc.z;
  u = hx / hz;
  v = hy / hz;
}
// e[0..3] = projected (cx, cy, w, h) - meas
PPO_HD void bbox_residual(const Pose& Tcw, const Cub& c, const double* K, const double* meas, double* e) {
  const Mat3 Rwo = rotmat(c.T.q), Rcw = rotmat(Tcw.q);
  double ulo = 0, uhi = 0, vlo = 0, vhi = 0;
#pragma unroll 1  // (on the GPU a residual is executed once per wave: unrolled, the eight corners are eight times the instruction fetch)
  for (int k = 0; k < 8; ++k) {
    double u, v;
    cuboid_corner_px(c, Rwo, Rcw, Tcw.t, K, k, u, v);
    if (k == 0) { ulo = uhi = u; vlo = vhi = v; }
    else { ulo = fmin(ulo, u); uhi = fmax(uhi, u); vlo = fmin(vlo, v); vhi = fmax(vhi, v); }
  }
  e[0] = (uhi + ulo) / 2 - meas[0];
  e[1] = (vhi + vlo) / 2 - meas[1];
  e[2] = (uhi - ulo) - meas[2];
  e[3] = (vhi - vlo) - meas[3];
}
PPO_HD void corner_residual(const Pose& Tcw, const Cub& c, const double* K, const double* meas, double* e) {
  const Mat3 Rwo = rotmat(c.T.q), Rcw = rotmat(Tcw.q);
  for (int k = 0; k < 8; ++k) {
    double u, v;
    cuboid_corner_px(c, Rwo, Rcw, Tcw.t, K, k, u, v);
    e[2 * k] = u - meas[2 * k];
    e[2 * k + 1] = v - meas[2 * k + 1];
  }
}

// EdgeSE3Cuboid::computeError (include/g2o_cuboid.h:332-341) with cuboid::min_log_error (:83-109): the measured cuboid (camera frame,
// meas = qx qy qz qw tx ty tz sx sy sz as given) goes to the world with Tcw^-1; of the four copies of it turned by -90 / 0 / 90 / 180
// degrees about its own z axis (the half extents x, y swapped for the quarter turns) the one with the smallest 9-D error
// [log(copy^-1 * cuboid), scale - copy scale] is the residual (the FIRST minimum, as Eigen's minCoeff).
PPO_HD void cub3d_residual(const Pose& Tcw, const Cub& c, const double* meas, double* e) {
  Cub m;
  m.T = compose(inverse(Tcw), Pose{Quat{meas[0], meas[1], meas[2], meas[3]}, Vec3{meas[4], meas[5], meas[6]}});
  m.s = v3(meas[7], meas[8], meas[9]);
  const double kPi2 = 1.5707963267948966;  // M_PI / 2.0
  double best = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double yaw = (double)(i - 1) * 3.141592653589793 / 2.0;  // rotate_angles(i) * M_PI / 2.0, left to right like the reference
    const Pose turn{canon(Quat{0.0, 0.0, sin(yaw * 0.5), cos(yaw * 0.5)}), Vec3{0.0, 0.0, 0.0}};
    const Pose mp = compose(m.T, turn);
    const bool swap = (yaw == kPi2) || (yaw == -kPi2);
    const Vec3 ms = swap ? v3(m.s.y, m.s.x, m.s.z) : m.s;
    double ei[9];
    log_se3(compose(inverse(mp), c.T), ei);
    ei[6] = c.s.x - ms.x; ei[7] = c.s.y - ms.y; ei[8] = c.s.z - ms.z;
    double n2 = 0.0;
    for (int k = 0; k < 9; ++k) n2 += ei[k] * ei[k];
    const double n = sqrt(n2);
    if (i == 0 || n < best) {
      best = n;
      for (int k = 0; k < 9; ++k) e[k] = ei[k];
    }
  }
}
// ---- analytic Jacobians (options.jacobian_mode = PPO_JAC_ANALYTIC; SURVEY.md section 7 H1: "numeric first, analytic later") ------------
// The reference differentiates these edges numerically (base_binary_edge.hpp:217-320, delta = 1e-9).  The functions below are the exact
// derivatives of the same composite maps -- vertex box-plus, transform, residual -- at a zero update; they agree with the central
// differences to the 1e-7 relative round-off those carry (tests/test_analytic_jacobians.py, csrc/hostcheck.cpp).
//
// EdgePlane / EdgeVerticalPlane / EdgeParallelPlane (include/G2O_Plane3D.h:176-310).  With n~ = sigma R n the camera-frame normal
// (sigma = -1 when the transformed offset is negative, :136-137) and c~ its offset, a zero-centred update moves them by
//   plane  (azimuth a, elevation b, distance dd; :74-87):  dn~ = sigma R (a Yw + b Zw),  dc~ = sigma (-dd - t . R (a Yw + b Zw)),
//          Yw, Zw = second and third column of rotation(n) (:66-72);
//   pose   (omega, upsilon; exp(delta) T):                 dn~ = sigma (omega x R n),     dc~ = -sigma (R n) . upsilon.
// The angular residual is (azimuth, elevation) of F(a)' n_m with F = rotation(a) and a = n~ (ominus, :89-95), +-n~ (ominus_par, :108-117)
// or the unit projection of n_m onto the plane perpendicular to n~ (ominus_ver, :97-106: a quarter turn of n~ about n~ x n_m).  Its
// derivative with respect to the axis is a pair of tangent vectors g0, g1 (below), chained through da = dn~ (ominus), +-dn~ (par),
// (dp - a (a . dp)) / |p| with dp = -(dn~ . n_m) n~ - (n~ . n_m) dn~ (ver).  J is row-major [3][9]: columns 0-2 plane, 3-8 pose;
// e as plane_residual.  (The d >= 0 flips of :120-125, :136-137 are constant in a neighbourhood, as the central difference sees them.)
PPO_HD void plane_edge_analytic(const Pose& Tcw, const PlaneC& p, const PlaneC& m, int kind, double* e, double* J) {
  const Mat3 R = rotmat(Tcw.q);
  const Vec3 nw = v3(p.nx, p.ny, p.nz), u = v3(m.nx, m.ny, m.nz);
  const Vec3 nc = matvec(R, nw);
  const double d2 = p.d - dot3(Tcw.t, nc);
  const double sg = d2 < 0.0 ? -1.0 : 1.0;
  const PlaneC loc = plane_canon(sg * nc.x, sg * nc.y, sg * nc.z, sg * d2);
  const Vec3 nt = v3(loc.nx, loc.ny, loc.nz);
  // axis of the residual frame and the linear map dn~ -> (de0, de1) as two gradient vectors G0, G1
  Vec3 a = nt;
  double s2 = 1.0, gdot = 0.0, ipn = 1.0;
  if (kind == 1) {
    gdot = dot3(nt, u);
    const Vec3 pr = u - nt * gdot;
    ipn = 1.0 / sqrt(dot3(pr, pr));
    a = pr * ipn;
  } else if (kind == 2) {
    if (dot3(u, nt) < 0) { s2 = -1.0; a = nt * -1.0; }
  }
  const Mat3 F = plane_frame(a);
  const Vec3 yh = v3(F.a[1], F.a[4], F.a[7]), zh = v3(F.a[2], F.a[5], F.a[8]);
  const double ce = F.a[8], se = F.a[6];  // cos / sin of the axis' elevation
  const Vec3 mm = matTvec(F, u);
  e[0] = az_of(mm);
  e[1] = el_of(mm);
  if (kind == 0) e[2] = (-loc.d) - (-m.d);
  const double rho2 = mm.x * mm.x + mm.y * mm.y, irho = 1.0 / sqrt(rho2);
  const double A0 = (-mm.x * (ce * mm.x - se * mm.z) - ce * mm.y * mm.y) / rho2, B0 = -mm.y * mm.z / rho2;
  const double A1 = -se * mm.y * irho, B1 = -mm.x * irho;
  const double ice = 1.0 / ce;
  Vec3 g0 = yh * (A0 * ice) + zh * B0, g1 = yh * (A1 * ice) + zh * B1;
  if (kind == 1) {  // g . da = (g . dp) / |p| (g is perpendicular to a): fold the projection into the gradients
    g0 = (u * dot3(g0, nt) + g0 * gdot) * -ipn;
    g1 = (u * dot3(g1, nt) + g1 * gdot) * -ipn;
  } else if (kind == 2) {
    g0 = g0 * s2;
    g1 = g1 * s2;
  }
  const Mat3 Fw = plane_frame(nw);
  const Vec3 ry = matvec(R, v3(Fw.a[1], Fw.a[4], Fw.a[7])) * sg, rz = matvec(R, v3(Fw.a[2], Fw.a[5], Fw.a[8])) * sg;  // dn~ of the plane's two angles
  const Vec3 snc = nc * sg;
  const Vec3 w0 = cross3(snc, g0), w1 = cross3(snc, g1);  // g . (e_k x s nc) = e_k . (s nc x g)
  J[0] = dot3(g0, ry); J[1] = dot3(g0, rz); J[2] = 0.0; J[3] = w0.x; J[4] = w0.y; J[5] = w0.z; J[6] = 0.0; J[7] = 0.0; J[8] = 0.0;
  J[9] = dot3(g1, ry); J[10] = dot3(g1, rz); J[11] = 0.0; J[12] = w1.x; J[13] = w1.y; J[14] = w1.z; J[15] = 0.0; J[16] = 0.0; J[17] = 0.0;
  if (kind == 0) {  // e2 = -c~ + c_m: -dc~
    J[18] = dot3(Tcw.t, ry); J[19] = dot3(Tcw.t, rz); J[20] = sg; J[21] = 0.0; J[22] = 0.0; J[23] = 0.0; J[24] = snc.x; J[25] = snc.y; J[26] = snc.z;
  } else {
    for (int k = 18; k < 27; ++k) J[k] = 0.0;
  }
}

// One projected cuboid corner (include/g2o_cuboid.h:198-225) with its derivatives: Ju / Jv [15] = d(u, v) / d(pose omega, upsilon |
// cuboid update 9).  Pose: exp(delta) Tcw moves the camera-frame corner by omega x pc + upsilon.  Cuboid (src/g2o_cuboid.cc:39-67):
// fix-roll/pitch (flags bit 0): pose . (Rz(yaw), V upsilon) -> the corner moves by Rwo (z x b) dyaw + Rwo upsilon, columns 0, 1 are
// zero; otherwise pose . exp(delta): Rwo (omega x b) + Rwo upsilon; fix-height (bit 1) restores the WORLD y of the translation:
// the upsilon columns lose their y row; scale: Rwo e_j sign_j unless a fixed scale is set.  (b = scale * corner signs.)
// whether_fixrotation (bit 2) resets the rotation in its box-plus (a zero update is not the identity there): not supported, the
// library refuses the analytic mode for such a cuboid.
PPO_HD void cuboid_corner_analytic(const Cub& c, const Mat3& Rwo, const Mat3& Rcw, Vec3 tcw, const double* K, unsigned flags, bool scale_fixed, int k, double& u,
                                   double& v, double* Ju, double* Jv) {
  double sx, sy, sz;
  corner_sign(k, sx, sy, sz);
  const Vec3 b = v3(c.s.x * sx, c.s.y * sy, c.s.z * sz);
  Vec3 pw;
  pw.x = (Rwo.a[0] * c.s.x) * sx + (Rwo.a[1] * c.s.y) * sy + (Rwo.a[2] * c.s.z) * sz + c.T.t.x;
  pw.y = (Rwo.a[3] * c.s.x) * sx + (Rwo.a[4] * c.s.y) * sy + (Rwo.a[5] * c.s.z) * sz + c.T.t.y;
  pw.z = (Rwo.a[6] * c.s.x) * sx + (Rwo.a[7] * c.s.y) * sy + (Rwo.a[8] * c.s.z) * sz + c.T.t.z;
  const Vec3 pc = matvec(Rcw, pw) + tcw;
  const double hx = K[0] * pc.x + K[1] * pc.y + K[2] * pc.z;
  const double hy = K[3] * pc.x + K[4] * pc.y + K[5] * pc.z;
  const double hz = K[6] * pc.x + K[7] * pc.y + K[8] * pc.z;
  const double ihz = 1.0 / hz;
  u = hx / hz;
  v = hy / hz;
  const Vec3 gu = v3((K[0] - u * K[6]) * ihz, (K[1] - u * K[7]) * ihz, (K[2] - u * K[8]) * ihz);  // du / dpc
  const Vec3 gv = v3((K[3] - v * K[6]) * ihz, (K[4] - v * K[7]) * ihz, (K[5] - v * K[8]) * ihz);
  const Vec3 wu = cross3(pc, gu), wv = cross3(pc, gv);  // g . (e_j x pc) = e_j . (pc x g)
  Ju[0] = wu.x; Ju[1] = wu.y; Ju[2] = wu.z; Ju[3] = gu.x; Ju[4] = gu.y; Ju[5] = gu.z;
  Jv[0] = wv.x; Jv[1] = wv.y; Jv[2] = wv.z; Jv[3] = gv.x; Jv[4] = gv.y; Jv[5] = gv.z;
  const Vec3 hu = matTvec(Rcw, gu), hv = matTvec(Rcw, gv);  // du / dpw
  const Vec3 bu = matTvec(Rwo, hu), bv = matTvec(Rwo, hv);  // du / d(body-frame displacement)
  // rotation columns: omega x b in the body frame
  const Vec3 ru = cross3(b, bu), rv = cross3(b, bv);      // bu . (e_j x b) = e_j . (b x bu)
  if (flags & 1u) { Ju[6] = 0.0; Ju[7] = 0.0; Jv[6] = 0.0; Jv[7] = 0.0; }
  else { Ju[6] = ru.x; Ju[7] = ru.y; Jv[6] = rv.x; Jv[7] = rv.y; }
  Ju[8] = ru.z; Jv[8] = rv.z;
  // translation columns: Rwo e_j in the world, y row dropped when the height is fixed
  const bool fh = (flags & 2u) != 0;
  const Vec3 hu2 = fh ? v3(hu.x, 0.0, hu.z) : hu, hv2 = fh ? v3(hv.x, 0.0, hv.z) : hv;
  const Vec3 tu = matTvec(Rwo, hu2), tv = matTvec(Rwo, hv2);
  Ju[9] = tu.x; Ju[10] = tu.y; Ju[11] = tu.z; Jv[9] = tv.x; Jv[10] = tv.y; Jv[11] = tv.z;
  if (scale_fixed) { Ju[12] = Ju[13] = Ju[14] = 0.0; Jv[12] = Jv[13] = Jv[14] = 0.0; }
  else { Ju[12] = bu.x * sx; Ju[13] = bu.y * sy; Ju[14] = bu.z * sz; Jv[12] = bv.x * sx; Jv[13] = bv.y * sy; Jv[14] = bv.z * sz; }
}
// bbox edge (EdgeSE3CuboidProj, src/g2o_cuboid.cc:70-91): e[4] and J[4][15]; the hull's derivative is that of the corners attaining it
PPO_HD void bbox_edge_analytic(const Pose& Tcw, const Cub& c, unsigned flags, bool scale_fixed, const double* K, const double* meas, double* e, double* J) {
  const Mat3 Rwo = rotmat(c.T.q), Rcw = rotmat(Tcw.q);
  double ulo = 0, uhi = 0, vlo = 0, vhi = 0;
  double Julo[15], Juhi[15], Jvlo[15], Jvhi[15];
  for (int k = 0; k < 8; ++k) {
    double u, v, Ju[15], Jv[15];
    cuboid_corner_analytic(c, Rwo, Rcw, Tcw.t, K, flags, scale_fixed, k, u, v, Ju, Jv);
    if (k == 0 || u < ulo) { ulo = u; for (int q = 0; q < 15; ++q) Julo[q] = Ju[q]; }
    if (k == 0 || u > uhi) { uhi = u; for (int q = 0; q < 15; ++q) Juhi[q] = Ju[q]; }
    if (k == 0 || v < vlo) { vlo = v; for (int q = 0; q < 15; ++q) Jvlo[q] = Jv[q]; }
    if (k == 0 || v > vhi) { vhi = v; for (int q = 0; q < 15; ++q) Jvhi[q] = Jv[q]; }
  }
  e[0] = (uhi + ulo) / 2 - meas[0];
  e[1] = (vhi + vlo) / 2 - meas[1];
  e[2] = (uhi - ulo) - meas[2];
  e[3] = (vhi - vlo) - meas[3];
  for (int q = 0; q < 15; ++q) {
    J[q] = (Juhi[q] + Julo[q]) / 2;
    J[15 + q] = (Jvhi[q] + Jvlo[q]) / 2;
    J[30 + q] = Juhi[q] - Julo[q];
    J[45 + q] = Jvhi[q] - Jvlo[q];
  }
}
// corner edge (EdgeSE3CuboidCornerProj, src/g2o_cuboid.cc:103-120): e[16] and J[16][15]
PPO_HD void corner_edge_analytic(const Pose& Tcw, const Cub& c, unsigned flags, bool scale_fixed, const double* K, const double* meas, double* e, double* J) {
  const Mat3 Rwo = rotmat(c.T.q), Rcw = rotmat(Tcw.q);
  for (int k = 0; k < 8; ++k) {
    double u, v;
    cuboid_corner_analytic(c, Rwo, Rcw, Tcw.t, K, flags, scale_fixed, k, u, v, &J[(2 * k) * 15], &J[(2 * k + 1) * 15]);
    e[2 * k] = u - meas[2 * k];
    e[2 * k + 1] = v - meas[2 * k + 1];
  }
}

// EdgePointCuboidOnlyObject::computeError, src/g2o_cuboid.cc:132-160, in two steps: the sum over the points first + k stride, k = 0, 1, ...
// (the whole list with first = 0, stride = 1; one of several interleaved partial sums otherwise), then the mean and the prior terms
PPO_HD void ptcub_partial(const Cub& c, const double* pts, int npts, int first, int stride, double margin, double* acc) {
  const Pose inv = inverse(c.T);
  acc[0] = acc[1] = acc[2] = 0.0;
  const double s[3] = {c.s.x, c.s.y, c.s.z};
  // (points are taken six at a time, all loads first: one point per trip is one dependent memory round trip per point on the GPU)
  constexpr int G = 6;
  for (int p0 = first; p0 < npts; p0 += G * stride) {
    double xyz[3 * G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int pi = (p0 + q * stride < npts) ? p0 + q * stride : npts - 1;  // unconditional loads (clamped index)
      xyz[3 * q] = pts[3 * pi]; xyz[3 * q + 1] = pts[3 * pi + 1]; xyz[3 * q + 2] = pts[3 * pi + 2];
    }
#pragma unroll
    for (int q = 0; q < G; ++q) {
      if (p0 + q * stride < npts) {
        const Vec3 l = apply(inv, v3(xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2]));
        const double a[3] = {fabs(l.x), fabs(l.y), fabs(l.z)};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          double pe;
          if (a[i] < s[i]) pe = 0;
          else if (a[i] < (margin + 1) * s[i]) pe = a[i] - s[i];
          else pe = margin * s[i];
          acc[i] += fabs(pe);
        }
      }
    }
  }
}
PPO_HD void ptcub_finish(const Cub& c, const double* sum, int npts, const double* prior_half, double* e) {
  const double s[3] = {c.s.x, c.s.y, c.s.z};
  for (int i = 0; i < 3; ++i) {
    double acc = sum[i];
    if (npts > 0) acc = acc / (double)npts;
    acc = acc / s[i];
    double pw = 0.2, pr = s[i];
    if (prior_half[0] > 0) { pw = 50.0; pr = fabs((s[i] - prior_half[i]) / prior_half[i]); }
    e[i] = 1.0 * acc + pw * pr;
  }
}
PPO_HD void ptcub_residual(const Cub& c, const double* pts, int npts, double margin, const double* prior_half, double* e) {
  double acc[3];
  ptcub_partial(c, pts, npts, 0, 1, margin, acc);
  ptcub_finish(c, acc, npts, prior_half, e);
}

// ---- reprojection edges ------------------------------------------------------------------------------
// EdgeSE3ProjectXYZ: residual (2) and Jacobians wrt point (2x3, row-major) and pose (2x6)
PPO_HD void mono_eval(const Mat3& R, Vec3 t, Vec3 X, double fx, double fy, double cx, double cy, double ou, double ov,
                      double* e, double* Jl, double* Jp, double& zc) {
  const Vec3 pc = matvec(R, X) + t;
  zc = pc.z;
  // Three divisions instead of the eight distinct quotients of types_six_dof_expmap.cpp:65-103 (an FP64 division is ~15
  // instructions on the device and this routine runs three times per edge and LM iteration).  The residual and every Jacobian
  // entry that is a single quotient in the reference keep their bits (x/z, y/z, 1/z); x y / z^2, x^2 / z^2, y^2 / z^2, x / z^2
  // and y / z^2 become products of those quotients (an ulp or two apart; the parity bar on J is 1e-13 relative).
  const double a = pc.x / pc.z, b = pc.y / pc.z, iz = 1. / pc.z;
  e[0] = ou - (a * fx + cx);
  e[1] = ov - (b * fy + cy);
  const double s = -iz;
  const double r0[3] = {s * fx, s * 0.0, s * (-a * fx)};
  const double r1[3] = {s * 0.0, s * fy, s * (-b * fy)};
  for (int c = 0; c < 3; ++c) {
    Jl[c] = r0[0] * R.a[c] + r0[1] * R.a[3 + c] + r0[2] * R.a[6 + c];
    Jl[3 + c] = r1[0] * R.a[c] + r1[1] * R.a[3 + c] + r1[2] * R.a[6 + c];
  }
  Jp[0] = a * b * fx;        Jp[1] = -(1 + a * a) * fx; Jp[2] = b * fx;
  Jp[3] = -iz * fx;          Jp[4] = 0;                 Jp[5] = a * iz * fx;
  Jp[6] = (1 + b * b) * fy;  Jp[7] = -(a * b) * fy;     Jp[8] = -a * fy;
  Jp[9] = 0;                 Jp[10] = -iz * fy;         Jp[11] = b * iz * fy;
}
PPO_HD void mono_residual(const Mat3& R, Vec3 t, Vec3 X, double fx, double fy, double cx, double cy, double ou, double ov, double* e, double& zc) {
  const Vec3 pc = matvec(R, X) + t;
  zc = pc.z;
  e[0] = ou - (pc.x / pc.z * fx + cx);
  e[1] = ov - (pc.y / pc.z * fy + cy);
}
// EdgeStereoSE3ProjectXYZ (1/z and bf pass through float, types_six_dof_expmap.cpp:182-189)
PPO_HD void stereo_residual(const Mat3& R, Vec3 t, Vec3 X, double fx, double fy, double cx, double cy, double bfd,
                            double ou, double ov, double our, double* e, double& zc) {
  const Vec3 pc = matvec(R, X) + t;
  zc = pc.z;
  const float bf = (float)bfd;
  const float invz = (float)(1.0 / pc.z);  // 1.0f / double = double division, then rounded to float
  const double r0 = pc.x * invz * fx + cx;
  const double r1 = pc.y * invz * fy + cy;
  const double r2 = r0 - (double)(bf * invz);
  e[0] = ou - r0; e[1] = ov - r1; e[2] = our - r2;
}
PPO_HD void stereo_eval(const Mat3& R, Vec3 t, Vec3 X, double fx, double fy, double cx, double cy, double bf,
                        double ou, double ov, double our, double* e, double* Jl, double* Jp, double& zc) {
  stereo_residual(R, t, X, fx, fy, cx, cy, bf, ou, ov, our, e, zc);
  const Vec3 pc = matvec(R, X) + t;
  const double x = pc.x, y = pc.y, z = pc.z, z2 = z * z;
  for (int c = 0; c < 3; ++c) {
    Jl[c] = -fx * R.a[c] / z + fx * x * R.a[6 + c] / z2;
    Jl[3 + c] = -fy * R.a[3 + c] / z + fy * y * R.a[6 + c] / z2;
    Jl[6 + c] = Jl[c] - bf * R.a[6 + c] / z2;
  }
  Jp[0] = x * y / z2 * fx; Jp[1] = -(1 + (x * x / z2)) * fx; Jp[2] = y / z * fx;
  Jp[3] = -1. / z * fx;    Jp[4] = 0;                        Jp[5] = x / z2 * fx;
  Jp[6] = (1 + y * y / z2) * fy; Jp[7] = -x * y / z2 * fy;   Jp[8] = -x / z * fy;
  Jp[9] = 0;               Jp[10] = -1. / z * fy;            Jp[11] = y / z2 * fy;
  Jp[12] = Jp[0] - bf * y / z2; Jp[13] = Jp[1] + bf * x / z2; Jp[14] = Jp[2];
  Jp[15] = Jp[3];          Jp[16] = 0;                       Jp[17] = Jp[5] - bf / z2;
}

// Quadratic-form pieces of one reprojection edge with a compile-time residual dimension (BaseBinaryEdge::constructQuadraticForm,
// G2O/core/base_binary_edge.hpp:55-120, with Omega = info * I and the Huber weight rho1 folded into w and g).
//   landmark side: D += Jl' w Jl (packed upper), bl += Jl' g          (g = -rho1 Omega e)
//   coupling:      W = Jp' w Jl (6x3)
//   pose side:     A += Jp' w Jp (packed lower, 21), bp += Jp' g (6)
template <int DIM>
PPO_HD void acc_landmark(const double* Jl, double w, const double* gr, double* D, double* bl) {
#pragma unroll
  for (int r = 0; r < DIM; ++r) {
    const double* j = &Jl[3 * r];
    D[0] += j[0] * w * j[0]; D[1] += j[0] * w * j[1]; D[2] += j[0] * w * j[2];
    D[3] += j[1] * w * j[1]; D[4] += j[1] * w * j[2]; D[5] += j[2] * w * j[2];
    bl[0] += j[0] * gr[r]; bl[1] += j[1] * gr[r]; bl[2] += j[2] * gr[r];
  }
}
template <int DIM>
PPO_HD void form_W(const double* Jp, const double* Jl, double w, double* o) {
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double h = 0;
#pragma unroll
      for (int r = 0; r < DIM; ++r) h += Jp[6 * r + a] * w * Jl[3 * r + c];
      o[a * 3 + c] = h;
    }
}
template <int DIM>
PPO_HD void acc_pose(const double* Jp, double w, const double* gr, double* A) {
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int r = 0; r < DIM; ++r) A[21 + a] += Jp[6 * r + a] * gr[r];
#pragma unroll
    for (int c = 0; c <= a; ++c) {
      double h = 0;
#pragma unroll
      for (int r = 0; r < DIM; ++r) h += Jp[6 * r + a] * w * Jp[6 * r + c];
      A[a * (a + 1) / 2 + c] += h;
    }
  }
}

struct alignas(32) FactorRec {
  double a, b, iz, ws;  // x/z, y/z, 1/z of the point in the camera frame; rho1 * info, negated for a stereo edge, 0 for an inactive one
};
// Factor record of a reprojection edge (32 bytes instead of the 144-byte coupling block W = Jp' w Jl it stands for).  With
// a = x/z, b = y/z, i = 1/z of the point in the camera frame, the Jacobians of EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ
// (types_six_dof_expmap.cpp:65-103, :192-246) are polynomials in (a, b, i) and the intrinsics, and Jl = Jc R with R the
// keyframe's rotation, so W = M R with the 6x3 matrix M = Jp' w Jc.  The pose side of the linearisation writes (a, b, i, +-w)
// per slot; the Schur pair kernel (ppo_kernels.hpp, schur_pairs_body) forms M_j (R_j (D + lambda I)^-1 R_i') M_i' with R_j, R_i uniform over its wave.
PPO_HD FactorRec factor_record(const Mat3& R, Vec3 t, Vec3 X, double ws) {
  const Vec3 pc = matvec(R, X) + t;
  const double iz = 1.0 / pc.z;
  return FactorRec{pc.x * iz, pc.y * iz, iz, ws};
}
// M (row-major 6x3) of a record; `stereo_rows`: wave-uniform, some lane holds a stereo record (third residual row, bf)
PPO_HD void factor_M(const FactorRec f, double fx, double fy, double bf, bool stereo_rows, double* M) {
  const double a = f.a, b = f.b, iz = f.iz, w = fabs(f.ws);
  const double afx = a * fx, bfy = b * fy, izfx = iz * fx, izfy = iz * fy;
  const double p0[6] = {afx * b, -__builtin_fma(afx, a, fx), b * fx, -izfx, 0.0, afx * iz};   // d e_u / d pose
  const double p1[6] = {__builtin_fma(bfy, b, fy), -(bfy * a), -(a * fy), 0.0, -izfy, bfy * iz};  // d e_v / d pose
  const double c00 = -(izfx * w), c02 = izfx * a * w, c11 = -(izfy * w), c12 = izfy * b * w;  // w Jc (camera-frame point)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    M[3 * k] = p0[k] * c00;
    M[3 * k + 1] = p1[k] * c11;
    M[3 * k + 2] = __builtin_fma(p0[k], c02, p1[k] * c12);
  }
  if (stereo_rows) {
    const double w2 = f.ws < 0.0 ? w : 0.0;
    const double bi = bf * iz;
    const double p2[6] = {__builtin_fma(-bi, b, p0[0]), __builtin_fma(bi, a, p0[1]), p0[2], p0[3], 0.0, __builtin_fma(-bi, iz, p0[5])};
    const double c20 = -(izfx * w2), c22 = (izfx * a - bi * iz) * w2;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      M[3 * k] = __builtin_fma(p2[k], c20, M[3 * k]);
      M[3 * k + 2] = __builtin_fma(p2[k], c22, M[3 * k + 2]);
    }
  }
}

// Huber weight: rho0 (robustified chi2) and rho1 (weight); delta <= 0 disables the kernel
PPO_HD void huber(double chi2, double delta, double& rho0, double& rho1) {
  rho0 = chi2;
  rho1 = 1.0;
  if (delta > 0) {
    const double dsqr = (double)(float)(delta * delta);  // g2o keeps delta^2 in a float member (robust_kernel_impl.h:84)
    if (!(chi2 <= dsqr)) {
      const double sq = sqrt(chi2);
      rho0 = 2 * sq * delta - dsqr;
      rho1 = delta / sq;
    }
  }
}

// inverse of a symmetric 3x3 given as (d00,d01,d02,d11,d12,d22) by cofactors; returns the same packing
PPO_HD void sym3_inverse(const double* D, double* I) {
  const double c00 = D[3] * D[5] - D[4] * D[4];
  const double c01 = D[2] * D[4] - D[1] * D[5];
  const double c02 = D[1] * D[4] - D[2] * D[3];
  const double det = D[0] * c00 + D[1] * c01 + D[2] * c02;
  const double id = 1.0 / det;
  I[0] = c00 * id;
  I[1] = c01 * id;
  I[2] = c02 * id;
  I[3] = (D[0] * D[5] - D[2] * D[2]) * id;
  I[4] = (D[1] * D[2] - D[0] * D[4]) * id;
  I[5] = (D[0] * D[3] - D[1] * D[1]) * id;
}

}  // namespace ppo
