// hostcheck.cpp -- TEST AID: the formulas of ppo_math.hpp (the header the HIP kernels use) compiled for the host so that
// the CPU test-suite can check them against the golden vectors of the reference's prebuilt libg2o.so and against the
// oracle.  Not linked into libppo_ba.so and never used by the product path.
#include "ppo_math.hpp"
#include "ppo_pose_math.hpp"
#include "ppo_lm.hpp"

using namespace ppo;

extern "C" {
void hc_se3_exp(const double* u, double* qt) {
  const Pose P = exp_se3(u);
  qt[0] = P.q.x; qt[1] = P.q.y; qt[2] = P.q.z; qt[3] = P.q.w; qt[4] = P.t.x; qt[5] = P.t.y; qt[6] = P.t.z;
}
void hc_pose_oplus(const double* qt, const double* u, double* out) {
  const Pose T{Quat{qt[0], qt[1], qt[2], qt[3]}, Vec3{qt[4], qt[5], qt[6]}};
  const Pose P = pose_boxplus(T, u);
  out[0] = P.q.x; out[1] = P.q.y; out[2] = P.q.z; out[3] = P.q.w; out[4] = P.t.x; out[5] = P.t.y; out[6] = P.t.z;
}
void hc_huber(double e, double delta, double* rho01) { huber(e, delta, rho01[0], rho01[1]); }
void hc_mono(const double* qt, const double* X, const double* cam, const double* obs, double* e, double* Jl, double* Jp) {
  const Mat3 R = rotmat(Quat{qt[0], qt[1], qt[2], qt[3]});
  double z;
  mono_eval(R, Vec3{qt[4], qt[5], qt[6]}, Vec3{X[0], X[1], X[2]}, cam[0], cam[1], cam[2], cam[3], obs[0], obs[1], e, Jl, Jp, z);
}
void hc_stereo(const double* qt, const double* X, const double* cam, const double* obs, double* e, double* Jl, double* Jp) {
  const Mat3 R = rotmat(Quat{qt[0], qt[1], qt[2], qt[3]});
  double z;
  stereo_eval(R, Vec3{qt[4], qt[5], qt[6]}, Vec3{X[0], X[1], X[2]}, cam[0], cam[1], cam[2], cam[3], cam[4], obs[0], obs[1], obs[2], e, Jl, Jp, z);
}
void hc_plane_oplus(const double* c, const double* u, double* out) {
  const PlaneC p = plane_boxplus(plane_canon(c[0], c[1], c[2], c[3]), u);
  out[0] = p.nx; out[1] = p.ny; out[2] = p.nz; out[3] = p.d;
}
void hc_plane_residual(const double* qt, const double* plane, const double* meas, int kind, double* e) {
  const Pose T{Quat{qt[0], qt[1], qt[2], qt[3]}, Vec3{qt[4], qt[5], qt[6]}};
  const PlaneC local = plane_into_camera(T, plane_canon(plane[0], plane[1], plane[2], plane[3]));
  plane_residual(local, plane_canon(meas[0], meas[1], meas[2], meas[3]), kind, e);
}
void hc_cuboid_oplus(const double* c10, const double* u, unsigned flags, const double* fs, double* out10) {
  const Cub c{Pose{Quat{c10[0], c10[1], c10[2], c10[3]}, Vec3{c10[4], c10[5], c10[6]}}, Vec3{c10[7], c10[8], c10[9]}};
  const Cub n = cuboid_boxplus(c, u, flags, fs ? fs[0] : 0.0, fs ? fs[1] : 0.0, fs ? fs[2] : 0.0);
  out10[0] = n.T.q.x; out10[1] = n.T.q.y; out10[2] = n.T.q.z; out10[3] = n.T.q.w; out10[4] = n.T.t.x; out10[5] = n.T.t.y; out10[6] = n.T.t.z;
  out10[7] = n.s.x; out10[8] = n.s.y; out10[9] = n.s.z;
}
void hc_bbox(const double* pose_qt, const double* c10, const double* K, const double* meas, double* e) {
  const Pose T{Quat{pose_qt[0], pose_qt[1], pose_qt[2], pose_qt[3]}, Vec3{pose_qt[4], pose_qt[5], pose_qt[6]}};
  const Cub c{Pose{Quat{c10[0], c10[1], c10[2], c10[3]}, Vec3{c10[4], c10[5], c10[6]}}, Vec3{c10[7], c10[8], c10[9]}};
  bbox_residual(T, c, K, meas, e);
}
void hc_corner(const double* pose_qt, const double* c10, const double* K, const double* meas, double* e) {
  const Pose T{Quat{pose_qt[0], pose_qt[1], pose_qt[2], pose_qt[3]}, Vec3{pose_qt[4], pose_qt[5], pose_qt[6]}};
  const Cub c{Pose{Quat{c10[0], c10[1], c10[2], c10[3]}, Vec3{c10[4], c10[5], c10[6]}}, Vec3{c10[7], c10[8], c10[9]}};
  corner_residual(T, c, K, meas, e);
}
double hc_cube(double x) { return cube_rounded_once(x); }
// the device's LM decision (ppo_lm.hpp).  ctl: lambda chi_cur ini_chi ni cubpl lm_upper lm_lower nbad qmax iter epoch max_trials (doubles), updated in place
void hc_lm_decide(double* ctl, double chi2_sum, double scale_raw, int info, double* rec) {
  LmCtl c{ctl[0], ctl[1], ctl[2], ctl[3], ctl[4], ctl[5], ctl[6], (int)ctl[7], (int)ctl[8], (int)ctl[9], (int)ctl[10], (int)ctl[11], 0};
  lm_decide(c, chi2_sum, scale_raw, info, rec);
  ctl[0] = c.lambda; ctl[1] = c.chi_cur; ctl[2] = c.ini_chi; ctl[3] = c.ni; ctl[7] = c.nbad; ctl[8] = c.qmax; ctl[9] = c.iter; ctl[10] = c.epoch;
}
void hc_cub3d(const double* pose_qt, const double* c10, const double* meas10, double* e) {
  const Pose T{Quat{pose_qt[0], pose_qt[1], pose_qt[2], pose_qt[3]}, Vec3{pose_qt[4], pose_qt[5], pose_qt[6]}};
  const Cub c{Pose{Quat{c10[0], c10[1], c10[2], c10[3]}, Vec3{c10[4], c10[5], c10[6]}}, Vec3{c10[7], c10[8], c10[9]}};
  cub3d_residual(T, c, meas10, e);
}
void hc_ptcub(const double* c10, const double* pts, int npts, double margin, const double* prior, double* e) {
  const Cub c{Pose{Quat{c10[0], c10[1], c10[2], c10[3]}, Vec3{c10[4], c10[5], c10[6]}}, Vec3{c10[7], c10[8], c10[9]}};
  ptcub_residual(c, pts, npts, margin, prior, e);
}
// W = Jp' w Jl (6x3, row-major) of a reprojection edge rebuilt from its 32-byte factor record: M(record, intrinsics) * R
void hc_factor_W(const double* qt, const double* X, const double* cam, double w, int stereo, double* W) {
  const Mat3 R = rotmat(Quat{qt[0], qt[1], qt[2], qt[3]});
  const FactorRec f = factor_record(R, Vec3{qt[4], qt[5], qt[6]}, Vec3{X[0], X[1], X[2]}, stereo ? -w : w);
  double M[18];
  factor_M(f, cam[0], cam[1], cam[4], stereo != 0, M);
  for (int a = 0; a < 6; ++a)
    for (int c = 0; c < 3; ++c) W[3 * a + c] = M[3 * a] * R.a[c] + M[3 * a + 1] * R.a[3 + c] + M[3 * a + 2] * R.a[6 + c];
}
// One reprojection edge through exactly the per-edge arithmetic of the linearisation kernels (lin_points_body / lin_poses_body of
// ppo_kernels.hpp): residual + Jacobians, chi2, Huber, the landmark block D (packed upper 00 01 02 11 12 22) and b_l, the pose
// block A (packed lower 21, then b_p 6) and the coupling block W (6x3 row-major) rebuilt from the 32-byte factor record as the
// Schur pair kernel does.  For the constructQuadraticForm vectors of the reference's prebuilt libg2o.so.
void hc_quadform(const double* qt, const double* X, const double* cam, const double* obs, int stereo, double info, double hub,
                 double* e, double* D, double* bl, double* A, double* W) {
  const Mat3 R = rotmat(Quat{qt[0], qt[1], qt[2], qt[3]});
  const Vec3 t{qt[4], qt[5], qt[6]}, P{X[0], X[1], X[2]};
  for (int k = 0; k < 6; ++k) D[k] = 0;
  for (int k = 0; k < 3; ++k) bl[k] = 0;
  for (int k = 0; k < 27; ++k) A[k] = 0;
  double Jl[9], Jp[18], zc, rho0, rho1, w;
  if (!stereo) {
    mono_eval(R, t, P, cam[0], cam[1], cam[2], cam[3], obs[0], obs[1], e, Jl, Jp, zc);
    const double chi2 = e[0] * (info * e[0]) + e[1] * (info * e[1]);
    huber(chi2, hub, rho0, rho1);
    w = rho1 * info;
    const double gr[2] = {-(info * e[0]) * rho1, -(info * e[1]) * rho1};
    acc_landmark<2>(Jl, w, gr, D, bl);
    acc_pose<2>(Jp, w, gr, A);
  } else {
    stereo_eval(R, t, P, cam[0], cam[1], cam[2], cam[3], cam[4], obs[0], obs[1], obs[2], e, Jl, Jp, zc);
    double chi2 = 0;
    for (int r = 0; r < 3; ++r) chi2 += e[r] * (info * e[r]);
    huber(chi2, hub, rho0, rho1);
    w = rho1 * info;
    const double gr[3] = {-(info * e[0]) * rho1, -(info * e[1]) * rho1, -(info * e[2]) * rho1};
    acc_landmark<3>(Jl, w, gr, D, bl);
    acc_pose<3>(Jp, w, gr, A);
  }
  const FactorRec f = factor_record(R, t, P, stereo ? -w : w);
  double M[18];
  factor_M(f, cam[0], cam[1], cam[4], stereo != 0, M);
  for (int a = 0; a < 6; ++a)
    for (int c = 0; c < 3; ++c) W[3 * a + c] = M[3 * a] * R.a[c] + M[3 * a + 1] * R.a[3 + c] + M[3 * a + 2] * R.a[6 + c];
}
// One OnlyPose observation through exactly the per-observation arithmetic of k_pose_only (ppo_pose_math.hpp): residual, chi2, and
// its contribution acc[28] (21 packed lower H | 6 b | rho0) to the 6x6 normal equations.  obs[2] < 0 marks a monocular observation.
double hc_pose_obs(const double* qt, const double* cam, const double* xw, const double* obs, double invs, double huber_delta, double* e, double* acc) {
  const Pose T{Quat{qt[0], qt[1], qt[2], qt[3]}, Vec3{qt[4], qt[5], qt[6]}};
  PoseEdge E;
  E.X = v3(xw[0], xw[1], xw[2]);
  E.u = obs[0]; E.v = obs[1]; E.ur = obs[2];
  E.invs = invs;
  E.stereo = !(E.ur < 0);
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  Vec3 pc;
  (void)pose_edge_chi2(T, cam, E, e, pc);
  return pose_obs_accumulate(T, cam, E, huber_delta, acc);
}
// analytic Jacobians (jacobian_mode = PPO_JAC_ANALYTIC) for tests/test_analytic_jacobians.py
void hc_plane_analytic(const double* qt, const double* plane, const double* meas, int kind, double* e, double* J27) {
  const Pose T{Quat{qt[0], qt[1], qt[2], qt[3]}, Vec3{qt[4], qt[5], qt[6]}};
  plane_edge_analytic(T, plane_canon(plane[0], plane[1], plane[2], plane[3]), plane_canon(meas[0], meas[1], meas[2], meas[3]), kind, e, J27);
}
void hc_cubproj_analytic(const double* pose_qt, const double* c10, const double* K, const double* meas, unsigned flags, int scale_fixed, int corner, double* e, double* J) {
  const Pose T{Quat{pose_qt[0], pose_qt[1], pose_qt[2], pose_qt[3]}, Vec3{pose_qt[4], pose_qt[5], pose_qt[6]}};
  const Cub c{Pose{Quat{c10[0], c10[1], c10[2], c10[3]}, Vec3{c10[4], c10[5], c10[6]}}, Vec3{c10[7], c10[8], c10[9]}};
  if (corner) corner_edge_analytic(T, c, flags, scale_fixed != 0, K, meas, e, J);
  else bbox_edge_analytic(T, c, flags, scale_fixed != 0, K, meas, e, J);
}
}
