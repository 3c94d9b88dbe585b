// ppo_pose_math.hpp -- per-observation arithmetic of the pose-only optimisation (Optimizer::PoseOptimization, src/Optimizer.cc:247-459):
// residual, chi2 and the contribution of one Edge(Stereo)SE3ProjectXYZOnlyPose to the 6x6 normal equations.  Host + device
// (PPO_HD): k_pose_only of ppo_pose.hip runs it per thread, csrc/hostcheck.cpp compiles it for the CPU test-suite, which holds it
// against BaseUnaryEdge::constructQuadraticForm of the reference's prebuilt libg2o.so (tests/test_golden.py).
#pragma once
#include "ppo_math.hpp"

namespace ppo {

struct PoseEdge {
  Vec3 X;
  double u, v, ur, invs;
  bool stereo;
};
// computeError + chi2 (base_edge.h:58-61 with Omega = invSigma2 * I); pc = Tcw * Xw
PPO_HD double pose_edge_chi2(const Pose& T, const double* cam, const PoseEdge& E, double* e, Vec3& pc) {
  pc = qrot(T.q, E.X) + T.t;
  if (!E.stereo) {
    e[0] = E.u - (pc.x / pc.z * cam[0] + cam[2]);
    e[1] = E.v - (pc.y / pc.z * cam[1] + cam[3]);
    e[2] = 0.0;
    return e[0] * (E.invs * e[0]) + e[1] * (E.invs * e[1]);
  }
  const float invz = (float)(1.0 / pc.z);  // `const float invz = 1.0f/trans_xyz[2]`: a double division rounded to float
  const double r0 = pc.x * invz * cam[0] + cam[2];
  const double r1 = pc.y * invz * cam[1] + cam[3];
  const double r2 = r0 - cam[4] * invz;    // bf is a double member of the OnlyPose edge
  e[0] = E.u - r0;
  e[1] = E.v - r1;
  e[2] = E.ur - r2;
  return e[0] * (E.invs * e[0]) + e[1] * (E.invs * e[1]) + e[2] * (E.invs * e[2]);
}


// One active observation at pose T: acc[0..20] += J' (rho1 Omega) J (packed lower: row c, column c2 <= c at c (c + 1) / 2 + c2),
// acc[21..26] -= J' (rho1 Omega e), acc[27] += rho0 (base_unary_edge.hpp:43-72 with RobustKernelHuber, robust_kernel_impl.cpp:78-91);
// huber_delta = 0: no kernel.  Returns chi2 = e' Omega e (the edge's stored error, base_edge.h:58-61).
// Jacobian: types_six_dof_expmap.cpp:298-320 (mono), 367-398 (stereo).
PPO_HD double pose_obs_accumulate(const Pose& T, const double* cam, const PoseEdge& E, double huber_delta, double* acc) {
  double er[3];
  Vec3 pc;
  const double chi2 = pose_edge_chi2(T, cam, E, er, pc);
  double rho0, rho1;
  huber(chi2, huber_delta, rho0, rho1);
  acc[27] += rho0;
  const double xx = pc.x, yy = pc.y, invz = 1.0 / pc.z, invz_2 = invz * invz;
  double J[18];
  J[0] = xx * yy * invz_2 * cam[0]; J[1] = -(1 + (xx * xx * invz_2)) * cam[0]; J[2] = yy * invz * cam[0];
  J[3] = -invz * cam[0];            J[4] = 0;                                  J[5] = xx * invz_2 * cam[0];
  J[6] = (1 + yy * yy * invz_2) * cam[1]; J[7] = -xx * yy * invz_2 * cam[1];   J[8] = -xx * invz * cam[1];
  J[9] = 0;                         J[10] = -invz * cam[1];                    J[11] = yy * invz_2 * cam[1];
  J[12] = J[0] - cam[4] * yy * invz_2; J[13] = J[1] + cam[4] * xx * invz_2;    J[14] = J[2];
  J[15] = J[3];                     J[16] = 0;                                 J[17] = J[5] - cam[4] * invz_2;
  const int dim = E.stereo ? 3 : 2;
  const double w = rho1 * E.invs;  // robustInformation(rho) = rho[1] * Omega (base_edge.h:96-102)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r < dim) {
      const double we = rho1 * (E.invs * er[r]);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        acc[21 + c] -= J[6 * r + c] * we;
#pragma unroll
        for (int c2 = 0; c2 <= c; ++c2) acc[c * (c + 1) / 2 + c2] += J[6 * r + c] * w * J[6 * r + c2];
      }
    }
  }
  return chi2;
}

}  // namespace ppo
