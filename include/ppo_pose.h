/* ppo_pose.h -- C ABI of the pose-only optimisation in libppo_ba.so.
 *
 * Replaces the body of `int Optimizer::PoseOptimization(Frame *pFrame)` (reference src/Optimizer.cc:247-459, declared
 * include/Optimizer.h; called from the Tracking thread for every frame: src/Tracking.cc TrackReferenceKeyFrame /
 * TrackWithMotionModel / TrackLocalMap): one free SE3 pose, N unary reprojection edges against fixed map points
 * (g2o::EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose, Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:227-290),
 * Levenberg-Marquardt with a dense 6x6 solve, four rounds of ten iterations that each restart from the initial pose
 * and re-classify every observation as inlier / outlier (Optimizer.cc:375-450).
 *
 * One call solves `n_frames` independent frames, one workgroup each, in a single kernel launch; the reference passes
 * one frame.  No torch types, no C++ types, no exceptions.  Like the rest of the library there is no CPU path: without
 * a HIP device the call fails with PPO_BA_ERR_HIP.  Error codes and ppo_ba_strerror() are those of ppo_ba.h.
 */
#ifndef PPO_POSE_H
#define PPO_POSE_H

#include <stdint.h>

#include "ppo_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PPO_POSE_ABI_VERSION 1
#define PPO_POSE_MAX_ROUNDS 8

/* caller-owned, read-only; nothing is retained after the call returns */
typedef struct ppo_pose_problem {
  int32_t n_frames;
  const double* pose_qt;        /* [n_frames][7] qx qy qz qw tx ty tz = Converter::toSE3Quat(pFrame->mTcw)  (OPT:263) */
  const double* cam;            /* [n_frames][5] fx fy cx cy bf                                              (OPT:314-317, 351-355) */
  const int32_t* edge_offsets;  /* [n_frames+1] observations of frame f are [edge_offsets[f], edge_offsets[f+1]) */
  const double* xw;             /* [E][3] pMP->GetWorldPos() (float32 in the reference, widened)             (OPT:318-321, 356-359) */
  const double* obs;            /* [E][3] kpUn.pt.x, kpUn.pt.y, mvuRight[i]; mvuRight < 0 : monocular edge   (OPT:294, 301, 337) */
  const double* inv_sigma2;     /* [E] pFrame->mvInvLevelSigma2[kpUn.octave]                                 (OPT:307, 343) */
} ppo_pose_problem;

typedef struct ppo_pose_options {
  int32_t abi_version;          /* PPO_POSE_ABI_VERSION */
  int32_t device_id;
  int32_t rounds;               /* 4   (OPT:377-382) */
  int32_t iters_per_round;      /* 10  (its[4], OPT:379) */
  int32_t robust_rounds;        /* 3: rounds 0..2 keep the Huber kernel, `if(it==2) e->setRobustKernel(0)` (OPT:415, 444) */
  int32_t min_correspondences;  /* 3:  fewer observations -> return 0 without optimising (OPT:372-373) */
  int32_t min_edges;            /* 10: `if(optimizer.edges().size()<10) break` after a round (OPT:448-449) */
  float chi2_mono;              /* 5.991 (compared as float against float(e->chi2()), OPT:377, 401-403) */
  float chi2_stereo;            /* 7.815 */
  float huber_mono;             /* sqrt(5.991) as float (deltaMono, OPT:281-282) */
  float huber_stereo;           /* sqrt(7.815) as float */
  double lm_tau;                /* 1e-5, g2o defaults as in ppo_ba_options */
  int32_t lm_max_trials;        /* 10 */
  double lm_lower, lm_upper;    /* 1/3, 2/3 */
} ppo_pose_options;

typedef struct ppo_pose_stats {
  int32_t rounds_run;
  int32_t lm_iterations;        /* OptimizationAlgorithmLevenberg::solve calls over all rounds */
  int32_t lm_trials;            /* linear solves over all rounds */
  int32_t n_bad;                /* nBad of the last round run */
  double chi2_initial;          /* active robust chi2 at the initial pose, first round */
  double chi2_final;            /* active robust chi2 after the last accepted step of the last round */
  double lambda_final;
} ppo_pose_stats;

/* caller-allocated; any pointer but pose_qt may be NULL */
typedef struct ppo_pose_result {
  double* pose_qt;              /* [n_frames][7] optimised Tcw (what pFrame->SetPose receives, OPT:453-456) */
  uint8_t* outlier;             /* [E] pFrame->mvbOutlier after the last round */
  int32_t* n_inliers;           /* [n_frames] the function's return value: nInitialCorrespondences - nBad (OPT:458) */
  double* chi2;                 /* [E] e->chi2() each observation was last classified with */
  ppo_pose_stats* stats;        /* [n_frames] */
} ppo_pose_result;

#ifndef PPO_API
#if defined(__GNUC__)
#define PPO_API __attribute__((visibility("default")))
#else
#define PPO_API
#endif
#endif
PPO_API void ppo_pose_default_options(ppo_pose_options* o);

/* returns PPO_BA_OK or a negative PPO_BA_ERR_* code */
PPO_API int ppo_pose_optimize(const ppo_pose_problem* p, const ppo_pose_options* o, ppo_pose_result* r);

#ifdef __cplusplus
}
#endif
#endif
