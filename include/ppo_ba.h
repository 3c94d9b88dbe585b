/*
 * ppo_ba.h -- C ABI of libppo_ba.so, the MI355X-native local bundle adjustment that replaces the
 * g2o section of Optimizer::LocalBACameraPlaneCuboids.
 *
 * Reference seam (paths relative to the reference checkout):
 *   include/Optimizer.h:62          static void LocalBACameraPlaneCuboids(KeyFrame*, bool*, Map*, bool, bool)
 *   src/LocalMapping.cc:100         the only call site
 *   src/Optimizer.cc:2108-2116      g2o solver set-up                      -> ppo_ba_create()
 *   src/Optimizer.cc:2120-2714      vertex / edge construction             -> the arrays of ppo_ba_problem
 *   src/Optimizer.cc:2723-2838      5 LM iterations, outlier pass, 10 more -> ppo_ba_run()
 *   src/Optimizer.cc:2840-2966      erase lists + write-back               <- ppo_ba_read()
 *
 * Everything is plain C: caller-owned pointers and sizes, no C++ types, no exceptions.  The library
 * copies what it needs inside ppo_ba_create(); no caller pointer is retained after it returns.
 *
 * Conventions (each cites the reference):
 *   pose     world->camera, (qx,qy,qz,qw,tx,ty,tz); update [omega(3), upsilon(3)], T <- exp(d)*T
 *            (Thirdparty/g2o/g2o/types/se3quat.h:274-308, types_six_dof_expmap.h:73-91)
 *   point    xyz, additive update (Thirdparty/g2o/g2o/types/types_sba.h:52-56)
 *   plane    (nx,ny,nz,d), |n|=1, d>=0; 3-DoF update (azimuth, elevation, distance)
 *            (include/G2O_Plane3D.h:45-87,120-125)
 *   cuboid   object->world (qx,qy,qz,qw,tx,ty,tz) + half extents; 9-DoF update
 *            [omega(3), upsilon(3), dscale(3)], right-multiplicative (src/g2o_cuboid.cc:39-67)
 */
#ifndef PPO_BA_H
#define PPO_BA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden and a version script: the entry points declared here (and in ppo_pose.h) are its ONLY dynamic
 * symbols, so that it can be loaded into a process that already holds g2o, Eigen, OpenCV and other ROCm libraries (no rocprim, STL or
 * ppo:: instantiation of this library can interpose on, or be interposed by, theirs). */
#if defined(__GNUC__)
#define PPO_API __attribute__((visibility("default")))
#else
#define PPO_API
#endif

#define PPO_BA_ABI_VERSION 3  /* 2: the 9-D camera-object edge (cub3d_*), th_cub3d_norm; 3: ppo_ba_stats.deterministic_fell_back, options.deterministic defaults to 1 */

/* return codes: 0 ok, <0 error, >0 early termination */
enum {
  PPO_BA_OK = 0,
  PPO_BA_ABORTED = 1,        /* *stop_flag became non-zero (Optimizer.cc:2723-2725, 2732-2734) */
  PPO_BA_ERR_ARG = -1,       /* null pointer, negative size, index out of range */
  PPO_BA_ERR_NONFINITE = -2, /* NaN/Inf in the inputs */
  PPO_BA_ERR_NOMEM = -3,
  PPO_BA_ERR_HIP = -4,       /* no device, kernel or runtime failure */
  PPO_BA_ERR_SOLVER = -5,    /* dense factorisation backend failure (not "matrix not positive") */
  PPO_BA_ERR_COMM = -6,      /* RCCL failure */
  PPO_BA_ERR_STATE = -7      /* call order (read before run, ...) */
};

/* cuboid vertex mode bits (include/g2o_cuboid.h:284-287; set at Optimizer.cc:2167-2168) */
enum { PPO_CUBOID_FIX_ROLLPITCH = 1, PPO_CUBOID_FIX_HEIGHT = 2, PPO_CUBOID_FIX_ROTATION = 4 };

/* Jacobian mode for the edges g2o differentiates numerically */
enum {
  PPO_JAC_NUMERIC = 0, /* central differences, delta = numeric_delta, as the reference (base_binary_edge.hpp:217-320): the default */
  PPO_JAC_ANALYTIC = 1 /* opt-in: exact derivatives of the same composite maps for the plane edges (EdgePlane / EdgeVerticalPlane /
                        * EdgeParallelPlane, G2O_Plane3D.h:176-310) and the cuboid projections (EdgeSE3CuboidProj / CornerProj,
                        * g2o_cuboid.cc:70-120); they agree with the central differences to the 1e-7 round-off those carry.  The
                        * point-cuboid edge stays numeric.  Cuboids with PPO_CUBOID_FIX_ROTATION are refused in this mode
                        * (their box-plus resets the rotation: a zero update is not the identity) -> PPO_BA_ERR_ARG */
};

/* dense reduced-system backend */
enum { PPO_SOLVER_AUTO = 0, PPO_SOLVER_ROCSOLVER = 1, PPO_SOLVER_NATIVE = 2 };

/*
 * The flattened factor graph.  All arrays are caller-owned and read-only.  Index arrays are 0-based
 * dense per-type indices (the reference's colliding graph ids, Optimizer.cc:2125,2170,2214,2342, are
 * not reproduced).  Edge blocks may be given in any order; order inside a block = the reference's
 * addEdge order and defines the deterministic summation order of the oracle.
 * info arrays hold the DIAGONAL of the information matrix (every information matrix built by
 * Optimizer.cc:2240-2244, 2273-2276, 2297-2300, 2377, 2405, 2465, 2533, 2645-2647, 2696-2700 is diagonal).
 * huber = RobustKernelHuber delta (residual-norm units, compared against chi2 as delta^2,
 * Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:78-91); huber <= 0 means "no robust kernel".
 */
typedef struct ppo_ba_problem {
  /* -- vertices ------------------------------------------------------------------------------- */
  int32_t n_poses;            /* local + fixed keyframes (Optimizer.cc:2120-2145) */
  const double* pose_qt;      /* [n_poses][7] qx qy qz qw tx ty tz */
  const uint8_t* pose_fixed;  /* [n_poses] 1 = setFixed(true) */
  const double* pose_cam;     /* [n_poses][5] fx fy cx cy bf (Optimizer.cc:2383-2386, 2412-2416) */

  int32_t n_points;           /* Optimizer.cc:2332-2348 */
  const double* point_xyz;    /* [n_points][3] */
  const uint8_t* point_fixed; /* [n_points] or NULL (= all free, marginalised) */

  int32_t n_planes;           /* Optimizer.cc:2208-2219 */
  const double* plane_coef;   /* [n_planes][4], any scale/sign: normalised as Plane3D::fromVector */

  int32_t n_cuboids;          /* Optimizer.cc:2150-2179; never marginalised */
  const double* cuboid_qt;    /* [n_cuboids][7] */
  const double* cuboid_scale; /* [n_cuboids][3] half extents */
  const uint8_t* cuboid_flags;      /* [n_cuboids] PPO_CUBOID_* bits, or NULL (= ROLLPITCH|HEIGHT) */
  const double* cuboid_fixedscale;  /* [n_cuboids][3] or NULL; [0]>0 pins the scale (g2o_cuboid.cc:61-62) */

  /* -- EdgeSE3ProjectXYZ (types_six_dof_expmap.h:164-193), Optimizer.cc:2366-2392 -------------- */
  int32_t n_mono;
  const int32_t* mono_point;  /* vertex slot 0 */
  const int32_t* mono_pose;   /* vertex slot 1 */
  const double* mono_obs;     /* [n_mono][2] */
  const double* mono_info;    /* [n_mono] invSigma2 (information = invSigma2 * I2) */
  const double* mono_huber;   /* [n_mono] */

  /* -- EdgeStereoSE3ProjectXYZ (types_six_dof_expmap.h:196-225), Optimizer.cc:2393-2422 --------- */
  int32_t n_stereo;
  const int32_t* stereo_point;
  const int32_t* stereo_pose;
  const double* stereo_obs;   /* [n_stereo][3] u v u_right */
  const double* stereo_info;  /* [n_stereo] */
  const double* stereo_huber; /* [n_stereo] */

  /* -- EdgePlane / EdgeVerticalPlane / EdgeParallelPlane (G2O_Plane3D.h:176-310),
   *    Optimizer.cc:2229-2307.  kind: 0 = EdgePlane (3-D), 1 = vertical (2-D), 2 = parallel (2-D) */
  int32_t n_plane_edges;
  const int32_t* pl_plane;    /* slot 0 */
  const int32_t* pl_pose;     /* slot 1 */
  const uint8_t* pl_kind;     /* [n] or NULL (= all 0) */
  const double* pl_meas;      /* [n][4] camera-frame plane, normalised on ingest like toPlane3D */
  const double* pl_info;      /* [n][3] (third entry unused for kind 1,2) */
  const double* pl_huber;     /* [n] */

  /* -- EdgeSE3CuboidProj (g2o_cuboid.h:291-303), Optimizer.cc:2433-2487 ------------------------- */
  int32_t n_bbox;
  const int32_t* bbox_pose;   /* slot 0 */
  const int32_t* bbox_cuboid; /* slot 1 */
  const double* bbox_meas;    /* [n][4] cx cy w h */
  const double* bbox_info;    /* [n][4] */
  const double* bbox_huber;   /* [n] */
  const double* bbox_K;       /* [n][9] row-major Kalib */

  /* -- EdgeSE3CuboidCornerProj (g2o_cuboid.h:306-318), Optimizer.cc:2493-2551 ------------------- */
  int32_t n_corner;
  const int32_t* corner_pose;
  const int32_t* corner_cuboid;
  const double* corner_meas;  /* [n][16] u0 v0 ... u7 v7 */
  const double* corner_info;  /* [n][16] */
  const double* corner_huber; /* [n] */
  const double* corner_K;     /* [n][9] */

  /* -- EdgePointCuboidOnlyObject (g2o_cuboid.h:355-376), Optimizer.cc:2556-2655: unary ---------- */
  int32_t n_ptcub;
  const int32_t* ptcub_cuboid;
  const int32_t* ptcub_offsets;     /* [n_ptcub+1] CSR into ptcub_points */
  const double* ptcub_points;       /* [offsets[n]][3] world points frozen at build time */
  const double* ptcub_margin_ratio; /* [n] max_outside_margin_ratio */
  const double* ptcub_prior_half;   /* [n][3] prior_object_half_size, [0]<=0 = unset; or NULL */
  const double* ptcub_info;         /* [n][3] */
  const double* ptcub_huber;        /* [n] or NULL (= none, as the reference) */

  /* -- EdgeCuboidPlane (G2O_Plane3D.h:332-474), Optimizer.cc:2662-2714: error == measurement ----- */
  int32_t n_cubpl;
  const int32_t* cubpl_cuboid; /* slot 0 */
  const int32_t* cubpl_plane;  /* slot 1 */
  const double* cubpl_meas;    /* [n][3] */
  const double* cubpl_info;    /* [n][3] */
  const double* cubpl_huber;   /* [n] */

  /* -- EdgeSE3Cuboid (g2o_cuboid.h:322-352): the 9-D camera-object 3-D measurement of Optimizer::LocalBACameraPointCuboids2D,
   *    Optimizer.cc:1767-1800 (optimize_with_cuboid_3d).  error = min over the measured box turned by -90 / 0 / 90 / 180 degrees
   *    about its z axis of [log(meas_world^-1 * cuboid), scale - meas scale] (cuboid::min_log_error, g2o_cuboid.h:83-109) -------- */
  int32_t n_cub3d;
  const int32_t* cub3d_pose;   /* slot 0 */
  const int32_t* cub3d_cuboid; /* slot 1 */
  const double* cub3d_meas;    /* [n][10] measured cuboid in the CAMERA frame (MapCuboid::cuboid_local_meas): qx qy qz qw tx ty tz sx sy sz */
  const double* cub3d_info;    /* [n][9]  (ba_weight_SE3 * meas_quality)^2, Optimizer.cc:1785-1789 */
  const double* cub3d_huber;   /* [n]     thHuberSE3 itself (not its square root), Optimizer.cc:1793 */
} ppo_ba_problem;

#define PPO_LM_DEVICE 0
#define PPO_LM_HOST 1
typedef struct ppo_ba_options {
  int32_t abi_version;      /* PPO_BA_ABI_VERSION */
  int32_t iters_pass1;      /* 5   Optimizer.cc:2728 */
  int32_t iters_pass2;      /* 10  Optimizer.cc:2837; 0 skips the outlier pass entirely */
  double lm_tau;            /* 1e-5 optimization_algorithm_levenberg.cpp:47 */
  int32_t lm_max_trials;    /* 10  :51 */
  double lm_lower;          /* 1/3 :49 */
  double lm_upper;          /* 2/3 :48 */
  double numeric_delta;     /* 1e-9 base_binary_edge.hpp:232 */
  int32_t jacobian_mode;    /* PPO_JAC_NUMERIC (default: what the reference does) or PPO_JAC_ANALYTIC; anything else -> PPO_BA_ERR_ARG */
  /* outlier thresholds of the between-pass classification, Optimizer.cc:2739-2833 */
  double chi2_mono;         /* 5.991 */
  double chi2_stereo;       /* 7.815 */
  double chi2_plane;        /* plane_chi */
  double chi2_plane_vp;     /* 200 */
  double th_bbox_norm;      /* thHuberBbox2d, compared with |e| (not chi2) Optimizer.cc:2774 */
  double th_corner_norm;    /* thHuberConer2d Optimizer.cc:2783 */
  int32_t deterministic;    /* 1 (default since round 6): every sum that crosses a wave adds 2^-24 fixed-point INTEGERS (exact, hence independent
                             * of the order the atomics arrive in): two runs on the same device and build give bit-identical results, like the
                             * single-threaded reference.  This is the mode every strict parity test runs and the mode bench.py times.
                             * The integers cover +-4e11: a window with a FINITE term beyond that (the reference sums doubles and takes it) leaves
                             * this mode at that linearisation and continues as deterministic = 0 (ppo_ba_stats.deterministic_fell_back).
                             * 0: the same sums as floating-point atomics (run-to-run differences ~1e-9 in chi2 that flat directions of a
                             * window amplify); ~1 % faster at the BASELINE size.  Any other value -> PPO_BA_ERR_ARG */
  int32_t solver;           /* PPO_SOLVER_* */
  int32_t eliminate_cuboids; /* 1 (default): cuboids are Schur-eliminated too and the dense system is poses only --
                              * the same linear solve as the reference (which keeps them dense, Optimizer.cc:2172),
                              * smaller factorisation; 0: cuboids stay in the dense system like the reference */
  /* multi-GPU (one process per GPU).  world_size<=1 = single GPU.  The 128-byte RCCL unique id is
   * created by rank 0 with ppo_ba_comm_unique_id() and broadcast by the caller. */
  int32_t device_id;        /* HIP device ordinal for this process */
  int32_t rank, world_size;
  uint8_t comm_id[128];
  int32_t verbose;
  double th_cub3d_norm;     /* thHuberSE3 (900), compared with |e| of the 9-D camera-object edges, Optimizer.cc:1875-1882 */
  int32_t solve_on_owner;   /* multi-GPU only.  0 (default): the packed (S, b) is ALL-REDUCED and every rank factors it redundantly (no broadcast:
                             * BASELINE.json north_star).  1: (S, b) is REDUCED to rank 0 (half the ring traffic), rank 0 alone factors and solves,
                             * x and the factorisation flag are broadcast.  Same results on every rank either way. */
  int32_t lm_control;       /* where the Levenberg-Marquardt decisions (gain ratio, accept / reject, lambda, nu, the nBad stop;
                             * optimization_algorithm_levenberg.cpp:102-163) are taken.  PPO_LM_DEVICE (0, default): by a kernel queued behind
                             * every trial, the host queues the next trial without waiting (single rank, native solver,
                             * verbose = 0; every other configuration takes the host path whatever this says).  PPO_LM_HOST (1): by the host
                             * after a round trip per trial.  Same arithmetic (pow(x, 3) is a cube within one ulp on the device: one ulp of
                             * lambda in about one accepted trial in a thousand), same results.  PPO_LM_DEVICE queues up to five solve attempts
                             * at once; the stop flag of ppo_ba_run is still honoured before every attempt (round 6: the device reads a mirror of it
                             * before every decision and cancels what is queued behind, levenberg.cpp:149); a rejected trial costs ~12 us more than
                             * on the host path.  Anything else -> PPO_BA_ERR_ARG */
} ppo_ba_options;

/* one record per LM iteration (one OptimizationAlgorithmLevenberg::solve call) */
typedef struct ppo_ba_iter_stat {
  int32_t pass;        /* 0 or 1 */
  int32_t iteration;   /* index inside the pass */
  int32_t trials;      /* qmax, levenberg.cpp:100-149 */
  int32_t accepted;    /* last trial accepted */
  double chi2_before;  /* currentChi at entry (robustified) */
  double chi2_after;   /* currentChi at exit */
  double lambda;       /* lambda used by the last trial */
  int32_t result;      /* 0 OK, 1 Terminate, 2 aborted */
  int32_t pad_;
} ppo_ba_iter_stat;

#define PPO_BA_MAX_ITER_STATS 64

typedef struct ppo_ba_stats {
  int32_t n_iters;                                  /* LM iterations actually run (both passes) */
  int32_t n_trials;                                 /* solve attempts */
  ppo_ba_iter_stat iters[PPO_BA_MAX_ITER_STATS];
  int32_t n_reduced;                                /* dimension of the dense reduced system */
  int32_t n_active_edges_pass1, n_active_edges_pass2;
  double wall_ms_total;                             /* host steady_clock around run() */
  double wall_ms_pass1, wall_ms_pass2, wall_ms_classify;  /* host clock; the classification kernels are only enqueued in
                                                          * `classify` (no synchronisation between the passes), they run inside pass2 */
  /* device time per phase summed over the run, from HIP events on the stream; collected when options.verbose != 0 (0 otherwise:
   * the events cost a few us each).  linearize = k_linearize_all (incl. speculated and repeated ones), schur = reduction to the
   * pose system, factor = dense factor + solve, backsub = landmark back-substitution + state update, chi2 = trial chi2 + publish */
  double ms_linearize, ms_schur, ms_factor, ms_backsub, ms_chi2;
  /* 1: options.deterministic was set, but a finite term of a committed state's linearisation did not fit the fixed-point range (|v| >= 4e11):
   * from that linearisation on -- and for the rest of the handle's life -- the sums are the floating-point atomics of deterministic = 0 (the reference
   * sums doubles and takes such a window without complaint; nothing of the abandoned linearisation is used).  0 otherwise. */
  int32_t deterministic_fell_back;
  int32_t pad_;
} ppo_ba_stats;

/* caller-allocated output buffers; any pointer may be NULL to skip that output */
typedef struct ppo_ba_result {
  double* pose_qt;        /* [n_poses][7] */
  double* point_xyz;      /* [n_points][3] */
  double* plane_coef;     /* [n_planes][4] */
  double* cuboid_qt;      /* [n_cuboids][7] */
  double* cuboid_scale;   /* [n_cuboids][3] */
  /* per-edge chi2 = e' * Omega * e of the LAST evaluated error (the semantics of g2o's stored _error,
   * SURVEY.md 8(a) note under A27), level (1 = classified outlier between the passes) and depth test */
  double* mono_chi2;      uint8_t* mono_level;    uint8_t* mono_depth_positive;
  double* stereo_chi2;    uint8_t* stereo_level;  uint8_t* stereo_depth_positive;
  double* pl_chi2;        uint8_t* pl_level;
  double* bbox_chi2;      double* bbox_errnorm;   uint8_t* bbox_level;
  double* corner_chi2;    double* corner_errnorm; uint8_t* corner_level;
  double* ptcub_chi2;
  double* cubpl_chi2;
  double* cub3d_chi2;     double* cub3d_errnorm;  uint8_t* cub3d_level;
} ppo_ba_result;

typedef struct ppo_ba_solver* ppo_ba_handle;

PPO_API int ppo_ba_version(void);
PPO_API const char* ppo_ba_strerror(int code);

/* fills *o with the reference's constants (the comments on ppo_ba_options) */
PPO_API void ppo_ba_default_options(ppo_ba_options* o);

/* validate + copy + upload once.  Fails with PPO_BA_ERR_HIP when no gfx950 device is usable.
 * Host threads: the flattening passes run on a process-wide worker team of min(hardware threads - 1, 7) threads, started on first
 * use and asleep between calls; the environment variable PPO_BA_HOST_THREADS=n caps the total (1: the calling thread only). */
PPO_API int ppo_ba_create(const ppo_ba_problem* p, const ppo_ba_options* o, ppo_ba_handle* out);

/* pass 1, outlier classification, pass 2.  stop_flag (may be NULL) is polled before the start,
 * between the passes, before every LM iteration and every LM trial, like g2o's _forceStopFlag. */
PPO_API int ppo_ba_run(ppo_ba_handle h, const volatile uint8_t* stop_flag, ppo_ba_stats* stats);

PPO_API int ppo_ba_read(ppo_ba_handle h, ppo_ba_result* r);

PPO_API void ppo_ba_destroy(ppo_ba_handle h); /* NULL-safe */

/* -- measurement and test hooks (not part of the reference seam) ---------------------------------- */

/* Re-upload the initial estimates and reset edge levels / robust kernels, so that run() can be timed
 * repeatedly on one handle with inputs already resident in HBM. */
PPO_API int ppo_ba_reset(ppo_ba_handle h);

/* One linearisation of the current state: fills the block-sparse system and returns sum(rho0).
 * Used by bench.py to time the Jacobian/assembly kernels in isolation and by the parity tests. */
PPO_API int ppo_ba_linearize(ppo_ba_handle h, double* robust_chi2);

/* Time `reps` back-to-back launches of the linearise/assemble kernel group with HIP events on the
 * library's own stream; returns average milliseconds per linearisation in *ms. */
PPO_API int ppo_ba_time_linearize(ppo_ba_handle h, int reps, double* ms);

/* Time ONE kernel of the path alone: `reps` back-to-back launches between two HIP events on the library's own
 * stream; *ms = average per launch (agrees with rocprofv3's average duration up to the inter-launch gap),
 * *bytes = the algorithmic bytes that kernel must move per launch.  State is not advanced;
 * call after ppo_ba_linearize() / ppo_ba_debug_reduced() so that the kernel's inputs exist. */
enum {
  PPO_BA_KERNEL_LIN_POSES = 0,   /* reprojection residual + Jacobians, pose side: factor records of the coupling blocks, Hpp, b_p */
  PPO_BA_KERNEL_LIN_POINTS = 1,  /* reprojection residual + Jacobians, landmark side: D, b_l, chi2 */
  PPO_BA_KERNEL_SCHUR_Z = 2,     /* retired (folded into the pair kernel): reports 0 ms */
  PPO_BA_KERNEL_SCHUR_PAIRS = 3,
  PPO_BA_KERNEL_BACKSUB = 4,
  PPO_BA_KERNEL_CHI2_POINTS = 5,
  PPO_BA_KERNEL_LINEARIZE_ALL = 6 /* the fused linearisation launch of the product path (no zeroing) */
};
PPO_API int ppo_ba_time_kernel(ppo_ba_handle h, int which, int reps, double* ms, int64_t* bytes);

/* Sharded runs (every rank calls it): average milliseconds of the exchange steps of ONE solve attempt, each timed alone with HIP events:
 * out[0] pack + all-reduce of (S, b) + unpack, out[1] pack + reduce to rank 0 + unpack there + broadcast of x and the flag
 * (options.solve_on_owner), out[2] the all-reduce of the three trial scalars, out[3] bytes of the packed (S, b).  Zeros on a single rank.
 * S, b, x hold garbage afterwards: ppo_ba_reset() before the next run. */
PPO_API int ppo_ba_time_collectives(ppo_ba_handle h, int reps, double* out);

/* Time `reps` factorisations of the current reduced system (HIP events); average ms in *ms. */
PPO_API int ppo_ba_time_factor(ppo_ba_handle h, int reps, double* ms);

/* The Schur reduction of windows with enough panels is OVERLAPPED with the dense factorisation (csrc/ppo_chol.hip): only the
 * pose-pair chunks of the first column blocks of the reduced matrix are reduced ahead of it, the first panel launches carry the rest.
 * Returns the number of 64-column blocks (= panel launches) of this window's dense system and fills fin[k], k < min(n, blocks + 1), with the
 * number of column blocks that are final when launch k starts (fin[0]: reduced ahead of the factorisation); 0 when the reduction of this
 * window is not overlapped (few panels, sharded runs, the rocSOLVER backend).  ppo_ba_time_factor() times the dense
 * solve ALONE, after a complete reduction. */
PPO_API int ppo_ba_overlap_schedule(ppo_ba_handle h, int32_t* fin, int32_t n);
/* The schedule rule itself (host only, no GPU needed): for n_blocks 64-column blocks whose pose-pair chunks start at first_chunk[c]
 * (n_blocks + 1 entries, non-decreasing), fin[0 .. n_blocks] as ppo_ba_create would choose it with `first_blocks` blocks reduced ahead and at
 * most share0 / share of all chunks carried by launch 0 / by every later launch.  Guarantees: fin is non-decreasing, fin[k] >= min(n_blocks,
 * k + 1) (the column workgroups of launch k only read final blocks), fin[n_blocks] = n_blocks. */
PPO_API int ppo_ba_overlap_plan(int32_t n_blocks, const int32_t* first_chunk, int32_t first_blocks, double share0, double share, int32_t* fin);

/* Re-linearise the current state and build the reduced system for the damping set with
 * ppo_ba_debug_set_lambda(): S is n*n column-major (returned symmetric), b the Schur right-hand side, x the
 * solution of S x = b.  n = ppo_ba_stats.n_reduced.  Any pointer may be NULL. */
PPO_API int ppo_ba_debug_set_lambda(ppo_ba_handle h, double lambda);
PPO_API int ppo_ba_debug_reduced(ppo_ba_handle h, double* S, double* b, double* x, int32_t* n);

/* Graph dump / replay (host only, no device needed).  ppo_ba_problem_save() writes every array of the
 * problem to one binary file (layout in csrc/ppo_io.cpp); ppo_ba_problem_load() returns a problem that owns its arrays and
 * must be released with ppo_ba_problem_free().  A maintainer of the reference calls save() next to ppo_ba_create() to
 * capture real local-BA windows; tools/replay.py runs a capture through the solver (and, where built, the oracle). */
PPO_API int ppo_ba_problem_save(const ppo_ba_problem* problem, const char* path);
PPO_API int ppo_ba_problem_load(const char* path, ppo_ba_problem** problem);
PPO_API void ppo_ba_problem_free(ppo_ba_problem* problem);

/* End-of-run text exports in the reference's formats (src/System.cc:402-494; the second half of SURVEY.md 8(f) N3).  Host only.
 *   tum:     one line per keyframe "timestamp tx ty tz qx qy qz qw" -- camera centre and camera-to-world rotation of the
 *            world->camera pose given, narrowed to float32 like the cv::Mat the reference prints (fixed, 6 / 7 decimals)
 *   cuboids: "class x y z roll pitch yaw sx sy sz" = cuboid::toMinimalVector (include/g2o_cuboid.h:143-162), fixed, 6 decimals;
 *            class_names may be NULL ("object")
 *   planes:  "index a b c d" as float32, fixed, 6 decimals */
PPO_API int ppo_ba_export_tum(const char* path, int32_t n, const double* timestamps, const double* pose_qt);
PPO_API int ppo_ba_export_cuboids(const char* path, int32_t n, const char* const* class_names, const double* cuboid_qt, const double* cuboid_scale);
PPO_API int ppo_ba_export_planes(const char* path, int32_t n, const double* plane_coef);

/* ppo_ba_destroy() keeps device blocks, streams and the mapped host block of the solver in a process-wide cache so that
 * the next ppo_ba_create() -- one per local-BA call in the reference -- does not pay for ~60 allocations again (capped at
 * 4 GiB).  This call returns everything cached to the driver; live solvers are not affected. */
PPO_API int ppo_ba_release_cached(void);

/* Pays the process-wide first-use costs up front: loading the library's code objects onto `device`, the stream and the mapped host block, the host
 * team, the block cache (~155 ms on MI355X, otherwise part of the FIRST ppo_ba_create of the process -- i.e. of the first local bundle
 * adjustment of a SLAM run, src/LocalMapping.cc:100).  Runs one tiny window with every vertex / edge class and one pose-only frame through the
 * public entry points, makes one large copy each way (the runtime's copy-engine path) and reserves 512 MiB of device memory for the block cache
 * (ppo_ba_release_cached() gives it back): the first create of a BASELINE-size window afterwards takes 2.2 ms instead of 155.  Call once, e.g. from System's constructor next to where the threads are spawned (src/System.cc:95-119).
 * Returns PPO_BA_OK or the negative code of the step that failed; calling it again is harmless (a few hundred microseconds). */
PPO_API int ppo_ba_warmup(int device);

/* Algorithmic bytes of one linearisation (SURVEY.md 8(d) formula) for the problem in h. */
PPO_API int64_t ppo_ba_linearize_bytes(ppo_ba_handle h);

/* Host-only helpers (no GPU needed). */
/* sizeof() of the ABI structs as THIS library was compiled: 0 ppo_ba_problem, 1 ppo_ba_options, 2 ppo_ba_stats, 3 ppo_ba_result,
 * 4 ppo_ba_iter_stat; -1 otherwise.  A binding (ctypes, cgo ...) compares its own mirror against it instead of trusting a count. */
PPO_API int64_t ppo_ba_sizeof(int32_t which);
/* Range-partition landmarks (points then planes) into world_size contiguous edge-balanced ranges:
 * fills first_point[world_size+1], first_plane[world_size+1]. */
PPO_API int ppo_ba_partition(const ppo_ba_problem* p, int32_t world_size, int32_t* first_point, int32_t* first_plane);
/* 128-byte RCCL unique id for ppo_ba_options.comm_id (call on rank 0, broadcast to the others). */
PPO_API int ppo_ba_comm_unique_id(uint8_t id[128]);

#ifdef __cplusplus
}
#endif
#endif /* PPO_BA_H */
