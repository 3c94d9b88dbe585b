// ppo_solver.hip -- C ABI of libppo_ba.so (include/ppo_ba.h): graph ingest, device upload, the
// Levenberg-Marquardt control loop of Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-164
// driving the kernels of ppo_kernels.hpp, the two-pass protocol of src/Optimizer.cc:2723-2838, read-back.
//
// There is no CPU fallback: without a usable HIP device ppo_ba_create() returns PPO_BA_ERR_HIP.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <numeric>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/ppo_ba.h"
#include "ppo_chol.hpp"
#include "ppo_comm.hpp"
#include "ppo_kernels.hpp"
#include "ppo_lm_host.hpp"
#include "ppo_plan.hpp"
#include "ppo_pool.hpp"

using namespace ppo;

#define HIP_TRY(expr)                                                                                            \
  do {                                                                                                           \
    hipError_t e_ = (expr);                                                                                      \
    if (e_ != hipSuccess) {                                                                                      \
      std::fprintf(stderr, "[ppo_ba] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return PPO_BA_ERR_HIP;                                                                                     \
    }                                                                                                            \
  } while (0)
#define RC_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ < 0) return rc_;    \
  } while (0)

struct ppo_ba_solver {
  ppo_ba_options opt;
  int device = 0;
  hipStream_t stream = nullptr;
  // options.verbose: phase boundaries recorded as events on the stream, summed into stats.ms_* at the end of run()
  std::vector<hipEvent_t> ph_events;
  std::vector<int> ph_tags;  // phase that starts at event i (PH_*)
  size_t ph_used = 0;
  bool S_is_zero = false;   // S (lower triangle) and bs are cleared and nothing has been accumulated into them yet
  bool lin_valid = false;   // the linearisation buffers hold the linearisation of state `cur` (speculated by the previous trial)
  double chi_cur = 0;       // robust chi2 of state `cur` when lin_valid
  rocblas_handle blas = nullptr;
  Comm comm;
  int n_cubpl = 0;
  int n_cub_edges = 0;
  // host copies kept after create
  Stage* stage = nullptr;  // page-locked host arena: every SVec below and in create / read lives here until destroy
  SVec<int32_t> mono_perm, stereo_perm;  // sorted position -> caller index
  SVec<MonoEdge> h_mono;      // (host builder only; the device builder of ppo_plan.hpp writes the records in place)
  SVec<StereoEdge> h_stereo;
  SVec<int32_t> hk_mono_pose, hk_mono_point, hk_stereo_pose, hk_stereo_point;  // pose / point of the k-th record, kept for ppo_ba_read
  SVec<PlaneEdge> h_pl;
  SVec<CubEdge> h_bbox, h_corner;
  SVec<PtCubEdge> h_ptcub;
  SVec<double> cubpl_chi2;  // EdgeCuboidPlane: error == measurement, so chi2 is a constant
  double cubpl_rho_sum = 0.0;      // its robustified sum: a constant term of every chi2
  SVec<int32_t> h_slot_off;
  // device
  DevBuf<double> d_pose_cam, d_cub_fixedscale, d_ptcub_pts;
  DevBuf<int32_t> d_pose_col, d_pose_free_id, d_cub_col, d_mo_off, d_st_off, d_slot_col, d_slot_lm, d_pl_slot_off, d_pl_slots;
  DevBuf<uint32_t> d_mono_flags, d_stereo_flags, d_mono_flags0, d_stereo_flags0;
  DevBuf<MonoEdgeP> d_monoP;
  DevBuf<StereoEdgeP> d_stereoP;
  DevBuf<int32_t> d_monoP_src, d_stereoP_src;  // index lists the pose-major copies were expanded from (kept until destroy: the block cache is shared between streams)
  DevBuf<uint8_t> d_point_fixed, d_cub_flags;
  DevBuf<MonoEdge> d_mono;
  DevBuf<StereoEdge> d_stereo;
  DevBuf<PlaneEdge> d_pl;
  DevBuf<CubEdge> d_bbox, d_corner;
  DevBuf<PtCubEdge> d_ptcub;
  // pristine copies (flags at level 0, kernels on) for reset()
  DevBuf<PlaneEdge> d_pl0;
  DevBuf<CubEdge> d_bbox0, d_corner0;
  DevBuf<double> d_chi2_all;  // per-edge chi2 / norms
  DevBuf<double> d_state[2];  // two copies of all estimates (current / trial)
  DevBuf<double> d_init;      // pristine estimates for reset()
  DevBuf<double> d_arena;     // HB | HC | acc | D | bl | W ; a prefix is zeroed before every linearisation
  DevBuf<double> d_Wpc, d_Dinv, d_ul, d_S, d_xr, d_cholws;
  double* d_bs_p = nullptr;  // = d_S.p + ld * ld
  // deterministic mode: the reduction sums S and b as 2^-24 fixed-point integers into a second array of the same shape (ppo_chol.hip: StepFix).  The
  // native solver converts on load (fix_pending: the system in d_Sfix has not been turned into doubles -- launch_solve_local takes the fixed path);
  // every other consumer (rocSOLVER, the sum over ranks, the debug hooks) gets a conversion launch into d_S.  S_is_zero then speaks of d_Sfix.
  DevBuf<double> d_Sfix;
  double* d_bsfix_p = nullptr;
  bool fix_pending = false;
  unsigned long long lin_tag = 0;  // deterministic + pipelined: tag of the last linearisation launch (k_decide_fix converts its ranges, see queue_trial)
  DevBuf<double> d_lin_tag;        // (one 8-byte word)
  DecideFix pending_fix{};         // ... and the ranges that launch summed into
  DevBuf<double> d_pack;     // sharded runs: lower-triangle tiles of S + b, contiguous, for the all-reduce
  DevBuf<int32_t> d_pc_start, d_pc_colj, d_pc_coli, d_pc_sj, d_pc_si, d_pc_lm;  // pose-pair chunks of the Schur accumulation
  PairChunks pc{};
  std::vector<int32_t> cb_start;  // [ld / 64 + 1] first pair chunk of every 64-column block of S (the chunks are ordered by column): the overlapped reduction
  std::vector<int> fin;           // [ld / 64 + 1] its schedule (chol_overlap_schedule); empty: the reduction is never overlapped for this window
  bool reduce_overlapped = false; // the last launch_reduce left the column blocks [fin[0], ld / 64) of S to the launches of the factorisation
  DevBuf<int32_t> d_po_start, d_po_end, d_po_pose;  // pose-major chunks of the reprojection edges
  PoseChunks posec{};
  DevBuf<double> d_W;
  DevBuf<int32_t> d_cub_edge_off, d_cub_edges, d_cub_edge_col, d_cp_cub, d_cp_a, d_cp_b, d_cp_cola, d_cp_colb, d_cp_off;  // cuboid elimination plan
  DevBuf<double> d_cub_Linv, d_cub_Dinv, d_xc;
  DevBuf<int> d_info;
  // LM control on the device (optimize_pipelined): the record, the epoch the host queues launches under, the sequence number of the
  // last record asked for (slot ring_seq % kRingSlots of the mapped ring behind h_pinned's accumulator block)
  DevBuf<LmCtl> d_ctl;
  DevBuf<int> d_tailcnt;    // [65 (+ pad)] arrival counters of the decide tail (ppo_kernels.hpp: lm_tail), zero between launches
  DecideTail tail{};        // armed by queue_trial: the next evaluation launch (k_linearize_all / k_chi2_all) carries the LM decision as its tail
  bool tail_armed = false;
  DevBuf<double> d_rec;  // device ring of trial records (kRecSlots x REC_DOUBLES), copied to the mapped host ring at the end of a batch
  int epoch = 0;
  unsigned long long ring_seq = 0;
  bool pipeline_ok = false;  // this window / these options take the pipelined path
  bool poisoned = false;     // a batch of the pipelined path never reported (wait_for_records): unusable until ppo_ba_reset()
#ifdef PPO_BA_TEST_HOOKS
  DevBuf<unsigned long long> d_stamps;  // PPO_BA_LIN_STAMPS (development aid)
  std::string stamp_path;
  int stamp_plan[LIN_PARTS] = {0, 0, 0, 0, 0, 0, 0}, stamp_order[LIN_PARTS] = {0, 1, 2, 3, 4, 5, 6};
#endif
  size_t state_doubles = 0;
  double* h_pinned = nullptr;  // mapped host memory: ACC_N doubles, the factorisation info (as a double), a sequence number
  double* h_pinned_dev = nullptr;
  unsigned long long pub_seq = 0;
  Graph g{};
  State st[2]{};
  int cur = 0;
  int lds_bytes = 0, lin_lds_bytes = 0;
  bool use_lds_lin = false;  // camera table in LDS inside the fused linearisation kernel (kept small: 4 workgroups per CU)
  bool use_lds = true;
  int n_cu = 256;
  // memset plan for a linearisation: [HB|HC|acc] always, and the plane part of D/bl/W (accumulated with atomics)
  // The accumulation targets of a linearisation (H_pp | H_cc, D_l, b_l, the planes' W) exist TWICE.  A linearisation always writes the
  // buffer that is not current; the reduction of the next solve attempt clears the other one in its first launch.  That removes the
  // clearing launch from between the update and the (speculated) linearisation of the trial state, and a rejected trial leaves the
  // kept state's system untouched: switching back costs nothing (it used to cost a re-linearisation).
  // (Everything a linearisation WRITES belongs to its buffer: besides the accumulation targets, the factor records of the point slots,
  // the linearisation-point rotations and the pose-cuboid blocks of the cuboid projections, which are plain stores.)
  struct LinBuf { double *HB, *HC, *D, *bl, *W, *F, *linR, *Wpc_bbox, *Wpc_corner; };
  LinBuf lb[2] = {};
  int lin_buf = 0;                       // the buffer s->g points at: the linearisation of the current estimate
  bool lin_clean[2] = {false, false};    // all-zero accumulation ranges (ready to be linearised into)
  size_t nHBC = 0, zD_n = 0, zbl_n = 0, zW_n = 0;  // doubles: H_pp | H_cc, the planes' part of D / b_l, the planes' W
  // LM state (levenberg.cpp:43-55)
  double lambda = -1, ni = 2;
  int nBad = 0;
  ppo_ba_stats stats{};
  bool ran = false;
  int64_t lin_bytes = 0;
  int solver_kind = PPO_SOLVER_ROCSOLVER;
  bool have_system = false;  // S/bs/xr hold the last solve attempt
  // sharded runs: the stop flag is a process-local byte, so the ranks VOTE on it -- every exchange of the (chi2, scale) scalars
  // carries each rank's current reading, and control flow follows only the summed value (ranks that branched on their own
  // reading would fall out of step and leave their peers waiting in the next collective)
  const volatile uint8_t* stop_ptr = nullptr;
  bool stop_agreed = false;
  bool acc_fixed = false;  // deterministic mode: acc[ACC_CHI2 / ACC_SCALE] currently hold fixed-point integers
  bool fell_back = false;  // the handle has left the deterministic mode (leave_fixed_point)
  // device LM control: the head of a pass rides in the hand-over launch (ppo_kernels.hpp: HeadInit); the host has queued the first batch without
  // looking at the numbers and looks when it first waits for records (head_arrived)
  bool head_pending = false;
  int head_special = 0;  // what head_arrived found: 0 nothing, 1 chi2 of the estimate is NaN (frozen window), 2 a term out of the fixed-point range
  double head_chi = 0.0, head_lambda = 0.0;
};

namespace {

// (branch-free over the array so that the loops vectorise: validation was 0.4 ms of a 6 ms create at the BASELINE size)
bool finite_range(const double* p, size_t n) {
  const uint64_t* u = reinterpret_cast<const uint64_t*>(p);
  uint64_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad |= (uint64_t)((u[i] & 0x7ff0000000000000ull) == 0x7ff0000000000000ull);  // exponent all ones: NaN / Inf
  return bad == 0;
}
bool idx_range_ok(const int32_t* p, size_t n, int hi) {
  uint32_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad |= (uint32_t)((uint32_t)p[i] >= (uint32_t)hi);  // negative values wrap above hi
  return bad == 0;
}
// the large arrays (reprojection edges) are checked by the host team in slices
template <class F>
bool all_slices(size_t n, F&& ok) {
  HostTeam& team = HostTeam::get();
  const int nt = n >= (size_t(1) << 16) ? std::min(team.size(), 8) : 1;
  if (nt == 1) return ok(size_t(0), n);
  std::atomic<int> bad{0};
  team.run(nt, [&](int t) { if (!ok(n * t / nt, n * (t + 1) / nt)) bad.store(1, std::memory_order_relaxed); });
  return bad.load() == 0;
}
bool finite_all(const double* p, size_t n) { return all_slices(n, [&](size_t a, size_t b) { return finite_range(p + a, b - a); }); }
bool idx_ok(const int32_t* p, int n, int hi) { return all_slices((size_t)n, [&](size_t a, size_t b) { return idx_range_ok(p + a, b - a, hi); }); }
#define REQ(cond) do { if (!(cond)) return PPO_BA_ERR_ARG; } while (0)
#define FIN(ptr, n) do { if (!finite_all((ptr), (size_t)(n))) return PPO_BA_ERR_NONFINITE; } while (0)

int validate(const ppo_ba_problem* p) {
  REQ(p->n_poses >= 0 && p->n_points >= 0 && p->n_planes >= 0 && p->n_cuboids >= 0);
  REQ(p->n_mono >= 0 && p->n_stereo >= 0 && p->n_plane_edges >= 0 && p->n_bbox >= 0 && p->n_corner >= 0 && p->n_ptcub >= 0 && p->n_cubpl >= 0 && p->n_cub3d >= 0);
  if (p->n_poses) { REQ(p->pose_qt && p->pose_fixed && p->pose_cam); FIN(p->pose_qt, 7 * (size_t)p->n_poses); FIN(p->pose_cam, 5 * (size_t)p->n_poses); }
  for (int i = 0; i < p->n_poses; ++i) { const double* q = &p->pose_qt[7 * (size_t)i]; REQ(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] > 0.0); }  // (canon_quat divides by the norm)
  if (p->n_points) { REQ(p->point_xyz); FIN(p->point_xyz, 3 * (size_t)p->n_points); }
  if (p->n_planes) { REQ(p->plane_coef); FIN(p->plane_coef, 4 * (size_t)p->n_planes); }
  if (p->n_cuboids) { REQ(p->cuboid_qt && p->cuboid_scale); FIN(p->cuboid_qt, 7 * (size_t)p->n_cuboids); FIN(p->cuboid_scale, 3 * (size_t)p->n_cuboids); }
  for (int i = 0; i < p->n_cuboids; ++i) { const double* q = &p->cuboid_qt[7 * (size_t)i]; REQ(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] > 0.0); }
  if (p->n_mono) {
    REQ(p->mono_point && p->mono_pose && p->mono_obs && p->mono_info && p->mono_huber);
    REQ(idx_ok(p->mono_point, p->n_mono, p->n_points) && idx_ok(p->mono_pose, p->n_mono, p->n_poses));
    FIN(p->mono_obs, 2 * (size_t)p->n_mono); FIN(p->mono_info, p->n_mono); FIN(p->mono_huber, p->n_mono);
  }
  if (p->n_stereo) {
    REQ(p->stereo_point && p->stereo_pose && p->stereo_obs && p->stereo_info && p->stereo_huber);
    REQ(idx_ok(p->stereo_point, p->n_stereo, p->n_points) && idx_ok(p->stereo_pose, p->n_stereo, p->n_poses));
    FIN(p->stereo_obs, 3 * (size_t)p->n_stereo); FIN(p->stereo_info, p->n_stereo); FIN(p->stereo_huber, p->n_stereo);
  }
  if (p->n_plane_edges) {
    REQ(p->pl_plane && p->pl_pose && p->pl_meas && p->pl_info && p->pl_huber);
    REQ(idx_ok(p->pl_plane, p->n_plane_edges, p->n_planes) && idx_ok(p->pl_pose, p->n_plane_edges, p->n_poses));
    FIN(p->pl_meas, 4 * (size_t)p->n_plane_edges); FIN(p->pl_info, 3 * (size_t)p->n_plane_edges); FIN(p->pl_huber, p->n_plane_edges);
    if (p->pl_kind) for (int i = 0; i < p->n_plane_edges; ++i) REQ(p->pl_kind[i] <= 2);
  }
  if (p->n_bbox) {
    REQ(p->bbox_pose && p->bbox_cuboid && p->bbox_meas && p->bbox_info && p->bbox_huber && p->bbox_K);
    REQ(idx_ok(p->bbox_pose, p->n_bbox, p->n_poses) && idx_ok(p->bbox_cuboid, p->n_bbox, p->n_cuboids));
    FIN(p->bbox_meas, 4 * (size_t)p->n_bbox); FIN(p->bbox_info, 4 * (size_t)p->n_bbox); FIN(p->bbox_K, 9 * (size_t)p->n_bbox); FIN(p->bbox_huber, p->n_bbox);
  }
  if (p->n_corner) {
    REQ(p->corner_pose && p->corner_cuboid && p->corner_meas && p->corner_info && p->corner_huber && p->corner_K);
    REQ(idx_ok(p->corner_pose, p->n_corner, p->n_poses) && idx_ok(p->corner_cuboid, p->n_corner, p->n_cuboids));
    FIN(p->corner_meas, 16 * (size_t)p->n_corner); FIN(p->corner_info, 16 * (size_t)p->n_corner); FIN(p->corner_K, 9 * (size_t)p->n_corner); FIN(p->corner_huber, p->n_corner);
  }
  if (p->n_cub3d) {
    REQ(p->cub3d_pose && p->cub3d_cuboid && p->cub3d_meas && p->cub3d_info && p->cub3d_huber);
    REQ(idx_ok(p->cub3d_pose, p->n_cub3d, p->n_poses) && idx_ok(p->cub3d_cuboid, p->n_cub3d, p->n_cuboids));
    FIN(p->cub3d_meas, 10 * (size_t)p->n_cub3d); FIN(p->cub3d_info, 9 * (size_t)p->n_cub3d); FIN(p->cub3d_huber, p->n_cub3d);
    for (int i = 0; i < p->n_cub3d; ++i) { const double* q = &p->cub3d_meas[10 * (size_t)i]; REQ(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] > 0.0); }
  }
  if (p->n_ptcub) {
    REQ(p->ptcub_cuboid && p->ptcub_offsets && p->ptcub_margin_ratio && p->ptcub_info);
    REQ(idx_ok(p->ptcub_cuboid, p->n_ptcub, p->n_cuboids));
    REQ(p->ptcub_offsets[0] >= 0);
    for (int i = 0; i < p->n_ptcub; ++i) REQ(p->ptcub_offsets[i] <= p->ptcub_offsets[i + 1]);
    if (p->ptcub_offsets[p->n_ptcub] > 0) { REQ(p->ptcub_points); FIN(p->ptcub_points, 3 * (size_t)p->ptcub_offsets[p->n_ptcub]); }
    FIN(p->ptcub_info, 3 * (size_t)p->n_ptcub); FIN(p->ptcub_margin_ratio, p->n_ptcub);
  }
  if (p->n_cubpl) {
    REQ(p->cubpl_cuboid && p->cubpl_plane && p->cubpl_meas && p->cubpl_info && p->cubpl_huber);
    REQ(idx_ok(p->cubpl_cuboid, p->n_cubpl, p->n_cuboids) && idx_ok(p->cubpl_plane, p->n_cubpl, p->n_planes));
    FIN(p->cubpl_meas, 3 * (size_t)p->n_cubpl); FIN(p->cubpl_info, 3 * (size_t)p->n_cubpl); FIN(p->cubpl_huber, p->n_cubpl);
  }
  return PPO_BA_OK;
}

void canon_quat(double* q) {  // SE3Quat ctor: w >= 0, unit norm (se3quat.h:331-336)
  if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}
void canon_plane_cv(const double* in, double* out) {  // Converter::toPlane3D (Converter.cc:172-181) + Plane3D::fromVector
  double t[4] = {in[0], in[1], in[2], in[3]};
  if (t[3] < 0.0) for (int i = 0; i < 4; ++i) t[i] = -t[i];
  const PlaneC c = plane_canon(t[0], t[1], t[2], t[3]);
  out[0] = c.nx; out[1] = c.ny; out[2] = c.nz; out[3] = c.d;
}

void free_all(ppo_ba_solver* s) {
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
#ifdef PPO_BA_TEST_HOOKS
  if (s->g.stamps && !s->stamp_path.empty()) {  // one line per workgroup: part, then the stamps in 10 ns ticks relative to the earliest entry
    const size_t n = s->d_stamps.n;
    std::vector<unsigned long long> h(n);
    if (hipMemcpy(h.data(), s->d_stamps.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
      unsigned long long t0 = ~0ull;
      for (size_t b = 0; b < n / 8; ++b) if (h[8 * b] && h[8 * b] < t0) t0 = h[8 * b];
      if (FILE* f = std::fopen(s->stamp_path.c_str(), "w")) {
        static const char* names[LIN_PARTS] = {"plane", "bbox", "corner", "ptcub", "pose", "points", "cub3d"};
        size_t b = 0;
        for (int oi = 0; oi < LIN_PARTS; ++oi)
          for (int k = 0, part = s->stamp_order[oi]; k < s->stamp_plan[part]; ++k, ++b) {
            std::fprintf(f, "%s", names[part]);
            for (int j = 0; j < 5; ++j) std::fprintf(f, " %lld", h[8 * b + j] ? (long long)(h[8 * b + j] - t0) : -1LL);
            std::fprintf(f, "\n");
          }
        std::fclose(f);
      }
    }
  }
  s->d_stamps.release();
#endif
  // (the captured Cholesky graph stays with the cached block of S: DevPool drops it when the block is really freed)
  s->d_pose_cam.release(); s->d_cub_fixedscale.release(); s->d_ptcub_pts.release();
  s->d_pose_col.release(); s->d_pose_free_id.release(); s->d_cub_col.release(); s->d_mo_off.release(); s->d_st_off.release();
  s->d_slot_col.release(); s->d_slot_lm.release(); s->d_pl_slot_off.release(); s->d_pl_slots.release();
  s->d_mono_flags.release(); s->d_stereo_flags.release(); s->d_mono_flags0.release(); s->d_stereo_flags0.release(); s->d_monoP.release(); s->d_stereoP.release(); s->d_monoP_src.release(); s->d_stereoP_src.release(); s->d_point_fixed.release(); s->d_cub_flags.release();
  s->d_mono.release(); s->d_stereo.release(); s->d_pl.release(); s->d_bbox.release(); s->d_corner.release(); s->d_ptcub.release();
  s->d_pl0.release(); s->d_bbox0.release(); s->d_corner0.release();
  s->d_chi2_all.release(); s->d_state[0].release(); s->d_state[1].release(); s->d_init.release(); s->d_arena.release();
  s->d_Wpc.release(); s->d_Dinv.release(); s->d_ul.release();
  s->d_S.release(); s->d_Sfix.release(); s->d_lin_tag.release(); s->d_pack.release(); s->d_xr.release(); s->d_cholws.release(); s->d_info.release(); s->d_ctl.release(); s->d_rec.release(); s->d_tailcnt.release();
  s->d_pc_start.release(); s->d_pc_colj.release(); s->d_pc_coli.release(); s->d_pc_sj.release(); s->d_pc_si.release(); s->d_pc_lm.release();
  s->d_po_start.release(); s->d_po_end.release(); s->d_po_pose.release(); s->d_W.release();
  s->d_cub_edge_off.release(); s->d_cub_edges.release(); s->d_cub_edge_col.release(); s->d_cp_cub.release(); s->d_cp_a.release(); s->d_cp_b.release(); s->d_cp_cola.release(); s->d_cp_colb.release(); s->d_cp_off.release();
  s->d_cub_Linv.release(); s->d_cub_Dinv.release(); s->d_xc.release();
  if (s->blas) rocblas_destroy_handle(s->blas);
  for (hipEvent_t e : s->ph_events) (void)hipEventDestroy(e);
  s->ph_events.clear();
  s->comm.destroy();
  if (s->stream) {  // hand the (drained) stream and the mapped host block back
    DevCtx c;
    c.dev = s->device; c.stream = s->stream;
    c.h_pinned = s->h_pinned; c.h_pinned_dev = s->h_pinned_dev;
    g_ctx_pool.put(c);
    s->stream = nullptr;
  }
  if (s->stage) {  // (the SVec members never touch their memory again: their deallocate is a no-op)
    s->stage->release();
    delete s->stage;
    s->stage = nullptr;
  }
}

enum { PH_LIN = 0, PH_SCHUR, PH_FACTOR, PH_BACKSUB, PH_CHI2, PH_OTHER };
inline void phase_mark(ppo_ba_solver* s, int tag) {
  if (!s->opt.verbose) return;
  if (s->ph_used == s->ph_events.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return;
    s->ph_events.push_back(e);
    s->ph_tags.push_back(PH_OTHER);
  }
  s->ph_tags[s->ph_used] = tag;
  (void)hipEventRecord(s->ph_events[s->ph_used++], s->stream);
}
// (stream already synchronised) device time between consecutive marks, added to the phase the first one opened
inline void phase_collect(ppo_ba_solver* s) {
  for (size_t i = 0; i + 1 < s->ph_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ph_events[i], s->ph_events[i + 1]) != hipSuccess) continue;
    switch (s->ph_tags[i]) {
      case PH_LIN: s->stats.ms_linearize += ms; break;
      case PH_SCHUR: s->stats.ms_schur += ms; break;
      case PH_FACTOR: s->stats.ms_factor += ms; break;
      case PH_BACKSUB: s->stats.ms_backsub += ms; break;
      case PH_CHI2: s->stats.ms_chi2 += ms; break;
      default: break;
    }
  }
  s->ph_used = 0;
}

State make_state(double* base, int np, int npt, int npl, int nc) {
  State st;
  st.pose_qt = base; base += (size_t)np * 7;
  st.pose_rt = base; base += (size_t)np * 12;
  st.point = base; base += (size_t)npt * 3;
  st.plane = base; base += (size_t)npl * 4;
  st.cub = base;
  (void)nc;
  return st;
}

inline int nblk(long n, int b) { return (int)std::max<long>(1, (n + b - 1) / b); }

void use_lin_buffer(ppo_ba_solver* s, int buf);
int reset_device(ppo_ba_solver* s) {
  // edge flags back to "level 0, kernel on", estimates back to the inputs
  // (device-to-device from the pristine copies made at create: a reset never crosses PCIe)
  CopyPlan cp{};
  int nc = 0;
  size_t words = 0;
  auto up = [&](void* dst, const void* src, size_t bytes) {  // (every record type is a multiple of 4 bytes)
    if (bytes == 0) return;
    cp.dst[nc] = static_cast<uint32_t*>(dst);
    cp.src[nc] = static_cast<const uint32_t*>(src);
    cp.n[nc] = bytes / 4;
    words += bytes / 4;
    ++nc;
  };
  up(s->d_mono_flags.p, s->d_mono_flags0.p, s->d_mono_flags.n * sizeof(uint32_t));
  up(s->d_stereo_flags.p, s->d_stereo_flags0.p, s->d_stereo_flags.n * sizeof(uint32_t));
  up(s->d_pl.p, s->d_pl0.p, s->d_pl.n * sizeof(PlaneEdge));
  up(s->d_bbox.p, s->d_bbox0.p, s->d_bbox.n * sizeof(CubEdge));
  up(s->d_corner.p, s->d_corner0.p, s->d_corner.n * sizeof(CubEdge));
  up(s->d_state[0].p, s->d_init.p, s->state_doubles * sizeof(double));
  up(s->d_state[1].p, s->d_init.p, s->state_doubles * sizeof(double));
  ZeroPlan zp{};
  zp.p[0] = s->d_chi2_all.p; zp.n[0] = s->d_chi2_all.n;
  zp.p[1] = s->d_arena.p; zp.n[1] = s->d_arena.n;
  zp.p[2] = s->lb[0].W; zp.n[2] = 2 * s->zW_n;  // both copies of the planes' W
  s->lin_clean[0] = s->lin_clean[1] = true;
  use_lin_buffer(s, 0);
  const size_t work = words + 2 * (zp.n[0] + zp.n[1] + zp.n[2]);
  hipLaunchKernelGGL(k_reset, dim3(std::min(nblk((long)work / 4, 256), 4 * s->n_cu)), dim3(256), 0, s->stream, cp, zp);
  HIP_TRY(hipGetLastError());
  s->cur = 0;
  s->ran = false;
  s->have_system = false;
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->poisoned = false;  // (the stream has drained: whatever a timed-out batch had left queued is done, the next pass sets a fresh epoch)
  s->head_pending = false;
  s->head_special = 0;
  return PPO_BA_OK;
}

// ---- one linearisation of the current state: fills HB/HC/D/bl/W, per-edge chi2, acc[ACC_CHI2] ------------
int launch_linearize_kernel(ppo_ba_solver* s, const Graph& g, const State& st);
ZeroPlan linearize_zero_plan(ppo_ba_solver* s, int buf, bool with_acc) {
  const ppo_ba_solver::LinBuf& b = s->lb[buf];
  ZeroPlan zp{};
  zp.p[0] = b.HB; zp.n[0] = s->nHBC;  // H_pp | H_cc
  zp.p[1] = b.D + (size_t)s->g.n_points * 6; zp.n[1] = s->zD_n;
  zp.p[2] = b.bl + (size_t)s->g.n_points * 3; zp.n[2] = s->zbl_n;
  zp.p[3] = b.W; zp.n[3] = s->zW_n;
  if (with_acc) { zp.p[4] = s->g.acc; zp.n[4] = ACC_TOTAL; }
  return zp;
}
size_t zero_plan_doubles(const ZeroPlan& zp) { return zp.n[0] + zp.n[1] + zp.n[2] + zp.n[3] + zp.n[4] + zp.n[5]; }
void point_graph_at(Graph& g, const ppo_ba_solver::LinBuf& b) {
  g.HB = b.HB; g.HC = b.HC; g.D = b.D; g.bl = b.bl; g.W = b.W;
  g.F = b.F; g.linR = b.linR; g.Wpc_bbox = b.Wpc_bbox; g.Wpc_corner = b.Wpc_corner;
}
void use_lin_buffer(ppo_ba_solver* s, int buf) {
  point_graph_at(s->g, s->lb[buf]);
  s->lin_buf = buf;
}
// Linearise state `which_state` (default: the current estimate) into the buffer that is not current.  commit: that buffer becomes the
// current one right away (the linearisation of the kept / initial state); otherwise the caller switches once the trial is accepted.
// zero_acc: the chi2 / scale accumulators start from zero too (not for the speculated linearisation: the scale of the update is in there).
int launch_linearize(ppo_ba_solver* s, int which_state = -1, bool commit = true, bool zero_acc = true, bool defer_fix = false) {
  const State& st = s->st[which_state < 0 ? s->cur : which_state];
  const int target = s->lin_buf ^ 1;
  if (!s->lin_clean[target] || zero_acc) {
    ZeroPlan zp = linearize_zero_plan(s, target, zero_acc);
    if (s->lin_clean[target]) { zp.n[0] = zp.n[1] = zp.n[2] = zp.n[3] = 0; }  // only the accumulator block
    hipLaunchKernelGGL(k_zero_ranges, dim3(std::max(1, std::min(nblk((long)zero_plan_doubles(zp) / 2, 256), 2 * s->n_cu))), dim3(256), 0, s->stream, zp);
  }
  Graph g = s->g;
  point_graph_at(g, s->lb[target]);
  if (g.det && defer_fix) {  // (queue_trial: the decide launch behind this one converts, if this one runs)
    g.lin_tag_word = reinterpret_cast<unsigned long long*>(s->d_lin_tag.p);
    g.lin_tag = ++s->lin_tag;
  }
  RC_TRY(launch_linearize_kernel(s, g, st));
  s->lin_clean[target] = false;
  if (g.det) {  // the ranges the launch summed into (H_pp | H_cc, the planes' D / b_l / W) back to doubles
    const ZeroPlan zp = linearize_zero_plan(s, target, false);
    if (defer_fix) {
      s->pending_fix = DecideFix{g.lin_tag_word, g.lin_tag, {zp.p[0], zp.p[1], zp.p[2], zp.p[3]}, {zp.n[0], zp.n[1], zp.n[2], zp.n[3]}};
    } else {
      hipLaunchKernelGGL(k_fix_to_double, dim3(std::max(1, std::min(nblk((long)zero_plan_doubles(zp), 256), 2 * s->n_cu))), dim3(256), 0, s->stream, zp);
    }
    s->acc_fixed = true;
  }
  if (commit) use_lin_buffer(s, target);
  HIP_TRY(hipGetLastError());
  return PPO_BA_OK;
}

// the fused linearisation launch alone (no zeroing): numeric edge classes | pose-major chunks | landmark-major points
int launch_linearize_kernel(ppo_ba_solver* s, const Graph& g, const State& st) {
  LinPlan lp{};
  const bool ana = g.jac != 0;  // analytic Jacobians: 8 lanes per plane / cuboid-projection edge instead of 19 / 248
  lp.nb[LIN_PLANE] = (g.plane_last > g.plane_first && g.n_pl > 0) ? nblk(g.n_pl, ana ? kAnaPlaneEPB : NumericShape<PlaneModel>::EPB) : 0;
  if (g.owns_reduced) {
    lp.nb[LIN_BBOX] = ana ? (g.n_bbox ? nblk(g.n_bbox, kAnaBboxEPB) : 0) : g.n_bbox;
    lp.nb[LIN_CORNER] = ana ? (g.n_corner2d ? nblk(g.n_corner2d, kAnaCornerEPB) : 0) : g.n_corner2d;
    lp.nb[LIN_CUB3D] = g.n_corner - g.n_corner2d;  // the 9-D camera-object edges behind the corner edges: one workgroup each, always numeric
    lp.nb[LIN_PTCUB] = g.n_ptcub;
  }
  const int npts = g.pt_last - g.pt_first;
  lp.nb[LIN_POSE] = (npts > 0 && s->posec.n_chunks > 0) ? nblk(s->posec.n_chunks, 4) : 0;
  lp.nb[LIN_POINTS] = npts > 0 ? std::min(nblk((long)PPO_LIN_LPP * npts, 256), 4 * s->n_cu) : 0;
  // longest-running workgroups first (tools/lin_stamps.py): landmark side 19 us, planes 16, pose side 14, cuboid projections 8, point-cuboid 6
  static const int kOrder[LIN_PARTS] = {LIN_POINTS, LIN_PLANE, LIN_POSE, LIN_CUB3D, LIN_CORNER, LIN_BBOX, LIN_PTCUB};
  for (int i = 0; i < LIN_PARTS; ++i) lp.order[i] = kOrder[i];
  if (const char* e = PPO_DEV_ENV("PPO_BA_LIN_ORDER")) {  // development aid: seven digits, a permutation of 0123456 (plane bbox corner ptcub pose points cub3d)
    int seen = 0;
    for (int i = 0; i < LIN_PARTS && e[i] >= '0' && e[i] < '0' + LIN_PARTS; ++i) { lp.order[i] = e[i] - '0'; seen |= 1 << (e[i] - '0'); }
    if (seen != (1 << LIN_PARTS) - 1) for (int i = 0; i < LIN_PARTS; ++i) lp.order[i] = kOrder[i];
  }
  if (const char* e = PPO_DEV_ENV("PPO_BA_LIN_PARTS")) {  // development aid: time the parts of the fused launch separately
    const int m = atoi(e);
    if (!(m & 1)) lp.nb[LIN_PLANE] = 0;
    if (!(m & 2)) lp.nb[LIN_BBOX] = lp.nb[LIN_CORNER] = lp.nb[LIN_CUB3D] = 0;
    if (!(m & 4)) lp.nb[LIN_PTCUB] = 0;
    if (!(m & 8)) lp.nb[LIN_POSE] = 0;
    if (!(m & 16)) lp.nb[LIN_POINTS] = 0;
  }
  int grid = 0;
  for (int i = 0; i < LIN_PARTS; ++i) grid += lp.nb[i];
  const DecideTail tl = (s->tail_armed && grid > 0) ? s->tail : DecideTail{};
  if (grid > 0) s->tail_armed = false;
#ifdef PPO_BA_TEST_HOOKS
  if (const char* path = PPO_DEV_ENV("PPO_BA_LIN_STAMPS")) {  // development aid: wall-clock stamps of every workgroup of the LAST launch
    if (!s->d_stamps.p && grid > 0 && s->d_stamps.alloc((size_t)grid * 8)) s->g.stamps = s->d_stamps.p;
    s->stamp_path = path;
    for (int i = 0; i < LIN_PARTS; ++i) { s->stamp_plan[i] = lp.nb[i]; s->stamp_order[i] = lp.order[i]; }
    if (s->g.stamps) (void)hipMemsetAsync(s->d_stamps.p, 0, (size_t)grid * 8 * sizeof(unsigned long long), s->stream);
  }
#endif
  if (grid > 0) {
    const bool cor = lp.nb[LIN_CORNER] > 0 || lp.nb[LIN_CUB3D] > 0;
    const dim3 gd(grid), bd(256);
    const double dl = s->opt.numeric_delta;
    if (ana) {  // (the point-cuboid edges stay numeric in this mode)
      if (s->use_lds_lin && cor) hipLaunchKernelGGL((k_linearize_all<true, true, true>), gd, bd, s->lin_lds_bytes, s->stream, g, st, lp, s->posec, dl, tl);
      else if (s->use_lds_lin) hipLaunchKernelGGL((k_linearize_all<true, false, true>), gd, bd, s->lin_lds_bytes, s->stream, g, st, lp, s->posec, dl, tl);
      else if (cor) hipLaunchKernelGGL((k_linearize_all<false, true, true>), gd, bd, s->lin_lds_bytes, s->stream, g, st, lp, s->posec, dl, tl);
      else hipLaunchKernelGGL((k_linearize_all<false, false, true>), gd, bd, s->lin_lds_bytes, s->stream, g, st, lp, s->posec, dl, tl);
    } else
    if (s->use_lds_lin && cor) hipLaunchKernelGGL((k_linearize_all<true, true>), dim3(grid), dim3(256), s->lin_lds_bytes, s->stream, g, st, lp, s->posec, s->opt.numeric_delta, tl);
    else if (s->use_lds_lin) hipLaunchKernelGGL((k_linearize_all<true, false>), dim3(grid), dim3(256), s->lin_lds_bytes, s->stream, g, st, lp, s->posec, s->opt.numeric_delta, tl);
    else if (cor) hipLaunchKernelGGL((k_linearize_all<false, true>), dim3(grid), dim3(256), s->lin_lds_bytes, s->stream, g, st, lp, s->posec, s->opt.numeric_delta, tl);
    else hipLaunchKernelGGL((k_linearize_all<false, false>), dim3(grid), dim3(256), s->lin_lds_bytes, s->stream, g, st, lp, s->posec, s->opt.numeric_delta, tl);
  }
  HIP_TRY(hipGetLastError());
  return PPO_BA_OK;
}

int launch_chi2(ppo_ba_solver* s, const State& st) {
  const Graph& g = s->g;
  const int npts = g.pt_last - g.pt_first;
  NumericPlan np{};
  np.nb_plane = (g.plane_last > g.plane_first && g.n_pl > 0) ? nblk(g.n_pl, 64) : 0;
  if (g.owns_reduced) {
    np.nb_bbox = g.n_bbox ? nblk(g.n_bbox, 64) : 0;
    np.nb_corner = g.n_corner ? nblk(g.n_corner, 64) : 0;
    np.nb_ptcub = g.n_ptcub ? nblk(g.n_ptcub, 64) : 0;
  }
  const int nb_num = np.nb_plane + np.nb_bbox + np.nb_corner + np.nb_ptcub;
  const int nb_pts = npts > 0 ? std::min(nblk((long)kLanesPerPoint * npts, 256), 4 * s->n_cu) : 0;
  if (nb_num + nb_pts > 0) {
    const DecideTail tl = s->tail_armed ? s->tail : DecideTail{};
    s->tail_armed = false;
    hipLaunchKernelGGL(k_chi2_all, dim3(nb_num + nb_pts), dim3(256), 0, s->stream, g, st, np, nb_num, nb_pts, tl);
  }
  if (g.det) s->acc_fixed = true;
  HIP_TRY(hipGetLastError());
  return PPO_BA_OK;
}

// reduced system for `lambda` into S (lower triangle) / bs      (block_solver.hpp:354-439)
// (deterministic mode: the array the reduction sums into is the fixed-point one)
Graph reduce_graph(const ppo_ba_solver* s) {
  Graph g = s->g;
  if (g.det) { g.S = s->d_Sfix.p; g.bs = s->d_bsfix_p; }
  return g;
}
int launch_reduce(ppo_ba_solver* s, double lambda, bool allow_overlap = false) {
  const Graph g = reduce_graph(s);
  // Overlapped reduction (ppo_chol.hip): only the pair chunks of the first column blocks of S are reduced here, the launches of the
  // factorisation carry the rest.  (One rank, native solver, and a caller that goes straight on to launch_solve.  Round 6: also in deterministic
  // mode -- integer sums commute with everything, the factorisation converts on load.)
  const bool to_native = allow_overlap && s->solver_kind == PPO_SOLVER_NATIVE && !s->comm.active();
  s->reduce_overlapped = to_native && !s->fin.empty();
  s->fix_pending = to_native && g.det;
  const size_t n = g.ld;
  if (!s->S_is_zero) {  // (normally k_update_all of the previous trial has cleared S and b already)
    ZeroPlan zp{};
    zp.p[0] = g.S; zp.n[0] = n * n;
    zp.p[1] = g.bs; zp.n[1] = n;
    hipLaunchKernelGGL(k_zero_ranges, dim3(4 * s->n_cu), dim3(256), 0, s->stream, zp);
  }
  s->S_is_zero = false;
  const bool cub_stage = g.elim_cub && g.owns_reduced && g.n_cuboids > 0;
  const int nl = (g.pt_last - g.pt_first) + (g.plane_last - g.plane_first);
  ReducePlan rp{};
  const long work = (long)g.n_pose_free * 27 + (long)g.n_cuboids * 54 + (long)(g.n_bbox + g.n_corner) * 54 + (g.ld - g.n_red);
  rp.nb_asm = nblk(work, 256);
  rp.nb_prep = nl > 0 ? nblk(nl, 256) : 0;
  rp.nb_cubprep = cub_stage ? nblk(g.n_cuboids, kCubPerBlock) : 0;
  rp.chunk_hi = s->reduce_overlapped ? s->cb_start[s->fin[0]] : s->pc.n_chunks;
  rp.nb_pairs = (nl > 0 && rp.chunk_hi > 0) ? (nblk(rp.chunk_hi, 4) + 7) / 8 * 8 : 0;
  rp.nb_cubschur = (cub_stage && g.n_cub_pairs > 0) ? (nblk((long)g.n_cub_pairs * 8, 256) + 7) / 8 * 8 : 0;  // (multiple of 8: see k_reduce_stage3)
  if (const char* e = PPO_DEV_PROBE("PPO_BA_STAGE1_PROBE")) {  // timing experiments (WRONG results): parts of stage 1 left out -- 'c' cuboid blocks, 'l' landmark blocks, 'a' assembly
    if (strchr(e, 'c')) rp.nb_cubprep = 0;
    if (strchr(e, 'l')) rp.nb_prep = 0;
    if (strchr(e, 'a')) rp.nb_asm = 0;
  }
  if (const char* e = PPO_DEV_PROBE("PPO_BA_STAGE3_PROBE")) {  // timing experiments (WRONG results): 'c' no cuboid pair workgroups, 'p' no pose-pair workgroups
    if (strchr(e, 'c')) rp.nb_cubschur = 0;
    if (strchr(e, 'p')) rp.nb_pairs = 0;
  }
  // ... and, in the same launch, clears the linearisation buffer that is not current: the (speculated) linearisation of the trial
  // state this reduction leads to is written there
  ZeroPlan zother{};
  rp.nb_zero = 0;
  if (!s->lin_clean[s->lin_buf ^ 1]) {
    zother = linearize_zero_plan(s, s->lin_buf ^ 1, false);
    rp.nb_zero = std::max(1, std::min(nblk((long)zero_plan_doubles(zother) / 2, 256), s->n_cu));
    s->lin_clean[s->lin_buf ^ 1] = true;
  }
  // Overlapped reduction: what has to be in S before the first panel launch -- the pair chunks of the first column block(s) and the cuboid
  // terms -- rides in stage 1's launch instead of a launch of its own behind it: the pair workgroups invert their landmarks' blocks
  // themselves (they cannot wait for the landmark part of the same launch), every cuboid gets ONE workgroup that factors its block and
  // then forms its edge pairs.  (8.8 + 10.9 us in two launches -> one launch of the longer chain.)
  bool fuse_a = s->reduce_overlapped;
  if (const char* e = PPO_DEV_ENV("PPO_BA_FUSE_A")) fuse_a = fuse_a && !(e[0] == '0' && e[1] == 0);  // (A/B: "0" keeps the two launches)
  if (fuse_a) {
    rp.nb_cubfused = rp.nb_cubprep > 0 ? g.n_cuboids * kCubFusedSplit : 0;
    rp.nb_cubprep = 0;
    rp.nb_pairs1 = rp.nb_pairs;
    rp.nb_pairs = 0;
    rp.nb_cubschur = 0;
  }
  const dim3 grid1(rp.nb_cubfused + rp.nb_pairs1 + rp.nb_asm + rp.nb_prep + rp.nb_cubprep + rp.nb_zero);
  if (fuse_a) hipLaunchKernelGGL(k_reduce_stage1<true>, grid1, dim3(256), 0, s->stream, g, lambda, rp, zother, s->pc);
  else hipLaunchKernelGGL(k_reduce_stage1<false>, grid1, dim3(256), 0, s->stream, g, lambda, rp, zother, s->pc);
  if (rp.nb_pairs + rp.nb_cubschur > 0) hipLaunchKernelGGL(k_reduce_stage3, dim3(rp.nb_pairs + rp.nb_cubschur), dim3(256), 0, s->stream, g, s->pc, rp);
  if (g.det && !s->fix_pending && !s->comm.active()) {
    // S and b were summed as fixed-point integers (assembly, pair chunks, cuboid terms).  The native solver converts on load (launch_solve_local);
    // for rocSOLVER and the debug hooks they become doubles in d_S here (sharded runs: while they are packed for the collective, below)
    hipLaunchKernelGGL(k_fix_to_double_copy, dim3(2 * s->n_cu), dim3(256), 0, s->stream, (const double*)s->d_Sfix.p, s->d_S.p, n * n + n);
  }
  HIP_TRY(hipGetLastError());
  if (s->comm.active()) {
    // ONE collective per solve attempt: the lower-triangle tiles of S and b, packed (the owner's identity on the padding
    // and lambda on the diagonal are part of its contribution)
    const int nb = g.ld / 64, ntiles = nb * (nb + 1) / 2;
    if (g.det) hipLaunchKernelGGL(k_pack_lower_tiles, dim3(ntiles + 1), dim3(256), 0, s->stream, s->d_Sfix.p, s->d_bsfix_p, s->d_pack.p, g.ld, 2);
    else hipLaunchKernelGGL(k_pack_lower_tiles, dim3(ntiles + 1), dim3(256), 0, s->stream, s->d_S.p, s->d_bs_p, s->d_pack.p, g.ld, 0);
    if (s->opt.solve_on_owner) {  // reduce to rank 0, which alone unpacks (and solves: launch_solve)
      if (s->comm.sum_to_root(s->d_pack.p, (size_t)ntiles * 4096 + n, s->stream)) return PPO_BA_ERR_COMM;
      if (s->comm.rank == 0) hipLaunchKernelGGL(k_pack_lower_tiles, dim3(ntiles + 1), dim3(256), 0, s->stream, s->d_S.p, s->d_bs_p, s->d_pack.p, g.ld, 1);
    } else {
      if (s->comm.sum(s->d_pack.p, (size_t)ntiles * 4096 + n, s->stream)) return PPO_BA_ERR_COMM;
      hipLaunchKernelGGL(k_pack_lower_tiles, dim3(ntiles + 1), dim3(256), 0, s->stream, s->d_S.p, s->d_bs_p, s->d_pack.p, g.ld, 1);
    }
  }
  return PPO_BA_OK;
}

// dense SPD solve of S x = bs (LinearSolverDense::solve, linear_solver_dense.h:65-113).  info != 0 <=> not positive.
LmGate lm_gate(const ppo_ba_solver* s) { return s->g.ctl ? LmGate{&s->g.ctl->epoch, s->g.epoch} : LmGate{nullptr, 0}; }
int launch_solve_local(ppo_ba_solver* s);
int launch_solve(ppo_ba_solver* s) {
  if (s->comm.active() && s->opt.solve_on_owner) {
    // options.solve_on_owner: rank 0 holds the summed system and solves it; the solution and the factorisation flag go to everybody
    if (s->comm.rank == 0) RC_TRY(launch_solve_local(s));
    if (s->comm.bcast_from_root(s->d_xr.p, (size_t)s->g.ld * sizeof(double), s->stream)) return PPO_BA_ERR_COMM;
    if (s->comm.bcast_from_root(s->d_info.p, sizeof(int), s->stream)) return PPO_BA_ERR_COMM;
    return PPO_BA_OK;
  }
  return launch_solve_local(s);
}
int launch_solve_local(ppo_ba_solver* s) {
  const Graph& g = s->g;
  const int n = g.n_red, ld = g.ld;
  if (s->solver_kind == PPO_SOLVER_NATIVE) {
    if (s->fix_pending) {  // deterministic mode: the system lies in d_Sfix as fixed-point integers, the factorisation converts on load
      const CholFixed fixed{reinterpret_cast<const long long*>(s->d_Sfix.p), reinterpret_cast<const long long*>(s->d_bsfix_p), s->g.acc + ACC_DETRANGE};
      CholOverlap ov{pair_ctx_host(reduce_graph(s)), s->pc, s->cb_start.data(), s->fin.data(), lm_gate(s)};
      RC_TRY(chol_factor_solve_fixed(s->d_S.p, ld, s->d_xr.p, s->d_info.p, s->d_cholws.p, fixed, s->reduce_overlapped ? &ov : nullptr, lm_gate(s), s->stream, n));
      s->reduce_overlapped = false;
      s->fix_pending = false;
    } else if (s->reduce_overlapped) {
      CholOverlap ov{pair_ctx_host(g), s->pc, s->cb_start.data(), s->fin.data(), lm_gate(s)};
      RC_TRY(chol_factor_solve_overlapped(s->d_S.p, ld, s->d_xr.p, s->d_info.p, s->d_cholws.p, ov, s->stream, n));
      s->reduce_overlapped = false;
    } else {
      RC_TRY(chol_factor_solve(s->d_S.p, ld, s->d_bs_p, s->d_xr.p, s->d_info.p, s->d_cholws.p, s->stream, lm_gate(s), n));
    }
  } else {
    HIP_TRY(hipMemcpyAsync(s->d_xr.p, s->d_bs_p, (size_t)ld * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_info.p, 0, sizeof(int), s->stream));
    if (rocsolver_dpotrf(s->blas, rocblas_fill_lower, n, s->d_S.p, ld, s->d_info.p) != rocblas_status_success) return PPO_BA_ERR_SOLVER;
    if (rocsolver_dpotrs(s->blas, rocblas_fill_lower, n, 1, s->d_S.p, ld, s->d_xr.p, ld) != rocblas_status_success) return PPO_BA_ERR_SOLVER;
  }
  return PPO_BA_OK;
}

int launch_update(ppo_ba_solver* s, double lambda) {
  const Graph& g = s->g;
  const State& cur = s->st[s->cur];
  const State& tr = s->st[s->cur ^ 1];
  const int nl = (g.pt_last - g.pt_first) + (g.plane_last - g.plane_first);
  const int nb = g.ld / 64;
  UpdatePlan up{};
  up.nb_red = g.n_cuboids + nblk(g.n_poses, 256);
  up.nb_lm = nl > 0 ? nblk((long)PPO_UPD_LPP * nl, 256) : 0;
  up.nb_zero = nb * (nb + 1) / 2;
  up.S = g.det ? s->d_Sfix.p : s->d_S.p; up.bs = g.det ? s->d_bsfix_p : s->d_bs_p; up.ld = g.ld;  // (what the next reduction sums into)
  if (const char* e = PPO_DEV_PROBE("PPO_BA_UPDATE_PROBE")) {  // timing experiments (WRONG results): parts of the update launch left out -- 'l' landmarks, 'z' clearing of S
    if (strchr(e, 'l')) up.nb_lm = 0;
    if (strchr(e, 'z')) up.nb_zero = 0;
    if (strchr(e, 'r')) up.nb_red = 0;  // ('r': cuboids and poses)
  }
#ifdef PPO_BA_TEST_HOOKS
  if (PPO_DEV_ENV("PPO_BA_FUSE_BOUND")) {  // round 6 timing experiment (ppo_kernels.hpp: UpdatePlan::fuse_lin_blocks); the extra work lands in the buffer that is not current
    const int npts = g.pt_last - g.pt_first;
    up.fuse_lin_blocks = npts > 0 ? std::min(up.nb_lm, std::min(nblk((long)PPO_LIN_LPP * npts, 256), 4 * s->n_cu)) : 0;
    up.fD = s->lb[s->lin_buf ^ 1].D; up.fbl = s->lb[s->lin_buf ^ 1].bl;
  }
#endif
  const int grid = up.nb_red + up.nb_lm + up.nb_zero;
  if (s->use_lds) hipLaunchKernelGGL(k_update_all<true>, dim3(grid), dim3(256), s->lds_bytes, s->stream, g, cur, tr, lambda, up);
  else hipLaunchKernelGGL(k_update_all<false>, dim3(grid), dim3(256), 0, s->stream, g, cur, tr, lambda, up);
  s->S_is_zero = true;  // the factor is dead after the backward substitution: the next reduction finds S and b cleared
  if (g.det) s->acc_fixed = true;  // (the scale of the gain ratio was summed in fixed point)
  HIP_TRY(hipGetLastError());
  return PPO_BA_OK;
}

// bring ACC_N accumulators (+ the factorisation info) to the host; sums over ranks first.  Split in two so that more work
// can be queued behind the copy before the host blocks on it.
int fetch_acc_enqueue(ppo_ba_solver* s, bool with_info, bool zero_for_next_linearize = false, bool with_maxdiag = false, const HeadInit* head = nullptr) {
  const HeadInit hi = head ? *head : HeadInit{};
  if (s->g.det && s->acc_fixed && !s->comm.active()) {
    // deterministic mode, one rank (round 6): ONE launch folds the integer sums, converts chi2 / scale, hands them over -- and its other workgroups
    // convert the ranges of a linearisation whose conversion was deferred to here (launch_linearize(..., defer_fix): s->pending_fix)
    DecideFix df = s->pending_fix;
    df.tag_word = nullptr;  // (the host queued that linearisation unconditionally: it ran)
    const size_t nconv = df.n[0] + df.n[1] + df.n[2] + df.n[3];
    const int nwg = nconv ? std::max(1, std::min(nblk((long)nconv, 1024), s->n_cu)) : 0;
    hipLaunchKernelGGL(k_publish_fix, dim3(1 + nwg), dim3(256), 0, s->stream, s->g.acc, (const int*)(with_info ? s->d_info.p : nullptr), s->h_pinned_dev, ++s->pub_seq, df, hi);
    s->pending_fix = DecideFix{};
    s->acc_fixed = false;
    (void)zero_for_next_linearize; (void)with_maxdiag;
    return PPO_BA_OK;
  }
  if ((s->g.det && s->acc_fixed) || s->comm.active())  // the workgroups' partial sums into acc[0..1] (otherwise k_publish does it itself)
    hipLaunchKernelGGL(k_fold_acc, dim3(1), dim3(64), 0, s->stream, s->g.acc, (s->g.det && s->acc_fixed) ? 1 : 0);
  if (s->g.det && s->acc_fixed) {  // chi2 and scale -> doubles (before they are summed over ranks / published)
    ZeroPlan zp{};
    zp.p[0] = s->g.acc; zp.n[0] = 2;
    hipLaunchKernelGGL(k_fix_to_double, dim3(1), dim3(64), 0, s->stream, zp);
    s->acc_fixed = false;
  }
  if (s->comm.active()) {
    // chi2 and scale are sums over ranks; the max-diagonal (first iteration of a pass only) is a max
    // chi2, scale and the stop vote: this rank's reading of the flag goes up through the mapped host block (slot ACC_N + 2)
    s->h_pinned[ACC_N + 2] = (s->stop_ptr && *s->stop_ptr) ? 1.0 : 0.0;
    HIP_TRY(hipMemcpyAsync(s->g.acc + ACC_STOP, s->h_pinned + ACC_N + 2, sizeof(double), hipMemcpyHostToDevice, s->stream));
    if (s->comm.sum(s->g.acc, 3, s->stream)) return PPO_BA_ERR_COMM;
    if (with_maxdiag && s->comm.max(s->g.acc + ACC_MAXDIAG, 1, s->stream)) return PPO_BA_ERR_COMM;
    if (s->g.det && s->comm.sum(s->g.acc + ACC_DETBAD, 2, s->stream)) return PPO_BA_ERR_COMM;  // every rank must see every rank's flags (ACC_DETBAD, ACC_DETRANGE)
  }
  (void)zero_for_next_linearize;  // (the linearisation targets are double-buffered: nothing to clear behind the hand-over any more)
  ZeroPlan zp{};
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s->stream, s->g.acc, (const int*)(with_info ? s->d_info.p : nullptr), s->h_pinned_dev, ++s->pub_seq, zp, hi);
  // (no event behind it: a hipEventRecord costs a 6-8 us bubble in the queue in front of the next kernel -- measured with
  //  tools/gap_report.py as idle time before k_reduce_stage1 in every iteration; the host's spin is backed up by the stream itself)
  return PPO_BA_OK;
}
int fetch_acc_wait(ppo_ba_solver* s) {
  // spin on the sequence number the device writes last; fall back to the event if it does not show up
  volatile unsigned long long* seq = reinterpret_cast<volatile unsigned long long*>(s->h_pinned + ACC_N + 1);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned it = 0;; ++it) {
    if (*seq == s->pub_seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      if (s->comm.active()) s->stop_agreed = s->stop_agreed || s->h_pinned[ACC_STOP] > 0.0;
      return PPO_BA_OK;
    }
    if ((it & 1023) == 1023) {
      if (hipStreamQuery(s->stream) == hipSuccess) break;  // the stream has drained: the value is there (or something failed)
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;
    }
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (*seq != s->pub_seq) return PPO_BA_ERR_HIP;
  if (s->comm.active()) s->stop_agreed = s->stop_agreed || s->h_pinned[ACC_STOP] > 0.0;
  return PPO_BA_OK;
}
// the stop decision every rank takes alike (single rank: the flag itself)
bool stop_now(const ppo_ba_solver* s) {
  if (s->comm.active()) return s->stop_agreed;
  return s->stop_ptr && *s->stop_ptr;
}
int fetch_acc(ppo_ba_solver* s, bool with_info, bool with_maxdiag = false) {
  RC_TRY(fetch_acc_enqueue(s, with_info, false, with_maxdiag));
  return fetch_acc_wait(s);
}

// max |H_jj| over every diagonal into acc[ACC_MAXDIAG] (this rank's part; fetch_acc(..., with_maxdiag) takes the max over
// ranks).  Sharded runs keep Hpp as per-rank partial sums, so its diagonal is summed over ranks first (d_xr is free here).
int launch_maxdiag(ppo_ba_solver* s) {
  Graph g = s->g;
  g.lin_raw = (g.det && (s->pending_fix.n[0] + s->pending_fix.n[1] + s->pending_fix.n[2] + s->pending_fix.n[3]) > 0) ? 1 : 0;  // (conversion deferred to the publish launch)
  const long npd = (long)g.n_pose_free * 6;
  if (s->comm.active() && npd > 0) {
    hipLaunchKernelGGL(k_hb_diag, dim3(nblk(npd, 256)), dim3(256), 0, s->stream, g, s->d_xr.p);
    if (s->comm.sum(s->d_xr.p, (size_t)npd, s->stream)) return PPO_BA_ERR_COMM;
    g.hb_diag = s->d_xr.p;
  }
  const long nd = std::max<long>({npd, (long)g.n_cuboids * 9, 3L * ((g.pt_last - g.pt_first) + (g.plane_last - g.plane_first)), 1L});
  hipLaunchKernelGGL(k_maxdiag, dim3(std::min(nblk(nd, 256), 64)), dim3(256), 0, s->stream, g);
  HIP_TRY(hipGetLastError());
  return PPO_BA_OK;
}

bool stop_requested(const volatile uint8_t* f) { return f && *f; }

// Deterministic mode, chi2 of the ESTIMATE is NaN (a non-finite addend of its linearisation was left out of the integer sums and flagged,
// ACC_DETBAD).  In the reference every comparison on a NaN chi2 is false (levenberg.cpp:113-149): each iteration makes one solve attempt,
// rejects it (`rho > 0` is false), leaves the retry loop (`rho < 0` is false too), is never counted as bad, and lambda *= ni, ni *= 2 -- the
// window is FROZEN for the rest of the pass, nothing moves, the stored per-edge errors stay the estimate's.  That outcome is produced here
// without queueing the solves whose results would be thrown away (in the floating-point mode the NaN simply flows through the kernels).
void frozen_iteration(ppo_ba_solver* s, int pass, int iteration) {
  ppo_ba_iter_stat it{};
  it.pass = pass;
  it.iteration = iteration;
  it.trials = 1;
  it.accepted = 0;
  it.result = 0;
  it.chi2_before = it.chi2_after = std::numeric_limits<double>::quiet_NaN();
  it.lambda = s->lambda;
  s->lambda *= s->ni;
  s->ni *= 2;
  s->stats.n_trials++;
  if (s->stats.n_iters < PPO_BA_MAX_ITER_STATS) s->stats.iters[s->stats.n_iters] = it;
  s->stats.n_iters++;
}

// OptimizationAlgorithmLevenberg::solve, levenberg.cpp:61-164.  Returns <0 error, 0 OK, 1 Terminate.
//
// Host/device pipelining: the only value the host must see per trial is (chi2, scale, info) of the trial state.  While
// that copy is in flight the NEXT iteration's linearisation is already queued on the trial state (an accepted trial is
// the common case), so the device never idles on the host's decision; a rejected trial re-linearises the kept state.
// `may_speculate` is false on the last iteration of a pass (the outlier classification or the caller comes next).
// Deterministic mode: a FINITE term of the linearisation of the COMMITTED state did not fit the fixed-point range (+-4e11 at 2^-24).  The reference sums
// such a window in doubles without complaint (block_solver.hpp:461-481), so this is not an error: the handle leaves the deterministic mode for good
// and the caller of this function linearises the same state again with floating-point sums (stats.deterministic_fell_back tells).  Nothing of the
// abandoned attempt is used: both linearisation buffers and the accumulator block are cleared by the next launch_linearize, S by the next reduction.
// (A TRIAL state whose linearisation does not fit is rejected like a failed factorisation -- k_decide / lm_iteration, info 1 << 29 -- and the pass goes on
// from the kept state in the mode it is in.)  Sharded runs: the flags are summed over the ranks before anyone looks, so every rank leaves together.
void leave_fixed_point(ppo_ba_solver* s) {
  s->g.det = 0;
  s->fell_back = true;
  s->stats.deterministic_fell_back = 1;
  s->acc_fixed = false;
  s->fix_pending = false;
  s->pending_fix = DecideFix{};
  s->g.lin_raw = 0;
  s->g.lin_tag_word = nullptr;
  s->lin_valid = false;
  s->S_is_zero = false;
  s->lin_clean[0] = s->lin_clean[1] = false;
}
int lm_iteration(ppo_ba_solver* s, int iteration, bool may_speculate, const volatile uint8_t* stop, ppo_ba_iter_stat& it) {
  const bool fresh = !s->lin_valid;
  const bool defer = s->g.det && !s->comm.active();  // (deterministic, one rank: the conversions ride in the hand-over launch, fetch_acc_enqueue)
  if (fresh) {
    phase_mark(s, PH_LIN);
    RC_TRY(launch_linearize(s, -1, true, true, defer));
    phase_mark(s, PH_OTHER);
  }
  if (iteration == 0) RC_TRY(launch_maxdiag(s));
  double currentChi;
  if (fresh || iteration == 0) {
    RC_TRY(fetch_acc(s, false, iteration == 0));
    currentChi = s->h_pinned[ACC_CHI2] + s->cubpl_rho_sum;
    if (s->g.det && s->h_pinned[ACC_DETBAD] != 0.0) currentChi = std::numeric_limits<double>::quiet_NaN();  // (a non-finite term of chi2: NaN in the reference's arithmetic -- frozen, below)
    else if (s->g.det && s->h_pinned[ACC_DETRANGE] != 0.0) {  // a finite addend of THIS linearisation (the committed state) did not fit the fixed-point range
      leave_fixed_point(s);
      return lm_iteration(s, iteration, may_speculate, stop, it);  // (once: g.det is 0 now)
    }
  } else {
    currentChi = s->chi_cur;  // computeActiveErrors of the state the previous trial produced (levenberg.cpp:72-75)
  }
  s->lin_valid = false;
  const double iniChi = currentChi;
  it.chi2_before = currentChi;
  if (iteration == 0) {  // computeLambdaInit, levenberg.cpp:166-180
    s->lambda = s->opt.lm_tau * s->h_pinned[ACC_MAXDIAG];
    s->ni = 2;
    s->nBad = 0;
  }
  if (s->g.det && std::isnan(currentChi)) {  // frozen window (see frozen_iteration): the caller records the iteration
    s->lin_valid = true;  // (the estimate does not move: its linearisation stays the current system)
    s->chi_cur = currentChi;
    return 2;
  }
  double rho = 0;
  int qmax = 0;
  do {
    const double lambda = s->lambda;
    phase_mark(s, PH_SCHUR);
    RC_TRY(launch_reduce(s, lambda, true));
    phase_mark(s, PH_FACTOR);
    RC_TRY(launch_solve(s));
    s->have_system = true;
    phase_mark(s, PH_BACKSUB);
    RC_TRY(launch_update(s, lambda));
    const bool speculated = may_speculate && !stop_now(s);
    // An accepted trial is the common case and its state is linearised next anyway.  That linearisation sums rho over every
    // active edge of the trial state -- exactly the chi2 the accept test needs (computeActiveErrors + activeRobustChi2,
    // levenberg.cpp:113-116) -- and stores the per-edge chi2 like the residual-only pass, so when the linearisation is
    // speculated it REPLACES that pass: update -> zero -> linearise(trial) -> publish.  (acc[ACC_CHI2] is still 0 from
    // k_reduce_stage1, acc[ACC_SCALE] comes from the update and is left alone by the zeroing.)  The residual-only pass
    // remains for the last trial of a pass and for PPO_BA_CHI2_PASS=1 (A/B switch).
    static const bool chi2_pass = []() { const char* e = PPO_DEV_ENV("PPO_BA_CHI2_PASS"); return e && e[0] == '1'; }();
    if (speculated && !chi2_pass) {
      phase_mark(s, PH_LIN);
      RC_TRY(launch_linearize(s, s->cur ^ 1, false, false, defer));  // (into the other buffer, which the reduction above has cleared)
      RC_TRY(fetch_acc_enqueue(s, true, false));
    } else {
      phase_mark(s, PH_CHI2);
      RC_TRY(launch_chi2(s, s->st[s->cur ^ 1]));
      RC_TRY(fetch_acc_enqueue(s, true, false));
      if (speculated) {
        phase_mark(s, PH_LIN);
        RC_TRY(launch_linearize(s, s->cur ^ 1, false, true));
      }
    }
    phase_mark(s, PH_OTHER);
    RC_TRY(fetch_acc_wait(s));
    int info = 0;
    info = (int)s->h_pinned[ACC_N];
    // A hand-off of the backward substitution that timed out (chol_back_all's spin cap, info = 1 << 30) is a SOLVER FAILURE, not a
    // "not positive" matrix: it must not be retried with a larger lambda as if the reference's LDLT had rejected the system.
    if (info >= (1 << 30)) return PPO_BA_ERR_SOLVER;
    if (s->g.det && s->h_pinned[ACC_DETRANGE] != 0.0) info = 1 << 29;  // a finite addend out of the fixed-point range: the system is not usable
    double tempChi = s->h_pinned[ACC_CHI2] + s->cubpl_rho_sum;
    if (s->g.det && s->h_pinned[ACC_DETBAD] != 0.0) tempChi = std::numeric_limits<double>::quiet_NaN();  // a non-finite addend: the trial's chi2 is NaN (k_decide_fix does the same)
    if (info != 0 || !std::isfinite(s->h_pinned[ACC_SCALE])) tempChi = std::numeric_limits<double>::max();  // solver not ok (:126-127)
    rho = currentChi - tempChi;
    double scale = s->h_pinned[ACC_SCALE];
    if (info != 0) scale = 0;
    scale += 1e-3;
    rho /= scale;
    it.lambda = lambda;
    if (rho > 0 && std::isfinite(tempChi) && info == 0) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = std::min(alpha, s->opt.lm_upper);
      const double scaleFactor = std::max(s->opt.lm_lower, alpha);
      s->lambda *= scaleFactor;
      s->ni = 2;
      currentChi = tempChi;
      s->cur ^= 1;  // discardTop(): the trial state becomes the estimate
      it.accepted = 1;
      if (speculated) use_lin_buffer(s, s->lin_buf ^ 1);  // ... and its (speculated) linearisation the current system
      s->lin_valid = speculated;
      s->chi_cur = tempChi;
    } else {
      s->lambda *= s->ni;
      s->ni *= 2;
      it.accepted = 0;  // pop(): keep the current state
      // (a failed factorisation gives rho = (chi2 - DBL_MAX) / 1e-3 = -inf: retried like every rejected trial -- unless chi2 itself is NaN,
      //  in which case rho is NaN and the loop ends after this one attempt, exactly as levenberg.cpp:128-149 behaves)
      if (info != 0) {
        // The update of a failed factorisation is garbage, and so are the per-edge errors its evaluation left behind.  The reference
        // applies the STALE solution of the last good solve instead and keeps THOSE errors (levenberg.cpp:110-146); the device restores
        // the kept state's: one residual-only pass on it (rare path; DESIGN.md section 2 and the test that pins this deviation).
        RC_TRY(launch_chi2(s, s->st[s->cur]));
      }
      // (the rejected state's linearisation sits in the OTHER buffer: the kept state's system is intact, the retry reduces it again
      //  with the larger lambda and clears the other buffer on the way)
    }
    qmax++;
    s->stats.n_trials++;
  } while (rho < 0 && qmax < s->opt.lm_max_trials && !stop_now(s));
  it.trials = qmax;
  it.chi2_after = currentChi;
  if (qmax == s->opt.lm_max_trials || rho == 0) return 1;
  if ((iniChi - currentChi) * 1e3 < iniChi) s->nBad++;
  else s->nBad = 0;
  if (s->nBad >= 3) return 1;
  return 0;
}

// ---- the same pass with the LM control on the device (ppo_kernels.hpp: LmCtl, k_decide) --------------------------------------------------
// lm_iteration above waits for (chi2, scale, info) of every trial before it queues the next reduction: the queue runs dry for a host round
// trip per trial (11 us of a 90 us iteration at 30 keyframes), and the hand-over itself -- a kernel that writes host memory -- holds the
// next kernel back until the write has crossed the bus.  Here k_decide takes the decision behind the evaluation of the trial state, in
// device memory, and the host queues a BATCH of trials at once, each on the assumption that the ones before it were accepted and ended
// their iteration -- the common case, and the only one in which the launches it queues are the right ones (they read lambda from the
// device record; state and linearisation buffers swap on an accepted trial, so their pointers are the assumption).  When the assumption
// fails the record's epoch moves on and the launches queued under the old one return at once.  The last k_decide of a batch hands the
// batch's trial records to the host, which then puts its bookkeeping back to the first trial that went differently, queues what that
// outcome asks for (a retry of the same iteration, a fresh linearisation) and goes on with a batch of one, doubling while the assumption
// holds.  The stop flag is read between batches.  Single rank, floating-point accumulation, native solver (ppo_ba_create decides).
constexpr int kStopMirror = ACC_N + 3;  // slot of the mapped host block that mirrors the caller's stop flag for the device (mirror_stop_flag, below)
TrialBook book_of(const ppo_ba_solver* s) { return TrialBook{s->cur, s->lin_buf, {s->lin_clean[0], s->lin_clean[1]}, s->S_is_zero, s->reduce_overlapped, s->have_system}; }
void restore_book(ppo_ba_solver* s, const TrialBook& b) {
  s->cur = b.cur;
  use_lin_buffer(s, b.lin_buf);
  s->lin_clean[0] = b.lin_clean[0]; s->lin_clean[1] = b.lin_clean[1];
  s->S_is_zero = b.S_is_zero; s->reduce_overlapped = b.reduce_overlapped; s->have_system = b.have_system;
}
// reduction, solve, update, evaluation of the trial state and the decision: one trial, nothing waited for
int queue_trial(ppo_ba_solver* s, bool speculate, int n_publish, unsigned long long* seq_out) {
  RC_TRY(launch_reduce(s, 0.0, true));  // (lambda: the kernels read the record)
  RC_TRY(launch_solve(s));
  s->have_system = true;
  RC_TRY(launch_update(s, 0.0));
  // round 5 experiment: the decision as the tail of the launch that evaluates the trial state (ppo_kernels.hpp: lm_tail) instead of k_decide as a
  // launch of its own (5 us of every trace).
  const DecidePlan dp{s->d_rec.p, s->h_pinned_dev + 16, ++s->ring_seq, n_publish, s->h_pinned_dev + kStopMirror};
  // Measured (profiles/r05_decide_tail.txt): NEUTRAL -- 15 KF 0.822 / 0.836 ms per call with the tail against 0.822 / 0.847 with the launch, 30 KF
  // 1.240 / 1.247 against 1.254 / 1.209, BASELINE size 2 771 against 2 785 LM iterations/s: a 5 us kernel between two long ones costs the queue
  // next to nothing, and the tail holds every workgroup of the evaluation until its atomics are acknowledged.  So the separate launch, whose
  // ordering needs no argument beyond the kernel boundary, stays the product path; PPO_BA_DECIDE_TAIL=1 (test-hooks build) selects the tail.
  const char* const tail_env = PPO_DEV_ENV("PPO_BA_DECIDE_TAIL");  // (read per trial in the test-hooks build: a test switches it)
  const bool tail_off = s->g.det || !(tail_env && tail_env[0] == '1');     // (deterministic mode: the decide launch also carries the conversion of the linearisation's ranges)
  s->tail = DecideTail{dp, (const int*)s->d_info.p, s->d_ctl.p, s->d_tailcnt.p};
  s->tail_armed = !tail_off;
  s->pending_fix = DecideFix{reinterpret_cast<const unsigned long long*>(s->d_lin_tag.p), 0, {}, {}};
  if (speculate) RC_TRY(launch_linearize(s, s->cur ^ 1, false, false, true));  // doubles as the evaluation of the trial state (see lm_iteration)
  else RC_TRY(launch_chi2(s, s->st[s->cur ^ 1]));
  if (s->g.det) {
    // round 6: the deterministic mode on the pipelined path.  chi2 and scale are folded and converted by the decide wave itself, the ranges of the
    // linearisation by the other workgroups of its launch: the same number of launches per trial as the floating-point mode.
    const DecideFix& df = s->pending_fix;
    const size_t nconv = df.n[0] + df.n[1] + df.n[2] + df.n[3];
    const int nwg = nconv ? std::max(1, std::min(nblk((long)nconv, 1024), s->n_cu)) : 0;
    hipLaunchKernelGGL(k_decide_fix, dim3(1 + nwg), dim3(256), 0, s->stream, s->g.acc, (const int*)s->d_info.p, s->d_ctl.p, s->g.epoch, dp, df);
    s->pending_fix = DecideFix{};
    s->acc_fixed = false;  // (consumed: the next reduction clears the sums)
  } else
  if (s->tail_armed || tail_off) {
    s->tail_armed = false;
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, s->stream, s->g.acc, (const int*)s->d_info.p, s->d_ctl.p, s->g.epoch, dp);
  }
  HIP_TRY(hipGetLastError());
  *seq_out = dp.seq;
  return PPO_BA_OK;
}
// The caller's stop flag is a byte in ordinary host memory; the device reads its MIRROR in the mapped block (slot kStopMirror), which the host keeps
// up to date while it waits for a batch (it spins there anyway): k_decide looks at it before every decision, so a flag raised mid-batch ends the pass at
// the attempt that is being evaluated -- as levenberg.cpp:149 does -- instead of at the end of the batch (up to five attempts, ~1.8 ms at the BASELINE size).
inline void mirror_stop_flag(ppo_ba_solver* s) {
  if (s->stop_ptr && *s->stop_ptr) {
    volatile double* m = s->h_pinned + kStopMirror;
    if (*m == 0.0) *m = 1.0;
  }
}
int wait_for_records(ppo_ba_solver* s, unsigned long long seq) {
  volatile const unsigned long long* w = reinterpret_cast<volatile const unsigned long long*>(s->h_pinned + 16 + kRingSlots * kRingSlotDoubles);
  const auto t0 = std::chrono::steady_clock::now();
  bool timed_out = false;
  for (unsigned it = 0; *w != seq; ++it) {
    cpu_relax();  // (a batch is up to five LM iterations long: leave the core's other hardware thread its issue slots)
    if ((it & 15) == 15) mirror_stop_flag(s);
    if ((it & 1023) == 1023) {
      if (hipStreamQuery(s->stream) == hipSuccess) break;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { timed_out = true; break; }
    }
  }
  if (*w != seq && timed_out) {
    // Not there after 5 s of spinning.  The longest batch the tests know takes 30 ms, but a GPU shared with other processes, a profiler that
    // serialises launches or a debugger can make a healthy batch that slow: one bounded recovery -- sleep-poll the stream for another 25 s --
    // before the batch is declared hung.  (Bounded: the caller's thread is never parked in an unbounded hipStreamSynchronize.)
    const auto t1 = std::chrono::steady_clock::now();
    while (*w != seq && hipStreamQuery(s->stream) == hipErrorNotReady && std::chrono::steady_clock::now() - t1 < std::chrono::seconds(25))
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    timed_out = *w != seq && hipStreamQuery(s->stream) == hipErrorNotReady;
  }
  if (*w != seq) {
    // Launches that write the mapped ring and the control record may still be queued: the handle's sequence numbers and epoch are out of step
    // with the device from here on.  It is marked unusable -- run / read fail until ppo_ba_reset() has drained the stream and re-armed it.
    if (timed_out) { s->poisoned = true; return PPO_BA_ERR_HIP; }
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (*w != seq) { s->poisoned = true; return PPO_BA_ERR_HIP; }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return PPO_BA_OK;
}
// the device interface of lm_pipelined_pass (ppo_lm_host.hpp) on a solver
// The head of a pipelined pass (HeadInit): the hand-over that initialised the control record on the device has arrived; what the host keeps of it, and whether the
// device left the record alone (head_special: everything queued behind the head is dead then, optimize_pipelined takes it from there).
constexpr int kHeadSpecial = 1001;  // wait_records: "the head did not start the pass" (not an error code: those are negative)
int head_arrived(ppo_ba_solver* s) {
  RC_TRY(fetch_acc_wait(s));
  s->head_pending = false;
  s->head_chi = s->h_pinned[ACC_CHI2] + s->cubpl_rho_sum;
  s->head_lambda = s->opt.lm_tau * s->h_pinned[ACC_MAXDIAG];
  s->head_special = 0;
  if (s->g.det && s->h_pinned[ACC_DETBAD] != 0.0) s->head_special = 1;
  else if (s->g.det && s->h_pinned[ACC_DETRANGE] != 0.0) s->head_special = 2;
  return PPO_BA_OK;
}
struct SolverDev {
  ppo_ba_solver* s;
  bool stop() { return stop_now(s); }
  TrialBook book() { return book_of(s); }
  void restore(const TrialBook& b) { restore_book(s, b); }
  void accept(bool speculated) {  // discardTop(): the trial state is the estimate, its linearisation the current system
    s->cur ^= 1;
    if (speculated) use_lin_buffer(s, s->lin_buf ^ 1);
  }
  int enqueue_trial(bool speculate, int n_publish, unsigned long long* seq) { return queue_trial(s, speculate, n_publish, seq); }
  int wait_records(unsigned long long seq) {
    if (s->head_pending) {  // the first wait of a pass: the head's numbers first (they arrived long ago: the batch was queued behind them)
      RC_TRY(head_arrived(s));
      if (s->head_special) return kHeadSpecial;
    }
    return wait_for_records(s, seq);
  }
  const double* record(unsigned long long seq) { return s->h_pinned + 16 + (seq % kRingSlots) * kRingSlotDoubles; }
  void set_epoch(int e) { s->epoch = e; s->g.epoch = e; }
  int restore_kept_errors() { return launch_chi2(s, s->st[s->cur]); }
  int relinearize(LmCtl& m) {
    s->g.ctl = nullptr;
    RC_TRY(launch_linearize(s, -1, true, true, s->g.det != 0));
    RC_TRY(fetch_acc(s, false));
    m.chi_cur = m.ini_chi = s->h_pinned[ACC_CHI2] + s->cubpl_rho_sum;
    if (s->g.det && s->h_pinned[ACC_DETBAD] != 0.0) m.chi_cur = m.ini_chi = std::numeric_limits<double>::quiet_NaN();
    else if (s->g.det && s->h_pinned[ACC_DETRANGE] != 0.0) {
      leave_fixed_point(s);
      return relinearize(m);
    }
    hipLaunchKernelGGL(k_ctl_set, dim3(1), dim3(1), 0, s->stream, s->d_ctl.p, m);
    HIP_TRY(hipGetLastError());
    s->g.ctl = s->d_ctl.p;
    return PPO_BA_OK;
  }
  void count_trial() { s->stats.n_trials++; }
  void push_stat(const ppo_ba_iter_stat& it) {
    if (s->stats.n_iters < PPO_BA_MAX_ITER_STATS) s->stats.iters[s->stats.n_iters] = it;
    s->stats.n_iters++;
  }
};
int optimize_pipelined(ppo_ba_solver* s, int pass, int iterations) {
  s->h_pinned[kStopMirror] = stop_now(s) ? 1.0 : 0.0;  // (the mirror the device reads: see wait_for_records)
  // The head of the pass as lm_iteration does it -- linearise the estimate, lambda from the largest diagonal entry (levenberg.cpp:166-180) -- except that the
  // DEVICE writes the control record (HeadInit, riding in the hand-over launch) and the host queues the first batch right behind it.
  s->g.ctl = nullptr;
  LmCtl m{};  // the host's mirror of the record: what the device holds whenever nothing is queued (lambda, chi2: from the records, or from the head)
  m.ni = 2;
  m.cubpl = s->cubpl_rho_sum;
  m.lm_upper = s->opt.lm_upper; m.lm_lower = s->opt.lm_lower;
  m.max_trials = s->opt.lm_max_trials;
  m.epoch = ++s->epoch;  // (larger than anything the record has held: a head that leaves the record alone leaves everything queued under this epoch dead)
  const HeadInit head{s->d_ctl.p, m, s->opt.lm_tau};
  RC_TRY(launch_linearize(s, -1, true, true, s->g.det != 0));  // (deterministic mode: converted by the hand-over launch)
  RC_TRY(launch_maxdiag(s));
  RC_TRY(fetch_acc_enqueue(s, false, false, true, &head));
  HIP_TRY(hipGetLastError());
  s->head_pending = true;
  s->head_special = 0;
  s->g.ctl = s->d_ctl.p;
  s->g.epoch = s->epoch;
  int lookahead = kLookahead;
  if (const char* e = PPO_DEV_ENV("PPO_BA_LOOKAHEAD")) lookahead = std::max(1, std::min(atoi(e), kLookahead));  // (A/B: 1 = a hand-over per trial)
  SolverDev dev{s};
  const TrialBook before = book_of(s);
  const int trials0 = s->stats.n_trials;
  int rc = lm_pipelined_pass(dev, pass, iterations, m, lookahead);  // (ppo_lm_host.hpp: the batch loop, also run on the CPU against a simulated device)
  // (on an error return launches may still be queued: they are gated or harmless, the caller gives the handle up or resets it)
  s->g.ctl = nullptr;
  if (rc == PPO_BA_OK && s->head_pending) rc = head_arrived(s);  // (no batch was queued -- the stop flag, no iterations: the head's numbers all the same)
  if ((rc == PPO_BA_OK || rc == kHeadSpecial) && s->head_special) {
    // The device left the record alone: whatever was queued behind the head found its epoch missing and did nothing.  Drain it (rare path), take the
    // bookkeeping back to the head, and do what the host-side head did in these cases.
    const int what = s->head_special;
    s->head_special = 0;
    HIP_TRY(hipStreamSynchronize(s->stream));
    restore_book(s, before);
    if (what == 2) {  // a term of the estimate's linearisation out of the fixed-point range: floating-point sums from here on (as lm_iteration)
      leave_fixed_point(s);
      return optimize_pipelined(s, pass, iterations);
    }
    // chi2 of the estimate is NaN: the window is frozen for this pass (frozen_iteration)
    s->lambda = s->head_lambda;
    s->ni = 2;
    s->nBad = 0;
    for (int i = 0; i < iterations && !stop_now(s); ++i) frozen_iteration(s, pass, i);
    s->lin_valid = false;
    return PPO_BA_OK;
  }
  if (s->stats.n_trials == trials0) { m.lambda = s->head_lambda; m.chi_cur = m.ini_chi = s->head_chi; }  // (no record came back: the mirror is the head)
  s->lambda = m.lambda; s->ni = m.ni; s->nBad = m.nbad; s->chi_cur = m.chi_cur;
  s->lin_valid = false;
  return rc;
}

// SparseOptimizer::optimize, sparse_optimizer.cpp:354-419
int optimize(ppo_ba_solver* s, int pass, int iterations, const volatile uint8_t* stop) {
  if (s->g.n_red == 0) return 0;
  if (s->pipeline_ok && iterations > 0 && !stop_now(s)) return optimize_pipelined(s, pass, iterations);
  bool ok = true;
  s->lin_valid = false;
  for (int i = 0; i < iterations && !stop_now(s) && ok; ++i) {
    ppo_ba_iter_stat it{};
    it.pass = pass;
    it.iteration = i;
    const int res = lm_iteration(s, i, i + 1 < iterations, stop, it);
    if (res < 0) return res;
    if (res == 2) { frozen_iteration(s, pass, i); continue; }
    it.result = res;
    ok = (res == 0);
    if (s->stats.n_iters < PPO_BA_MAX_ITER_STATS) s->stats.iters[s->stats.n_iters] = it;
    s->stats.n_iters++;
  }
  return 0;
}

}  // namespace

// ================================================================================================
// the recorded small uploads of this thread's open batch (ppo_pool.hpp: UploadBatch) in as few launches of k_gather_uploads as their number needs
static int flush_uploads(hipStream_t stream) {
  UploadBatch* const b = t_upload_batch;
  if (!b || b->items.empty()) return PPO_BA_OK;
  for (size_t at = 0; at < b->items.size(); at += GatherPlan::kItems) {
    GatherPlan gp{};
    const int n = (int)std::min<size_t>(GatherPlan::kItems, b->items.size() - at);
    size_t largest = 0;
    for (int i = 0; i < n; ++i) {
      const UploadBatch::Item& it = b->items[at + i];
      gp.dst[i] = it.dst; gp.src[i] = it.src; gp.bytes[i] = (unsigned int)it.bytes;
      largest = std::max(largest, it.bytes);
    }
    const int bx = (int)std::max<size_t>(1, std::min<size_t>(16, (largest / 16 + 255) / 256));
    hipLaunchKernelGGL(k_gather_uploads, dim3(bx, n), dim3(256), 0, stream, gp);
  }
  b->items.clear();
  HIP_TRY(hipGetLastError());
  return PPO_BA_OK;
}
struct UploadBatchScope {  // opens a batch on this thread.  Whatever is still recorded when the scope ends is DROPPED: that is a failure path of
  UploadBatch batch;       // ppo_ba_create (its device blocks are on their way back to the pool); the success path flushes explicitly
  UploadBatch* prev;
  UploadBatchScope() : prev(t_upload_batch) { t_upload_batch = &batch; }
  ~UploadBatchScope() { t_upload_batch = prev; }
};

// ppo_ba_warmup (ppo_warmup.cpp): 512 MiB of device memory reserved for the block cache of `device` (hidden symbol: not part of the C ABI)
namespace ppo {
int pool_reserve(int device) {
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return PPO_BA_ERR_HIP;
  const bool ok = g_pool.reserve(device, (size_t)512 << 20);
  // ... and one large copy each way: the warm-up window's copies all go through the copy KERNEL (UploadBatch), so the runtime's copy-engine path for
  // arrays above 1 MB was first set up inside the first BASELINE-size create (6.4 of its 8.7 ms, in the phase that uploads the raw edge arrays)
  {
    const size_t bytes = (size_t)8 << 20;
    size_t got = 0, dgot = 0;
    bool pinned = false;
    void* h = g_host_pool.get(bytes, &got, &pinned);
    void* d = g_pool.get(bytes, device, &dgot);
    hipStream_t st = nullptr;
    if (h && d && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
      (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
      (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st);
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
    (void)hipGetLastError();
    if (d) g_pool.put(d, dgot, device);
    if (h) g_host_pool.put(h, got, pinned);
  }
  (void)hipSetDevice(prev);
  return ok ? PPO_BA_OK : PPO_BA_ERR_NOMEM;
}
}  // namespace ppo

static thread_local ppo_ba_solver* t_creating = nullptr;  // the solver ppo_ba_create is building on this thread

extern "C" {

int ppo_ba_version(void) { return PPO_BA_ABI_VERSION; }

int ppo_ba_create(const ppo_ba_problem* p, const ppo_ba_options* o, ppo_ba_handle* out) try {
  if (!p || !o || !out) return PPO_BA_ERR_ARG;
  *out = nullptr;
  if (o->abi_version != PPO_BA_ABI_VERSION) return PPO_BA_ERR_ARG;
  if ((o->deterministic != 0 && o->deterministic != 1) || (o->jacobian_mode != PPO_JAC_NUMERIC && o->jacobian_mode != PPO_JAC_ANALYTIC)) {
    std::fprintf(stderr, "[ppo_ba] options.deterministic must be 0 or 1, jacobian_mode PPO_JAC_NUMERIC or PPO_JAC_ANALYTIC\n");
    return PPO_BA_ERR_ARG;
  }
  if (o->world_size < 1 || o->rank < 0 || o->rank >= o->world_size) return PPO_BA_ERR_ARG;
  if (o->solve_on_owner != 0 && o->solve_on_owner != 1) return PPO_BA_ERR_ARG;
  if (o->lm_control != PPO_LM_DEVICE && o->lm_control != PPO_LM_HOST) return PPO_BA_ERR_ARG;
  const auto t_begin = std::chrono::steady_clock::now();
  RC_TRY(validate(p));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    std::fprintf(stderr, "[ppo_ba] no HIP device: libppo_ba has no CPU path\n");
    return PPO_BA_ERR_HIP;
  }
  if (o->device_id < 0 || o->device_id >= ndev) return PPO_BA_ERR_ARG;
  ppo_ba_solver* s = new (std::nothrow) ppo_ba_solver();
  if (!s) return PPO_BA_ERR_NOMEM;
  t_creating = s;  // (the handler of this function's try block tears it down if a staging vector throws)
  s->opt = *o;
  s->device = o->device_id;
  s->stage = new Stage();
  StageScope stage_scope(s->stage, false);  // what create builds stays until destroy: the uploads below read it asynchronously
  auto fail = [&](int code) { t_creating = nullptr; free_all(s); delete s; return code; };
  if (hipSetDevice(s->device) != hipSuccess) return fail(PPO_BA_ERR_HIP);
  {
    int ncu = 0;  // (hipGetDeviceProperties fills a large structure and costs more than a millisecond)
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, s->device) == hipSuccess && ncu > 0) s->n_cu = ncu;
  }
  {
    DevCtx c;
    if (!g_ctx_pool.get(s->device, &c)) return fail(PPO_BA_ERR_HIP);
    s->stream = c.stream;
    s->h_pinned = c.h_pinned; s->h_pinned_dev = c.h_pinned_dev;
    std::memset(s->h_pinned, 0, kPinnedDoubles * sizeof(double));
    s->pub_seq = 0;
  }
  if (s->comm.init(o->comm_id, o->rank, o->world_size)) return fail(PPO_BA_ERR_COMM);
  s->solver_kind = (o->solver == PPO_SOLVER_NATIVE) ? PPO_SOLVER_NATIVE : (o->solver == PPO_SOLVER_ROCSOLVER ? PPO_SOLVER_ROCSOLVER : chol_default_backend());
  if (s->solver_kind == PPO_SOLVER_ROCSOLVER) {  // the rocBLAS handle costs ~3 ms: only the rocSOLVER backend pays for it
    if (rocblas_create_handle(&s->blas) != rocblas_status_success) return fail(PPO_BA_ERR_SOLVER);
    rocblas_set_stream(s->blas, s->stream);
  }

  auto t_mark = t_begin;
  auto mark = [&](const char* what) {  // phase timer of create (options.verbose)
    if (!o->verbose) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[ppo_ba] create: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_mark).count());
    t_mark = t;
  };
  mark("validate, device, streams");
  Graph& g = s->g;
  g.n_poses = p->n_poses; g.n_points = p->n_points; g.n_planes = p->n_planes; g.n_cuboids = p->n_cuboids;
  g.n_mono = p->n_mono; g.n_stereo = p->n_stereo; g.n_pl = p->n_plane_edges; g.n_bbox = p->n_bbox; g.n_ptcub = p->n_ptcub;
  // The 9-D camera-object edges (EdgeSE3Cuboid, cub3d_*) are stored BEHIND the corner-projection edges in the same records: every
  // pose x cuboid structure below (blocks, W storage, elimination lists, classification, per-edge outputs) covers both.
  g.n_corner2d = p->n_corner;
  g.n_corner = p->n_corner + p->n_cub3d;
  const int n_corner_all = g.n_corner;
  auto corner_pose_of = [p](int e) { return e < p->n_corner ? p->corner_pose[e] : p->cub3d_pose[e - p->n_corner]; };
  auto corner_cuboid_of = [p](int e) { return e < p->n_corner ? p->corner_cuboid[e] : p->cub3d_cuboid[e - p->n_corner]; };
  s->n_cubpl = p->n_cubpl;
  const int Np = p->n_poses, Npt = p->n_points, Npl = p->n_planes, Nc = p->n_cuboids;

  // ---- which vertices enter the reduced system: non-fixed and touched by at least one edge -----------
  SVec<uint8_t> pose_used(Np, 0), cub_used(Nc, 0), point_fixed(Npt, 0);
  if (p->point_fixed) std::copy(p->point_fixed, p->point_fixed + Npt, point_fixed.begin());
  {  // (the reprojection edges in slices on the host team, one private mark array per slice)
    auto mark_poses = [&](const int32_t* pose, int n) {
      HostTeam& tm = HostTeam::get();
      const int nt = n >= (1 << 16) ? std::min(tm.size(), 8) : 1;
      std::vector<std::vector<uint8_t>> local(nt, std::vector<uint8_t>(Np, 0));
      tm.run(nt, [&](int t) { uint8_t* u = local[t].data(); for (int e = (int)((long)n * t / nt); e < (int)((long)n * (t + 1) / nt); ++e) u[pose[e]] = 1; });
      for (int t = 0; t < nt; ++t) for (int i = 0; i < Np; ++i) pose_used[i] |= local[t][i];
    };
    mark_poses(p->mono_pose, p->n_mono);
    mark_poses(p->stereo_pose, p->n_stereo);
  }
  for (int e = 0; e < p->n_plane_edges; ++e) pose_used[p->pl_pose[e]] = 1;
  for (int e = 0; e < p->n_bbox; ++e) { pose_used[p->bbox_pose[e]] = 1; cub_used[p->bbox_cuboid[e]] = 1; }
  for (int e = 0; e < n_corner_all; ++e) { pose_used[corner_pose_of(e)] = 1; cub_used[corner_cuboid_of(e)] = 1; }
  for (int e = 0; e < p->n_ptcub; ++e) cub_used[p->ptcub_cuboid[e]] = 1;
  SVec<int32_t> pose_col(Np, -1), pose_free_id(Np, -1), cub_col(Nc, -1);
  int nfree = 0;
  for (int i = 0; i < Np; ++i)
    if (!p->pose_fixed[i] && pose_used[i]) { pose_free_id[i] = nfree; pose_col[i] = 6 * nfree; ++nfree; }
  int ncol = 6 * nfree, n_cub_used = 0;
  g.elim_cub = o->eliminate_cuboids ? 1 : 0;
  g.hb_diag = nullptr;
  for (int i = 0; i < Nc; ++i)
    if (cub_used[i]) {
      ++n_cub_used;
      if (g.elim_cub) cub_col[i] = 0;  // marker only: eliminated cuboids have no column in the dense system
      else { cub_col[i] = ncol; ncol += 9; }
    }
  g.n_pose_free = nfree;
  g.n_red = ncol;
  g.ld = chol_padded_ld(ncol);
  const int ld = g.ld;
  // chol_back_all hands x over between its ld/64 workgroups inside one launch; the head of the chain is dispatched first, so it makes
  // progress with any number of them resident (tested to 64 blocks).  The native solver is a latency design for local-BA sizes: past 256
  // blocks (2730 free poses) the automatic choice is rocSOLVER, an explicit request for the native solver is refused
  if (s->solver_kind == PPO_SOLVER_NATIVE && ld / 64 > 256) {
    if (o->solver == PPO_SOLVER_NATIVE) return fail(PPO_BA_ERR_ARG);
    s->solver_kind = PPO_SOLVER_ROCSOLVER;
    if (rocblas_create_handle(&s->blas) != rocblas_status_success) return fail(PPO_BA_ERR_SOLVER);
    rocblas_set_stream(s->blas, s->stream);
  }
  g.n_lm = Npt + Npl;
  s->stats.n_reduced = 6 * nfree + 9 * n_cub_used;  // the reference's reduced dimension (cuboids are not marginalised there)

  // ---- reprojection edges sorted by point (stable counting sort), CSR offsets -------------------------
  auto csr_sort = [&](int n, const int32_t* key, SVec<int32_t>& off, SVec<int32_t>& perm) -> bool {
    // the reference adds the observations point by point (Optimizer.cc:2350-2422), so the edges normally ARRIVE sorted: then the
    // permutation is the identity and the offsets are the positions where the key changes (both filled in slices by the team)
    if (n >= 1 && all_slices((size_t)n - 1, [&](size_t a, size_t b) { int32_t bad = 0; for (size_t e = a; e < b; ++e) bad |= (int32_t)(key[e] > key[e + 1]); return bad == 0; })) {
      off.resize(Npt + 1);
      perm.resize(n);
      const int nt = n >= (1 << 16) ? std::min(HostTeam::get().size(), 8) : 1;
      HostTeam::get().run(nt, [&](int t) {
        const int e0 = (int)((long)n * t / nt), e1 = (int)((long)n * (t + 1) / nt);
        for (int e = e0; e < e1; ++e) perm[e] = e;
        // off[k] = first edge with key >= k: the slice owns the keys in (key[e0 - 1], key[e1 - 1]]
        int k = e0 == 0 ? 0 : key[e0 - 1] + 1;
        for (int e = e0; e < e1; ++e)
          for (; k <= key[e]; ++k) off[k] = e;
        if (e1 == n) for (; k <= Npt; ++k) off[k] = n;
      });
      return true;
    }
    off.assign(Npt + 1, 0);
    for (int e = 0; e < n; ++e) off[key[e] + 1]++;
    for (int i = 0; i < Npt; ++i) off[i + 1] += off[i];
    perm.resize(n);
    SVec<int32_t> cursor(off.begin(), off.end() - 1);
    for (int e = 0; e < n; ++e) perm[cursor[key[e]]++] = e;
    return n == 0;
  };
  SVec<int32_t> mo_off, st_off;
  const bool mono_sorted = csr_sort(p->n_mono, p->mono_point, mo_off, s->mono_perm);
  const bool stereo_sorted = csr_sort(p->n_stereo, p->stereo_point, st_off, s->stereo_perm);
  // Edge lists that arrive sorted by point (the reference's loop, include/ppo_flatten.hpp) have their records, pose-major copy and
  // Schur pair plan built ON THE DEVICE (ppo_plan.hpp); anything else, and small windows, take the host builder below.  Both produce
  // the same device state.
  // The device builder wins at every size measured (tools/small_window_times.py, profiles/r03_small_windows.txt: create 0.47 vs 0.51 ms at
  // 15 keyframes / 2 900 edges, 0.59 vs 0.89 at 30 / 9 500 -- the ICL-sized windows the reference really runs --, 0.75 vs 1.40 at 50 / 24 800),
  // so every window whose edge lists arrive sorted by point takes it; the host builder remains for unsorted input.
  // (round 5, with the uploads batched: at 15 KF / 2 900 edges the two builders are now level -- 0.35 / 0.40 ms host against 0.39 / 0.38 device in two
  //  runs --, from 30 KF on the device builder wins as before: 0.45 against 0.76)
  const long dev_plan_min = PPO_DEV_ENV("PPO_BA_DEV_PLAN_MIN") ? atol(PPO_DEV_ENV("PPO_BA_DEV_PLAN_MIN")) : 0;
  const bool dev_plan = mono_sorted && stereo_sorted && (long)p->n_mono + p->n_stereo >= dev_plan_min && !PPO_DEV_ENV("PPO_BA_HOST_PLAN");

  mark("csr sort");
  // round 5: the small uploads from here on are recorded and go up in one gather launch each time a kernel is about to read them (flush_uploads)
  UploadBatchScope upload_batch;
  // ---- host threads for the plans below -------------------------------------------------------------------------------
  // Everything from here to the uploads is counting / bucketing over the ~4e5 edges and ~9e5 landmark pose pairs of a
  // BASELINE-size window.  It runs on a few threads over contiguous, ascending unit ranges, so every bucket keeps the order
  // the sequential loops produced (the device layout does not depend on the thread count).
  HostTeam& team = HostTeam::get();
  const int n_host_threads = ((long)p->n_mono + p->n_stereo >= 40000) ? std::min(team.size(), 8) : 1;
  auto run_threads = [&](int nt, auto&& fn) { team.run(nt, fn); };
  // stable bucket sort of the items the units [0, n_units) generate: count(u0, u1, hist) tallies keys, fill(u0, u1, cursor)
  // places items at cursor[key]++ (after prepare(total) has sized the outputs); start gets n_keys + 1 bucket offsets
  auto bucket_by_key = [&](int n_units, long n_keys, SVec<int32_t>& start, auto&& count, auto&& prepare, auto&& fill) {
    const int nt = n_host_threads;
    std::vector<SVec<int32_t>> hist(nt);
    run_threads(nt, [&](int t) {
      hist[t].assign(n_keys, 0);
      count((int)((long)n_units * t / nt), (int)((long)n_units * (t + 1) / nt), hist[t].data());
    });
    start.assign(n_keys + 1, 0);
    int32_t run = 0;
    for (long k = 0; k < n_keys; ++k) {
      start[k] = run;
      for (int t = 0; t < nt; ++t) { const int32_t c = hist[t][k]; hist[t][k] = run; run += c; }
    }
    start[n_keys] = run;
    prepare(run);
    run_threads(nt, [&](int t) { fill((int)((long)n_units * t / nt), (int)((long)n_units * (t + 1) / nt), hist[t].data()); });
  };

  // ---- (landmark, pose) slots: unique free poses seen by each free landmark, ascending column ----------
  // per landmark from the CSR orders (points: mo_off / st_off; planes: a small CSR built here): gather, sort + unique the
  // short segment in place, then compact
  SVec<int32_t> pl_off, pl_perm;
  {
    pl_off.assign(Npl + 1, 0);
    for (int e = 0; e < p->n_plane_edges; ++e) pl_off[p->pl_plane[e] + 1]++;
    for (int j = 0; j < Npl; ++j) pl_off[j + 1] += pl_off[j];
    pl_perm.resize(p->n_plane_edges);
    SVec<int32_t> cursor(pl_off.begin(), pl_off.end() - 1);
    for (int e = 0; e < p->n_plane_edges; ++e) pl_perm[cursor[p->pl_plane[e]]++] = e;
  }
  // (device builder: all of this happens in k_plan_slot_* of ppo_plan.hpp; the host only sizes the arrays by upper bounds)
  SVec<int32_t> seg_off, seg_col, seg_cnt, slot_col, slot_lm, new_id, d_slot_col_v, d_slot_lm_v, pl_slot_off, pl_slots;
  SVec<int32_t>& slot_off = s->h_slot_off;
  int n_point_slots = p->n_mono + p->n_stereo;
  g.n_slots = n_point_slots + p->n_plane_edges;
  if (!dev_plan) {
  seg_off.assign(g.n_lm + 1, 0); seg_cnt.assign(g.n_lm, 0);
  for (int pt = 0; pt < Npt; ++pt) seg_off[pt + 1] = seg_off[pt] + (mo_off[pt + 1] - mo_off[pt]) + (st_off[pt + 1] - st_off[pt]);
  for (int j = 0; j < Npl; ++j) seg_off[Npt + j + 1] = seg_off[Npt + j] + (pl_off[j + 1] - pl_off[j]);
  seg_col.resize(seg_off[g.n_lm]);
  run_threads(n_host_threads, [&](int t) {
    const int l0 = (int)((long)g.n_lm * t / n_host_threads), l1 = (int)((long)g.n_lm * (t + 1) / n_host_threads);
    for (int l = l0; l < l1; ++l) {
      int32_t* b0 = seg_col.data() + seg_off[l];
      int32_t* q = b0;
      if (l < Npt) {
        if (point_fixed[l]) continue;
        for (int k = mo_off[l]; k < mo_off[l + 1]; ++k) { const int pc = pose_col[p->mono_pose[s->mono_perm[k]]]; if (pc >= 0) *q++ = pc; }
        for (int k = st_off[l]; k < st_off[l + 1]; ++k) { const int pc = pose_col[p->stereo_pose[s->stereo_perm[k]]]; if (pc >= 0) *q++ = pc; }
      } else {
        for (int k = pl_off[l - Npt]; k < pl_off[l - Npt + 1]; ++k) { const int pc = pose_col[p->pl_pose[pl_perm[k]]]; if (pc >= 0) *q++ = pc; }
      }
      bool ascending = true;  // (observations usually arrive in keyframe order: then there is nothing to sort or to merge)
      for (int32_t* x = b0 + 1; x < q; ++x) ascending &= x[-1] < x[0];
      if (ascending) { seg_cnt[l] = (int32_t)(q - b0); continue; }
      std::sort(b0, q);
      seg_cnt[l] = (int32_t)(std::unique(b0, q) - b0);
    }
  });
  mark("  slots: gather + sort + unique");
  slot_off.assign(g.n_lm + 1, 0);
  for (int l = 0; l < g.n_lm; ++l) slot_off[l + 1] = slot_off[l] + seg_cnt[l];
  g.n_slots = slot_off[g.n_lm];
  slot_col.resize(g.n_slots); slot_lm.resize(g.n_slots);
  run_threads(n_host_threads, [&](int t) {
    const int l0 = (int)((long)g.n_lm * t / n_host_threads), l1 = (int)((long)g.n_lm * (t + 1) / n_host_threads);
    for (int l = l0; l < l1; ++l)
      for (int k = 0; k < seg_cnt[l]; ++k) { slot_col[slot_off[l] + k] = seg_col[seg_off[l] + k]; slot_lm[slot_off[l] + k] = l; }
  });
  mark("  slots: compact");
  // device numbering is POSE-MAJOR: point slots ordered by (pose column, point), then plane slots by (pose column, plane),
  // so that everything a keyframe touches is contiguous (the pose side of the linearisation and the Schur pair kernel stream it)
  n_point_slots = slot_off[Npt];
  new_id.resize(g.n_slots); d_slot_col_v.resize(g.n_slots); d_slot_lm_v.resize(g.n_slots);
  {
    auto renumber = [&](int first, int last, int base) {
      SVec<int32_t> start;
      bucket_by_key(last - first, ncol + 1, start,
                    [&](int u0, int u1, int32_t* h) { for (int i = first + u0; i < first + u1; ++i) h[slot_col[i]]++; }, [](int32_t) {},
                    [&](int u0, int u1, int32_t* c) { for (int i = first + u0; i < first + u1; ++i) new_id[i] = base + c[slot_col[i]]++; });
    };
    renumber(0, n_point_slots, 0);
    renumber(n_point_slots, g.n_slots, n_point_slots);
    run_threads(n_host_threads, [&](int t) {
      const int i0 = (int)((long)g.n_slots * t / n_host_threads), i1 = (int)((long)g.n_slots * (t + 1) / n_host_threads);
      for (int i = i0; i < i1; ++i) { d_slot_col_v[new_id[i]] = slot_col[i]; d_slot_lm_v[new_id[i]] = slot_lm[i]; }
    });
  }
  pl_slot_off.assign(Npl + 1, 0); pl_slots.resize(g.n_slots - n_point_slots);
  for (int j = 0; j <= Npl; ++j) pl_slot_off[j] = slot_off[Npt + j] - n_point_slots;
  for (int i = n_point_slots; i < g.n_slots; ++i) pl_slots[i - n_point_slots] = new_id[i];
  }  // (host builder)
  auto find_slot = [&](int l, int pc) -> int32_t {
    if (pc < 0) return -1;
    const int32_t* b = slot_col.data() + slot_off[l];
    const int32_t* e = slot_col.data() + slot_off[l + 1];
    const int32_t* it = std::lower_bound(b, e, pc);
    return (it != e && *it == pc) ? new_id[it - slot_col.data()] : -1;
  };

  mark("slots, pose-major numbering");
  // ---- landmark ownership (multi-GPU): contiguous edge-balanced ranges; rank 0 owns the reduced-only terms ----
  g.pt_first = 0; g.pt_last = Npt; g.plane_first = 0; g.plane_last = Npl; g.owns_reduced = 1;
  g.det = o->deterministic ? 1 : 0;
  g.jac = o->jacobian_mode == PPO_JAC_ANALYTIC ? 1 : 0;
#ifdef PPO_BA_TEST_HOOKS
  g.dbg = PPO_DEV_ENV("PPO_BA_LIN_DBG") ? atoi(PPO_DEV_ENV("PPO_BA_LIN_DBG")) : 0;
  g.stamps = nullptr;
#endif
  if (o->world_size > 1) {
    SVec<int32_t> fp(o->world_size + 1), fl(o->world_size + 1);
    if (ppo_ba_partition(p, o->world_size, fp.data(), fl.data()) != PPO_BA_OK) return fail(PPO_BA_ERR_ARG);
    g.pt_first = fp[o->rank]; g.pt_last = fp[o->rank + 1];
    g.plane_first = fl[o->rank]; g.plane_last = fl[o->rank + 1];
    g.owns_reduced = (o->rank == 0);
  }

  // ---- Schur accumulation plan: (slot_j, slot_i) pairs of every owned landmark filed by pose pair, then chunked -----
  // (built on a helper thread while this one fills the edge records: both are ~3 ms of host work at the BASELINE size)
  SVec<int32_t> pc_start, pc_colj, pc_coli, pc_sj, pc_si, pc_lm;
  std::atomic<bool> pair_plan_failed{false};
  const int pair_chunk = []() { const char* e = PPO_DEV_ENV("PPO_BA_SCHUR_CHUNK"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
  std::thread pair_plan_thread;
  if (!dev_plan) pair_plan_thread = std::thread([&]() { try {
    t_stage = s->stage;
    const int chunk = pair_chunk;
    const long nkeys = (long)nfree * (nfree + 1) / 2;
    // (slot_col is a multiple of 6: pose = col / 6.  Keys run down the COLUMNS of the lower triangle -- key(pj >= pi) = col_base[pi] + pj - pi
    //  -- so that the chunks come in the order in which the dense factorisation consumes the columns of S: chol_factor_solve_overlapped)
    SVec<int32_t> col_base(nfree + 1);
    for (long pi = 0; pi <= nfree; ++pi) col_base[pi] = (int32_t)(pi * nfree - pi * (pi - 1) / 2);
    // owned landmarks as one unit range: points [pt_first, pt_last) then planes [plane_first, plane_last)
    const int n_own_pt = g.pt_last - g.pt_first, n_own = n_own_pt + (g.plane_last - g.plane_first);
    auto lm_of = [&](int u) { return u < n_own_pt ? g.pt_first + u : Npt + g.plane_first + (u - n_own_pt); };
    SVec<int32_t> cnt;
    bucket_by_key(n_own, nkeys, cnt,
                  [&](int u0, int u1, int32_t* h) {
                    for (int u = u0; u < u1; ++u) {
                      const int l = lm_of(u);
                      for (int a = slot_off[l]; a < slot_off[l + 1]; ++a) {
                        const int pa = slot_col[a] / 6;
                        for (int b = slot_off[l]; b <= a; ++b) {
                          const int pb = slot_col[b] / 6;
                          h[col_base[pb] + pa - pb]++;
                        }
                      }
                    }
                  },
                  [&](int32_t total) { pc_sj.resize(total); pc_si.resize(total); pc_lm.resize(total); },
                  [&](int u0, int u1, int32_t* c) {
                    for (int u = u0; u < u1; ++u) {
                      const int l = lm_of(u);
                      for (int a = slot_off[l]; a < slot_off[l + 1]; ++a) {
                        const int pa = slot_col[a] / 6;
                        const int32_t ida = new_id[a];
                        for (int b = slot_off[l]; b <= a; ++b) {
                          const int pb = slot_col[b] / 6;
                          const int32_t pos = c[col_base[pb] + pa - pb]++;
                          pc_sj[pos] = ida;
                          pc_si[pos] = new_id[b];
                          pc_lm[pos] = l;
                        }
                      }
                    }
                  });
    pc_start.push_back(0);
    for (long pi = 0, k = 0; pi < nfree; ++pi)
      for (long pj = pi; pj < nfree; ++pj, ++k)
        for (int32_t e = cnt[k]; e < cnt[k + 1]; e += chunk) {
          pc_colj.push_back((int32_t)(6 * pj));
          pc_coli.push_back((int32_t)(6 * pi));
          pc_start.push_back(std::min<int32_t>(e + chunk, cnt[k + 1]));
        }
  } catch (...) { pair_plan_failed.store(true); } });
  struct JoinGuard {  // an exception on this thread (std::bad_alloc from a staging vector) must not meet a joinable std::thread
    std::thread& t;
    ~JoinGuard() { if (t.joinable()) t.join(); }
  } pair_plan_guard{pair_plan_thread};

  // ---- edge records -----------------------------------------------------------------------------------------
  // (independent per edge: large blocks are filled by a few host threads; this was the longest phase of create)
  auto parallel_for = [&](int n, auto&& body) {
    const int nt = n >= 40000 ? n_host_threads : 1;
    team.run(nt, [&](int t) { body((int)((long)n * t / nt), (int)((long)n * (t + 1) / nt)); });
  };
  SVec<uint32_t> h_mono_flags, h_stereo_flags;
  SVec<int32_t>& mono_pose_k = s->hk_mono_pose;  // pose of the k-th record, compact (the pose-major bucket passes below and ppo_ba_read use it)
  SVec<int32_t>& stereo_pose_k = s->hk_stereo_pose;
  mono_pose_k.resize(p->n_mono); stereo_pose_k.resize(p->n_stereo);
  s->hk_mono_point.resize(p->n_mono); s->hk_stereo_point.resize(p->n_stereo);
  // device builder: the caller's arrays are staged as they are (page-locked copies, made in slices) and start their upload now
  SVec<double> raw_mono_obs, raw_mono_info, raw_mono_huber, raw_stereo_obs, raw_stereo_info, raw_stereo_huber;
  ScopedDevBuf<double> d_raw_mono_obs, d_raw_mono_info, d_raw_mono_huber, d_raw_stereo_obs, d_raw_stereo_info, d_raw_stereo_huber;
  ScopedDevBuf<int32_t> d_raw_mono_pose, d_raw_mono_point, d_raw_stereo_pose, d_raw_stereo_point;
  bool ok = true;
  if (dev_plan) {
    auto stage_copy = [&](SVec<double>& dst, const double* src, size_t n) {
      dst.resize(n);
      parallel_for((int)(n / 2), [&](int a, int b) { std::memcpy(dst.data() + 2 * (size_t)a, src + 2 * (size_t)a, 2 * (size_t)(b - a) * sizeof(double)); });
      if (n & 1) dst[n - 1] = src[n - 1];
    };
    stage_copy(raw_mono_obs, p->mono_obs, 2 * (size_t)p->n_mono); stage_copy(raw_mono_info, p->mono_info, p->n_mono); stage_copy(raw_mono_huber, p->mono_huber, p->n_mono);
    stage_copy(raw_stereo_obs, p->stereo_obs, 3 * (size_t)p->n_stereo); stage_copy(raw_stereo_info, p->stereo_info, p->n_stereo); stage_copy(raw_stereo_huber, p->stereo_huber, p->n_stereo);
    parallel_for(p->n_mono, [&](int k0, int k1) { for (int k = k0; k < k1; ++k) { mono_pose_k[k] = p->mono_pose[k]; s->hk_mono_point[k] = p->mono_point[k]; } });
    parallel_for(p->n_stereo, [&](int k0, int k1) { for (int k = k0; k < k1; ++k) { stereo_pose_k[k] = p->stereo_pose[k]; s->hk_stereo_point[k] = p->stereo_point[k]; } });
    ok = d_raw_mono_obs.upload(raw_mono_obs, s->stream) && d_raw_mono_info.upload(raw_mono_info, s->stream) && d_raw_mono_huber.upload(raw_mono_huber, s->stream) &&
         d_raw_stereo_obs.upload(raw_stereo_obs, s->stream) && d_raw_stereo_info.upload(raw_stereo_info, s->stream) && d_raw_stereo_huber.upload(raw_stereo_huber, s->stream) &&
         d_raw_mono_pose.upload(mono_pose_k, s->stream) && d_raw_mono_point.upload(s->hk_mono_point, s->stream) &&
         d_raw_stereo_pose.upload(stereo_pose_k, s->stream) && d_raw_stereo_point.upload(s->hk_stereo_point, s->stream);
  } else {
  s->h_mono.resize(p->n_mono);
  h_mono_flags.resize(p->n_mono); h_stereo_flags.resize(p->n_stereo);
  parallel_for(p->n_mono, [&](int k0, int k1) {
  for (int k = k0; k < k1; ++k) {
    const int e = s->mono_perm[k];
    MonoEdge& m = s->h_mono[k];
    m.u = p->mono_obs[2 * e]; m.v = p->mono_obs[2 * e + 1]; m.info = p->mono_info[e]; m.huber = p->mono_huber[e];
    m.pose = p->mono_pose[e];
    m.slot = point_fixed[p->mono_point[e]] ? -1 : find_slot(p->mono_point[e], pose_col[m.pose]);
    const bool all_fixed = point_fixed[p->mono_point[e]] && pose_col[m.pose] < 0;
    m.flags = (all_fixed ? 0u : F_ACTIVE) | F_ROBUST;
    m.point = p->mono_point[e];
    h_mono_flags[k] = m.flags;
    mono_pose_k[k] = m.pose;
    s->hk_mono_point[k] = m.point;
  }
  });
  s->h_stereo.resize(p->n_stereo);
  parallel_for(p->n_stereo, [&](int k0, int k1) {
  for (int k = k0; k < k1; ++k) {
    const int e = s->stereo_perm[k];
    StereoEdge& m = s->h_stereo[k];
    std::memset(&m, 0, sizeof(m));
    m.u = p->stereo_obs[3 * e]; m.v = p->stereo_obs[3 * e + 1]; m.ur = p->stereo_obs[3 * e + 2];
    m.info = p->stereo_info[e]; m.huber = p->stereo_huber[e];
    m.pose = p->stereo_pose[e];
    m.point = p->stereo_point[e];
    m.slot = point_fixed[p->stereo_point[e]] ? -1 : find_slot(p->stereo_point[e], pose_col[m.pose]);
    const bool all_fixed = point_fixed[p->stereo_point[e]] && pose_col[m.pose] < 0;
    m.flags = (all_fixed ? 0u : F_ACTIVE) | F_ROBUST;
    h_stereo_flags[k] = m.flags;
    stereo_pose_k[k] = m.pose;
    s->hk_stereo_point[k] = m.point;
  }
  });
  }  // (host builder)
  mark("reprojection edge records");
  s->h_pl.resize(p->n_plane_edges);
  for (int e = 0; e < p->n_plane_edges; ++e) {
    PlaneEdge& m = s->h_pl[e];
    std::memset(&m, 0, sizeof(m));
    canon_plane_cv(&p->pl_meas[4 * e], m.meas);
    for (int i = 0; i < 3; ++i) m.info[i] = p->pl_info[3 * e + i];
    m.huber = p->pl_huber[e];
    m.plane = p->pl_plane[e]; m.pose = p->pl_pose[e];
    m.slot = dev_plan ? -1 : find_slot(Npt + m.plane, pose_col[m.pose]);  // (device builder: k_plan_plane_edge_slots)
    m.flags = F_ACTIVE | F_ROBUST | ((uint32_t)(p->pl_kind ? p->pl_kind[e] : 0) << 8);
  }
  auto fill_cub = [&](SVec<CubEdge>& v, int n, const int32_t* pose, const int32_t* cub, const double* meas, const double* info,
                      const double* hub, const double* K, int dim) {
    v.resize(n);
    for (int e = 0; e < n; ++e) {
      CubEdge& m = v[e];
      std::memset(&m, 0, sizeof(m));
      for (int i = 0; i < dim; ++i) { m.meas[i] = meas[dim * e + i]; m.info[i] = info[dim * e + i]; }
      m.huber = hub[e];
      for (int i = 0; i < 9; ++i) m.K[i] = K[9 * e + i];
      m.pose = pose[e]; m.cuboid = cub[e]; m.dim = dim;
      m.flags = F_ACTIVE | F_ROBUST;
    }
  };
  fill_cub(s->h_bbox, p->n_bbox, p->bbox_pose, p->bbox_cuboid, p->bbox_meas, p->bbox_info, p->bbox_huber, p->bbox_K, 4);
  fill_cub(s->h_corner, p->n_corner, p->corner_pose, p->corner_cuboid, p->corner_meas, p->corner_info, p->corner_huber, p->corner_K, 16);
  s->h_corner.resize(n_corner_all);
  for (int e = 0; e < p->n_cub3d; ++e) {  // EdgeSE3Cuboid: measured cuboid (10 values as given, like MapCuboid::cuboid_local_meas), 9 information entries
    CubEdge& m = s->h_corner[p->n_corner + e];
    std::memset(&m, 0, sizeof(m));
    for (int i = 0; i < 10; ++i) m.meas[i] = p->cub3d_meas[10 * (size_t)e + i];
    for (int i = 0; i < 9; ++i) m.info[i] = p->cub3d_info[9 * (size_t)e + i];
    m.huber = p->cub3d_huber[e];
    m.pose = p->cub3d_pose[e]; m.cuboid = p->cub3d_cuboid[e]; m.dim = 9;
    m.flags = F_ACTIVE | F_ROBUST;
  }
  s->h_ptcub.resize(p->n_ptcub);
  for (int e = 0; e < p->n_ptcub; ++e) {
    PtCubEdge& m = s->h_ptcub[e];
    std::memset(&m, 0, sizeof(m));
    m.margin = p->ptcub_margin_ratio[e];
    for (int i = 0; i < 3; ++i) { m.prior[i] = p->ptcub_prior_half ? p->ptcub_prior_half[3 * e + i] : 0.0; m.info[i] = p->ptcub_info[3 * e + i]; }
    m.huber = p->ptcub_huber ? p->ptcub_huber[e] : 0.0;
    m.cuboid = p->ptcub_cuboid[e]; m.p0 = p->ptcub_offsets[e]; m.p1 = p->ptcub_offsets[e + 1];
    m.flags = F_ACTIVE | (m.huber > 0 ? F_ROBUST : 0u);
  }
  // EdgeCuboidPlane: constant error (G2O_Plane3D.h:470-473).  Its chi2 and robustified value never change and its
  // numeric Jacobians are identically zero, so it only shifts every chi2 by a constant.
  s->cubpl_chi2.resize(p->n_cubpl);
  s->cubpl_rho_sum = 0;
  for (int e = 0; e < p->n_cubpl; ++e) {
    double c = 0;
    for (int i = 0; i < 3; ++i) c += p->cubpl_meas[3 * e + i] * (p->cubpl_info[3 * e + i] * p->cubpl_meas[3 * e + i]);
    s->cubpl_chi2[e] = c;
    double r0, r1;
    huber(c, p->cubpl_huber[e], r0, r1);
    s->cubpl_rho_sum += r0;
  }
  // (the constant chi2 of the cuboid-plane edges is added on the host by EVERY rank: all ranks must see the same totals,
  //  their termination tests compare absolute chi2 values)

  mark("other edge records");
  // the copy engine moves the landmark-major records (19 MB at the BASELINE size) while the host builds the rest
  if (dev_plan) ok = ok && s->d_mono.alloc(p->n_mono) && s->d_stereo.alloc(p->n_stereo);
  else ok = ok && s->d_mono.upload(s->h_mono, s->stream) && s->d_stereo.upload(s->h_stereo, s->stream);
  // ---- pose-major static copies of the owned reprojection edges + chunks (pose side of the linearisation) -----------
  SVec<int32_t> po_start, po_end, po_pose;
  SVec<int32_t> monoP_src, stereoP_src;  // position in the landmark-major array of the pose-major copy's q-th edge (expanded on the device)
  int32_t n_monoP = 0, n_stereoP = 0;
  {
    const int chunk = []() { const char* e = PPO_DEV_ENV("PPO_BA_POSE_CHUNK"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    auto build = [&](int type) {
      SVec<int32_t> cnt;
      const SVec<int32_t>& off = type == 0 ? mo_off : st_off;
      SVec<int32_t>& src = type == 0 ? monoP_src : stereoP_src;
      const int32_t* const pose_k = type == 0 ? mono_pose_k.data() : stereo_pose_k.data();
      auto pose_of = [&](int k) { return pose_k[k]; };
      bucket_by_key(g.pt_last - g.pt_first, Np, cnt,
                    [&](int u0, int u1, int32_t* h) {
                      for (int k = off[g.pt_first + u0]; k < off[g.pt_first + u1]; ++k)
                        if (pose_free_id[pose_of(k)] >= 0) h[pose_of(k)]++;
                    },
                    [&](int32_t total) { if (!dev_plan) src.resize(total); (type == 0 ? n_monoP : n_stereoP) = total; },
                    [&](int u0, int u1, int32_t* c) {
                      if (dev_plan) return;  // (the device sorts the edge positions by pose itself: only the counts are needed here)
                      for (int k = off[g.pt_first + u0]; k < off[g.pt_first + u1]; ++k) {
                        const int pose = pose_of(k);
                        if (pose_free_id[pose] >= 0) src[c[pose]++] = k;
                      }
                    });
      for (int i = 0; i < Np; ++i)
        for (int32_t e = cnt[i]; e < cnt[i + 1]; e += chunk) {
          po_pose.push_back((i << 1) | type);
          po_start.push_back(e);
          po_end.push_back(std::min<int32_t>(e + chunk, cnt[i + 1]));
        }
    };
    build(0);
    build(1);
  }
  mark("  pose-major index lists");
  ok = ok && s->d_monoP.alloc(n_monoP) && s->d_stereoP.alloc(n_stereoP);
  if (!dev_plan) {
    ok = ok && s->d_monoP_src.upload(monoP_src, s->stream) && s->d_stereoP_src.upload(stereoP_src, s->stream);
    if (ok && n_monoP + n_stereoP > 0) {
      if (flush_uploads(s->stream) != PPO_BA_OK) return fail(PPO_BA_ERR_HIP);
      hipLaunchKernelGGL(k_expand_pose_major, dim3(std::min(nblk((long)std::max(n_monoP, n_stereoP), 256), 4 * s->n_cu)), dim3(256), 0, s->stream,
                         s->d_mono.p, s->d_monoP_src.p, s->d_monoP.p, (int)n_monoP, s->d_stereo.p, s->d_stereoP_src.p, s->d_stereoP.p, (int)n_stereoP);
    }
  }

  mark("  flags");
  // ---- cuboid elimination plan: edges per cuboid and all (a >= b) pairs of them -----------------------------------------
  SVec<int32_t> cub_edge_off(Nc + 1, 0), cub_edges, cub_edge_col, cp_cub, cp_a, cp_b, cp_cola, cp_colb, cp_off(Nc + 1, 0);
  {
    std::vector<std::vector<int32_t>> per(Nc);
    for (int e = 0; e < p->n_bbox; ++e) per[p->bbox_cuboid[e]].push_back(e << 1);
    for (int e = 0; e < n_corner_all; ++e) per[corner_cuboid_of(e)].push_back((e << 1) | 1);
    for (int c = 0; c < Nc; ++c) {
      cub_edge_off[c] = (int32_t)cub_edges.size();
      cub_edges.insert(cub_edges.end(), per[c].begin(), per[c].end());
    }
    cub_edge_off[Nc] = (int32_t)cub_edges.size();
    cub_edge_col.resize(cub_edges.size());  // reduced column of every listed edge's keyframe (static; saves the edge -> pose -> column walk of the cuboid back-substitution)
    for (size_t q = 0; q < cub_edges.size(); ++q) {
      const int32_t code = cub_edges[q];
      cub_edge_col[q] = pose_col[(code & 1) ? corner_pose_of(code >> 1) : p->bbox_pose[code >> 1]];
    }
    s->n_cub_edges = (int)cub_edges.size();
    if (g.elim_cub)
      for (int c = 0; c < Nc; ++c) {
        cp_off[c] = (int32_t)cp_cub.size();
        for (int a = cub_edge_off[c]; a < cub_edge_off[c + 1]; ++a)
          for (int b = cub_edge_off[c]; b <= a; ++b) {
            // a pair carries everything that does not change after create -- the two edge codes and their keyframes' columns -- so that the
            // kernel starts its operand loads one round trip after its first load (it used to walk pair -> edge list -> edge -> pose -> column)
            auto col_of = [&](int32_t code) { return pose_col[(code & 1) ? corner_pose_of(code >> 1) : p->bbox_pose[code >> 1]]; };
            cp_cub.push_back(c); cp_a.push_back(cub_edges[a]); cp_b.push_back(cub_edges[b]);
            cp_cola.push_back(col_of(cub_edges[a])); cp_colb.push_back(col_of(cub_edges[b]));
          }
      }
    cp_off[Nc] = (int32_t)cp_cub.size();
    if (!g.elim_cub) for (int c = 0; c <= Nc; ++c) cp_off[c] = 0;
  }

  // ---- initial estimates, canonicalised like the reference's converters ------------------------------------------
  s->state_doubles = (size_t)Np * 19 + (size_t)Npt * 3 + (size_t)Npl * 4 + (size_t)Nc * 10;
  SVec<double> h_state(s->state_doubles, 0.0);
  {
    State hs = make_state(h_state.data(), Np, Npt, Npl, Nc);
    for (int i = 0; i < Np; ++i) {
      double* q = &hs.pose_qt[7 * i];
      std::memcpy(q, &p->pose_qt[7 * i], 7 * sizeof(double));
      canon_quat(q);
    }
    if (Npt) std::memcpy(hs.point, p->point_xyz, (size_t)Npt * 3 * sizeof(double));
    for (int i = 0; i < Npl; ++i) canon_plane_cv(&p->plane_coef[4 * i], &hs.plane[4 * i]);
    for (int i = 0; i < Nc; ++i) {
      double* c = &hs.cub[10 * i];
      std::memcpy(c, &p->cuboid_qt[7 * i], 7 * sizeof(double));
      canon_quat(c);
      std::memcpy(c + 7, &p->cuboid_scale[3 * i], 3 * sizeof(double));
    }
  }

  mark("pose-major copies, cuboid plan, state");
  if (pair_plan_thread.joinable()) pair_plan_thread.join();
  if (pair_plan_failed.load()) return fail(PPO_BA_ERR_NOMEM);
  mark("pair plan join");
  // ---- uploads (the two big record arrays are already on their way) -----------------------------------------------------
  SVec<double> v_cam(p->pose_cam, p->pose_cam + (size_t)5 * Np);
  SVec<double> v_fs((size_t)3 * Nc, 0.0);
  if (p->cuboid_fixedscale) std::copy(p->cuboid_fixedscale, p->cuboid_fixedscale + (size_t)3 * Nc, v_fs.begin());
  SVec<uint8_t> v_cf(Nc, (uint8_t)(PPO_CUBOID_FIX_ROLLPITCH | PPO_CUBOID_FIX_HEIGHT));
  if (p->cuboid_flags) std::copy(p->cuboid_flags, p->cuboid_flags + Nc, v_cf.begin());
  if (o->jacobian_mode == PPO_JAC_ANALYTIC)
    for (int i = 0; i < Nc; ++i)
      if (v_cf[i] & PPO_CUBOID_FIX_ROTATION) {
        std::fprintf(stderr, "[ppo_ba] PPO_JAC_ANALYTIC does not support cuboids with PPO_CUBOID_FIX_ROTATION (cuboid %d)\n", i);
        return fail(PPO_BA_ERR_ARG);
      }
  SVec<double> v_pts;
  if (p->n_ptcub && p->ptcub_offsets[p->n_ptcub] > 0) v_pts.assign(p->ptcub_points, p->ptcub_points + (size_t)3 * p->ptcub_offsets[p->n_ptcub]);
  ok = ok && s->d_pose_cam.upload(v_cam, s->stream) && s->d_cub_fixedscale.upload(v_fs, s->stream) && s->d_ptcub_pts.upload(v_pts, s->stream);
  ok = ok && s->d_pose_col.upload(pose_col, s->stream) && s->d_pose_free_id.upload(pose_free_id, s->stream) && s->d_cub_col.upload(cub_col, s->stream);
  ok = ok && s->d_mo_off.upload(mo_off, s->stream) && s->d_st_off.upload(st_off, s->stream);
  if (dev_plan) {
    ok = ok && s->d_slot_col.alloc(g.n_slots) && s->d_slot_lm.alloc(g.n_slots) && s->d_pl_slot_off.alloc(Npl + 1) && s->d_pl_slots.alloc(p->n_plane_edges);
  } else {
    ok = ok && s->d_slot_col.upload(d_slot_col_v, s->stream) && s->d_slot_lm.upload(d_slot_lm_v, s->stream);
    ok = ok && s->d_pl_slot_off.upload(pl_slot_off, s->stream) && s->d_pl_slots.upload(pl_slots, s->stream);
  }
  if (dev_plan) {
    ok = ok && s->d_mono_flags.alloc(p->n_mono) && s->d_mono_flags0.alloc(p->n_mono) && s->d_stereo_flags.alloc(p->n_stereo) && s->d_stereo_flags0.alloc(p->n_stereo);
  } else {
    ok = ok && s->d_mono_flags.upload(h_mono_flags, s->stream) && s->d_mono_flags0.upload(h_mono_flags, s->stream);
    ok = ok && s->d_stereo_flags.upload(h_stereo_flags, s->stream) && s->d_stereo_flags0.upload(h_stereo_flags, s->stream);
  }
  ok = ok && s->d_point_fixed.upload(point_fixed, s->stream) && s->d_cub_flags.upload(v_cf, s->stream);
  ok = ok && s->d_pl.upload(s->h_pl, s->stream);
  ok = ok && s->d_bbox.upload(s->h_bbox, s->stream) && s->d_corner.upload(s->h_corner, s->stream) && s->d_ptcub.upload(s->h_ptcub, s->stream);
  ok = ok && s->d_pl0.upload(s->h_pl, s->stream);
  ok = ok && s->d_bbox0.upload(s->h_bbox, s->stream) && s->d_corner0.upload(s->h_corner, s->stream);
  ok = ok && s->d_init.upload(h_state, s->stream) && s->d_state[0].alloc(s->state_doubles) && s->d_state[1].alloc(s->state_doubles);
  const size_t n_chi = (size_t)p->n_mono + p->n_stereo + p->n_plane_edges + 2 * (size_t)p->n_bbox + 2 * (size_t)n_corner_all + p->n_ptcub;
  ok = ok && s->d_chi2_all.alloc(n_chi);
  // arena: HB | HC | acc | D | bl;  d_W: factor records of the point slots | linearisation-point rotations | W of the plane slots
  g.n_pt_slots = n_point_slots;  // (device builder: upper bounds here, the exact counts come back with the final synchronisation)
  const size_t nF = (size_t)g.n_pt_slots * 4, nLinR = (size_t)(ld / 6 + 2) * 12, nWpl = (size_t)(g.n_slots - g.n_pt_slots) * 18;
  const size_t nHB = (size_t)nfree * 27, nHC = (size_t)Nc * 54, nD = (size_t)g.n_lm * 6, nbl = (size_t)g.n_lm * 3, nW = 2 * (nF + nLinR + nWpl);
  ok = ok && s->d_arena.alloc(2 * (nHB + nHC + nD + nbl) + ACC_TOTAL) && s->d_W.alloc(nW);
  ok = ok && s->d_po_start.upload(po_start, s->stream) && s->d_po_end.upload(po_end, s->stream) && s->d_po_pose.upload(po_pose, s->stream);
  s->posec = PoseChunks{s->d_po_start.p, s->d_po_end.p, s->d_po_pose.p, (int)po_pose.size()};
  ok = ok && s->d_Wpc.alloc(2 * (size_t)(p->n_bbox + n_corner_all) * 54) && s->d_Dinv.alloc(nD) && s->d_ul.alloc(nbl);
  if (o->deterministic) ok = ok && s->d_Sfix.alloc((size_t)ld * ld + ld) && s->d_lin_tag.alloc(2);
  ok = ok && s->d_S.alloc((size_t)ld * ld + ld) && s->d_xr.alloc(ld) && s->d_info.alloc(1) && s->d_ctl.alloc(1) && s->d_rec.alloc((size_t)kRecSlots * REC_DOUBLES) && s->d_tailcnt.alloc(80);
  if (s->comm.active()) ok = ok && s->d_pack.alloc((size_t)(ld / 64) * (ld / 64 + 1) / 2 * 4096 + ld);  // S, then b right behind it: ONE all-reduce when sharded
  ok = ok && s->d_cholws.alloc(chol_workspace_doubles(ncol));
  ok = ok && s->d_cub_edge_off.upload(cub_edge_off, s->stream) && s->d_cub_edges.upload(cub_edges, s->stream) && s->d_cub_edge_col.upload(cub_edge_col, s->stream);
  ok = ok && s->d_cp_cub.upload(cp_cub, s->stream) && s->d_cp_a.upload(cp_a, s->stream) && s->d_cp_b.upload(cp_b, s->stream) &&
       s->d_cp_cola.upload(cp_cola, s->stream) && s->d_cp_colb.upload(cp_colb, s->stream) && s->d_cp_off.upload(cp_off, s->stream);
  ok = ok && s->d_cub_Linv.alloc((size_t)Nc * 81) && s->d_cub_Dinv.alloc((size_t)Nc * 81) && s->d_xc.alloc((size_t)Nc * 9);
  PlanScratch plan_scratch;
  ScopedDevBuf<int32_t> d_slot_off_lm, d_slot_col_lm, d_new_id, d_slot_lm_lm, d_pair_off, d_npairs, d_key_count, d_key_start, d_nchunks, d_chunk_off, d_n_chunks_out, d_cb;
  ScopedDevBuf<int32_t> d_pl_off, d_pl_perm, d_pl_pose, d_seg, d_seg_cnt, d_counts;
  ScopedDevBuf<uint32_t> d_pose_key, d_pose_key_out, d_pair_key, d_pair_key_out, d_pair_perm, d_slot_key, d_slot_key_out, d_slot_order;
  int32_t* h_n_chunks = nullptr;  // (page-locked: the device's chunk count comes back with the final synchronisation)
  if (!dev_plan) {
    ok = ok && s->d_pc_start.upload(pc_start, s->stream) && s->d_pc_colj.upload(pc_colj, s->stream) && s->d_pc_coli.upload(pc_coli, s->stream);
    ok = ok && s->d_pc_sj.upload(pc_sj, s->stream) && s->d_pc_si.upload(pc_si, s->stream) && s->d_pc_lm.upload(pc_lm, s->stream);
    s->pc = PairChunks{s->d_pc_start.p, s->d_pc_colj.p, s->d_pc_coli.p, s->d_pc_sj.p, s->d_pc_si.p, s->d_pc_lm.p, (int)pc_colj.size()};
    s->cb_start.assign(g.ld / 64 + 1, 0);
    for (int c = 0; c <= g.ld / 64; ++c) s->cb_start[c] = (int32_t)(std::lower_bound(pc_coli.begin(), pc_coli.end(), (int32_t)(64 * c)) - pc_coli.begin());
  } else {
    // ---- device builder (ppo_plan.hpp): records + flags, pose-major index lists, Schur pair plan ------------------------------------
    // (landmark, pose) slots: gather + sort + unique per landmark, scan, compact, pose-major numbering by a stable sort
    const int n_slot_bound = g.n_slots, pose_bits = plan_bits((unsigned long long)Np);
    ok = ok && d_pl_off.upload(pl_off, s->stream) && d_pl_perm.upload(pl_perm, s->stream);
    {
      SVec<int32_t> pl_pose_v(p->pl_pose, p->pl_pose + p->n_plane_edges);
      ok = ok && d_pl_pose.upload(pl_pose_v, s->stream);
    }
    ok = ok && d_seg.alloc(n_slot_bound) && d_seg_cnt.alloc(g.n_lm + 1) && d_slot_off_lm.alloc(g.n_lm + 1) && d_slot_col_lm.alloc(n_slot_bound) && d_slot_lm_lm.alloc(n_slot_bound) &&
         d_new_id.alloc(n_slot_bound) && d_counts.alloc(4) && d_slot_key.alloc(n_slot_bound) && d_slot_key_out.alloc(n_slot_bound) && d_slot_order.alloc(n_slot_bound);
    if (!ok) return fail(PPO_BA_ERR_NOMEM);
    const SlotInputs sin{s->d_mo_off.p, s->d_st_off.p, d_raw_mono_pose.p, d_raw_stereo_pose.p, d_pl_off.p, d_pl_perm.p, d_pl_pose.p, s->d_pose_col.p, s->d_point_fixed.p,
                         Npt, Npl, p->n_mono, p->n_stereo};
    if (flush_uploads(s->stream) != PPO_BA_OK) return fail(PPO_BA_ERR_HIP);
    hipLaunchKernelGGL(k_plan_slot_gather, dim3(nblk(g.n_lm + 1, 256)), dim3(256), 0, s->stream, sin, d_seg.p, d_seg_cnt.p);
    if (!plan_exclusive_scan(plan_scratch, d_seg_cnt.p, d_slot_off_lm.p, (size_t)g.n_lm + 1, s->stream)) return fail(PPO_BA_ERR_NOMEM);
    hipLaunchKernelGGL(k_plan_slot_compact, dim3(nblk(std::max(g.n_lm, 1), 256)), dim3(256), 0, s->stream, sin, d_seg.p, d_slot_off_lm.p, d_slot_col_lm.p, d_slot_lm_lm.p, d_counts.p);
    if (n_slot_bound > 0) {
      hipLaunchKernelGGL(k_plan_slot_keys, dim3(nblk(n_slot_bound, 256)), dim3(256), 0, s->stream, d_slot_col_lm.p, d_slot_lm_lm.p, d_counts.p, n_slot_bound, Npt, pose_bits, d_slot_key.p);
      if (!plan_sort_positions(plan_scratch, d_slot_key.p, d_slot_key_out.p, d_slot_order.p, (size_t)n_slot_bound, pose_bits + 2, s->stream)) return fail(PPO_BA_ERR_NOMEM);
      hipLaunchKernelGGL(k_plan_slot_number, dim3(nblk(n_slot_bound, 256)), dim3(256), 0, s->stream, d_slot_order.p, d_slot_col_lm.p, d_slot_lm_lm.p, d_counts.p, s->d_slot_col.p,
                         s->d_slot_lm.p, d_new_id.p);
    }
    hipLaunchKernelGGL(k_plan_plane_lists, dim3(nblk(std::max(Npl + 1, p->n_plane_edges), 256)), dim3(256), 0, s->stream, d_slot_off_lm.p, d_new_id.p, d_counts.p, Npt, Npl,
                       s->d_pl_slot_off.p, s->d_pl_slots.p);
    if (p->n_plane_edges > 0)
      hipLaunchKernelGGL(k_plan_plane_edge_slots, dim3(nblk(p->n_plane_edges, 256)), dim3(256), 0, s->stream, s->d_pl.p, s->d_pl0.p, p->n_plane_edges, d_slot_off_lm.p, d_slot_col_lm.p,
                         d_new_id.p, s->d_pose_col.p, Npt);
    const SlotTables tabs{d_slot_off_lm.p, d_slot_col_lm.p, d_new_id.p, s->d_pose_col.p, s->d_point_fixed.p};
    const ReprojRaw rm{d_raw_mono_obs.p, d_raw_mono_info.p, d_raw_mono_huber.p, d_raw_mono_pose.p, d_raw_mono_point.p, p->n_mono};
    const ReprojRaw rs{d_raw_stereo_obs.p, d_raw_stereo_info.p, d_raw_stereo_huber.p, d_raw_stereo_pose.p, d_raw_stereo_point.p, p->n_stereo};
    hipLaunchKernelGGL(k_plan_records, dim3(std::min(nblk((long)std::max(p->n_mono, p->n_stereo), 256), 4 * s->n_cu)), dim3(256), 0, s->stream, rm, rs, tabs, s->d_mono.p,
                       s->d_stereo.p, s->d_mono_flags.p, s->d_mono_flags0.p, s->d_stereo_flags.p, s->d_stereo_flags0.p);
    // pose-major copies: edge positions sorted by pose (stable: landmark order inside a keyframe), edges that do not belong at the end
    const size_t n_key = (size_t)std::max(p->n_mono, p->n_stereo);
    ok = ok && d_pose_key.alloc(n_key) && d_pose_key_out.alloc(n_key) && s->d_monoP_src.alloc(p->n_mono) && s->d_stereoP_src.alloc(p->n_stereo);
    if (!ok) return fail(PPO_BA_ERR_NOMEM);
    for (int type = 0; type < 2; ++type) {
      const int n = type == 0 ? p->n_mono : p->n_stereo;
      if (n == 0) continue;
      hipLaunchKernelGGL(k_plan_pose_keys, dim3(nblk(n, 256)), dim3(256), 0, s->stream, type == 0 ? d_raw_mono_pose.p : d_raw_stereo_pose.p,
                         type == 0 ? d_raw_mono_point.p : d_raw_stereo_point.p, n, s->d_pose_free_id.p, g.pt_first, g.pt_last, (uint32_t)Np, d_pose_key.p);
      if (!plan_sort_positions(plan_scratch, d_pose_key.p, d_pose_key_out.p, reinterpret_cast<uint32_t*>(type == 0 ? s->d_monoP_src.p : s->d_stereoP_src.p), (size_t)n, pose_bits, s->stream))
        return fail(PPO_BA_ERR_NOMEM);
    }
    if (n_monoP + n_stereoP > 0) {
      if (flush_uploads(s->stream) != PPO_BA_OK) return fail(PPO_BA_ERR_HIP);
      hipLaunchKernelGGL(k_expand_pose_major, dim3(std::min(nblk((long)std::max(n_monoP, n_stereoP), 256), 4 * s->n_cu)), dim3(256), 0, s->stream,
                         s->d_mono.p, s->d_monoP_src.p, s->d_monoP.p, (int)n_monoP, s->d_stereo.p, s->d_stereoP_src.p, s->d_stereoP.p, (int)n_stereoP);
    }
    // Schur pair plan
    OwnedLandmarks own{Npt, g.pt_first, g.pt_last - g.pt_first, g.plane_first, (g.pt_last - g.pt_first) + (g.plane_last - g.plane_first)};
    long n_pairs = 0;  // (upper bound: every edge of a landmark a slot; the device knows the exact number, positions past it sort to the end)
    for (int u = 0; u < own.n_own; ++u) {
      const int l = own.lm_of(u);
      const long k = l < Npt ? (long)(mo_off[l + 1] - mo_off[l]) + (st_off[l + 1] - st_off[l]) : (long)(pl_off[l - Npt + 1] - pl_off[l - Npt]);
      n_pairs += k * (k + 1) / 2;
    }
    const long nkeys = (long)nfree * (nfree + 1) / 2;
    const long max_chunks = nkeys + n_pairs / pair_chunk + 1;
    if (n_pairs >= (1L << 31) || nkeys >= (1L << 31)) return fail(PPO_BA_ERR_ARG);
    ok = ok && d_npairs.alloc(own.n_own + 1) && d_pair_off.alloc(own.n_own + 1) && d_key_count.alloc(nkeys + 1) && d_key_start.alloc(nkeys + 1) && d_nchunks.alloc(nkeys + 1) &&
         d_chunk_off.alloc(nkeys + 1) && d_n_chunks_out.alloc(1) && d_pair_key.alloc(n_pairs) && d_pair_key_out.alloc(n_pairs) && d_pair_perm.alloc(n_pairs);
    ok = ok && s->d_pc_sj.alloc(n_pairs) && s->d_pc_si.alloc(n_pairs) && s->d_pc_lm.alloc(n_pairs) && s->d_pc_colj.alloc(max_chunks) && s->d_pc_coli.alloc(max_chunks) &&
         s->d_pc_start.alloc(max_chunks + 1);
    if (!ok) return fail(PPO_BA_ERR_NOMEM);
    (void)hipMemsetAsync(d_key_count.p, 0, (size_t)(nkeys + 1) * sizeof(int32_t), s->stream);
    (void)hipMemsetAsync(d_n_chunks_out.p, 0, sizeof(int32_t), s->stream);
    if (own.n_own > 0 && nkeys > 0) {
      hipLaunchKernelGGL(k_plan_pair_counts, dim3(nblk(own.n_own + 1, 256)), dim3(256), 0, s->stream, own, d_slot_off_lm.p, d_npairs.p);
      if (!plan_exclusive_scan(plan_scratch, d_npairs.p, d_pair_off.p, (size_t)own.n_own + 1, s->stream)) return fail(PPO_BA_ERR_NOMEM);
      if (n_slot_bound > 0)
        hipLaunchKernelGGL(k_plan_pair_keys, dim3(nblk(n_slot_bound, 256)), dim3(256), 0, s->stream, own, d_slot_off_lm.p, d_slot_col_lm.p, d_slot_lm_lm.p, d_counts.p, d_pair_off.p,
                           (uint32_t)nfree, d_pair_key.p, d_key_count.p);
      if (n_pairs > 0) {
        hipLaunchKernelGGL(k_plan_fill_u32, dim3(nblk(n_pairs, 256)), dim3(256), 0, s->stream, d_pair_key.p, d_pair_off.p + own.n_own, (int)n_pairs, (uint32_t)nkeys);
        if (!plan_sort_positions(plan_scratch, d_pair_key.p, d_pair_key_out.p, d_pair_perm.p, (size_t)n_pairs, plan_bits((unsigned long long)nkeys), s->stream)) return fail(PPO_BA_ERR_NOMEM);
        hipLaunchKernelGGL(k_plan_pair_fill, dim3(nblk(n_pairs, 256)), dim3(256), 0, s->stream, own, d_slot_off_lm.p, d_new_id.p, d_pair_off.p, d_pair_perm.p, s->d_pc_sj.p,
                           s->d_pc_si.p, s->d_pc_lm.p);
      }
      hipLaunchKernelGGL(k_plan_chunk_counts, dim3(nblk(nkeys + 1, 256)), dim3(256), 0, s->stream, d_key_count.p, (int)nkeys, pair_chunk, d_nchunks.p);
      if (!plan_exclusive_scan(plan_scratch, d_key_count.p, d_key_start.p, (size_t)nkeys + 1, s->stream) ||
          !plan_exclusive_scan(plan_scratch, d_nchunks.p, d_chunk_off.p, (size_t)nkeys + 1, s->stream))
        return fail(PPO_BA_ERR_NOMEM);
      hipLaunchKernelGGL(k_plan_chunk_fill, dim3(nblk(nkeys + 1, 256)), dim3(256), 0, s->stream, d_key_start.p, d_chunk_off.p, (int)nkeys, (int)nfree, pair_chunk, s->d_pc_colj.p, s->d_pc_coli.p,
                         s->d_pc_start.p, d_n_chunks_out.p);
    }
    const int n_cb = g.ld / 64 + 1;
    h_n_chunks = static_cast<int32_t*>(s->stage->take((4 + (size_t)n_cb) * sizeof(int32_t)));  // [0] chunks, [1] slots, [2] point slots, [4 ..] first chunk of every column block
    if (!h_n_chunks || !d_cb.alloc(n_cb)) return fail(PPO_BA_ERR_NOMEM);
    for (int i = 0; i < 4 + n_cb; ++i) h_n_chunks[i] = 0;
    if (own.n_own > 0 && nkeys > 0) {
      hipLaunchKernelGGL(k_plan_col_blocks, dim3(nblk(n_cb, 64)), dim3(64), 0, s->stream, s->d_pc_coli.p, d_n_chunks_out.p, n_cb - 1, d_cb.p);
      if (hipMemcpyAsync(h_n_chunks + 4, d_cb.p, (size_t)n_cb * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess) return fail(PPO_BA_ERR_HIP);
    }
    // (never HIP_TRY / RC_TRY once `s` exists: a plain return would leak the half-built solver and leave t_creating dangling)
    if (hipMemcpyAsync(h_n_chunks, d_n_chunks_out.p, sizeof(int32_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess) return fail(PPO_BA_ERR_HIP);
    if (hipMemcpyAsync(h_n_chunks + 1, d_counts.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess) return fail(PPO_BA_ERR_HIP);
    s->pc = PairChunks{s->d_pc_start.p, s->d_pc_colj.p, s->d_pc_coli.p, s->d_pc_sj.p, s->d_pc_si.p, s->d_pc_lm.p, 0};  // (n_chunks: after the synchronisation below)
  }
  if (flush_uploads(s->stream) != PPO_BA_OK) return fail(PPO_BA_ERR_HIP);
  mark("hipMalloc + H2D enqueue");
  if (!ok) return fail(PPO_BA_ERR_NOMEM);
  {  // blocks come from the cache with stale contents: the work arrays that are only partially written start from zero
    // (round 5: two launches of k_zero_ranges instead of eleven hipMemsetAsync -- each of those took the runtime's blit path, k_reset's comment)
    ZeroPlan zp{};
    int nz = 0;
    auto flush_zero = [&]() {
      if (nz == 0) return;
      size_t total = 0;
      for (int i = 0; i < nz; ++i) total += zp.n[i];
      hipLaunchKernelGGL(k_zero_ranges, dim3(std::max(1, std::min(nblk((long)total / 2, 256), 2 * s->n_cu))), dim3(256), 0, s->stream, zp);
      zp = ZeroPlan{};
      nz = 0;
    };
    auto zero = [&](DevBuf<double>& b) {
      if (!(b.p && b.n)) return;
      zp.p[nz] = b.p; zp.n[nz] = b.n;
      if (++nz == 6) flush_zero();
    };
    zero(s->d_Wpc); zero(s->d_Dinv); zero(s->d_ul); zero(s->d_xr); zero(s->d_cholws);
    zero(s->d_cub_Linv); zero(s->d_cub_Dinv); zero(s->d_xc); zero(s->d_S); zero(s->d_Sfix); zero(s->d_lin_tag);
    static_assert(sizeof(int) * 2 == sizeof(double), "the two int blocks below are cleared as doubles");
    zp.p[nz] = reinterpret_cast<double*>(s->d_tailcnt.p); zp.n[nz] = 40;  // 80 ints
    if (++nz == 6) flush_zero();
    flush_zero();
  }
  static_assert(ACC_N + 4 <= 16 && REC_DOUBLES == kRingSlotDoubles && kRecSlots == kRingSlots && kLookahead <= kRecSlots, "mapped host block too small");
  (void)hipMemsetAsync(s->d_info.p, 0, sizeof(int), s->stream);

  g.pose_cam = s->d_pose_cam.p; g.pose_col = s->d_pose_col.p; g.pose_free_id = s->d_pose_free_id.p; g.cub_col = s->d_cub_col.p;
  g.point_fixed = s->d_point_fixed.p; g.cub_flags = s->d_cub_flags.p; g.cub_fixedscale = s->d_cub_fixedscale.p;
  g.mo_off = s->d_mo_off.p; g.st_off = s->d_st_off.p;
  g.mono = s->d_mono.p; g.stereo = s->d_stereo.p; g.pl = s->d_pl.p; g.bbox = s->d_bbox.p; g.corner = s->d_corner.p; g.ptcub = s->d_ptcub.p;
  g.ptcub_pts = s->d_ptcub_pts.p;
  g.slot_col = s->d_slot_col.p; g.slot_lm = s->d_slot_lm.p; g.pl_slot_off = s->d_pl_slot_off.p; g.pl_slots = s->d_pl_slots.p;
  g.mono_flags = s->d_mono_flags.p; g.stereo_flags = s->d_stereo_flags.p; g.monoP = s->d_monoP.p; g.stereoP = s->d_stereoP.p;
  {
    double* c = s->d_chi2_all.p;
    g.mono_chi2 = c; c += p->n_mono;
    g.stereo_chi2 = c; c += p->n_stereo;
    g.pl_chi2 = c; c += p->n_plane_edges;
    g.bbox_chi2 = c; c += p->n_bbox;
    g.bbox_norm = c; c += p->n_bbox;
    g.corner_chi2 = c; c += n_corner_all;
    g.corner_norm = c; c += n_corner_all;
    g.ptcub_chi2 = c;
  }
  {
    double* a = s->d_arena.p;  // [H_pp | H_cc | D | b_l] x 2, then the accumulator block
    for (int b = 0; b < 2; ++b) {
      s->lb[b].HB = a; a += nHB;
      s->lb[b].HC = a; a += nHC;
      s->lb[b].D = a; a += nD;
      s->lb[b].bl = a; a += nbl;
    }
    g.acc = a; a += ACC_TOTAL;
    s->nHBC = nHB + nHC; s->zD_n = (size_t)Npl * 6; s->zbl_n = (size_t)Npl * 3; s->zW_n = nWpl;
    {  // d_W: [the planes' W | the planes' W] (cleared together), then [factor records | linearisation-point rotations] x 2
      double* w = s->d_W.p;
      s->lb[0].W = w; s->lb[1].W = w + nWpl; w += 2 * nWpl;
      for (int b = 0; b < 2; ++b) { s->lb[b].F = w; w += nF; s->lb[b].linR = w; w += nLinR; }
      double* c = s->d_Wpc.p;
      for (int b = 0; b < 2; ++b) { s->lb[b].Wpc_bbox = c; c += (size_t)p->n_bbox * 54; s->lb[b].Wpc_corner = c; c += (size_t)n_corner_all * 54; }
    }
    use_lin_buffer(s, 0);
    if (nW) (void)hipMemsetAsync(s->d_W.p, 0, nW * sizeof(double), s->stream);
  }
  g.cub_edge_off = s->d_cub_edge_off.p; g.cub_edges = s->d_cub_edges.p; g.cub_edge_col = s->d_cub_edge_col.p; g.cp_cub = s->d_cp_cub.p; g.cp_a = s->d_cp_a.p; g.cp_b = s->d_cp_b.p; g.cp_cola = s->d_cp_cola.p; g.cp_colb = s->d_cp_colb.p; g.cp_off = s->d_cp_off.p;
  g.n_cub_pairs = (int)cp_cub.size(); g.cub_Linv = s->d_cub_Linv.p; g.cub_Dinv = s->d_cub_Dinv.p; g.xc = s->d_xc.p;
  g.Dinv = s->d_Dinv.p; g.ul = s->d_ul.p; s->d_bs_p = s->d_S.p + (size_t)ld * ld; s->d_bsfix_p = s->d_Sfix.p ? s->d_Sfix.p + (size_t)ld * ld : nullptr; g.S = s->d_S.p; g.bs = s->d_bs_p; g.xr = s->d_xr.p;
  s->st[0] = make_state(s->d_state[0].p, Np, Npt, Npl, Nc);
  s->st[1] = make_state(s->d_state[1].p, Np, Npt, Npl, Nc);

  // LDS plan for k_lin_points: camera table + pose-block accumulators
  s->lds_bytes = (int)((size_t)Np * kCamRow * sizeof(double));
  s->use_lds = s->lds_bytes <= 150 * 1024;
  if (s->use_lds && s->lds_bytes > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)k_lin_points<true>, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_backsub<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_update_all<true>, hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes) != hipSuccess) s->use_lds = false;
  }
  s->use_lds_lin = s->use_lds && s->lds_bytes <= 40 * 1024;
  s->lin_lds_bytes = std::max({(int)(4 * kPoseWaveLds * sizeof(double)), (int)((g.jac ? std::max(kNumericLdsDoubles, kAnalyticLdsDoubles) : kNumericLdsDoubles) * sizeof(double)),
                               s->use_lds_lin ? s->lds_bytes + 64 : 64});
  // derive the rotation matrices of the pristine state once
  {
    State is = make_state(s->d_init.p, Np, Npt, Npl, Nc);
    if (Np) hipLaunchKernelGGL(k_pose_rt, dim3(nblk(Np, 256)), dim3(256), 0, s->stream, is, Np);
  }
  if (reset_device(s) != PPO_BA_OK) return fail(PPO_BA_ERR_HIP);
  if (dev_plan) {  // (reset_device has synchronised the stream: the plan is complete, its temporaries can go back to the block cache)
    s->pc.n_chunks = h_n_chunks[0];
    s->cb_start.assign(h_n_chunks + 4, h_n_chunks + 4 + s->g.ld / 64 + 1);
    s->g.n_slots = h_n_chunks[1];
    s->g.n_pt_slots = h_n_chunks[2];
    plan_scratch.release();
    for (DevBuf<int32_t>* b : {&d_slot_off_lm, &d_slot_col_lm, &d_new_id, &d_slot_lm_lm, &d_pair_off, &d_npairs, &d_key_count, &d_key_start, &d_nchunks, &d_chunk_off, &d_n_chunks_out, &d_cb,
                               &d_raw_mono_pose, &d_raw_mono_point, &d_raw_stereo_pose, &d_raw_stereo_point, &d_pl_off, &d_pl_perm, &d_pl_pose, &d_seg, &d_seg_cnt, &d_counts}) b->release();
    for (DevBuf<uint32_t>* b : {&d_pose_key, &d_pose_key_out, &d_pair_key, &d_pair_key_out, &d_pair_perm, &d_slot_key, &d_slot_key_out, &d_slot_order}) b->release();
    for (DevBuf<double>* b : {&d_raw_mono_obs, &d_raw_mono_info, &d_raw_mono_huber, &d_raw_stereo_obs, &d_raw_stereo_info, &d_raw_stereo_huber}) b->release();
  }
  mark("reset + sync");
  if (o->verbose) {
    size_t cap = 0, used = 0;
    for (const auto& b : s->stage->blocks) { cap += b.cap; used += b.used; }
    std::fprintf(stderr, "[ppo_ba] create: page-locked staging held by this handle %.1f MB (%.1f MB used)\n", cap / 1e6, used / 1e6);
  }

  {  // schedule of the overlapped reduction (ppo_chol.hip): worth its bookkeeping when the factorisation has enough launches to hide the pair chunks in
    // (measured, tools/overlap_ab.sh: config 4, 19 panels, 0.412 -> 0.388 ms per LM iteration; config 3, 10 panels, 0.231 -> 0.224; config 2, 5 panels, no difference)
    // round 4 (tools/overlap_ab.sh, tools/small_window_trace.py): it also pays on FEW panels -- the fused first stage saves the launch of the pair
    // stage, and with larger shares per launch the pair chunks still hide behind the panels: config 2 (5 panels) 8 010 -> 8 330 LM iterations/s,
    // 50 KF / 5 000 points / 16 planes (5 panels) 164 -> 154 us per iteration, 30 KF / 2 000 points (3 panels) 91 -> 88, 15 KF / 800 points (2 panels)
    // 75.6 -> 73.3.  Gate: two panels and a handful of chunks.
    const int nb = s->g.ld / 64;
    int first_blocks = 1, min_chunks = 16, min_nb = 2;
    double share0 = nb >= 8 ? 0.14 : 0.50, share = nb >= 8 ? 0.10 : 0.35;
    bool enabled = true;
    if (const char* e = PPO_DEV_ENV("PPO_BA_OVERLAP")) {  // development aid: "0" = off, or "first_blocks,share0,share[,min_chunks[,min_nb]]"
      if (e[0] == '0' && e[1] == 0) enabled = false;
      else std::sscanf(e, "%d,%lf,%lf,%d,%d", &first_blocks, &share0, &share, &min_chunks, &min_nb);
    }
    // round 5: a small system (ppo_chol.hip: chol_small_applies) is solved by ONE launch with the matrix in LDS (ppo_chol_small.inc): no panel launches to hide pair
    // chunks in, so the WHOLE reduction rides in stage 1's launch (fin[0] = nb: what the fused first stage already does for the first column block)
    const bool small = s->solver_kind == PPO_SOLVER_NATIVE && chol_small_applies(s->g.ld, s->g.n_red);
    if (small) min_nb = 1;
    if (enabled && nb >= std::max(small ? 1 : 2, min_nb) && (int)s->cb_start.size() == nb + 1 && s->cb_start[nb] == s->pc.n_chunks && s->pc.n_chunks >= std::max(1, min_chunks)) {
      s->fin.assign(nb + 1, nb);
      if (!small) chol_overlap_schedule(nb, s->cb_start.data(), first_blocks, share0, share, s->fin.data());
    }
  }
  // algorithmic bytes of one linearisation, SURVEY.md 8(d)
  {
    long n_vp = 0;
    for (int e = 0; e < p->n_plane_edges; ++e) n_vp += (p->pl_kind && p->pl_kind[e]) ? 1 : 0;
    int n_fixed = 0;
    for (int i = 0; i < Np; ++i) n_fixed += (pose_col[i] < 0);
    s->lin_bytes = 176L * p->n_mono + 184L * p->n_stereo + 208L * (p->n_plane_edges - n_vp) + 200L * n_vp + 576L * p->n_bbox + 768L * p->n_corner + 720L * p->n_cub3d +  // (9-D edge: 2 indices + 10 measurement + 9 information + the 6 x 9 block)
                  
                   96L * Npt + 104L * Npl + 512L * Nc + 272L * nfree + 56L * n_fixed;
  }
  // LM control on the device (optimize_pipelined): the configurations whose trials are queued without waiting for the host
  s->pipeline_ok = o->lm_control == PPO_LM_DEVICE && s->solver_kind == PPO_SOLVER_NATIVE && !s->comm.active() && !o->verbose;
  t_creating = nullptr;
  *out = s;
  return PPO_BA_OK;
} catch (const std::bad_alloc&) {  // (helper threads have been joined by their guards; device blocks, stream and staging go back to the caches)
  std::fprintf(stderr, "[ppo_ba] ppo_ba_create: out of host memory\n");
  if (ppo_ba_solver* s = t_creating) {
    t_creating = nullptr;
    t_stage = nullptr;
    free_all(s);
    delete s;
  }
  return PPO_BA_ERR_NOMEM;
}

int ppo_ba_reset(ppo_ba_handle s) {
  if (!s) return PPO_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  return reset_device(s);
}

// Optimizer.cc:2723-2838
int ppo_ba_run(ppo_ba_handle s, const volatile uint8_t* stop, ppo_ba_stats* stats) {
  if (!s) return PPO_BA_ERR_ARG;
  if (s->poisoned) return PPO_BA_ERR_HIP;  // (a batch of an earlier run never reported: ppo_ba_reset() first)
  HIP_TRY(hipSetDevice(s->device));
  const int n_red = s->stats.n_reduced;
  std::memset(&s->stats, 0, sizeof(s->stats));
  s->stats.n_reduced = n_red;
  s->stats.deterministic_fell_back = s->fell_back ? 1 : 0;  // (a handle that left the fixed-point sums stays out: leave_fixed_point)
  s->ph_used = 0;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  s->stop_ptr = stop;
  s->stop_agreed = false;
  if (s->comm.active()) RC_TRY(fetch_acc(s, false));  // agree on the flag before anything depends on it
  if (stop_now(s)) return PPO_BA_ABORTED;
  int rc = PPO_BA_OK;
  s->stats.n_active_edges_pass1 = s->g.n_mono + s->g.n_stereo + s->g.n_pl + s->g.n_bbox + s->g.n_corner + s->g.n_ptcub + s->n_cubpl;
  RC_TRY(optimize(s, 0, s->opt.iters_pass1, stop));
  // (the last trial of a pass is not speculated: the host has already waited for its result, the stream is idle here.
  //  Synchronising again -- and once more behind the classification kernels -- only added round trips between the passes.)
  const auto t1 = std::chrono::steady_clock::now();
  auto t2 = t1;
  if (stop_now(s)) rc = PPO_BA_ABORTED;
  if (rc == PPO_BA_OK && s->opt.iters_pass2 > 0) {
    const Graph& g = s->g;
    const State& st = s->st[s->cur];
    if (g.pt_last > g.pt_first)
      hipLaunchKernelGGL(k_classify_points, dim3(std::min(nblk(g.pt_last - g.pt_first, 256), 4 * s->n_cu)), dim3(256), 0, s->stream, g, st,
                         s->opt.chi2_mono, s->opt.chi2_stereo);
    const int nsm = std::max({g.n_pl, g.n_bbox, g.n_corner});
    if (nsm > 0)
      hipLaunchKernelGGL(k_classify_small, dim3(nblk(nsm, 256)), dim3(256), 0, s->stream, g, s->opt.chi2_plane, s->opt.chi2_plane_vp,
                         s->opt.th_bbox_norm, s->opt.th_corner_norm, s->opt.th_cub3d_norm);
    HIP_TRY(hipGetLastError());
    t2 = std::chrono::steady_clock::now();
    RC_TRY(optimize(s, 1, s->opt.iters_pass2, stop));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (stop_now(s)) rc = PPO_BA_ABORTED;
  }
  const auto t3 = std::chrono::steady_clock::now();
  if (s->opt.verbose) phase_collect(s);
  s->stats.wall_ms_total = ms(t0, t3);
  s->stats.wall_ms_pass1 = ms(t0, t1);
  s->stats.wall_ms_classify = ms(t1, t2);
  s->stats.wall_ms_pass2 = ms(t2, t3);
  s->ran = true;
  if (stats) *stats = s->stats;
  return rc;
}

int ppo_ba_read(ppo_ba_handle s, ppo_ba_result* r) try {
  if (!s || !r) return PPO_BA_ERR_ARG;
  if (s->poisoned) return PPO_BA_ERR_HIP;
  HIP_TRY(hipSetDevice(s->device));
  StageScope stage_scope(s->stage, true);  // the staging vectors below are gone (and their memory reusable) when read returns
  const Graph& g = s->g;
  const int Np = g.n_poses, Npt = g.n_points, Npl = g.n_planes, Nc = g.n_cuboids;
  // everything that is needed from the device in one batch of asynchronous copies into page-locked staging memory, one wait.
  // (The static parts of the reprojection records -- pose and point of an edge -- are still on the host from create.)
  SVec<double> hs(s->state_doubles);
  SVec<double> chi(s->d_chi2_all.n);
  SVec<uint32_t> hmf(g.n_mono), hsf(g.n_stereo);
  SVec<PlaneEdge> hp(g.n_pl);
  SVec<CubEdge> hb(g.n_bbox), hc(g.n_corner);
  // (round 5: the small ones -- every one of them on an ICL-sized window -- come down in ONE launch of the copy kernel writing the page-locked staging
  //  memory directly, instead of up to seven hipMemcpyAsync; arrays above UploadBatch::kMaxItemBytes keep the copy engine)
  UploadBatch down;
  auto fetch = [&](void* dst, const void* src, size_t bytes) -> int {
    if (bytes == 0) return PPO_BA_OK;
    if (bytes <= UploadBatch::kMaxItemBytes && s->stage->owns_pinned(dst, bytes)) { down.items.push_back(UploadBatch::Item{dst, src, bytes}); return PPO_BA_OK; }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->stream));
    return PPO_BA_OK;
  };
  if (s->state_doubles) RC_TRY(fetch(hs.data(), s->d_state[s->cur].p, s->state_doubles * sizeof(double)));
  // (only what the caller's non-NULL pointers ask for crosses the bus: the per-edge chi2 block is the largest part of a read)
  const bool want_chi = r->mono_chi2 || r->stereo_chi2 || r->pl_chi2 || r->bbox_chi2 || r->bbox_errnorm || r->corner_chi2 || r->corner_errnorm || r->ptcub_chi2 ||
                        r->cub3d_chi2 || r->cub3d_errnorm;
  if (chi.size() && want_chi) RC_TRY(fetch(chi.data(), s->d_chi2_all.p, chi.size() * sizeof(double)));
  if (g.n_mono && r->mono_level) RC_TRY(fetch(hmf.data(), s->d_mono_flags.p, hmf.size() * sizeof(uint32_t)));
  if (g.n_stereo && r->stereo_level) RC_TRY(fetch(hsf.data(), s->d_stereo_flags.p, hsf.size() * sizeof(uint32_t)));
  if (g.n_pl) RC_TRY(fetch(hp.data(), s->d_pl.p, hp.size() * sizeof(PlaneEdge)));
  if (g.n_bbox) RC_TRY(fetch(hb.data(), s->d_bbox.p, hb.size() * sizeof(CubEdge)));
  if (g.n_corner) RC_TRY(fetch(hc.data(), s->d_corner.p, hc.size() * sizeof(CubEdge)));
  {
    UploadBatch* const prev = t_upload_batch;
    t_upload_batch = &down;
    const int rc = flush_uploads(s->stream);
    t_upload_batch = prev;
    if (rc != PPO_BA_OK) return rc;
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  State h = make_state(hs.data(), Np, Npt, Npl, Nc);
  if (s->comm.active()) {
    // landmarks live on their owner, and with eliminated cuboids only the rank that owns the reduced terms back-substitutes
    // them: sum the (disjoint) owned parts over ranks through a zero-padded device buffer
    const size_t n_pt = (size_t)Npt * 3, n_pl = (size_t)Npl * 4, n_cb = (size_t)Nc * 10;
    SVec<double> tmp(n_pt + n_pl + n_cb, 0.0);
    for (int i = g.pt_first; i < g.pt_last; ++i) for (int a = 0; a < 3; ++a) tmp[(size_t)3 * i + a] = h.point[3 * i + a];
    for (int i = g.plane_first; i < g.plane_last; ++i) for (int a = 0; a < 4; ++a) tmp[n_pt + 4 * i + a] = h.plane[4 * i + a];
    if (g.owns_reduced) std::copy(h.cub, h.cub + n_cb, tmp.begin() + n_pt + n_pl);
    DevBuf<double> d;
    if (!d.alloc(tmp.size())) return PPO_BA_ERR_NOMEM;
    HIP_TRY(hipMemcpyAsync(d.p, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, s->stream));
    if (s->comm.sum(d.p, tmp.size(), s->stream)) return PPO_BA_ERR_COMM;
    HIP_TRY(hipMemcpyAsync(tmp.data(), d.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    d.release();
    std::copy(tmp.begin(), tmp.begin() + n_pt, h.point);
    std::copy(tmp.begin() + n_pt, tmp.begin() + n_pt + n_pl, h.plane);
    std::copy(tmp.begin() + n_pt + n_pl, tmp.end(), h.cub);
  }
  if (r->pose_qt) for (int i = 0; i < Np; ++i) std::memcpy(&r->pose_qt[7 * i], &h.pose_qt[7 * i], 7 * sizeof(double));
  if (r->point_xyz && Npt) std::memcpy(r->point_xyz, h.point, (size_t)Npt * 3 * sizeof(double));
  if (r->plane_coef && Npl) std::memcpy(r->plane_coef, h.plane, (size_t)Npl * 4 * sizeof(double));
  for (int i = 0; i < Nc; ++i) {
    if (r->cuboid_qt) std::memcpy(&r->cuboid_qt[7 * i], &h.cub[10 * i], 7 * sizeof(double));
    if (r->cuboid_scale) std::memcpy(&r->cuboid_scale[3 * i], &h.cub[10 * i + 7], 3 * sizeof(double));
  }
  // per-edge outputs
  const double* c = chi.data();
  const double* c_mono = c; c += g.n_mono;
  const double* c_st = c; c += g.n_stereo;
  const double* c_pl = c; c += g.n_pl;
  const double* c_bb = c; c += g.n_bbox;
  const double* n_bb = c; c += g.n_bbox;
  const double* c_co = c; c += g.n_corner;
  const double* n_co = c; c += g.n_corner;
  const double* c_pc = c;
  // isDepthPositive from the final estimates (types_six_dof_expmap.h:181-185)
  auto depth = [&](int pose, int pt) {
    const double* q = &h.pose_qt[7 * pose];
    const Pose T{Quat{q[0], q[1], q[2], q[3]}, Vec3{q[4], q[5], q[6]}};
    return apply(T, v3(h.point[3 * pt], h.point[3 * pt + 1], h.point[3 * pt + 2])).z > 0.0;
  };
  // (the reprojection edges in slices on the host team: 4e5 scattered writes and depth tests were half of a 1 ms read)
  auto in_slices = [&](int n, auto&& body) {
    const int nt = n >= (1 << 16) ? std::min(HostTeam::get().size(), 8) : 1;
    HostTeam::get().run(nt, [&](int t) { body((int)((long)n * t / nt), (int)((long)n * (t + 1) / nt)); });
  };
  in_slices(g.n_mono, [&](int k0, int k1) {
    for (int k = k0; k < k1; ++k) {
      const int e = s->mono_perm[k];
      if (r->mono_chi2) r->mono_chi2[e] = c_mono[k];
      if (r->mono_level) r->mono_level[e] = (hmf[k] & F_ACTIVE) ? 0 : 1;
      if (r->mono_depth_positive) r->mono_depth_positive[e] = depth(s->hk_mono_pose[k], s->hk_mono_point[k]);
    }
  });
  in_slices(g.n_stereo, [&](int k0, int k1) {
    for (int k = k0; k < k1; ++k) {
      const int e = s->stereo_perm[k];
      if (r->stereo_chi2) r->stereo_chi2[e] = c_st[k];
      if (r->stereo_level) r->stereo_level[e] = (hsf[k] & F_ACTIVE) ? 0 : 1;
      if (r->stereo_depth_positive) r->stereo_depth_positive[e] = depth(s->hk_stereo_pose[k], s->hk_stereo_point[k]);
    }
  });
  for (int e = 0; e < g.n_pl; ++e) {
    if (r->pl_chi2) r->pl_chi2[e] = c_pl[e];
    if (r->pl_level) r->pl_level[e] = (hp[e].flags & F_ACTIVE) ? 0 : 1;
  }
  for (int e = 0; e < g.n_bbox; ++e) {
    if (r->bbox_chi2) r->bbox_chi2[e] = c_bb[e];
    if (r->bbox_errnorm) r->bbox_errnorm[e] = n_bb[e];
    if (r->bbox_level) r->bbox_level[e] = (hb[e].flags & F_ACTIVE) ? 0 : 1;
  }
  for (int e = 0; e < g.n_corner2d; ++e) {
    if (r->corner_chi2) r->corner_chi2[e] = c_co[e];
    if (r->corner_errnorm) r->corner_errnorm[e] = n_co[e];
    if (r->corner_level) r->corner_level[e] = (hc[e].flags & F_ACTIVE) ? 0 : 1;
  }
  const int n_cub3d = g.n_corner - g.n_corner2d;  // the 9-D camera-object edges sit behind the corner edges
  for (int e = 0; e < n_cub3d; ++e) {
    if (r->cub3d_chi2) r->cub3d_chi2[e] = c_co[g.n_corner2d + e];
    if (r->cub3d_errnorm) r->cub3d_errnorm[e] = n_co[g.n_corner2d + e];
    if (r->cub3d_level) r->cub3d_level[e] = (hc[g.n_corner2d + e].flags & F_ACTIVE) ? 0 : 1;
  }
  if (r->ptcub_chi2) for (int e = 0; e < g.n_ptcub; ++e) r->ptcub_chi2[e] = c_pc[e];
  if (r->cubpl_chi2) for (int e = 0; e < s->n_cubpl; ++e) r->cubpl_chi2[e] = s->cubpl_chi2[e];
  if (s->comm.active()) {
    // per-edge outputs are computed by the rank that owns the edge's landmark (cuboid edges: the rank that owns the reduced
    // terms): mask the others and sum over ranks, one collective for everything
    struct Item { double* d; uint8_t* u; size_t n; int kind; };  // kind: 0 mono, 1 stereo, 2 plane, 3 owner-rank
    std::vector<Item> items = {
        {r->mono_chi2, nullptr, (size_t)g.n_mono, 0}, {nullptr, r->mono_level, (size_t)g.n_mono, 0}, {nullptr, r->mono_depth_positive, (size_t)g.n_mono, 0},
        {r->stereo_chi2, nullptr, (size_t)g.n_stereo, 1}, {nullptr, r->stereo_level, (size_t)g.n_stereo, 1}, {nullptr, r->stereo_depth_positive, (size_t)g.n_stereo, 1},
        {r->pl_chi2, nullptr, (size_t)g.n_pl, 2}, {nullptr, r->pl_level, (size_t)g.n_pl, 2},
        {r->bbox_chi2, nullptr, (size_t)g.n_bbox, 3}, {r->bbox_errnorm, nullptr, (size_t)g.n_bbox, 3}, {nullptr, r->bbox_level, (size_t)g.n_bbox, 3},
        {r->corner_chi2, nullptr, (size_t)g.n_corner2d, 3}, {r->corner_errnorm, nullptr, (size_t)g.n_corner2d, 3}, {nullptr, r->corner_level, (size_t)g.n_corner2d, 3},
        {r->cub3d_chi2, nullptr, (size_t)n_cub3d, 3}, {r->cub3d_errnorm, nullptr, (size_t)n_cub3d, 3}, {nullptr, r->cub3d_level, (size_t)n_cub3d, 3},
        {r->ptcub_chi2, nullptr, (size_t)g.n_ptcub, 3}};
    SVec<uint8_t> own_mono(g.n_mono, 0), own_stereo(g.n_stereo, 0), own_pl(g.n_pl, 0);
    for (int k = 0; k < g.n_mono; ++k) own_mono[s->mono_perm[k]] = s->hk_mono_point[k] >= g.pt_first && s->hk_mono_point[k] < g.pt_last;
    for (int k = 0; k < g.n_stereo; ++k) own_stereo[s->stereo_perm[k]] = s->hk_stereo_point[k] >= g.pt_first && s->hk_stereo_point[k] < g.pt_last;
    for (int e = 0; e < g.n_pl; ++e) own_pl[e] = hp[e].plane >= g.plane_first && hp[e].plane < g.plane_last;
    size_t total = 0;
    for (const Item& it : items) total += it.n;
    SVec<double> buf(total, 0.0);
    size_t off = 0;
    for (const Item& it : items) {
      for (size_t e = 0; e < it.n; ++e) {
        const bool mine = it.kind == 0 ? own_mono[e] : it.kind == 1 ? own_stereo[e] : it.kind == 2 ? own_pl[e] : (bool)g.owns_reduced;
        if (mine) buf[off + e] = it.d ? it.d[e] : (it.u ? (double)it.u[e] : 0.0);
      }
      off += it.n;
    }
    if (total) {
      DevBuf<double> d;
      if (!d.alloc(total)) return PPO_BA_ERR_NOMEM;
      HIP_TRY(hipMemcpyAsync(d.p, buf.data(), total * sizeof(double), hipMemcpyHostToDevice, s->stream));
      if (s->comm.sum(d.p, total, s->stream)) return PPO_BA_ERR_COMM;
      HIP_TRY(hipMemcpyAsync(buf.data(), d.p, total * sizeof(double), hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      d.release();
      off = 0;
      for (const Item& it : items) {
        for (size_t e = 0; e < it.n; ++e) {
          if (it.d) it.d[e] = buf[off + e];
          if (it.u) it.u[e] = (uint8_t)(buf[off + e] + 0.5);
        }
        off += it.n;
      }
    }
  }
  return PPO_BA_OK;
} catch (const std::bad_alloc&) {
  std::fprintf(stderr, "[ppo_ba] ppo_ba_read: out of host memory\n");
  return PPO_BA_ERR_NOMEM;
}

void ppo_ba_destroy(ppo_ba_handle s) {
  if (!s) return;
  free_all(s);
  delete s;
}

// ---- measurement / test hooks -------------------------------------------------------------------------------------------
int ppo_ba_linearize(ppo_ba_handle s, double* robust_chi2) {
  if (!s) return PPO_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  RC_TRY(launch_linearize(s));
  RC_TRY(fetch_acc(s, false));
  if (robust_chi2) *robust_chi2 = s->h_pinned[ACC_CHI2] + s->cubpl_rho_sum;
  return PPO_BA_OK;
}

int ppo_ba_time_linearize(ppo_ba_handle s, int reps, double* ms_out) {
  if (!s || reps <= 0 || !ms_out) return PPO_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  RC_TRY(launch_linearize(s));  // warm
  HIP_TRY(hipEventRecord(e0, s->stream));
  for (int i = 0; i < reps; ++i) RC_TRY(launch_linearize(s));
  HIP_TRY(hipEventRecord(e1, s->stream));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = (double)ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return PPO_BA_OK;
}

// One kernel of the hot path alone, `reps` back-to-back launches between two HIP events on the library's stream.
int ppo_ba_time_kernel(ppo_ba_handle s, int which, int reps, double* ms_out, int64_t* bytes_out) {
  if (!s || reps <= 0 || !ms_out) return PPO_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  const Graph& g = s->g;
  const State& st = s->st[s->cur];
  const State& tr = s->st[s->cur ^ 1];
  const int npts = g.pt_last - g.pt_first;
  const int nl = npts + (g.plane_last - g.plane_first);
  const long n_edges = (long)g.n_mono + g.n_stereo;
  int64_t bytes = 0;
  auto launch = [&]() -> bool {
    switch (which) {
      case PPO_BA_KERNEL_LIN_POSES:
        if (s->posec.n_chunks <= 0) return false;
        hipLaunchKernelGGL(k_lin_poses, dim3(nblk(s->posec.n_chunks, 4)), dim3(256), 0, s->stream, g, st, s->posec);
        return true;
      case PPO_BA_KERNEL_LIN_POINTS:
        if (npts <= 0) return false;
        if (s->use_lds) hipLaunchKernelGGL(k_lin_points<true>, dim3(std::min(nblk(4L * npts, 256), 4 * s->n_cu)), dim3(256), s->lds_bytes, s->stream, g, st);
        else hipLaunchKernelGGL(k_lin_points<false>, dim3(std::min(nblk(4L * npts, 256), 4 * s->n_cu)), dim3(256), 0, s->stream, g, st);
        return true;
      case PPO_BA_KERNEL_SCHUR_Z: return false;  // (the Z = W L^-T pass no longer exists: folded into the pair kernel)
      case PPO_BA_KERNEL_SCHUR_PAIRS:
        if (s->pc.n_chunks <= 0) return false;
        hipLaunchKernelGGL(k_schur_pairs, dim3((nblk(s->pc.n_chunks, 4) + 7) / 8 * 8), dim3(256), 0, s->stream, g, s->pc);
        return true;
      case PPO_BA_KERNEL_BACKSUB:
        if (nl <= 0) return false;
        if (s->use_lds) hipLaunchKernelGGL((k_backsub<true, 4>), dim3(nblk(4L * nl, 256)), dim3(256), s->lds_bytes, s->stream, g, st, tr, s->lambda);
        else hipLaunchKernelGGL((k_backsub<false, 4>), dim3(nblk(4L * nl, 256)), dim3(256), 0, s->stream, g, st, tr, s->lambda);
        return true;
      case PPO_BA_KERNEL_LINEARIZE_ALL: {
        return launch_linearize_kernel(s, g, st) == PPO_BA_OK;
      }
      case PPO_BA_KERNEL_CHI2_POINTS:
        if (npts <= 0) return false;
        hipLaunchKernelGGL(k_chi2_points, dim3(std::min(nblk((long)kLanesPerPoint * npts, 256), 4 * s->n_cu)), dim3(256), 0, s->stream, g, st);
        return true;
      default: return false;
    }
  };
  // algorithmic bytes: what the kernel must read and write once (records, flags, vertex estimates, output blocks)
  const long nP = (long)s->d_monoP.n, nSP = (long)s->d_stereoP.n;
  switch (which) {
    case PPO_BA_KERNEL_LIN_POSES:  // what THIS kernel moves: pose-major record + flag per edge, one 32-byte factor record per point slot (the 144-byte
                                   // W blocks of SURVEY 8d are no longer written), point per landmark; R|t + intrinsics in, packed Hpp|b + linR out per keyframe
      bytes = nP * (long)(sizeof(MonoEdgeP) + 4) + nSP * (long)(sizeof(StereoEdgeP) + 4) + 32L * g.n_pt_slots + 24L * npts + (96L + 40 + 216 + 96) * g.n_pose_free;
      break;
    case PPO_BA_KERNEL_LIN_POINTS:  // landmark-major record + flag + chi2 per edge; point in, D + b_l out
      bytes = (long)g.n_mono * (sizeof(MonoEdge) + 4 + 8) + (long)g.n_stereo * (sizeof(StereoEdge) + 4 + 8) + 96L * npts + 136L * g.n_poses;
      break;
    case PPO_BA_KERNEL_SCHUR_Z: bytes = 0; break;
    case PPO_BA_KERNEL_SCHUR_PAIRS:  // unique bytes: factor record per point slot, W block per plane slot, the pair plan (3 x int32 per pair), (D + lambda I)^-1 + v per landmark, S + b
      bytes = 32L * g.n_pt_slots + 144L * (g.n_slots - g.n_pt_slots) + 12L * s->d_pc_sj.n + 72L * nl + 8L * ((long)g.n_red * (g.n_red + 1) / 2 + g.n_red);
      break;
    case PPO_BA_KERNEL_BACKSUB: bytes = (long)g.n_mono * (sizeof(MonoEdge) + 4) + (long)g.n_stereo * (sizeof(StereoEdge) + 4) + 96L * nl + 8L * g.n_red; break;
    case PPO_BA_KERNEL_LINEARIZE_ALL: bytes = s->lin_bytes; break;
    case PPO_BA_KERNEL_CHI2_POINTS: bytes = (long)g.n_mono * (sizeof(MonoEdge) + 4 + 8) + (long)g.n_stereo * (sizeof(StereoEdge) + 4 + 8) + 24L * npts + 136L * g.n_poses; break;
    default: return PPO_BA_ERR_ARG;
  }
  (void)n_edges;
  if (bytes_out) *bytes_out = bytes;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  if (!launch()) { *ms_out = 0; return PPO_BA_OK; }
  HIP_TRY(hipEventRecord(e0, s->stream));
  for (int i = 0; i < reps; ++i) launch();
  HIP_TRY(hipEventRecord(e1, s->stream));
  HIP_TRY(hipEventSynchronize(e1));
  HIP_TRY(hipGetLastError());
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = (double)ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return PPO_BA_OK;
}

int ppo_ba_time_factor(ppo_ba_handle s, int reps, double* ms_out) {
  if (!s || reps <= 0 || !ms_out) return PPO_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  if (s->lambda <= 0) {  // no run yet: linearise and take lambda_0
    RC_TRY(launch_linearize(s));
    RC_TRY(launch_maxdiag(s));
    RC_TRY(fetch_acc(s, false, true));
    s->lambda = s->opt.lm_tau * s->h_pinned[ACC_MAXDIAG];
  }
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  float total = 0;
  for (int i = 0; i < reps + 1; ++i) {
    RC_TRY(launch_reduce(s, s->lambda));  // the factorisation overwrites S: rebuild it outside the timed span
    HIP_TRY(hipEventRecord(e0, s->stream));
    RC_TRY(launch_solve(s));
    HIP_TRY(hipEventRecord(e1, s->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (i > 0) total += ms;
  }
  *ms_out = (double)total / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return PPO_BA_OK;
}

// Sharded runs: average milliseconds of the exchange steps of one solve attempt, each timed alone with HIP events on the library's stream
// (every rank must call it; the collectives are real).  out[0] pack + ALL-REDUCE of (S, b) + unpack (options.solve_on_owner = 0),
// out[1] pack + REDUCE to rank 0 + unpack there + broadcast of x and the factorisation flag (solve_on_owner = 1), out[2] the all-reduce of
// the three trial scalars (chi2, scale, stop vote), out[3] bytes of the packed (S, b).  The contents of S, b and x are garbage afterwards:
// call ppo_ba_reset() before the next run.
int ppo_ba_time_collectives(ppo_ba_handle s, int reps, double* out) {
  if (!s || reps <= 0 || !out) return PPO_BA_ERR_ARG;
  out[0] = out[1] = out[2] = out[3] = 0.0;
  if (!s->comm.active()) return PPO_BA_OK;
  HIP_TRY(hipSetDevice(s->device));
  const Graph& g = s->g;
  const int nb = g.ld / 64, ntiles = nb * (nb + 1) / 2;
  const size_t n = g.ld, count = (size_t)ntiles * 4096 + n;
  out[3] = (double)count * sizeof(double);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct Guard {  // every exit: the events go back, and the handle knows that S, b, x hold garbage (a following time_factor / debug call must rebuild them)
    ppo_ba_solver* s;
    hipEvent_t *a, *b;
    ~Guard() {
      if (*a) (void)hipEventDestroy(*a);
      if (*b) (void)hipEventDestroy(*b);
      s->S_is_zero = false;
      s->lin_valid = false;
      s->have_system = false;
    }
  } guard{s, &e0, &e1};
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  auto timed = [&](int which, double* ms_out) -> int {
    float total = 0;
    for (int i = 0; i < reps + 1; ++i) {
      HIP_TRY(hipEventRecord(e0, s->stream));
      if (which < 2) hipLaunchKernelGGL(k_pack_lower_tiles, dim3(ntiles + 1), dim3(256), 0, s->stream, s->d_S.p, s->d_bs_p, s->d_pack.p, g.ld, 0);
      if (which == 0) {
        if (s->comm.sum(s->d_pack.p, count, s->stream)) return PPO_BA_ERR_COMM;
        hipLaunchKernelGGL(k_pack_lower_tiles, dim3(ntiles + 1), dim3(256), 0, s->stream, s->d_S.p, s->d_bs_p, s->d_pack.p, g.ld, 1);
      } else if (which == 1) {
        if (s->comm.sum_to_root(s->d_pack.p, count, s->stream)) return PPO_BA_ERR_COMM;
        if (s->comm.rank == 0) hipLaunchKernelGGL(k_pack_lower_tiles, dim3(ntiles + 1), dim3(256), 0, s->stream, s->d_S.p, s->d_bs_p, s->d_pack.p, g.ld, 1);
        if (s->comm.bcast_from_root(s->d_xr.p, n * sizeof(double), s->stream)) return PPO_BA_ERR_COMM;
        if (s->comm.bcast_from_root(s->d_info.p, sizeof(int), s->stream)) return PPO_BA_ERR_COMM;
      } else {
        if (s->comm.sum(s->g.acc, 3, s->stream)) return PPO_BA_ERR_COMM;
      }
      HIP_TRY(hipEventRecord(e1, s->stream));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      if (i > 0) total += ms;
    }
    *ms_out = (double)total / reps;
    return PPO_BA_OK;
  };
  for (int which = 0; which < 3; ++which) RC_TRY(timed(which, &out[which]));
  return PPO_BA_OK;
}

// reduced system of the given lambda at the CURRENT state (re-linearises); S is returned as a full symmetric matrix
int ppo_ba_debug_reduced(ppo_ba_handle s, double* S, double* b, double* x, int32_t* n_out) {
  if (!s) return PPO_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  const int n = s->g.n_red;
  if (n_out) *n_out = n;
  if (!S && !b && !x) return PPO_BA_OK;
  const double lambda = s->lambda > 0 ? s->lambda : 0.0;
  RC_TRY(launch_linearize(s));
  RC_TRY(launch_reduce(s, lambda));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (S) {
    const int ld = s->g.ld;
    std::vector<double> tmp((size_t)ld * ld);
    HIP_TRY(hipMemcpy(tmp.data(), s->d_S.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int c = 0; c < n; ++c)
      for (int r = c; r < n; ++r) S[(size_t)c * n + r] = S[(size_t)r * n + c] = tmp[(size_t)c * ld + r];
    if (s->g.det) {
      // deterministic mode: a term that is not finite or does not fit is LEFT OUT of the integer sums and flagged (ACC_DETRANGE: the system is not
      // usable, every trial on it is rejected like a failed factorisation).  The matrix handed out says the same thing the floating-point sums would.
      double flags[2] = {0.0, 0.0};
      HIP_TRY(hipMemcpy(flags, s->g.acc + ACC_DETBAD, sizeof(flags), hipMemcpyDeviceToHost));
      if (flags[1] != 0.0 && n > 0) S[0] = std::numeric_limits<double>::quiet_NaN();
    }
  }
  if (b) HIP_TRY(hipMemcpy(b, s->d_bs_p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  if (x) {
    if (PPO_DEV_ENV("PPO_BA_DEBUG_OVERLAP")) RC_TRY(launch_reduce(s, lambda, true));  // (test hook: x through the overlapped reduction, S and b above through the plain one)
    RC_TRY(launch_solve(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(x, s->d_xr.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  }
  return PPO_BA_OK;
}

int ppo_ba_debug_set_lambda(ppo_ba_handle s, double lambda) {
  if (!s) return PPO_BA_ERR_ARG;
  s->lambda = lambda;
  return PPO_BA_OK;
}

int64_t ppo_ba_linearize_bytes(ppo_ba_handle s) { return s ? s->lin_bytes : -1; }

int ppo_ba_overlap_plan(int32_t n_blocks, const int32_t* first_chunk, int32_t first_blocks, double share0, double share, int32_t* fin) {
  if (n_blocks < 1 || !first_chunk || !fin || !(share0 >= 0.0) || !(share >= 0.0)) return PPO_BA_ERR_ARG;
  for (int c = 0; c < n_blocks; ++c)
    if (first_chunk[c] < 0 || first_chunk[c + 1] < first_chunk[c]) return PPO_BA_ERR_ARG;
  std::vector<int> f(n_blocks + 1, n_blocks);
  chol_overlap_schedule(n_blocks, first_chunk, first_blocks, share0, share, f.data());
  for (int k = 0; k <= n_blocks; ++k) fin[k] = f[k];
  return PPO_BA_OK;
}

int ppo_ba_overlap_schedule(ppo_ba_handle s, int32_t* fin, int32_t n) {
  if (!s) return PPO_BA_ERR_ARG;
  if (s->fin.empty() || s->solver_kind != PPO_SOLVER_NATIVE || s->comm.active()) return 0;
  for (int k = 0; fin && k < n && k < (int)s->fin.size(); ++k) fin[k] = s->fin[k];
  return s->g.ld / 64;
}

int ppo_ba_release_cached(void) {
  g_pool.trim();
  g_ctx_pool.trim();
  g_host_pool.trim();
  return PPO_BA_OK;
}

int ppo_ba_comm_unique_id(uint8_t id[128]) {
  Comm c;
  return c.unique_id(id) == 0 ? PPO_BA_OK : PPO_BA_ERR_COMM;
}

}  // extern "C"
