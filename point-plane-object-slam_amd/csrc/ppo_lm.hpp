// ppo_lm.hpp -- the Levenberg-Marquardt decision behind one solve attempt, as ONE function for the device (k_decide, ppo_kernels.hpp)
// and for the CPU test-suite (csrc/hostcheck.cpp -> tests/test_lm_control.py, which holds it against a restatement of
// Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:102-163 on random inputs).
#pragma once
#include "ppo_math.hpp"

namespace ppo {

// ---- LM control on the device (ppo_solver.hip, optimize_pipelined; levenberg.cpp:61-164) ----------------------------------------------
// The damping, the gain-ratio test, nu, the nBad count and the iteration counters live in ONE device record.  k_decide, queued behind the
// evaluation of every trial state, takes the decision the host used to take after a round trip through mapped memory; the host queues the
// next trial BEFORE it knows the outcome, on the assumption "accepted, the pass goes on".  Whenever that assumption fails (a rejected trial,
// a terminated pass, a solver error) k_decide bumps `epoch`: every kernel of a trial carries the epoch it was queued under (Graph::epoch,
// LmGate) and returns at once when the record has moved on, so the launches queued on the wrong assumption change nothing.
struct LmCtl {
  double lambda;    // damping of the NEXT trial (_currentLambda)
  double chi_cur;   // robust chi2 of the current estimate (currentChi)
  double ini_chi;   // ... at the start of the running iteration (iniChi)
  double ni;        // _ni
  double cubpl;     // constant chi2 term of the cuboid-plane edges
  double lm_upper, lm_lower;
  int nbad, qmax, iter, epoch, max_trials, pad_;
};
// what k_decide records per trial (doubles; a slot of the device ring, copied to the mapped host ring at the end of a batch of trials)
enum { REC_CHI2 = 0, REC_SCALE, REC_INFO, REC_LAMBDA, REC_RHO, REC_ACCEPTED, REC_CODE, REC_CHI_BEFORE, REC_CHI_AFTER, REC_TRIALS, REC_ITER,
       REC_DEAD, REC_LAMBDA_NEXT, REC_NI, REC_NBAD, REC_INI_CHI, REC_EPOCH, REC_QMAX, REC_STOPPED, REC_SEQ = 31, REC_DOUBLES = 32 };
enum { LM_RETRY = 0, LM_ITER_DONE = 1, LM_TERMINATE = 2, LM_SOLVER_ERROR = 3 };


// One solve attempt has been evaluated: chi2_sum = sum of rho over the active edges of the trial state (without the constant term of the
// cuboid-plane edges, c.cubpl), scale_raw = the denominator of the gain ratio as the update accumulated it, info = the factorisation flag
// (0: positive definite, >= 1 << 30: the solver itself failed).  Updates the control record and fills the trial's record (REC_*, except
// REC_SEQ).  The statements follow levenberg.cpp:110-163 one to one; pow(x, 3) is cube_rounded_once(x) (ppo_math.hpp).
PPO_HD void lm_decide(LmCtl& c, double chi2_sum, double scale_raw, int info, double* rec) {
  const double lambda = c.lambda;
  double currentChi = c.chi_cur;
  double tempChi = chi2_sum + c.cubpl;
  const double chi_trial = tempChi;
  if (info != 0 || !isfinite(scale_raw)) tempChi = 1.7976931348623157e308;  // solver not ok (levenberg.cpp:126-127)
  double rho = currentChi - tempChi;
  double scale = scale_raw;
  if (info != 0) scale = 0;
  scale += 1e-3;
  rho /= scale;
  const bool accepted = rho > 0 && isfinite(tempChi) && info == 0;
  if (accepted) {
    double alpha = 1. - cube_rounded_once(2 * rho - 1);
    alpha = (c.lm_upper < alpha) ? c.lm_upper : alpha;                     // std::min(alpha, upper)
    const double scaleFactor = (c.lm_lower < alpha) ? alpha : c.lm_lower;  // std::max(lower, alpha)
    c.lambda *= scaleFactor;
    c.ni = 2;
    currentChi = tempChi;
  } else {
    c.lambda *= c.ni;
    c.ni *= 2;
  }
  c.qmax++;
  const int trials = c.qmax;
  const double chi_before = c.ini_chi;
  int code;
  if (info >= (1 << 30)) {
    code = LM_SOLVER_ERROR;  // the hand-off of the backward substitution timed out: not a "not positive" matrix
  } else if (rho < 0 && c.qmax < c.max_trials) {
    code = LM_RETRY;
  } else {  // the iteration is over (levenberg.cpp:150-163)
    if (c.qmax == c.max_trials || rho == 0) {
      code = LM_TERMINATE;
    } else {
      if ((c.ini_chi - currentChi) * 1e3 < c.ini_chi) c.nbad++;
      else c.nbad = 0;
      code = c.nbad >= 3 ? LM_TERMINATE : LM_ITER_DONE;
    }
    c.iter++;
    c.qmax = 0;
    c.ini_chi = currentChi;
  }
  c.chi_cur = currentChi;
  if (!(accepted && code == LM_ITER_DONE)) c.epoch++;  // not what the host assumed when it queued the next trials: those launches are dead
  rec[REC_CHI2] = chi_trial;
  rec[REC_SCALE] = scale_raw;
  rec[REC_INFO] = (double)info;
  rec[REC_LAMBDA] = lambda;
  rec[REC_RHO] = rho;
  rec[REC_ACCEPTED] = accepted ? 1.0 : 0.0;
  rec[REC_CODE] = (double)code;
  rec[REC_CHI_BEFORE] = chi_before;
  rec[REC_CHI_AFTER] = currentChi;
  rec[REC_TRIALS] = (double)trials;
  rec[REC_ITER] = (double)c.iter;
  rec[REC_DEAD] = 0.0;
  rec[REC_LAMBDA_NEXT] = c.lambda;
  rec[REC_NI] = c.ni;
  rec[REC_NBAD] = (double)c.nbad;
  rec[REC_INI_CHI] = c.ini_chi;
  rec[REC_EPOCH] = (double)c.epoch;
  rec[REC_QMAX] = (double)c.qmax;
  rec[REC_STOPPED] = 0.0;
}

// The caller's stop flag was up when this attempt was decided (k_decide reads its mirror in mapped host memory before EVERY decision, as
// levenberg.cpp:149 and sparse_optimizer.cpp:375 read terminate() before every retry and every iteration): nothing queued behind this attempt may
// run -- the epoch moves on if the decision itself has not moved it -- and the record says so; the host ends the pass at this attempt.
// `epoch_in`: the epoch the attempt ran under.
PPO_HD void lm_decide_stop(LmCtl& c, int epoch_in, double* rec) {
  if (c.epoch == epoch_in) c.epoch++;
  rec[REC_EPOCH] = (double)c.epoch;
  rec[REC_STOPPED] = 1.0;
}

}  // namespace ppo
