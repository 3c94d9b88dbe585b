// ppo_lm_host.hpp -- the host side of the LM control on the device: the loop that queues BATCHES of solve attempts on the
// assumption "accepted, the pass goes on", reads the batch's trial records, and puts its bookkeeping right at the first attempt that went
// differently.  Plain C++ over a small device interface so that the CPU test-suite can run it against a simulated device
// (tests/lm_pipeline_test.cpp: the simulated device runs lm_decide -- the very function k_decide runs -- on scripted outcomes, checks that
// every attempt that is LIVE was queued on bookkeeping that matches the true state, and compares the iteration statistics with a serial walk
// through OptimizationAlgorithmLevenberg::solve, levenberg.cpp:61-164).  The product's interface is SolverDev in ppo_solver.hip.
#pragma once
#include <algorithm>

#include "../../include/ppo_ba.h"
#include "ppo_lm.hpp"

namespace ppo {

constexpr int kLookahead = 5;  // solve attempts per batch at most (<= the slots of the record ring)

// the host-side bookkeeping a queued attempt changes (which state copy is the estimate, which linearisation buffer is current, what is clear)
struct TrialBook {
  int cur, lin_buf;
  bool lin_clean[2], S_is_zero, reduce_overlapped, have_system;
};

// Dev -- what the loop needs from a device:
//   bool stop();                                        the caller's stop flag
//   TrialBook book(); void restore(const TrialBook&);   bookkeeping snapshot / roll-back
//   void accept(bool speculated);                       discardTop() on the host: the trial state is the estimate (and its linearisation current)
//   int  enqueue_trial(bool speculate, int n_publish, unsigned long long* seq);   reduce + solve + update + evaluate + decide, nothing waited for;
//                                                       n_publish > 0: the decide step hands the records of the last n_publish attempts over
//   int  wait_records(unsigned long long seq);          until the hand-over that ends with attempt `seq` has arrived
//   const double* record(unsigned long long seq);       REC_* of attempt seq
//   void set_epoch(int e);                              the epoch further launches are queued under
//   int  restore_kept_errors();                         per-edge errors of the kept state after a failed factorisation (lm_iteration)
//   int  relinearize(LmCtl& m);                         fresh linearisation of the estimate, m.chi_cur / m.ini_chi from it, the record := m
//   void count_trial(); void push_stat(const ppo_ba_iter_stat&);
//
// m: the host's mirror of the device record at entry (iteration 0 of the pass set up: lambda, chi2 of the estimate, epoch); at exit what the
// device holds.  lookahead_cap: 1 ... kLookahead.  Returns PPO_BA_OK or an error code.
// back_off_for_the_pass (round 5; the product's choice): after the first batch of a pass that did not go as assumed -- a rejected trial, a
// termination -- every further batch of that pass is ONE attempt.  Rejections come in runs (a damping that is too small stays too small for a
// few attempts), and every attempt queued behind a rejected one is ~25 launches that find their epoch gone (~12 us): doubling back up to five lost
// 4 % on a 15-keyframe window that rejects a quarter of its trials (profiles/r04_lm_control.txt), a batch of one costs what the host path costs.
// false: the round-4 rule (one attempt after a miss, then doubling while the assumption holds).
template <class Dev>
int lm_pipelined_pass(Dev& dev, int pass, int iterations, LmCtl& m, int lookahead_cap, bool back_off_for_the_pass = true) {
  struct Queued { unsigned long long seq; TrialBook after; bool speculated; };
  Queued q[kLookahead];
  lookahead_cap = std::max(1, std::min(lookahead_cap, kLookahead));
  int lookahead = lookahead_cap;
  bool pass_over = false;  // the pass has ended (termination; the stop flag inside a retry loop)
  ppo_ba_iter_stat it{};
  it.pass = pass;
  int rc = PPO_BA_OK;
  while (rc == PPO_BA_OK && !pass_over && m.iter < iterations && !dev.stop()) {
    // a batch: attempt j is queued as if attempts 0 .. j - 1 had been accepted and had ended their iterations
    const int n = std::min(lookahead, iterations - m.iter);
    for (int j = 0; j < n && rc == PPO_BA_OK; ++j) {
      if (j > 0) dev.accept(q[j - 1].speculated);
      q[j].speculated = m.iter + j + 1 < iterations;  // (the last iteration of a pass is evaluated, not linearised: nothing follows it)
      rc = dev.enqueue_trial(q[j].speculated, j == n - 1 ? n : 0, &q[j].seq);
      q[j].after = dev.book();
    }
    if (rc != PPO_BA_OK) break;
    if ((rc = dev.wait_records(q[n - 1].seq)) != PPO_BA_OK) break;
    bool clean = true;
    for (int j = 0; j < n && clean && rc == PPO_BA_OK; ++j) {
      const double* r = dev.record(q[j].seq);
      if (r[REC_SEQ] != (double)q[j].seq || r[REC_DEAD] != 0.0) { rc = PPO_BA_ERR_HIP; break; }  // (a dead attempt is never seen before the one that went differently)
      const int code = (int)r[REC_CODE];
      const bool accepted = r[REC_ACCEPTED] != 0.0;
      dev.count_trial();
      it.iteration = (int)r[REC_ITER] - (code == LM_RETRY || code == LM_SOLVER_ERROR ? 0 : 1);
      it.chi2_before = r[REC_CHI_BEFORE];
      it.lambda = r[REC_LAMBDA];
      it.trials = (int)r[REC_TRIALS];
      it.accepted = accepted ? 1 : 0;
      it.chi2_after = r[REC_CHI_AFTER];
      m.lambda = r[REC_LAMBDA_NEXT]; m.ni = r[REC_NI]; m.nbad = (int)r[REC_NBAD]; m.chi_cur = r[REC_CHI_AFTER]; m.ini_chi = r[REC_INI_CHI];
      m.iter = (int)r[REC_ITER]; m.qmax = (int)r[REC_QMAX]; m.epoch = (int)r[REC_EPOCH];
      // (round 6) the device saw the stop flag when it decided this attempt (lm_decide_stop): everything queued behind it is dead, the pass ends here
      const bool stopped = r[REC_STOPPED] != 0.0;
      const bool as_assumed = accepted && code == LM_ITER_DONE && !stopped;
      if (as_assumed) {
        if (j == n - 1) dev.accept(q[j].speculated);  // (the others: applied when the next attempt was queued)
      } else {
        // the attempts queued behind it did not run: the bookkeeping goes back to "attempt j queued, outcome open", then takes j's real outcome
        clean = false;
        dev.restore(q[j].after);
        dev.set_epoch(m.epoch);  // (the decide step has moved the record on)
        if (accepted) dev.accept(q[j].speculated);
      }
      if (code == LM_SOLVER_ERROR) { rc = PPO_BA_ERR_SOLVER; break; }
      if (!accepted && (int)r[REC_INFO] != 0 && (rc = dev.restore_kept_errors()) != PPO_BA_OK) break;
      if (code == LM_RETRY) {
        if (stopped || dev.stop()) {  // the retry loop ends on the stop flag (levenberg.cpp:149): the iteration is over, rejected
          it.result = 0;
          dev.push_stat(it);
          pass_over = true;
        }
        continue;  // (clean is false: the batch ends here, the next one starts with the retry)
      }
      // the iteration is over
      it.result = code == LM_TERMINATE ? 1 : 0;
      dev.push_stat(it);
      if (code == LM_TERMINATE || stopped) pass_over = true;
      if (!pass_over && !as_assumed && m.iter < iterations) {
        // an iteration that ended on a rejected attempt without terminating (rho is NaN): the next one starts from a fresh linearisation of
        // the kept state, as lm_iteration's `fresh` branch does
        if ((rc = dev.relinearize(m)) != PPO_BA_OK) break;
      }
    }
    if (!clean && back_off_for_the_pass) lookahead_cap = 1;
    lookahead = clean ? std::min(lookahead_cap, 2 * lookahead) : 1;
  }
  return rc;
}

}  // namespace ppo
