// lm_pipeline_test.cpp -- CPU test of the host side of the LM control on the device (csrc/ppo_lm_host.hpp: lm_pipelined_pass, the loop
// ppo_solver.hip runs) against a SIMULATED device.  Built and run by tests/test_lm_control.py (plain g++, no HIP).
//
// The simulated device keeps the TRUE state (which state copy is the estimate, which linearisation buffer is current, the control record
// with its epoch), executes the queued solve attempts when the host waits -- live ones through lm_decide (csrc/ppo_lm.hpp, the function
// k_decide runs) on scripted outcomes, attempts queued under an outdated epoch as no-ops -- and checks for every LIVE attempt that the
// bookkeeping it was queued on is the true state: the property that makes the batch queueing correct.  The iteration statistics the
// loop reports are compared with a serial walk through OptimizationAlgorithmLevenberg::solve (Thirdparty/g2o/g2o/core/
// optimization_algorithm_levenberg.cpp:61-164, restated below independently of lm_decide) over the same outcomes.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <vector>

#include "../point-plane-object-slam_amd/csrc/ppo_lm_host.hpp"

using namespace ppo;

#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (line %d, script %d)\n", #c, __LINE__, g_script); std::exit(1); } } while (0)
static int g_script = -1;

struct Outcome { int kind; double mag; };  // what the evaluation of a solve attempt will show, relative to the chi2 of the estimate at that time
// kinds: 0 accepted, clear gain | 1 accepted, gain below 1e-3 (counts towards nBad) | 2 rejected | 3 chi2 NaN | 4 failed factorisation | 5 no change at all (rho == 0)
static void realise(const Outcome& o, double chi_cur, double cubpl, double* chi2_sum, double* scale, int* info) {
  *info = 0;
  *scale = 3.0 + o.mag;
  double chi = chi_cur;
  switch (o.kind) {
    case 0: chi = chi_cur * (1.0 - 0.01 - 0.3 * o.mag); break;
    case 1: chi = chi_cur * (1.0 - 1e-5 * (0.1 + o.mag)); break;
    case 2: chi = chi_cur * (1.0 + 0.05 * (0.1 + o.mag)); break;
    case 3: chi = std::numeric_limits<double>::quiet_NaN(); break;
    case 4: *info = 7; chi = chi_cur * 0.5; break;
    case 5: chi = chi_cur; break;
  }
  *chi2_sum = chi - cubpl;
  if (o.kind == 5) *chi2_sum = chi_cur - cubpl;
}

struct SimDevice {
  // truth
  TrialBook truth{0, 0, {false, false}, true, false, false};
  LmCtl ctl{};
  // what the host believes / has set
  TrialBook host = truth;
  int host_epoch = 0;
  // queue
  struct Entry { unsigned long long seq; int tag; bool speculate; TrialBook before; int n_publish; };
  std::vector<Entry> pending;
  unsigned long long ring_seq = 0;
  double dev_ring[16][REC_DOUBLES] = {}, host_ring[16][REC_DOUBLES] = {};
  // script
  std::vector<Outcome> script;
  size_t next = 0;
  int stop_after_polls = -1, polls = 0;
  int stop_at_live = -1;  // round 6: the flag goes up while live attempt number stop_at_live is evaluated -- the device's decide step sees it (REC_STOPPED), the host from then on
  // what came out
  std::vector<ppo_ba_iter_stat> stats;
  int n_trials = 0, n_dead = 0, n_live = 0, n_kept_error_passes = 0, n_relinearize = 0;

  static void trial_effects(TrialBook& b, bool speculate) {  // what launch_reduce / launch_solve / launch_update / launch_linearize do to the bookkeeping
    b.S_is_zero = false;
    b.lin_clean[b.lin_buf ^ 1] = true;   // stage 1 clears the buffer that is not current
    b.have_system = true;
    b.S_is_zero = true;                  // the update clears S behind the backward substitution
    if (speculate) b.lin_clean[b.lin_buf ^ 1] = false;  // the trial state is linearised into it
  }
  bool stop() { ++polls; return (stop_after_polls >= 0 && polls > stop_after_polls) || (stop_at_live >= 0 && n_live > stop_at_live); }
  TrialBook book() { return host; }
  void restore(const TrialBook& b) { host = b; }
  void accept(bool speculated) { host.cur ^= 1; if (speculated) host.lin_buf ^= 1; }
  int enqueue_trial(bool speculate, int n_publish, unsigned long long* seq) {
    Entry e{++ring_seq, host_epoch, speculate, host, n_publish};
    trial_effects(host, speculate);
    pending.push_back(e);
    *seq = e.seq;
    CHECK(pending.size() <= (size_t)kLookahead);
    return PPO_BA_OK;
  }
  int wait_records(unsigned long long seq) {
    CHECK(!pending.empty() && pending.back().seq == seq && pending.back().n_publish == (int)pending.size());
    for (const Entry& e : pending) {
      double* rec = dev_ring[e.seq % 16];
      for (int i = 0; i < REC_DOUBLES; ++i) rec[i] = 0.0;
      if (e.tag != ctl.epoch) {  // queued on an assumption that failed: every kernel of it returns at once
        rec[REC_DEAD] = 1.0;
        rec[REC_SEQ] = (double)e.seq;
        ++n_dead;
        continue;
      }
      // LIVE: it must have been queued on the true state
      CHECK(e.before.cur == truth.cur && e.before.lin_buf == truth.lin_buf);
      CHECK(e.before.S_is_zero == truth.S_is_zero && e.before.lin_clean[0] == truth.lin_clean[0] && e.before.lin_clean[1] == truth.lin_clean[1]);
      trial_effects(truth, e.speculate);
      CHECK(next < script.size());
      double chi2_sum, scale;
      int info;
      realise(script[next++], ctl.chi_cur, ctl.cubpl, &chi2_sum, &scale, &info);
      lm_decide(ctl, chi2_sum, scale, info, rec);
      if (stop_at_live >= 0 && n_live >= stop_at_live) lm_decide_stop(ctl, e.tag, rec);  // (what k_decide does when the mirror of the flag is up)
      rec[REC_SEQ] = (double)e.seq;
      if (rec[REC_ACCEPTED] != 0.0) {  // the trial state is the estimate now; its linearisation (if there is one) the current system
        truth.cur ^= 1;
        if (e.speculate) truth.lin_buf ^= 1;
      }
      ++n_live;
    }
    for (const Entry& e : pending)  // the hand-over of the batch
      for (int i = 0; i < REC_DOUBLES; ++i) host_ring[e.seq % 16][i] = dev_ring[e.seq % 16][i];
    pending.clear();
    return PPO_BA_OK;
  }
  const double* record(unsigned long long seq) { return host_ring[seq % 16]; }
  void set_epoch(int e) { CHECK(pending.empty() && e == ctl.epoch); host_epoch = e; }
  int restore_kept_errors() { CHECK(pending.empty() && host.cur == truth.cur); ++n_kept_error_passes; return PPO_BA_OK; }
  int relinearize(LmCtl& m) {
    CHECK(pending.empty() && host.cur == truth.cur && host.lin_buf == truth.lin_buf && host_epoch == ctl.epoch);
    for (TrialBook* b : {&host, &truth}) { const int target = b->lin_buf ^ 1; b->lin_clean[target] = false; b->lin_buf = target; }
    m.chi_cur = m.ini_chi = ctl.chi_cur;  // (the same state: the same chi2)
    ctl = m;
    ++n_relinearize;
    return PPO_BA_OK;
  }
  void count_trial() { ++n_trials; }
  void push_stat(const ppo_ba_iter_stat& it) { stats.push_back(it); }
};

// the serial algorithm over the same outcomes: levenberg.cpp:61-164 + the outer loop of sparse_optimizer.cpp:354-419 (no stop flag)
struct SerialLm {
  double lambda, ni = 2, chi, cubpl, upper = 2.0 / 3.0, lower = 1.0 / 3.0;
  int nBad = 0, max_trials = 10;
  std::vector<ppo_ba_iter_stat> stats;
  int n_trials = 0;
  size_t run(const std::vector<Outcome>& script, int pass, int iterations) {
    size_t next = 0;
    bool ok = true;
    for (int i = 0; i < iterations && ok; ++i) {
      ppo_ba_iter_stat it{};
      it.pass = pass;
      it.iteration = i;
      double currentChi = chi;
      const double iniChi = currentChi;
      it.chi2_before = currentChi;
      if (i == 0) { ni = 2; nBad = 0; }
      double rho = 0;
      int qmax = 0;
      do {
        double chi2_sum, scale;
        int info;
        realise(script[next++], currentChi, cubpl, &chi2_sum, &scale, &info);
        double tempChi = chi2_sum + cubpl;
        const bool ok2 = info == 0 && std::isfinite(scale);
        if (!ok2) tempChi = std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double sc = (info == 0 ? scale : 0.0) + 1e-3;
        rho /= sc;
        it.lambda = lambda;
        if (rho > 0 && std::isfinite(tempChi) && info == 0) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, upper);
          lambda *= std::max(lower, alpha);
          ni = 2;
          currentChi = tempChi;
          it.accepted = 1;
        } else {
          lambda *= ni;
          ni *= 2;
          it.accepted = 0;
        }
        qmax++;
        n_trials++;
      } while (rho < 0 && qmax < max_trials);
      it.trials = qmax;
      it.chi2_after = currentChi;
      chi = currentChi;
      int res = 0;
      if (qmax == max_trials || rho == 0) res = 1;
      else {
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) res = 1;
      }
      it.result = res;
      ok = res == 0;
      stats.push_back(it);
    }
    return next;
  }
};

static bool same(double a, double b) { return a == b || (std::isnan(a) && std::isnan(b)); }

int main() {
  std::mt19937_64 rng(20261001);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  long dead_by_policy[2] = {0, 0};
  long dead_total = 0, live_total = 0, rejected_total = 0, relin_total = 0, kept_total = 0, terminated = 0;
  for (g_script = 0; g_script < 4000; ++g_script) {
    const int iterations = 1 + (int)(U(rng) * 12);
    const int max_trials = U(rng) < 0.2 ? 2 : 10;
    const int cap = 1 + (int)(U(rng) * kLookahead);
    const double p_rej = U(rng) < 0.4 ? 0.0 : U(rng) * 0.6, p_small = U(rng) < 0.5 ? 0.0 : U(rng) * 0.7, p_odd = U(rng) < 0.6 ? 0.0 : 0.15;
    std::vector<Outcome> script;
    for (int i = 0; i < iterations * max_trials + 4; ++i) {
      const double u = U(rng);
      int kind = 0;
      if (u < p_odd) kind = 3 + (int)(U(rng) * 3);
      else if (u < p_odd + p_rej) kind = 2;
      else if (u < p_odd + p_rej + p_small) kind = 1;
      script.push_back(Outcome{kind, U(rng)});
    }
    SimDevice dev;
    dev.script = script;
    LmCtl m{};
    m.lambda = 1e-3 * (1 + U(rng));
    m.chi_cur = m.ini_chi = 1000.0 * (1 + 9 * U(rng));
    m.ni = 2;
    m.cubpl = U(rng) < 0.5 ? 0.0 : 12.5;
    m.lm_upper = 2.0 / 3.0; m.lm_lower = 1.0 / 3.0;
    m.max_trials = max_trials;
    m.epoch = 41;
    dev.ctl = m;
    dev.host_epoch = m.epoch;
    if (U(rng) < 0.5) { dev.truth.S_is_zero = dev.host.S_is_zero = false; }  // (first pass after create: S not cleared by an update yet)
    SerialLm ref;
    ref.lambda = m.lambda; ref.chi = m.chi_cur; ref.cubpl = m.cubpl; ref.max_trials = max_trials;
    const int pass = (int)(U(rng) * 2);
    // both batch policies walk the same script: same statistics, same final record; what differs is how many queued attempts find their epoch gone
    {
      SimDevice other = dev;
      LmCtl mo = m;
      CHECK(lm_pipelined_pass(other, pass, iterations, mo, cap, false) == PPO_BA_OK);
      CHECK(other.n_live == 0 || other.stats.size() > 0);
      dead_by_policy[0] += other.n_dead;
    }
    const int rc = lm_pipelined_pass(dev, pass, iterations, m, cap, true);
    dead_by_policy[1] += dev.n_dead;
    const size_t used = ref.run(script, pass, iterations);
    CHECK(rc == PPO_BA_OK);
    CHECK(dev.pending.empty() && dev.next == used);
    CHECK(dev.n_trials == ref.n_trials && dev.n_live == ref.n_trials);
    CHECK(dev.stats.size() == ref.stats.size());
    for (size_t i = 0; i < ref.stats.size(); ++i) {
      const ppo_ba_iter_stat &a = dev.stats[i], &b = ref.stats[i];
      CHECK(a.pass == b.pass && a.iteration == b.iteration && a.trials == b.trials && a.accepted == b.accepted && a.result == b.result);
      CHECK(same(a.chi2_before, b.chi2_before) && same(a.chi2_after, b.chi2_after));
      CHECK(std::fabs(a.lambda - b.lambda) <= 1e-13 * b.lambda);  // (one ulp per accepted trial: pow against the rounded cube)
    }
    // the host's view at the end is the truth
    CHECK(dev.host.cur == dev.truth.cur && dev.host.lin_buf == dev.truth.lin_buf && dev.host_epoch == dev.ctl.epoch);
    CHECK(dev.host.S_is_zero == dev.truth.S_is_zero && dev.host.lin_clean[0] == dev.truth.lin_clean[0] && dev.host.lin_clean[1] == dev.truth.lin_clean[1]);
    CHECK(m.lambda == dev.ctl.lambda && m.ni == dev.ctl.ni && m.nbad == dev.ctl.nbad && m.iter == dev.ctl.iter && m.qmax == dev.ctl.qmax && m.epoch == dev.ctl.epoch);
    CHECK(same(m.chi_cur, dev.ctl.chi_cur) && same(m.ini_chi, dev.ctl.ini_chi));
    CHECK(std::fabs(m.lambda - ref.lambda) <= 1e-12 * ref.lambda && ref.nBad == m.nbad);
    dead_total += dev.n_dead; live_total += dev.n_live; relin_total += dev.n_relinearize; kept_total += dev.n_kept_error_passes;
    for (const auto& s : dev.stats) { rejected_total += s.trials - s.accepted; terminated += s.result; }
  }
  CHECK(dead_by_policy[1] < dead_by_policy[0]);  // backing off for the rest of the pass wastes fewer queued attempts than doubling back up
  CHECK(dead_total > 500 && rejected_total > 1000 && relin_total > 50 && kept_total > 50 && terminated > 200);
  // the stop flag: raised at a random poll, the pass ends between batches or inside a retry loop; what is reported is consistent and complete
  for (g_script = 10000; g_script < 10500; ++g_script) {
    SimDevice dev;
    for (int i = 0; i < 200; ++i) dev.script.push_back(Outcome{U(rng) < 0.3 ? 2 : 0, U(rng)});
    LmCtl m{};
    m.lambda = 1e-3; m.chi_cur = m.ini_chi = 5000.0; m.ni = 2; m.lm_upper = 2.0 / 3.0; m.lm_lower = 1.0 / 3.0; m.max_trials = 10; m.epoch = 3;
    dev.ctl = m; dev.host_epoch = 3;
    dev.stop_after_polls = (int)(U(rng) * 8);
    CHECK(lm_pipelined_pass(dev, 0, 10, m, kLookahead) == PPO_BA_OK);
    CHECK(dev.pending.empty() && dev.host.cur == dev.truth.cur && dev.host.lin_buf == dev.truth.lin_buf && dev.host_epoch == dev.ctl.epoch);
    CHECK(dev.stats.size() <= 10);
    for (size_t i = 0; i < dev.stats.size(); ++i) CHECK(dev.stats[i].iteration == (int)i && dev.stats[i].chi2_after <= dev.stats[i].chi2_before);
  }
  // round 6: the flag goes up MID-BATCH, while some live attempt is being evaluated: the device's decision sees it, nothing behind that attempt runs (no
  // live attempt after it), and the host's report ends with that attempt
  for (g_script = 11000; g_script < 11500; ++g_script) {
    SimDevice dev;
    for (int i = 0; i < 200; ++i) dev.script.push_back(Outcome{U(rng) < 0.3 ? 2 : 0, U(rng)});
    LmCtl m{};
    m.lambda = 1e-3; m.chi_cur = m.ini_chi = 5000.0; m.ni = 2; m.lm_upper = 2.0 / 3.0; m.lm_lower = 1.0 / 3.0; m.max_trials = 10; m.epoch = 3;
    dev.ctl = m; dev.host_epoch = 3;
    dev.stop_at_live = (int)(U(rng) * 12);
    CHECK(lm_pipelined_pass(dev, 0, 10, m, kLookahead) == PPO_BA_OK);
    CHECK(dev.pending.empty() && dev.host.cur == dev.truth.cur && dev.host.lin_buf == dev.truth.lin_buf && dev.host_epoch == dev.ctl.epoch);
    CHECK(dev.n_live <= dev.stop_at_live + 1);  // the attempt that saw the flag is the last one that ran
    CHECK(dev.n_trials == dev.n_live && dev.stats.size() <= 10);
    for (size_t i = 0; i < dev.stats.size(); ++i) CHECK(dev.stats[i].iteration == (int)i && dev.stats[i].chi2_after <= dev.stats[i].chi2_before);
  }
  std::printf("lm_pipeline_test ok: 4000 scripted passes, %ld live and %ld dead solve attempts, %ld rejected trials, %ld terminated iterations, %ld fresh linearisations, "
              "%ld kept-state error passes (dead attempts with the round-4 doubling rule: %ld); 500 passes with the stop flag between batches, 500 with the flag seen by the device mid-batch\n", live_total, dead_total, rejected_total, terminated, relin_total, kept_total,
              dead_by_policy[0]);
  return 0;
}
