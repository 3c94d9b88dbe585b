// CPU test of the host worker team of libppo_ba (csrc/ppo_pool.hpp, HostTeam): many fork-join jobs from several caller threads at
// once, every task must run exactly once, exceptions of tasks must surface in the caller, and nothing may touch a finished job.
// Built by tests/test_host_team.py with g++ (plain and with -fsanitize=thread).
#include <atomic>
#include <cstdio>
#include <numeric>
#include <thread>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>

#include "ppo_pool.hpp"

int main() {
  ppo::HostTeam& team = ppo::HostTeam::get();
  std::printf("team size %d\n", team.size());
  std::atomic<long> bad{0};
  auto caller = [&](int id) {
    for (int rep = 0; rep < 400; ++rep) {
      const int nt = 1 + (rep + id) % 9;
      std::vector<int> hits(nt, 0);
      std::vector<long> part(nt, 0);
      const int n = 1000 + 37 * rep;
      team.run(nt, [&](int t) {
        hits[t]++;
        for (int i = (int)((long)n * t / nt); i < (int)((long)n * (t + 1) / nt); ++i) part[t] += i;
      });
      long sum = std::accumulate(part.begin(), part.end(), 0L);
      if (sum != (long)n * (n - 1) / 2) bad++;
      for (int t = 0; t < nt; ++t) if (hits[t] != 1) bad++;
    }
  };
  std::vector<std::thread> th;
  for (int id = 0; id < 4; ++id) th.emplace_back(caller, id);
  for (auto& x : th) x.join();
  // a task that throws std::bad_alloc: the caller sees it, after all tasks of the job have ended
  int thrown = 0;
  std::atomic<int> ran{0};
  try {
    team.run(6, [&](int t) { ran++; if (t == 3) throw std::bad_alloc(); });
  } catch (const std::bad_alloc&) { thrown = 1; }
  if (!thrown || ran.load() != 6) bad++;
  // nested use from inside a task (the pair-plan helper does this from its own thread; from a task it must not deadlock either)
  std::atomic<int> inner{0};
  team.run(4, [&](int) { team.run(3, [&](int) { inner++; }); });
  if (inner.load() != 12) bad++;
  // a child made by fork() has no worker threads: jobs must run on the caller instead of waiting for workers that do not exist
  {
    fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) {
      std::atomic<int> c{0};
      team.run(5, [&](int) { c++; });
      _exit(c.load() == 5 ? 0 : 3);
    }
    int status = 0;
    if (pid < 0 || waitpid(pid, &status, 0) != pid || !WIFEXITED(status) || WEXITSTATUS(status) != 0) bad++;
  }
  std::printf("%s\n", bad.load() == 0 ? "OK" : "FAILED");
  return bad.load() == 0 ? 0 : 1;
}
