// ppo_dev.hpp -- development / test switches are compiled in only with -DPPO_BA_TEST_HOOKS (csrc/libppo_ba_test.so, used by the
// A/B tests and by tools/): the product library libppo_ba.so reads no PPO_BA_* environment variable and has no loopback
// communicator, so nothing in a caller's environment or in a caller-supplied comm_id can change what it computes.
#pragma once
#include <cstdio>
#include <cstdlib>
#ifdef PPO_BA_TEST_HOOKS
#define PPO_DEV_ENV(name) getenv(name)
// Switches that make the library compute WRONG results on purpose (timing probes that leave parts of a launch out).  The hooks library
// is also the one the A/B parity tests load, so a variable leaked into the environment must not go unnoticed: the value is read ONCE per
// process and switch, and announced on stderr when it is set.
inline const char* ppo_dev_probe(const char* name, const char** slot, bool* read) {
  if (!*read) {
    *read = true;
    *slot = getenv(name);
    if (*slot) std::fprintf(stderr, "[ppo_ba TEST HOOKS] %s=%s is set: timing probe, RESULTS OF THIS PROCESS ARE WRONG\n", name, *slot);
  }
  return *slot;
}
#define PPO_DEV_PROBE(name) ([]() -> const char* { static const char* v = nullptr; static bool rd = false; return ppo_dev_probe(name, &v, &rd); }())
#else
#define PPO_DEV_ENV(name) (static_cast<const char*>(nullptr))
#define PPO_DEV_PROBE(name) (static_cast<const char*>(nullptr))
#endif
