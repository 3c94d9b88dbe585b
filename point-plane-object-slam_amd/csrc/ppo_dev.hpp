// ppo_dev.hpp -- development / test switches are compiled in only with -DPPO_BA_TEST_HOOKS (csrc/libppo_ba_test.so, used by the
// A/B tests and by tools/): the product library libppo_ba.so reads no PPO_BA_* environment variable and has no loopback
// communicator, so nothing in a caller's environment or in a caller-supplied comm_id can change what it computes.
#pragma once
#include <cstdlib>
#ifdef PPO_BA_TEST_HOOKS
#define PPO_DEV_ENV(name) getenv(name)
#else
#define PPO_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif
