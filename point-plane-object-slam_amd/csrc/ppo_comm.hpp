// ppo_comm.hpp -- the one exchange step of the sharded solver: sum the partial reduced systems over ranks.
//
// One process per GPU.  librccl.so is opened lazily (dlopen) so that the library loads on machines without RCCL
// and single-GPU runs never touch it.  Only sum/max all-reduce on doubles are needed:
//   - after linearisation: pose blocks HB (n_pose_free * 27 doubles) and the partial robust chi2,
//   - per solve attempt:   S (lower triangle travels inside the full column-major buffer) and bs,
//   - per solve attempt:   the trial chi2 / scale scalars.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ppo {

struct Comm {
  // minimal RCCL surface (ABI of rccl.h: ncclUniqueId is 128 opaque bytes; enums ncclFloat64 = 8, ncclSum = 0, ncclMax = 2)
  typedef struct { char internal[128]; } UniqueId;
  typedef int (*fn_get_id)(UniqueId*);
  typedef int (*fn_init_rank)(void**, int, UniqueId, int);
  typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
  typedef int (*fn_destroy)(void*);
  void* lib = nullptr;
  void* comm = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy comm_destroy = nullptr;
  int rank = 0, world = 1;
  bool forced = false;  // PPO_BA_COMM_SELFTEST: a one-rank communicator, so that the whole N > 1 code path runs on a single GPU

  bool load() {
    if (lib) return true;
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
      std::fprintf(stderr, "[ppo_ba] cannot open librccl.so: %s\n", dlerror());
      return false;
    }
    get_id = (fn_get_id)dlsym(lib, "ncclGetUniqueId");
    init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
    allreduce = (fn_allreduce)dlsym(lib, "ncclAllReduce");
    comm_destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
    return get_id && init_rank && allreduce && comm_destroy;
  }
  int unique_id(uint8_t out[128]) {
    if (!load()) return -1;
    UniqueId id;
    if (get_id(&id) != 0) return -1;
    std::memcpy(out, id.internal, 128);
    return 0;
  }
  int init(const uint8_t id_bytes[128], int rank_, int world_) {
    rank = rank_;
    world = world_;
    forced = false;
    if (world <= 1) {
      const char* e = getenv("PPO_BA_COMM_SELFTEST");
      if (!(e && e[0] == '1')) return 0;
      if (!load()) return -1;
      UniqueId self;
      if (get_id(&self) != 0 || init_rank(&comm, 1, self, 0) != 0) return -1;
      forced = true;
      return 0;
    }
    if (!load()) return -1;
    UniqueId id;
    std::memcpy(id.internal, id_bytes, 128);
    return init_rank(&comm, world, id, rank) == 0 ? 0 : -1;
  }
  bool active() const { return world > 1 || forced; }
  int sum(double* buf, size_t n, hipStream_t s) {
    if (!active() || n == 0) return 0;
    return allreduce(buf, buf, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, s) == 0 ? 0 : -1;
  }
  int max(double* buf, size_t n, hipStream_t s) {
    if (!active() || n == 0) return 0;
    return allreduce(buf, buf, n, 8, /*ncclMax*/ 2, comm, s) == 0 ? 0 : -1;
  }
  void destroy() {
    if (comm && comm_destroy) comm_destroy(comm);
    comm = nullptr;
    forced = false;
  }
};

}  // namespace ppo
