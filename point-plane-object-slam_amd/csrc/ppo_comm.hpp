// ppo_comm.hpp -- the one exchange step of the sharded solver: sum the partial reduced systems over ranks.
//
// One process per GPU.  librccl.so is opened lazily (dlopen) so that the library loads on machines without RCCL
// and single-GPU runs never touch it.  Only sum/max all-reduce on doubles are needed:
//   - per solve attempt:   the packed lower-triangle 64x64 tiles of S and b (each rank's S holds its partial Hpp, kept per
//                          rank, and its landmarks' Schur terms),
//   - per solve attempt:   the trial chi2 / scale scalars and the stop vote,
//   - first iteration of a pass: the diagonal of Hpp (sum) and the largest diagonal entry (max) for lambda_0.
#pragma once
#include "ppo_dev.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <mutex>
#include <vector>

namespace ppo {

// ---- loopback communicator (tests only: compiled in with -DPPO_BA_TEST_HOOKS, csrc/libppo_ba_test.so) ---------------------------
// A comm_id that starts with "PPOLOOP0" selects an in-process communicator: `world` solver handles of ONE process, each driven
// by its own host thread on the SAME device, meet at every collective; the last thread to arrive reduces the buffers with a
// small kernel and copies the result back.  It exists so that the sharded arithmetic of the N > 1 path (partial systems,
// landmark ownership, the gather in ppo_ba_read) can be tested for real on a single-GPU box, where RCCL cannot put two
// ranks on one device.
#ifdef PPO_BA_TEST_HOOKS
__global__ void k_loop_reduce(double* __restrict__ a, const double* __restrict__ b, size_t n, int op) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a[i] = op == 0 ? a[i] + b[i] : (a[i] > b[i] ? a[i] : b[i]);
}
struct LoopGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, arrived = 0;
  unsigned long long gen = 0;
  std::vector<double*> bufs;
  bool failed = false;
};
inline std::shared_ptr<LoopGroup> loop_group(unsigned long long key, int world) {
  static std::mutex mu;
  static std::map<unsigned long long, std::shared_ptr<LoopGroup>> groups;
  std::lock_guard<std::mutex> lk(mu);
  auto& g = groups[key];
  if (!g) {
    g = std::make_shared<LoopGroup>();
    g->world = world;
    g->bufs.assign(world, nullptr);
  }
  return g;
}
#endif  // PPO_BA_TEST_HOOKS

struct Comm {
  // minimal RCCL surface (ABI of rccl.h: ncclUniqueId is 128 opaque bytes; enums ncclFloat64 = 8, ncclSum = 0, ncclMax = 2)
  typedef struct { char internal[128]; } UniqueId;
  typedef int (*fn_get_id)(UniqueId*);
  typedef int (*fn_init_rank)(void**, int, UniqueId, int);
  typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
  typedef int (*fn_reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t);   // ncclReduce(send, recv, count, type, op, root, comm, stream)
  typedef int (*fn_bcast)(const void*, void*, size_t, int, int, void*, hipStream_t);         // ncclBroadcast(send, recv, count, type, root, comm, stream)
  typedef int (*fn_destroy)(void*);
  void* lib = nullptr;
  void* comm = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_reduce reduce = nullptr;
  fn_bcast bcast = nullptr;
  fn_destroy comm_destroy = nullptr;
  int rank = 0, world = 1;
#ifdef PPO_BA_TEST_HOOKS
  std::shared_ptr<LoopGroup> loop;  // non-null: loopback mode
#endif
  bool forced = false;  // PPO_BA_COMM_SELFTEST: a one-rank communicator, so that the whole N > 1 code path runs on a single GPU

  bool load() {
    if (lib) return true;
    // RCCL must sit on the SAME HIP runtime instance as this library (a process that also imported PyTorch holds two: ROCm's and
    // the one bundled with torch, each with its own librccl): look next to the libamdhip64 this library is bound to first.
    {
      Dl_info di;
      if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
          dir.resize(slash + 1);
          for (const char* name : {"librccl.so", "librccl.so.1"}) {
            lib = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
          }
        }
      }
    }
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
      std::fprintf(stderr, "[ppo_ba] cannot open librccl.so: %s\n", dlerror());
      return false;
    }
    get_id = (fn_get_id)dlsym(lib, "ncclGetUniqueId");
    init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
    allreduce = (fn_allreduce)dlsym(lib, "ncclAllReduce");
    reduce = (fn_reduce)dlsym(lib, "ncclReduce");
    bcast = (fn_bcast)dlsym(lib, "ncclBroadcast");
    comm_destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
    return get_id && init_rank && allreduce && reduce && bcast && comm_destroy;
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
#ifdef PPO_BA_TEST_HOOKS
    loop.reset();
    if (world > 1 && std::memcmp(id_bytes, "PPOLOOP0", 8) == 0) {
      unsigned long long key = 0;
      std::memcpy(&key, id_bytes + 8, 8);
      loop = loop_group(key, world);
      return loop->world == world && rank >= 0 && rank < world ? 0 : -1;
    }
#endif
    if (world <= 1) {
      const char* e = PPO_DEV_ENV("PPO_BA_COMM_SELFTEST");
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
#ifdef PPO_BA_TEST_HOOKS
  int loop_reduce(double* buf, size_t n, int op, hipStream_t s) {
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    LoopGroup& g = *loop;
    std::unique_lock<std::mutex> lk(g.mu);
    g.bufs[rank] = buf;
    const unsigned long long my_gen = g.gen;
    if (++g.arrived == g.world) {
      bool ok = true;
      for (int r = 1; r < g.world; ++r) hipLaunchKernelGGL(k_loop_reduce, dim3(256), dim3(256), 0, s, g.bufs[0], (const double*)g.bufs[r], n, op);
      for (int r = 1; r < g.world; ++r) ok = ok && hipMemcpyAsync(g.bufs[r], g.bufs[0], n * sizeof(double), hipMemcpyDeviceToDevice, s) == hipSuccess;
      ok = ok && hipStreamSynchronize(s) == hipSuccess;
      g.failed = g.failed || !ok;
      g.arrived = 0;
      ++g.gen;
      g.cv.notify_all();
    } else {
      g.cv.wait(lk, [&] { return g.gen != my_gen; });
    }
    return g.failed ? -1 : 0;
  }
#endif
  int sum(double* buf, size_t n, hipStream_t s) {
    if (!active() || n == 0) return 0;
#ifdef PPO_BA_TEST_HOOKS
    if (loop) return loop_reduce(buf, n, 0, s);
#endif
    return allreduce(buf, buf, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, s) == 0 ? 0 : -1;
  }
  // options.solve_on_owner: the sum lands on rank 0 only (half the traffic of an all-reduce on a ring) ...
  int sum_to_root(double* buf, size_t n, hipStream_t s) {
    if (!active() || n == 0) return 0;
#ifdef PPO_BA_TEST_HOOKS
    if (loop) return loop_reduce(buf, n, 0, s);  // (loopback: every rank receives the sum; only rank 0 reads it)
#endif
    return reduce(buf, buf, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, 0, comm, s) == 0 ? 0 : -1;
  }
  // ... and rank 0's bytes go to everybody
  int bcast_from_root(void* buf, size_t bytes, hipStream_t s) {
    if (!active() || bytes == 0) return 0;
#ifdef PPO_BA_TEST_HOOKS
    if (loop) {
      if (hipStreamSynchronize(s) != hipSuccess) return -1;
      LoopGroup& g = *loop;
      std::unique_lock<std::mutex> lk(g.mu);
      g.bufs[rank] = static_cast<double*>(buf);
      const unsigned long long my_gen = g.gen;
      if (++g.arrived == g.world) {
        bool ok = true;
        for (int r = 1; r < g.world; ++r) ok = ok && hipMemcpyAsync(g.bufs[r], g.bufs[0], bytes, hipMemcpyDeviceToDevice, s) == hipSuccess;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        g.failed = g.failed || !ok;
        g.arrived = 0;
        ++g.gen;
        g.cv.notify_all();
      } else {
        g.cv.wait(lk, [&] { return g.gen != my_gen; });
      }
      return g.failed ? -1 : 0;
    }
#endif
    return bcast(buf, buf, bytes, /*ncclUint8*/ 1, 0, comm, s) == 0 ? 0 : -1;
  }
  int max(double* buf, size_t n, hipStream_t s) {
    if (!active() || n == 0) return 0;
#ifdef PPO_BA_TEST_HOOKS
    if (loop) return loop_reduce(buf, n, 1, s);
#endif
    return allreduce(buf, buf, n, 8, /*ncclMax*/ 2, comm, s) == 0 ? 0 : -1;
  }
  void destroy() {
    if (comm && comm_destroy) comm_destroy(comm);
    comm = nullptr;
    forced = false;
#ifdef PPO_BA_TEST_HOOKS
    loop.reset();
#endif
  }
};

}  // namespace ppo
