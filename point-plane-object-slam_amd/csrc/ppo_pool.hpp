// ppo_pool.hpp -- process-wide caches shared by the entry points of libppo_ba.so: device blocks, stream/event contexts,
// page-locked host staging memory (and the arena + vector type built on it).
#pragma once
#include "ppo_dev.hpp"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "ppo_chol.hpp"

namespace ppo {

// Device memory comes from a process-wide cache of freed blocks (size-bucketed, per device): a SLAM process creates and
// destroys one solver per local-BA call, and ~60 hipMalloc + hipFree pairs cost ~9 ms per call against ~10 ms of solve.
// Blocks keep their addresses while cached, so the captured Cholesky launch graph of a recurring window shape is reused too.
struct DevPool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void*>> free_;
  size_t cached = 0;
  static constexpr size_t kMaxCached = 4ull << 30;
  static size_t bucket(size_t bytes) {  // next of {1, 1.25, 1.5, 1.75} x 2^k, at least 4 KiB: <= 25 % slack
    size_t b = 4096;
    while (b * 2 <= bytes) b *= 2;
    if (b >= bytes) return b;
    for (int q = 5; q <= 8; ++q)
      if (b / 4 * q >= bytes) return b / 4 * q;
    return b * 2;
  }
  void* get(size_t bytes, int dev, size_t* got) {
    const size_t b = bucket(bytes);
    *got = b;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_.find({dev, b});
      if (it != free_.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        cached -= b;
        return p;
      }
    }
    void* p = nullptr;
    if (hipMalloc(&p, b) != hipSuccess) {
      trim();  // give cached blocks back and retry once
      if (hipMalloc(&p, b) != hipSuccess) return nullptr;
    }
    return p;
  }
  void put(void* p, size_t b, int dev) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (cached + b <= kMaxCached) {
        free_[{dev, b}].push_back(p);
        cached += b;
        return;
      }
    }
    chol_release(static_cast<double*>(p));
    (void)hipFree(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : free_)
      for (void* p : kv.second) {
        chol_release(static_cast<double*>(p));
        (void)hipFree(p);
      }
    free_.clear();
    cached = 0;
  }
};
inline DevPool g_pool;

// The stream and the mapped host block of a solver are recycled the same way (creating and destroying them costs
// several ms per solver, mostly the pinned allocation and the stream teardown).
struct DevCtx {
  int dev = -1;
  hipStream_t stream = nullptr;
  double* h_pinned = nullptr;
  double* h_pinned_dev = nullptr;
};
constexpr int kPinnedDoubles = 16;
struct CtxPool {
  std::mutex mu;
  std::vector<DevCtx> free_;
  bool get(int dev, DevCtx* c) {
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].dev == dev) {
          *c = free_[i];
          free_.erase(free_.begin() + i);
          return true;
        }
    }
    DevCtx n;
    n.dev = dev;
    bool ok = hipStreamCreateWithFlags(&n.stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&n.h_pinned, kPinnedDoubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void**)&n.h_pinned_dev, n.h_pinned, 0) == hipSuccess;
    if (!ok) { destroy(n); return false; }
    *c = n;
    return true;
  }
  void put(const DevCtx& c) {
    std::lock_guard<std::mutex> lk(mu);
    if (free_.size() < 8) { free_.push_back(c); return; }
    destroy(c);
  }
  static void destroy(const DevCtx& c) {
    if (c.h_pinned) (void)hipHostFree(c.h_pinned);
    if (c.stream) (void)hipStreamDestroy(c.stream);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (const DevCtx& c : free_) destroy(c);
    free_.clear();
  }
};
inline CtxPool g_ctx_pool;

// Host staging memory.  Everything ppo_ba_create builds for the device (edge records, CSR tables, plans) and everything
// ppo_ba_read brings back lives in page-locked blocks that the process keeps: hipMemcpyAsync from ordinary heap memory has
// to pin the pages first, and a caller whose malloc trims and regrows the heap between calls (glibc's default, measured
// from examples/replay.cpp) paid 15-25 ms per create for that against 1.5 ms once the memory was already pinned.
struct HostPool {
  std::mutex mu;
  std::map<size_t, std::vector<void*>> free_;
  size_t cached = 0;
  static constexpr size_t kMaxCached = 2ull << 30;
  void* get(size_t bytes, size_t* got, bool* pinned) {
    const size_t b = DevPool::bucket(bytes);
    *got = b;
    *pinned = true;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_.find(b);
      if (it != free_.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        cached -= b;
        return p;
      }
    }
    void* p = nullptr;
    static const bool heap = PPO_DEV_ENV("PPO_BA_STAGE_HEAP") != nullptr;  // A/B switch: stage through ordinary heap memory
    if (!heap && hipHostMalloc(&p, b, hipHostMallocDefault) == hipSuccess) return p;
    (void)hipGetLastError();
    *pinned = false;  // page-locked memory exhausted: pageable memory still works, only slower
    return std::malloc(b);
  }
  void put(void* p, size_t b, bool pinned) {
    if (!pinned) { std::free(p); return; }
    {
      std::lock_guard<std::mutex> lk(mu);
      if (cached + b <= kMaxCached) {
        free_[b].push_back(p);
        cached += b;
        return;
      }
    }
    (void)hipHostFree(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : free_)
      for (void* p : kv.second) (void)hipHostFree(p);
    free_.clear();
    cached = 0;
  }
};
inline HostPool g_host_pool;

// Bump arena of one solver over HostPool blocks.  Memory is handed out again only after the mark it was taken behind is
// rewound (ppo_ba_read) or the solver is destroyed, both after the stream has drained, so an asynchronous copy never sees its
// source reused.  Small requests share 16 MB chunks, large ones get a block of their own.
struct Stage {
  struct Block { char* p; size_t cap, used; bool pinned; };
  struct Mark { size_t n_blocks; int open; size_t open_used; };
  static constexpr size_t kChunk = 16ull << 20;
  std::mutex mu;
  std::vector<Block> blocks;
  int open = -1;  // chunk the small requests are carved from
  void* take(size_t bytes) {
    bytes = std::max<size_t>(256, (bytes + 255) & ~size_t(255));
    std::lock_guard<std::mutex> lk(mu);
    if (bytes < kChunk / 2 && open >= 0 && blocks[open].used + bytes <= blocks[open].cap) {
      void* r = blocks[open].p + blocks[open].used;
      blocks[open].used += bytes;
      return r;
    }
    Block b;
    b.p = static_cast<char*>(g_host_pool.get(bytes < kChunk / 2 ? kChunk : bytes, &b.cap, &b.pinned));
    if (!b.p) return nullptr;
    b.used = bytes;
    blocks.push_back(b);
    if (bytes < kChunk / 2) open = (int)blocks.size() - 1;
    return b.p;
  }
  Mark mark() {
    std::lock_guard<std::mutex> lk(mu);
    return Mark{blocks.size(), open, open >= 0 ? blocks[open].used : 0};
  }
  void rewind(const Mark& m) {
    std::lock_guard<std::mutex> lk(mu);
    while (blocks.size() > m.n_blocks) {
      g_host_pool.put(blocks.back().p, blocks.back().cap, blocks.back().pinned);
      blocks.pop_back();
    }
    open = m.open;
    if (open >= 0) blocks[open].used = m.open_used;
  }
  void release() { rewind(Mark{0, -1, 0}); }
};
inline thread_local Stage* t_stage = nullptr;  // the arena SVec allocations of this thread go to (set by StageScope)

struct StageScope {  // entry guard of an API call that builds SVecs; rewinds to the entry state when asked to
  Stage* prev;
  Stage* st;
  Stage::Mark m;
  bool rewind_on_exit;
  StageScope(Stage* a, bool rewind) : prev(t_stage), st(a), rewind_on_exit(rewind) {
    t_stage = a;
    if (rewind) m = a->mark();
  }
  ~StageScope() {
    if (rewind_on_exit) st->rewind(m);
    t_stage = prev;
  }
};

template <class T>
struct StageAlloc {
  using value_type = T;
  StageAlloc() = default;
  template <class U>
  StageAlloc(const StageAlloc<U>&) {}
  T* allocate(size_t n) {
    void* p = t_stage ? t_stage->take(n * sizeof(T)) : nullptr;
    if (!p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T*, size_t) {}  // see Stage: a copy enqueued from this memory may still be in flight
  template <class U>
  bool operator==(const StageAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const StageAlloc<U>&) const { return false; }
};
template <class T>
using SVec = std::vector<T, StageAlloc<T>>;

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t cap_bytes = 0;
  int dev = 0;
  bool alloc(size_t count) {
    n = count;
    p = nullptr;
    cap_bytes = 0;
    if (count == 0) return true;
    (void)hipGetDevice(&dev);
    p = static_cast<T*>(g_pool.get(count * sizeof(T), dev, &cap_bytes));
    return p != nullptr;
  }
  template <class V>
  bool upload(const V& h, hipStream_t s) {
    if (!alloc(h.size())) return false;
    if (h.empty()) return true;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s) == hipSuccess;
  }
  void release() {  // the owner has synchronised its streams (free_all)
    if (p) g_pool.put(p, cap_bytes, dev);
    p = nullptr;
    n = 0;
    cap_bytes = 0;
  }
};


}  // namespace ppo
