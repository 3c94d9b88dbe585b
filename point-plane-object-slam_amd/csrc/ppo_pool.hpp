// ppo_pool.hpp -- process-wide caches shared by the entry points of libppo_ba.so: device blocks, stream/event contexts,
// page-locked host staging memory (and the arena + vector type built on it).
#pragma once
#include "ppo_dev.hpp"
#include <exception>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <pthread.h>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "ppo_chol.hpp"

namespace ppo {

// one spin-wait hint for the host's polling loops (the host side is plain C++: nothing here may assume an x86 host)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}

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
  // A slab per device that ppo_ba_warmup() reserves (round 5): a miss of the cache is then a pointer bump instead of a hipMalloc (~0.1 ms each -- 60 of
  // them made the first create of a BASELINE-size window 8 ms after the warm-up had already paid for everything else).  Blocks carved from the slab
  // go through the same size-class free lists as every other block; they are never handed to hipFree one by one: the slab goes back as a whole once
  // nobody holds a block of it (trim, i.e. ppo_ba_release_cached).
  struct Slab { char* base = nullptr; size_t cap = 0, used = 0, outstanding = 0; bool owned = true, keep = false; };
  std::map<int, Slab> slabs_;
  Slab* slab_of(const void* p, int dev) {  // (mu held)
    auto it = slabs_.find(dev);
    if (it == slabs_.end() || !it->second.base) return nullptr;
    const char* c = static_cast<const char*>(p);
    return (c >= it->second.base && c < it->second.base + it->second.cap) ? &it->second : nullptr;
  }
  bool reserve(int dev, size_t bytes) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (slabs_[dev].base) return true;
    }
    // (the allocation itself outside the lock: half a gigabyte takes the driver milliseconds, and every create / destroy of the process takes `mu`)
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    {
      std::lock_guard<std::mutex> lk(mu);
      Slab& sl = slabs_[dev];
      if (!sl.base) {
        sl.base = static_cast<char*>(p); sl.cap = bytes; sl.used = 0; sl.outstanding = 0; sl.owned = true;
        return true;
      }
    }
    (void)hipFree(p);  // somebody else reserved meanwhile
    return true;
  }
  // (tests/pool_test.cpp: the slab bookkeeping on the CPU, over an address range nobody dereferences and nobody frees)
  void reserve_external(int dev, void* base, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu);
    Slab& sl = slabs_[dev];
    sl.base = static_cast<char*>(base); sl.cap = bytes; sl.used = 0; sl.outstanding = 0; sl.owned = false;
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
        if (Slab* sl = slab_of(p, dev)) ++sl->outstanding;
        return p;
      }
      auto st = slabs_.find(dev);
      if (st != slabs_.end() && st->second.base && st->second.used + b <= st->second.cap) {
        void* p = st->second.base + st->second.used;
        st->second.used += b;  // (size classes are multiples of 1 KiB: every block stays 1 KiB-aligned)
        ++st->second.outstanding;
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
      Slab* const sl = slab_of(p, dev);
      if (sl) --sl->outstanding;
      if (cached + b <= kMaxCached || sl) {  // (a slab block cannot go anywhere else)
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
    // A slab nobody holds a block of goes back to the driver as a whole (with it every free block carved from it).  A slab that is still in use
    // KEEPS its free blocks in their size-class lists: its bump pointer cannot be rewound while blocks are out, so forgetting them would lose
    // their space until the last holder lets go.
    for (auto& kv : slabs_) {
      Slab& sl = kv.second;
      sl.keep = sl.base && sl.outstanding != 0;
    }
    size_t kept = 0;
    for (auto& kv : free_) {
      std::vector<void*> stay;
      for (void* p : kv.second) {
        Slab* const sl = slab_of(p, kv.first.first);
        if (sl && sl->keep) { stay.push_back(p); kept += kv.first.second; continue; }
        chol_release(static_cast<double*>(p));
        if (!sl) (void)hipFree(p);
      }
      kv.second.swap(stay);
    }
    for (auto it = free_.begin(); it != free_.end();) it = it->second.empty() ? free_.erase(it) : std::next(it);
    cached = kept;
    for (auto& kv : slabs_) {
      Slab& sl = kv.second;
      if (sl.base && !sl.keep) {
        if (sl.owned) (void)hipFree(sl.base);
        sl = Slab{};
      }
    }
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
constexpr int kRingSlots = 16, kRingSlotDoubles = 32;  // records of the device-side LM control (ppo_kernels.hpp: k_decide), behind the accumulator block
constexpr int kPinnedDoubles = 16 + kRingSlots * kRingSlotDoubles + 8;  // (+ the sequence word of the ring)
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
    if (!heap && hipHostMalloc(&p, b, hipHostMallocDefault) == hipSuccess) {
      // (round 5) touch it now: the pages of a fresh page-locked block are faulted in on first WRITE -- 6.3 of the 8.7 ms of the first BASELINE-size create
      // after ppo_ba_warmup were the staging copies of the raw arrays walking into a 16 MB chunk the warm-up window had used 0.2 MB of
      std::memset(p, 0, b);
      return p;
    }
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
  // [p, p + bytes) lies in a page-locked block of this arena (device-readable as it is: hipHostMalloc memory is mapped into the device's address space)
  bool owns_pinned(const void* p, size_t bytes) {
    const char* c = static_cast<const char*>(p);
    std::lock_guard<std::mutex> lk(mu);
    for (const Block& b : blocks)
      if (b.pinned && c >= b.p && c + bytes <= b.p + b.used) return true;
    return false;
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
  // resize(n) / vector(n) DEFAULT-initialise (no zero fill): the big staging arrays (19 MB of edge records, 11 MB of pair plan)
  // are written completely by the passes that size them, and a serial memset of each was ~0.3 ms of ppo_ba_create.
  // Code that needs zeros says so: assign(n, 0) / vector(n, 0).
  template <class U>
  void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
  template <class U, class... A>
  void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
  template <class U>
  bool operator==(const StageAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const StageAlloc<U>&) const { return false; }
};
template <class T>
using SVec = std::vector<T, StageAlloc<T>>;

// Host worker team.  ppo_ba_create is counting / bucketing over ~4e5 edges and ~9e5 landmark pose pairs, cut into a dozen
// fork-join phases; spawning std::threads per phase cost ~50 us per thread and phase.  The team is started once per process
// (on first use), its workers sleep on a condition variable between phases.  Several callers may fork at the same time (two
// solvers being created, or the pair-plan helper thread next to the edge records): jobs queue up, every caller also works on
// its own job, so a call never waits for workers that are busy elsewhere.
class HostTeam {
 public:
  static HostTeam& get() {
    // (never destroyed: the workers end with the process; a child made by fork() has no workers -- it runs every job on the caller)
    static HostTeam* const t = new HostTeam();
    return *t;
  }
  int size() const { return (int)workers_.size() + 1; }
  // fn(t) for t in [0, nt): t = 0 on the caller, the rest on whoever is free (the caller included).  std::bad_alloc thrown by a
  // task is rethrown here once the whole job has ended.
  template <class F>
  void run(int nt, F&& fn) {
    if (nt <= 1 || workers_.empty() || forked().load(std::memory_order_relaxed)) {
      for (int t = 0; t < nt; ++t) fn(t);
      return;
    }
    Stage* const st = t_stage;
    Job job;
    job.call = [&fn, st](int t) {
      Stage* const prev = t_stage;
      t_stage = st;
      try { fn(t); } catch (...) { t_stage = prev; throw; }
      t_stage = prev;
    };
    job.n = nt;
    job.next = 1;
    job.pending.store(nt, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(mu_);
      queue_.push_back(&job);
      n_queued_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    execute(&job, 0);
    for (;;) {  // unclaimed tasks of my own job: do them instead of waiting
      int t = -1;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (job.next < job.n) t = claim(&job);
      }
      if (t < 0) break;
      execute(&job, t);
    }
    // the phases of a create are 50-300 us long: spin for the stragglers before sleeping (a condition-variable wake-up costs 20-50 us)
    for (int i = 0; i < kSpin && job.pending.load(std::memory_order_acquire) != 0; ++i) relax();
    if (job.pending.load(std::memory_order_acquire) != 0) {
      std::unique_lock<std::mutex> lk(mu_);
      done_cv_.wait(lk, [&]() { return job.pending.load(std::memory_order_acquire) == 0; });
    }
    if (job.failed.load(std::memory_order_acquire)) throw std::bad_alloc();
  }

 private:
  static constexpr int kSpin = 600;  // x ~40 ns.  (Kept short: a container with a CPU quota pays for every spinning thread, and long spins
                                     // of 15 workers were measured as multi-millisecond throttling stalls inside ppo_ba_create.)
  static void relax() { cpu_relax(); }
  struct Job {
    std::function<void(int)> call;
    int n = 0, next = 0;            // (mu_)
    std::atomic<int> pending{0};    // tasks not finished yet; the job (on its caller's stack) dies when this reaches 0
    std::atomic<bool> failed{false};
  };
  HostTeam() {
    unsigned hw = std::thread::hardware_concurrency();
    int n = hw > 1 ? (int)std::min(hw - 1, 7u) : 0;
    if (const char* e = PPO_DEV_ENV("PPO_BA_HOST_THREADS")) n = std::max(0, std::min(atoi(e) - 1, 31));  // total threads a create may use
    for (int i = 0; i < n; ++i) workers_.emplace_back([this]() { loop(); });
    if (n > 0) (void)pthread_atfork(nullptr, nullptr, []() { forked().store(true); });
  }
  static std::atomic<bool>& forked() {
    static std::atomic<bool> f{false};
    return f;
  }
  ~HostTeam() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  int claim(Job* j) {  // (mu_ held) next task of j; a fully claimed job leaves the queue
    const int t = j->next++;
    if (j->next >= j->n) {
      queue_.erase(std::find(queue_.begin(), queue_.end(), j));
      n_queued_.fetch_sub(1, std::memory_order_release);
    }
    return t;
  }
  void execute(Job* j, int t) {
    try { j->call(t); } catch (...) { j->failed.store(true, std::memory_order_release); }
    if (j->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {  // (j may be gone from here on)
      { std::lock_guard<std::mutex> lk(mu_); }
      done_cv_.notify_all();
    }
  }
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      if (queue_.empty() && !stop_) {  // the next phase usually follows within microseconds: look for it before going to sleep
        lk.unlock();
        for (int i = 0; i < kSpin && n_queued_.load(std::memory_order_acquire) == 0; ++i) relax();
        lk.lock();
        cv_.wait(lk, [this]() { return stop_ || !queue_.empty(); });
      }
      if (stop_) return;
      Job* j = queue_.front();
      const int t = claim(j);
      lk.unlock();
      execute(j, t);
      lk.lock();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<Job*> queue_;
  std::atomic<int> n_queued_{0};
  std::vector<std::thread> workers_;
  bool stop_ = false;
};

// Uploads of ppo_ba_create, batched (round 5).  An ICL-sized window is ~60 arrays of a few KB each: as one hipMemcpyAsync per array the host spent
// 0.2 ms of a 0.5 ms create in the runtime's copy path.  While a batch is open on this thread, DevBuf::upload only RECORDS (destination, source,
// bytes) for sources that lie in the page-locked staging arena; the owner flushes the list with ONE gather kernel that reads the staging
// memory directly (ppo_solver.hip: flush_uploads -> k_gather_uploads) before the first kernel that reads any of the destinations.  Large arrays
// (> kMaxItemBytes: at the BASELINE size the copy engine is the better path) and sources outside the arena take hipMemcpyAsync as before.
struct UploadBatch {
  struct Item { void* dst; const void* src; size_t bytes; };
  static constexpr size_t kMaxItemBytes = 1u << 20;
  std::vector<Item> items;
};
inline thread_local UploadBatch* t_upload_batch = nullptr;

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
    const size_t bytes = h.size() * sizeof(T);
    if (t_upload_batch && bytes <= UploadBatch::kMaxItemBytes && t_stage && t_stage->owns_pinned(h.data(), bytes)) {
      t_upload_batch->items.push_back(UploadBatch::Item{p, h.data(), bytes});
      return true;
    }
    return hipMemcpyAsync(p, h.data(), bytes, hipMemcpyHostToDevice, s) == hipSuccess;
  }
  void release() {  // the owner has synchronised its streams (free_all)
    if (p) g_pool.put(p, cap_bytes, dev);
    p = nullptr;
    n = 0;
    cap_bytes = 0;
  }
};


// a DevBuf that hands its block back when it goes out of scope (temporaries of ppo_ba_create: every early return releases them; the
// owner has synchronised the stream by then -- free_all on the failure paths, reset_device on the way out)
template <class T>
struct ScopedDevBuf : DevBuf<T> {
  ScopedDevBuf() = default;
  ScopedDevBuf(const ScopedDevBuf&) = delete;
  ScopedDevBuf& operator=(const ScopedDevBuf&) = delete;
  ~ScopedDevBuf() {
    // stack unwinding (std::bad_alloc from a staging vector) reaches this destructor BEFORE the handler's free_all has synchronised
    // the stream: uploads or plan kernels may still be using the block, and a block handed back now could be taken by a concurrent
    // ppo_ba_create on another stream.  Rare path: wait for the whole device.
    if (this->p && std::uncaught_exceptions() > 0) (void)hipDeviceSynchronize();
    this->release();
  }
};

}  // namespace ppo
