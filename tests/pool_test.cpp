// CPU test of the bookkeeping round 5 added to the device block cache and the staging arena (csrc/ppo_pool.hpp): the slab that ppo_ba_warmup
// reserves -- a cache miss is a pointer bump, blocks carved from it live in the ordinary size-class free lists, they are never freed one by one,
// the slab goes back only when nobody holds a block of it -- and Stage::owns_pinned, which decides whether an upload may be batched (the copy
// kernel reads the source directly: only page-locked arena memory qualifies).  The slab here is an address range that is never dereferenced.
// Built by tests/test_host_team.py with g++; needs no GPU.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "ppo_pool.hpp"

namespace ppo {
void chol_release(double*) {}  // (ppo_chol.hip in the library: drops the cached launch graph of a matrix that is about to be freed)
}

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
  using namespace ppo;
  DevPool pool;
  const int dev = 0;
  char* const base = reinterpret_cast<char*>(0x7f0000000000ull);  // never touched
  const size_t cap = (size_t)64 << 20;
  pool.reserve_external(dev, base, cap);
  // misses are carved from the slab, back to back, in size-class granularity
  size_t got = 0;
  std::vector<std::pair<void*, size_t>> held;
  size_t expect_used = 0;
  for (size_t bytes : {1000ul, 5000ul, 70000ul, 1ul << 20, 3ul << 20, 4097ul}) {
    void* p = pool.get(bytes, dev, &got);
    CHECK(p == base + expect_used);
    CHECK(got == DevPool::bucket(bytes) && got >= bytes && got % 1024 == 0);
    expect_used += got;
    held.push_back({p, got});
  }
  CHECK(pool.slabs_[dev].used == expect_used && pool.slabs_[dev].outstanding == held.size());
  // a block that comes back is found again by the next request of its size class, not carved anew
  pool.put(held[2].first, held[2].second, dev);
  CHECK(pool.slabs_[dev].outstanding == held.size() - 1);
  void* again = pool.get(66000, dev, &got);  // same class as 70 000
  CHECK(again == held[2].first && got == held[2].second && pool.slabs_[dev].used == expect_used && pool.slabs_[dev].outstanding == held.size());
  // giving the cache back while blocks are held: the slab stays AND keeps its free blocks in their size-class lists (its bump pointer
  // cannot be rewound while blocks are out: forgetting them would lose their space until the last holder lets go)
  pool.put(held[0].first, held[0].second, dev);
  pool.trim();
  CHECK(pool.slabs_[dev].base == base && pool.slabs_[dev].outstanding == held.size() - 1 && pool.cached == held[0].second);
  void* fresh = pool.get(1000, dev, &got);
  CHECK(fresh == held[0].first && got == held[0].second && pool.slabs_[dev].used == expect_used && pool.cached == 0);  // the kept block, not a new carve
  CHECK(pool.slabs_[dev].outstanding == held.size());
  // a request the slab cannot hold falls through to hipMalloc (no device here: nullptr), the slab is untouched
  void* big = pool.get(cap, dev, &got);
  CHECK(big == nullptr && pool.slabs_[dev].used == expect_used);
  (void)hipGetLastError();
  // everything returned, then the cache given back: the slab goes (an owned one would be hipFree'd here, once)
  for (auto& h : held) pool.put(h.first, h.second, dev);
  CHECK(pool.slabs_[dev].outstanding == 0);
  pool.trim();
  CHECK(pool.slabs_[dev].base == nullptr && pool.cached == 0);
  // another device's slab is another range
  void* q = pool.get(5000, 0, &got);
  CHECK(q == nullptr);  // (device 0 has no slab any more and no GPU to allocate from; a failed hipMalloc also trims: idle slabs go back)
  (void)hipGetLastError();
  pool.reserve_external(1, base, cap);
  void* q1 = pool.get(5000, 1, &got);
  CHECK(q1 == base);
  pool.put(q1, got, 1);
  pool.trim();

  // ---- Stage::owns_pinned: without a device hipHostMalloc fails, the arena falls back to pageable memory -- which must NOT qualify
  Stage st;
  void* a = st.take(4096);
  CHECK(a != nullptr);
  bool any_pinned = false;
  for (const auto& b : st.blocks) any_pinned = any_pinned || b.pinned;
  CHECK(st.owns_pinned(a, 4096) == any_pinned);
  CHECK(!st.owns_pinned(reinterpret_cast<char*>(a) + (32u << 20), 16));  // outside every block
  int on_stack = 0;
  CHECK(!st.owns_pinned(&on_stack, sizeof(on_stack)));
  if (any_pinned) CHECK(!st.owns_pinned(reinterpret_cast<char*>(a) + 4000, 4096));  // runs past what has been handed out
  st.release();
  std::printf("pool_test OK\n");
  return 0;
}
