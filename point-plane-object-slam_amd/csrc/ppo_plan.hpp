// ppo_plan.hpp -- the per-window plans of ppo_ba_create that are built ON THE DEVICE (included by ppo_solver.hip only).
//
// ppo_ba_create flattens a window into edge records, (landmark, pose) slots, a pose-major copy of the reprojection edges and the Schur
// pair plan.  All of it -- 4e5 slots (a sort + unique per landmark, then a pose-major renumbering), 2.5e5 records, 2.5e5 pose-major
// indices, 9e5 pairs bucketed by pose pair -- is built here from the caller's raw arrays when the reprojection edge lists arrive
// sorted by point (the reference's own loop and
// include/ppo_flatten.hpp produce them that way; any other order takes the host builder in ppo_solver.hip, which produces the same
// device state -- tests/test_parity_gpu.py runs both on the same window).  Orders are the host builder's: edges of a keyframe in
// landmark order, pairs of a pose pair in (landmark, a, b) order -- both come out of STABLE radix sorts (rocPRIM), so the layout does
// not depend on how the device schedules anything.
#pragma once
#include <exception>
#include <cstring>  // (rocPRIM's texture iterator calls memset from host code)
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "ppo_kernels.hpp"
#include "ppo_pool.hpp"

namespace ppo {

// slot of (landmark l, reduced pose column pc) in the landmark-major slot table, as its device (pose-major) number; -1 if none
__device__ __forceinline__ int32_t plan_find_slot(const int32_t* __restrict__ slot_off, const int32_t* __restrict__ slot_col_lm, const int32_t* __restrict__ new_id, int l, int pc) {
  if (pc < 0) return -1;
  int lo = slot_off[l], hi = slot_off[l + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (slot_col_lm[mid] < pc) lo = mid + 1;
    else hi = mid;
  }
  return (lo < slot_off[l + 1] && slot_col_lm[lo] == pc) ? new_id[lo] : -1;
}

struct ReprojRaw {  // the caller's arrays of one reprojection edge class, staged and uploaded as they are
  const double* obs;   // [n][2] or [n][3]
  const double* info;  // [n]
  const double* huber; // [n]
  const int32_t* pose; // [n]
  const int32_t* point;// [n]
  int n;
};
struct SlotTables {
  const int32_t* slot_off;     // [n_lm + 1]
  const int32_t* slot_col_lm;  // [n_slots] reduced pose column, landmark-major
  const int32_t* new_id;       // [n_slots] landmark-major index -> device slot number
  const int32_t* pose_col;     // [Np]
  const uint8_t* point_fixed;  // [Npt]
};

// edge records + flags of the reprojection edges (the edge lists are sorted by point: record k is caller edge k)
__global__ __launch_bounds__(256) void k_plan_records(ReprojRaw mo, ReprojRaw st, SlotTables t, MonoEdge* __restrict__ mono, StereoEdge* __restrict__ stereo,
                                                      uint32_t* __restrict__ mono_flags, uint32_t* __restrict__ mono_flags0, uint32_t* __restrict__ stereo_flags,
                                                      uint32_t* __restrict__ stereo_flags0) {
  const int stride = gridDim.x * blockDim.x;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < mo.n; k += stride) {
    MonoEdge m;
    m.u = mo.obs[2 * (size_t)k]; m.v = mo.obs[2 * (size_t)k + 1]; m.info = mo.info[k]; m.huber = mo.huber[k];
    m.pose = mo.pose[k]; m.point = mo.point[k];
    const int pc = t.pose_col[m.pose];
    const bool pf = t.point_fixed[m.point];
    m.slot = pf ? -1 : plan_find_slot(t.slot_off, t.slot_col_lm, t.new_id, m.point, pc);
    m.flags = ((pf && pc < 0) ? 0u : F_ACTIVE) | F_ROBUST;
    mono[k] = m;
    mono_flags[k] = m.flags;
    mono_flags0[k] = m.flags;
  }
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < st.n; k += stride) {
    StereoEdge m{};
    m.u = st.obs[3 * (size_t)k]; m.v = st.obs[3 * (size_t)k + 1]; m.ur = st.obs[3 * (size_t)k + 2]; m.info = st.info[k]; m.huber = st.huber[k];
    m.pose = st.pose[k]; m.point = st.point[k];
    const int pc = t.pose_col[m.pose];
    const bool pf = t.point_fixed[m.point];
    m.slot = pf ? -1 : plan_find_slot(t.slot_off, t.slot_col_lm, t.new_id, m.point, pc);
    m.flags = ((pf && pc < 0) ? 0u : F_ACTIVE) | F_ROBUST;
    stereo[k] = m;
    stereo_flags[k] = m.flags;
    stereo_flags0[k] = m.flags;
  }
}

// sort key of edge k for the pose-major copy: its pose if the edge belongs there (owned point, free pose), a sentinel above every pose otherwise
__global__ __launch_bounds__(256) void k_plan_pose_keys(const int32_t* __restrict__ pose, const int32_t* __restrict__ point, int n, const int32_t* __restrict__ pose_free_id,
                                                        int pt_first, int pt_last, uint32_t sentinel, uint32_t* __restrict__ key) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int pt = point[k];
  key[k] = (pt >= pt_first && pt < pt_last && pose_free_id[pose[k]] >= 0) ? (uint32_t)pose[k] : sentinel;
}

// ---- (landmark, pose) slots --------------------------------------------------------------------------------------------------------
// One thread per landmark: the reduced columns of the free poses that observe it, gathered from its edges (points: the mono and stereo
// ranges of the sorted lists; planes: a small CSR the host builds), sorted and made unique in place.  A landmark has ~8 observers at the
// BASELINE size (a few hundred at most): an insertion sort by its own thread.  seg is addressed by the landmark's EDGE offset, so it needs
// no scan; a fixed point gets no slots.
struct SlotInputs {
  const int32_t* mo_off; const int32_t* st_off;      // [Npt + 1]
  const int32_t* mono_pose; const int32_t* stereo_pose;
  const int32_t* pl_off; const int32_t* pl_perm; const int32_t* pl_pose;  // planes: CSR over the plane edges
  const int32_t* pose_col; const uint8_t* point_fixed;
  int n_points, n_planes, n_mono, n_stereo;
};
__global__ __launch_bounds__(256) void k_plan_slot_gather(SlotInputs in, int32_t* __restrict__ seg, int32_t* __restrict__ seg_cnt) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x, n_lm = in.n_points + in.n_planes;
  if (l > n_lm) return;
  if (l == n_lm) { seg_cnt[l] = 0; return; }  // (the scan wants n_lm + 1 entries)
  int32_t* b0;
  int q = 0;
  if (l < in.n_points) {
    if (in.point_fixed[l]) { seg_cnt[l] = 0; return; }
    b0 = seg + in.mo_off[l] + in.st_off[l];
    for (int k = in.mo_off[l]; k < in.mo_off[l + 1]; ++k) { const int pc = in.pose_col[in.mono_pose[k]]; if (pc >= 0) b0[q++] = pc; }
    for (int k = in.st_off[l]; k < in.st_off[l + 1]; ++k) { const int pc = in.pose_col[in.stereo_pose[k]]; if (pc >= 0) b0[q++] = pc; }
  } else {
    const int j = l - in.n_points;
    b0 = seg + in.n_mono + in.n_stereo + in.pl_off[j];
    for (int k = in.pl_off[j]; k < in.pl_off[j + 1]; ++k) { const int pc = in.pose_col[in.pl_pose[in.pl_perm[k]]]; if (pc >= 0) b0[q++] = pc; }
  }
  for (int i = 1; i < q; ++i) {  // insertion sort (observations usually arrive in keyframe order: one comparison per entry)
    const int32_t v = b0[i];
    int j = i - 1;
    while (j >= 0 && b0[j] > v) { b0[j + 1] = b0[j]; --j; }
    b0[j + 1] = v;
  }
  int u = 0;
  for (int i = 0; i < q; ++i)
    if (u == 0 || b0[i] != b0[u - 1]) b0[u++] = b0[i];
  seg_cnt[l] = u;
}
// slots in landmark-major order: column and landmark of each; counts[0] = n_slots, counts[1] = n_pt_slots
__global__ __launch_bounds__(256) void k_plan_slot_compact(SlotInputs in, const int32_t* __restrict__ seg, const int32_t* __restrict__ slot_off, int32_t* __restrict__ slot_col_lm,
                                                           int32_t* __restrict__ slot_lm_lm, int32_t* __restrict__ counts) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x, n_lm = in.n_points + in.n_planes;
  if (l == 0) { counts[0] = slot_off[n_lm]; counts[1] = slot_off[in.n_points]; }
  if (l >= n_lm) return;
  const int32_t* b0 = l < in.n_points ? seg + in.mo_off[l] + in.st_off[l] : seg + in.n_mono + in.n_stereo + in.pl_off[l - in.n_points];
  const int s0 = slot_off[l], k = slot_off[l + 1] - s0;
  for (int i = 0; i < k; ++i) { slot_col_lm[s0 + i] = b0[i]; slot_lm_lm[s0 + i] = l; }
}
// device numbering is POSE-MAJOR: point slots by (pose column, point), then plane slots by (pose column, plane).  Sort key of the
// landmark-major slot i; positions past the last slot (the arrays are sized by an upper bound) get a sentinel above every key.
__global__ __launch_bounds__(256) void k_plan_slot_keys(const int32_t* __restrict__ slot_col_lm, const int32_t* __restrict__ slot_lm_lm, const int32_t* __restrict__ counts, int n_bound,
                                                        int n_points, int pose_bits, uint32_t* __restrict__ key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_bound) return;
  key[i] = i < counts[0] ? (((uint32_t)(slot_lm_lm[i] >= n_points) << pose_bits) | (uint32_t)(slot_col_lm[i] / 6)) : (2u << pose_bits);
}
__global__ __launch_bounds__(256) void k_plan_slot_number(const uint32_t* __restrict__ order, const int32_t* __restrict__ slot_col_lm, const int32_t* __restrict__ slot_lm_lm,
                                                          const int32_t* __restrict__ counts, int32_t* __restrict__ slot_col, int32_t* __restrict__ slot_lm, int32_t* __restrict__ new_id) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counts[0]) return;
  const int src = (int)order[i];
  slot_col[i] = slot_col_lm[src];
  slot_lm[i] = slot_lm_lm[src];
  new_id[src] = i;
}
// planes: their slots as a CSR over device slot numbers; plane edges: the slot of (plane, observing pose)
__global__ __launch_bounds__(256) void k_plan_plane_lists(const int32_t* __restrict__ slot_off, const int32_t* __restrict__ new_id, const int32_t* __restrict__ counts, int n_points,
                                                          int n_planes, int32_t* __restrict__ pl_slot_off, int32_t* __restrict__ pl_slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_pt_slots = counts[1], n_pl_slots = counts[0] - counts[1];
  if (i <= n_planes) pl_slot_off[i] = slot_off[n_points + i] - n_pt_slots;
  if (i < n_pl_slots) pl_slots[i] = new_id[n_pt_slots + i];
}
__global__ __launch_bounds__(256) void k_plan_plane_edge_slots(PlaneEdge* __restrict__ pl, PlaneEdge* __restrict__ pl0, int n, const int32_t* __restrict__ slot_off,
                                                               const int32_t* __restrict__ slot_col_lm, const int32_t* __restrict__ new_id, const int32_t* __restrict__ pose_col, int n_points) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int32_t sl = plan_find_slot(slot_off, slot_col_lm, new_id, n_points + pl[e].plane, pose_col[pl[e].pose]);
  pl[e].slot = sl;
  pl0[e].slot = sl;
}
__global__ __launch_bounds__(256) void k_plan_fill_u32(uint32_t* __restrict__ p, const int32_t* __restrict__ first, int n_bound, uint32_t v) {  // p[*first .. n_bound) = v
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *first && i < n_bound) p[i] = v;
}

// ---- Schur pair plan ---------------------------------------------------------------------------------------------------------------
struct OwnedLandmarks {  // points [pt_first, pt_last) then planes [plane_first, plane_last): one unit range, like the host builder's
  int n_points, pt_first, n_own_pt, plane_first, n_own;
  __host__ __device__ int lm_of(int u) const { return u < n_own_pt ? pt_first + u : n_points + plane_first + (u - n_own_pt); }
  __host__ __device__ int unit_of(int l) const {  // -1: not owned
    if (l < n_points) return (l >= pt_first && l < pt_first + n_own_pt) ? l - pt_first : -1;
    const int j = l - n_points - plane_first;
    return (j >= 0 && j < n_own - n_own_pt) ? n_own_pt + j : -1;
  }
};
__global__ __launch_bounds__(256) void k_plan_pair_counts(OwnedLandmarks own, const int32_t* __restrict__ slot_off, int32_t* __restrict__ npairs) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u > own.n_own) return;
  if (u == own.n_own) { npairs[u] = 0; return; }  // (the scan wants n_own + 1 entries: the last offset is the total)
  const int l = own.lm_of(u), k = slot_off[l + 1] - slot_off[l];
  npairs[u] = k * (k + 1) / 2;
}
// (a, c) of the packed lower-triangle index t = a (a + 1) / 2 + c for large t
__device__ __forceinline__ void unpack_lower_big(long long t, int& a, int& c) {
  long long r = (long long)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (r * (r + 1) / 2 > t) --r;
  while ((r + 1) * (r + 2) / 2 <= t) ++r;
  a = (int)r;
  c = (int)(t - r * (r + 1) / 2);
}
// one thread per landmark-major slot a: the keys of its pairs (a, b <= a) -- down the columns of the lower triangle of the pose-pair
// matrix, key = pose_b n_free - pose_b (pose_b - 1) / 2 + pose_a - pose_b -- at the position the host builder's (landmark, a, b) loop would
// visit them; a histogram of the keys next to it
__global__ __launch_bounds__(256) void k_plan_pair_keys(OwnedLandmarks own, const int32_t* __restrict__ slot_off, const int32_t* __restrict__ slot_col_lm,
                                                        const int32_t* __restrict__ slot_lm_lm, const int32_t* __restrict__ counts, const int32_t* __restrict__ pair_off,
                                                        uint32_t n_free, uint32_t* __restrict__ key, int32_t* __restrict__ key_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counts[0]) return;
  const int l = slot_lm_lm[i], u = own.unit_of(l);
  if (u < 0) return;
  const int s0 = slot_off[l], a = i - s0;
  const uint32_t pa = (uint32_t)(slot_col_lm[i] / 6);
  const int base = pair_off[u] + a * (a + 1) / 2;
  for (int b = 0; b <= a; ++b) {
    const uint32_t pb = (uint32_t)(slot_col_lm[s0 + b] / 6);
    const uint32_t k = pb * n_free - (pb * (pb - 1u)) / 2u + (pa - pb);
    key[base + b] = k;
    atomicAdd(&key_count[k], 1);
  }
}
// the plan entries in sorted order: perm[i] = position in the (landmark, a, b) enumeration of the i-th pair by key
__global__ __launch_bounds__(256) void k_plan_pair_fill(OwnedLandmarks own, const int32_t* __restrict__ slot_off, const int32_t* __restrict__ new_id,
                                                        const int32_t* __restrict__ pair_off, const uint32_t* __restrict__ perm, int32_t* __restrict__ pc_sj,
                                                        int32_t* __restrict__ pc_si, int32_t* __restrict__ pc_lm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pair_off[own.n_own]) return;  // (the grid is sized by an upper bound of the pair count)
  const int pos = (int)perm[i];
  int lo = 0, hi = own.n_own;  // largest u with pair_off[u] <= pos
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pair_off[mid] <= pos) lo = mid;
    else hi = mid;
  }
  int a, b;
  unpack_lower_big(pos - pair_off[lo], a, b);
  const int l = own.lm_of(lo), s0 = slot_off[l];
  pc_sj[i] = new_id[s0 + a];
  pc_si[i] = new_id[s0 + b];
  pc_lm[i] = l;
}
__global__ __launch_bounds__(256) void k_plan_chunk_counts(const int32_t* __restrict__ key_count, int n_keys, int chunk, int32_t* __restrict__ n_chunks) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_keys) return;
  n_chunks[k] = k < n_keys ? (key_count[k] + chunk - 1) / chunk : 0;
}
// chunk headers: key k (column pi, row pj >= pi of the pose-pair matrix) owns the pairs [key_start[k], key_start[k + 1]), cut into chunks of `chunk`
__global__ __launch_bounds__(256) void k_plan_chunk_fill(const int32_t* __restrict__ key_start, const int32_t* __restrict__ chunk_off, int n_keys, int n_free, int chunk,
                                                         int32_t* __restrict__ pc_colj, int32_t* __restrict__ pc_coli, int32_t* __restrict__ pc_start, int32_t* __restrict__ n_chunks_out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_keys) return;
  if (k == n_keys) {
    pc_start[chunk_off[n_keys]] = key_start[n_keys];
    *n_chunks_out = chunk_off[n_keys];
    return;
  }
  // counted from the far end the column-major keys are the packed ROW-major lower triangle of the reversed numbering
  int ra, rc;
  unpack_lower_big((long long)n_keys - 1 - k, ra, rc);
  const int pi = n_free - 1 - ra, pj = n_free - 1 - rc;
  const int c0 = chunk_off[k], c1 = chunk_off[k + 1];
  for (int c = c0; c < c1; ++c) {
    pc_colj[c] = 6 * pj;
    pc_coli[c] = 6 * pi;
    pc_start[c] = key_start[k] + (c - c0) * chunk;
  }
}

// first chunk of every 64-column block of S (the chunks are ordered by column): cb[c] = number of chunks with coli < 64 c, c = 0 .. nb
__global__ __launch_bounds__(64) void k_plan_col_blocks(const int32_t* __restrict__ pc_coli, const int32_t* __restrict__ n_chunks, int nb, int32_t* __restrict__ cb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > nb) return;
  int lo = 0, hi = *n_chunks;  // first position with coli >= 64 c
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (pc_coli[mid] < 64 * c) lo = mid + 1;
    else hi = mid;
  }
  cb[c] = lo;
}

// device-wide helpers over rocPRIM (temporary storage from the block cache; everything on the caller's stream)
struct PlanScratch {
  std::vector<DevBuf<unsigned char>> tmp;  // released by the caller after its final synchronisation
  void* take(size_t bytes) {
    tmp.emplace_back();
    if (!tmp.back().alloc(bytes ? bytes : 1)) return nullptr;
    return tmp.back().p;
  }
  void release() {
    for (auto& b : tmp) b.release();
    tmp.clear();
  }
  ~PlanScratch() {
    if (!tmp.empty() && std::uncaught_exceptions() > 0) (void)hipDeviceSynchronize();  // unwinding: the sorts / scans may still be running
    release();
  }
};
inline bool plan_exclusive_scan(PlanScratch& sc, const int32_t* in, int32_t* out, size_t n, hipStream_t s) {
  size_t bytes = 0;
  if (rocprim::exclusive_scan(nullptr, bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), s) != hipSuccess) return false;
  void* t = sc.take(bytes);
  return t && rocprim::exclusive_scan(t, bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), s) == hipSuccess;
}
// stable sort of (key, position) by key; perm_out[i] = position of the i-th smallest
inline bool plan_sort_positions(PlanScratch& sc, const uint32_t* key_in, uint32_t* key_out, uint32_t* perm_out, size_t n, int bits, hipStream_t s) {
  rocprim::counting_iterator<uint32_t> pos(0);
  size_t bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, bytes, key_in, key_out, pos, perm_out, n, 0, bits, s) != hipSuccess) return false;
  void* t = sc.take(bytes);
  return t && rocprim::radix_sort_pairs(t, bytes, key_in, key_out, pos, perm_out, n, 0, bits, s) == hipSuccess;
}
inline int plan_bits(unsigned long long max_value) {
  int b = 1;
  while (b < 32 && (max_value >> b)) ++b;
  return b;
}

}  // namespace ppo
