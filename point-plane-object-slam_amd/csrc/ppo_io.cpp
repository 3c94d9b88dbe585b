// ppo_io.cpp -- graph dump / replay: one binary file per local-BA call holding the arrays of
// ppo_ba_problem, so that windows captured on a machine that runs the reference (one ppo_ba_problem_save() next to the
// flattening of INTEGRATION.md) can be replayed on a GPU box.  g2o's own .g2o text format cannot carry these graphs: the
// cuboid vertex read/write are stubs in the reference (include/g2o_cuboid.h:280-281).  Host-only, no device work.
//
// File layout (little endian): "PPOBA001", uint32 n_records, then per record
//   uint32 name_len, name bytes, uint32 dtype (0 = f64, 1 = i32, 2 = u8), uint64 count, count * sizeof(dtype) bytes.
// Scalars (the n_* counts) are i32 records of count 1.  Optional arrays that are NULL are simply absent.  Unknown record
// names are skipped by the loader, so fields can be added later.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ppo_ba.h"

namespace {

enum { T_F64 = 0, T_I32 = 1, T_U8 = 2 };
const size_t kElem[3] = {8, 4, 1};

struct Field {
  const char* name;
  int dtype;
  size_t ptr_off;    // offset of the pointer member
  size_t count_off;  // offset of the int32 count member
  int width;         // elements per unit
  bool special;      // ptcub_offsets (count + 1) / ptcub_points (3 * offsets[n])
};
#define PF(name, dtype, count, width) {#name, dtype, offsetof(ppo_ba_problem, name), offsetof(ppo_ba_problem, count), width, false}
const Field kFields[] = {
    PF(pose_qt, T_F64, n_poses, 7), PF(pose_fixed, T_U8, n_poses, 1), PF(pose_cam, T_F64, n_poses, 5),
    PF(point_xyz, T_F64, n_points, 3), PF(point_fixed, T_U8, n_points, 1),
    PF(plane_coef, T_F64, n_planes, 4),
    PF(cuboid_qt, T_F64, n_cuboids, 7), PF(cuboid_scale, T_F64, n_cuboids, 3), PF(cuboid_flags, T_U8, n_cuboids, 1), PF(cuboid_fixedscale, T_F64, n_cuboids, 3),
    PF(mono_point, T_I32, n_mono, 1), PF(mono_pose, T_I32, n_mono, 1), PF(mono_obs, T_F64, n_mono, 2), PF(mono_info, T_F64, n_mono, 1), PF(mono_huber, T_F64, n_mono, 1),
    PF(stereo_point, T_I32, n_stereo, 1), PF(stereo_pose, T_I32, n_stereo, 1), PF(stereo_obs, T_F64, n_stereo, 3), PF(stereo_info, T_F64, n_stereo, 1),
    PF(stereo_huber, T_F64, n_stereo, 1),
    PF(pl_plane, T_I32, n_plane_edges, 1), PF(pl_pose, T_I32, n_plane_edges, 1), PF(pl_kind, T_U8, n_plane_edges, 1), PF(pl_meas, T_F64, n_plane_edges, 4),
    PF(pl_info, T_F64, n_plane_edges, 3), PF(pl_huber, T_F64, n_plane_edges, 1),
    PF(bbox_pose, T_I32, n_bbox, 1), PF(bbox_cuboid, T_I32, n_bbox, 1), PF(bbox_meas, T_F64, n_bbox, 4), PF(bbox_info, T_F64, n_bbox, 4), PF(bbox_huber, T_F64, n_bbox, 1),
    PF(bbox_K, T_F64, n_bbox, 9),
    PF(corner_pose, T_I32, n_corner, 1), PF(corner_cuboid, T_I32, n_corner, 1), PF(corner_meas, T_F64, n_corner, 16), PF(corner_info, T_F64, n_corner, 16),
    PF(corner_huber, T_F64, n_corner, 1), PF(corner_K, T_F64, n_corner, 9),
    PF(ptcub_cuboid, T_I32, n_ptcub, 1),
    {"ptcub_offsets", T_I32, offsetof(ppo_ba_problem, ptcub_offsets), offsetof(ppo_ba_problem, n_ptcub), 1, true},
    {"ptcub_points", T_F64, offsetof(ppo_ba_problem, ptcub_points), offsetof(ppo_ba_problem, n_ptcub), 3, true},
    PF(ptcub_margin_ratio, T_F64, n_ptcub, 1), PF(ptcub_prior_half, T_F64, n_ptcub, 3), PF(ptcub_info, T_F64, n_ptcub, 3), PF(ptcub_huber, T_F64, n_ptcub, 1),
    PF(cubpl_cuboid, T_I32, n_cubpl, 1), PF(cubpl_plane, T_I32, n_cubpl, 1), PF(cubpl_meas, T_F64, n_cubpl, 3), PF(cubpl_info, T_F64, n_cubpl, 3),
    PF(cubpl_huber, T_F64, n_cubpl, 1),
    PF(cub3d_pose, T_I32, n_cub3d, 1), PF(cub3d_cuboid, T_I32, n_cub3d, 1), PF(cub3d_meas, T_F64, n_cub3d, 10), PF(cub3d_info, T_F64, n_cub3d, 9),
    PF(cub3d_huber, T_F64, n_cub3d, 1),
};
struct Count {
  const char* name;
  size_t off;
};
#define PC(name) {#name, offsetof(ppo_ba_problem, name)}
const Count kCounts[] = {PC(n_poses), PC(n_points), PC(n_planes), PC(n_cuboids), PC(n_mono), PC(n_stereo), PC(n_plane_edges),
                         PC(n_bbox), PC(n_corner), PC(n_ptcub), PC(n_cubpl), PC(n_cub3d)};

int32_t count_of(const ppo_ba_problem* p, size_t off) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p) + off); }
const void* ptr_of(const ppo_ba_problem* p, size_t off) { return *reinterpret_cast<const void* const*>(reinterpret_cast<const char*>(p) + off); }

size_t field_count(const ppo_ba_problem* p, const Field& f) {
  const int32_t n = count_of(p, f.count_off);
  if (!f.special) return (size_t)n * f.width;
  if (std::strcmp(f.name, "ptcub_offsets") == 0) return (size_t)n + 1;
  return (n > 0 && p->ptcub_offsets) ? (size_t)3 * p->ptcub_offsets[n] : 0;  // ptcub_points
}

bool put(FILE* f, const void* p, size_t n) { return n == 0 || std::fwrite(p, 1, n, f) == n; }
bool put_record(FILE* f, const char* name, uint32_t dtype, uint64_t count, const void* data) {
  const uint32_t len = (uint32_t)std::strlen(name);
  return put(f, &len, 4) && put(f, name, len) && put(f, &dtype, 4) && put(f, &count, 8) && put(f, data, (size_t)count * kElem[dtype]);
}

// a loaded problem owns its arrays: the struct is followed in memory by the list of blocks to free
struct Loaded {
  ppo_ba_problem p;
  std::vector<void*> blocks;
};

}  // namespace

extern "C" {

int ppo_ba_problem_save(const ppo_ba_problem* p, const char* path) {
  if (!p || !path) return PPO_BA_ERR_ARG;
  FILE* f = std::fopen(path, "wb");
  if (!f) return PPO_BA_ERR_ARG;
  uint32_t n_rec = (uint32_t)(sizeof(kCounts) / sizeof(kCounts[0]));
  for (const Field& fd : kFields)
    if (ptr_of(p, fd.ptr_off) && field_count(p, fd) > 0) ++n_rec;
  bool ok = put(f, "PPOBA001", 8) && put(f, &n_rec, 4);
  for (const Count& c : kCounts) {
    const int32_t v = count_of(p, c.off);
    ok = ok && put_record(f, c.name, T_I32, 1, &v);
  }
  for (const Field& fd : kFields) {
    const void* q = ptr_of(p, fd.ptr_off);
    const size_t n = field_count(p, fd);
    if (q && n > 0) ok = ok && put_record(f, fd.name, (uint32_t)fd.dtype, n, q);
  }
  ok = (std::fclose(f) == 0) && ok;
  return ok ? PPO_BA_OK : PPO_BA_ERR_ARG;
}

int ppo_ba_problem_load(const char* path, ppo_ba_problem** out) {
  if (!path || !out) return PPO_BA_ERR_ARG;
  *out = nullptr;
  FILE* f = std::fopen(path, "rb");
  if (!f) return PPO_BA_ERR_ARG;
  Loaded* L = new Loaded();
  std::memset(&L->p, 0, sizeof(L->p));
  auto fail = [&](int code) {
    if (f) std::fclose(f);
    f = nullptr;
    for (void* b : L->blocks) std::free(b);
    delete L;
    return code;
  };
  // a record can never be larger than what is left of the file: a corrupt count must not turn into a multi-gigabyte allocation
  long file_bytes = 0;
  if (std::fseek(f, 0, SEEK_END) != 0 || (file_bytes = std::ftell(f)) < 0 || std::fseek(f, 0, SEEK_SET) != 0) return fail(PPO_BA_ERR_ARG);
  std::vector<std::pair<std::string, uint64_t>> loaded_counts;
  char magic[8];
  uint32_t n_rec = 0;
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "PPOBA001", 8) != 0 || std::fread(&n_rec, 4, 1, f) != 1 || n_rec > 4096) return fail(PPO_BA_ERR_ARG);
  for (uint32_t r = 0; r < n_rec; ++r) {
    uint32_t len = 0, dtype = 0;
    uint64_t count = 0;
    if (std::fread(&len, 4, 1, f) != 1 || len > 256) return fail(PPO_BA_ERR_ARG);
    std::string name(len, '\0');
    if (len && std::fread(&name[0], 1, len, f) != len) return fail(PPO_BA_ERR_ARG);
    if (std::fread(&dtype, 4, 1, f) != 1 || dtype > 2 || std::fread(&count, 8, 1, f) != 1 || count > (1ull << 33)) return fail(PPO_BA_ERR_ARG);
    const size_t bytes = (size_t)count * kElem[dtype];
    const long here = std::ftell(f);
    if (here < 0 || bytes > (size_t)(file_bytes - here)) return fail(PPO_BA_ERR_ARG);
    void* block = std::malloc(bytes ? bytes : 1);
    if (!block) return fail(PPO_BA_ERR_NOMEM);
    L->blocks.push_back(block);
    if (bytes && std::fread(block, 1, bytes, f) != bytes) return fail(PPO_BA_ERR_ARG);
    bool used = false;
    loaded_counts.emplace_back(name, count);
    for (const Count& c : kCounts)
      if (name == c.name && dtype == T_I32 && count == 1) {
        *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(&L->p) + c.off) = *static_cast<int32_t*>(block);
        used = true;
      }
    for (const Field& fd : kFields)
      if (name == fd.name && (int)dtype == fd.dtype) {
        *reinterpret_cast<const void**>(reinterpret_cast<char*>(&L->p) + fd.ptr_off) = block;
        used = true;
      }
    (void)used;  // unknown records are kept allocated and ignored
  }
  std::fclose(f);
  f = nullptr;
  // consistency: every present array must have exactly the length its count implies (a truncated or foreign file must not
  // turn into out-of-bounds reads inside ppo_ba_create)
  for (const Count& c : kCounts)
    if (count_of(&L->p, c.off) < 0) return fail(PPO_BA_ERR_ARG);
  for (const Field& fd : kFields) {
    if (!ptr_of(&L->p, fd.ptr_off)) continue;
    uint64_t have = 0;
    for (const auto& lc : loaded_counts)
      if (lc.first == fd.name) have = lc.second;
    if (have != field_count(&L->p, fd)) return fail(PPO_BA_ERR_ARG);
  }
  if (L->p.n_ptcub == 0 && !L->p.ptcub_offsets) {  // the ABI wants a valid offsets array even when it is empty
    int32_t* z = static_cast<int32_t*>(std::calloc(1, sizeof(int32_t)));
    if (!z) return fail(PPO_BA_ERR_NOMEM);
    L->blocks.push_back(z);
    L->p.ptcub_offsets = z;
  }
  *out = &L->p;
  return PPO_BA_OK;
}

void ppo_ba_problem_free(ppo_ba_problem* p) {
  if (!p) return;
  Loaded* L = reinterpret_cast<Loaded*>(p);  // p is the first member
  for (void* b : L->blocks) std::free(b);
  delete L;
}


// ---- end-of-run exports, src/System.cc:402-494 -----------------------------------------------------------------------------
// SaveKeyFrameTrajectoryTUM (:402-436): R = Rcw^T as quaternion (x y z w), t = camera centre, both float32, `fixed`,
// timestamp with 6 and the rest with 7 decimals
int ppo_ba_export_tum(const char* path, int32_t n, const double* timestamps, const double* pose_qt) {
  if (!path || n < 0 || (n && (!timestamps || !pose_qt))) return PPO_BA_ERR_ARG;
  FILE* f = std::fopen(path, "w");
  if (!f) return PPO_BA_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    const double* q = &pose_qt[7 * (size_t)i];
    const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = q[0] / nq, y = q[1] / nq, z = q[2] / nq, w = q[3] / nq;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    float c[3];  // Ow = -Rcw^T tcw (KeyFrame::GetCameraCenter)
    for (int a = 0; a < 3; ++a) c[a] = (float)-(R[a] * q[4] + R[3 + a] * q[5] + R[6 + a] * q[6]);
    // the conjugate is the camera-to-world rotation; Eigen's matrix -> quaternion yields w >= 0 when the trace is positive
    const double sgn = w < 0 ? -1.0 : 1.0;
    auto pz = [](double v) { return (double)(float)v + 0.0; };  // (+ 0.0: no "-0.0000000" in the file)
    std::fprintf(f, "%.6f %.7f %.7f %.7f %.7f %.7f %.7f %.7f\n", timestamps[i], pz(c[0]), pz(c[1]), pz(c[2]), pz(-x * sgn), pz(-y * sgn), pz(-z * sgn),
                 pz(w * sgn));
  }
  std::fclose(f);
  return PPO_BA_OK;
}

// SaveCuboidOptimized (:439-466) with cuboid::toMinimalVector (include/g2o_cuboid.h:143-162)
int ppo_ba_export_cuboids(const char* path, int32_t n, const char* const* class_names, const double* cuboid_qt, const double* cuboid_scale) {
  if (!path || n < 0 || (n && (!cuboid_qt || !cuboid_scale))) return PPO_BA_ERR_ARG;
  FILE* f = std::fopen(path, "w");
  if (!f) return PPO_BA_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    const double* q = &cuboid_qt[7 * (size_t)i];
    const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const double roll = std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
    const double pitch = std::asin(2 * (qw * qy - qz * qx));
    const double yaw = std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
    const double* sc = &cuboid_scale[3 * (size_t)i];
    std::fprintf(f, "%s %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", (class_names && class_names[i]) ? class_names[i] : "object", q[4], q[5], q[6], roll,
                 pitch, yaw, sc[0], sc[1], sc[2]);
  }
  std::fclose(f);
  return PPO_BA_OK;
}

// SavePlaneOptimized (:468-494)
int ppo_ba_export_planes(const char* path, int32_t n, const double* plane_coef) {
  if (!path || n < 0 || (n && !plane_coef)) return PPO_BA_ERR_ARG;
  FILE* f = std::fopen(path, "w");
  if (!f) return PPO_BA_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    const double* c = &plane_coef[4 * (size_t)i];
    std::fprintf(f, "%d %.6f %.6f %.6f %.6f\n", i, (double)(float)c[0], (double)(float)c[1], (double)(float)c[2], (double)(float)c[3]);
  }
  std::fclose(f);
  return PPO_BA_OK;
}

}  // extern "C"
