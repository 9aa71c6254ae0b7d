// edigpu_cache.hip -- per-solve cache of sector handles (second half).
//
// The reference rebuilds the sector Hamiltonian in every tridiag_Hv_sector_* call: one build per Green's-function
// channel and eigenstate (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:351-368 -> build_Hv_sector_normal ... delete_Hv_sector_normal;
// the same in ED_SUPERC / ED_NONSU2).  A GF pass over Norb orbitals, 2 spins and a few states asks for the same handful
// of sectors again and again.  The cache keys a handle on (the bytes of struct edigpu_model, the kind of image, the
// sector labels): a repeated request returns the handle that is already on the device.  Entries are evicted least
// recently used when the device-memory budget is exceeded; the two most recently returned handles are never evicted
// (the GF loop holds the eigenstate's sector and the target sector at the same time).  A new bath (the next DMFT
// iteration) is a new model, hence new keys; edigpu_cache_clear drops the old ones.
#include <hip/hip_runtime.h>

#include <cstring>
#include <list>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/edigpu.h"
#include "edigpu_internal.hpp"

using edigpu::set_error;

struct edigpu_cache_s {
  struct Entry {
    edigpu_model model;
    int kind;        // 0 normal (ed_total_ud = T), 1 stored superc / nonsu2, 2 on-the-fly, 3 normal with complex algebra
    int q1, q2;
    edigpu_handle h;
    int64_t bytes;
  };
  std::mutex mu;
  std::list<Entry> lru;  // front = most recently used
  int64_t budget = 0, used = 0;
  int64_t hits = 0, misses = 0, evictions = 0;
};

static int64_t handle_bytes_estimate(int64_t free_before, edigpu_handle h) {
  size_t fr = 0, tot = 0;
  int64_t built = 0;
  if (hipMemGetInfo(&fr, &tot) == hipSuccess) built = free_before - (int64_t)fr;
  if (built < 0) built = 0;
  int64_t info[10] = {0};
  (void)edigpu_info(h, info);
  // + the Lanczos workspace the first recurrence on the handle allocates (three vectors)
  return built + 3 * info[1] * (info[3] ? 16 : 8);
}

extern "C" {

int edigpu_cache_create(edigpu_cache* c, int64_t max_device_bytes) {
  if (!c || max_device_bytes < 0) {
    set_error("edigpu_cache_create: bad argument");
    return 1;
  }
  *c = new edigpu_cache_s();
  (*c)->budget = max_device_bytes;
  return 0;
}

int edigpu_cache_clear(edigpu_cache c) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lk(c->mu);
  for (auto& e : c->lru) (void)edigpu_destroy(e.h);
  c->lru.clear();
  c->used = 0;
  return 0;
}

int edigpu_cache_destroy(edigpu_cache c) {
  if (!c) return 0;
  (void)edigpu_cache_clear(c);
  delete c;
  return 0;
}

int edigpu_cache_stats(edigpu_cache c, int64_t stats[5]) {
  if (!c || !stats) {
    set_error("edigpu_cache_stats: NULL argument");
    return 1;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  stats[0] = c->hits;
  stats[1] = c->misses;
  stats[2] = c->evictions;
  stats[3] = c->used;
  stats[4] = (int64_t)c->lru.size();
  return 0;
}

int edigpu_cache_get(edigpu_cache c, const edigpu_model* model, int kind, int q1, int q2, edigpu_handle* h) {
  if (!c || !model || !h || kind < 0 || kind > 3) {
    set_error("edigpu_cache_get: bad argument");
    return 1;
  }
  *h = nullptr;
  std::lock_guard<std::mutex> lk(c->mu);
  for (auto it = c->lru.begin(); it != c->lru.end(); ++it)
    if (it->kind == kind && it->q1 == q1 && it->q2 == q2 && std::memcmp(&it->model, model, sizeof(edigpu_model)) == 0) {
      c->lru.splice(c->lru.begin(), c->lru, it);
      c->hits++;
      *h = it->h;
      return 0;
    }
  c->misses++;
  size_t fr = 0, tot = 0;
  const int64_t free_before = hipMemGetInfo(&fr, &tot) == hipSuccess ? (int64_t)fr : 0;
  edigpu_handle nh = nullptr;
  int rc = 1;
  switch (kind) {
    case 0: rc = edigpu_normal_build(&nh, model, q1, q2, 0, -1); break;
    case 1: rc = edigpu_flat_build(&nh, model, q1, 0, -1); break;
    case 2: rc = edigpu_direct_build(&nh, model, q1, 0, -1); break;
    case 3: rc = edigpu_normal_build_z(&nh, model, q1, q2); break;
  }
  if (rc) return 1;  // the builder's message stands
  edigpu_cache_s::Entry e;
  e.model = *model;
  e.kind = kind;
  e.q1 = q1;
  e.q2 = q2;
  e.h = nh;
  e.bytes = handle_bytes_estimate(free_before, nh);
  c->lru.push_front(e);
  c->used += e.bytes;
  // evict from the cold end, never the two most recently returned handles
  while (c->used > c->budget && c->lru.size() > 2) {
    auto& victim = c->lru.back();
    (void)edigpu_destroy(victim.h);
    c->used -= victim.bytes;
    c->lru.pop_back();
    c->evictions++;
  }
  *h = nh;
  return 0;
}

}  // extern "C"
