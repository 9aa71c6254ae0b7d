// edigpu_shard.hip -- the N > 1 path inside the library: communicator, exchange and the sharded recurrence.
//
// Takes the place of the MPI data flow of the reference's distributed products and of SciFortran's MPI Lanczos
// driver (reference paths relative to /root/reference/src/singlesite):
//   spMatVec_mpi_normal_main   ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:765-929  (two vector_transpose_MPI per
//                              product, ED_NORMAL/ED_HAMILTONIAN_NORMAL_COMMON.f90:66-167, + allgather for spH0nd)
//   spMatVec_mpi_superc_main   ED_SUPERC/ED_HAMILTONIAN_SUPERC_STORED_HxV.f90:366-432  (MPI_Allgatherv :418-421)
//   spMatVec_mpi_nonsu2_main   ED_NONSU2/ED_HAMILTONIAN_NONSU2_STORED_HxV.f90:213-267  (MPI_Allgatherv :256-259)
//   directMatVec_MPI_*         ED_*/ED_HAMILTONIAN_*_DIRECT_HxV.f90 (gather first, then compute)
//   sp_lanc_tridiag(MpiComm, spHtimesV_p, vvloc, alanc, blanc)   ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:357-365
//   scatter / gather of the seed: ED_AUX_FUNX.f90:598-928 (each rank hands over its own shard here)
//
// One process per GPU.  The communicator is RCCL over xGMI (loaded at run time: the library a host already mapped --
// PyTorch ships its own copy -- or /opt/rocm's); a second, host-staged transport through POSIX shared memory lets
// several ranks of one node share a GPU (tests of the N > 1 data flow on a one-GPU box; no RCCL needed).
// Every collective is enqueued on a HIP stream; the exchange of a product runs on a side stream and overlaps the part
// of H*v that needs no remote data.  A whole tridiagonalisation is enqueued without host synchronisation (RCCL) --
// the coefficients are read back once at the end.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "exchange_index.hpp"
#include "kernels.hpp"

namespace edigpu {

// ---------------------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------------------------------------
// types and enum values from the vendor header; the entry points themselves are looked up with dlsym
typedef ncclUniqueId rccl_unique_id;
typedef ncclComm_t rccl_comm_t;
#define RCCL_SUM ncclSum
#define RCCL_FLOAT64 ncclFloat64

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;  // optional
};

static RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.lib ? &api : nullptr;
  tried = true;
  // a copy the host process already mapped wins (one RCCL per process), then the system one
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names)
    if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char* n : names)
    if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!api.lib) api.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!api.lib) return nullptr;
#define EDIGPU_SYM(field, name)                                  \
  *(void**)(&api.field) = dlsym(api.lib, name);                 \
  if (!api.field) {                                              \
    api.lib = nullptr;                                           \
    return nullptr;                                              \
  }
  EDIGPU_SYM(GetUniqueId, "ncclGetUniqueId")
  EDIGPU_SYM(CommInitRank, "ncclCommInitRank")
  EDIGPU_SYM(CommDestroy, "ncclCommDestroy")
  EDIGPU_SYM(AllReduce, "ncclAllReduce")
  EDIGPU_SYM(AllGather, "ncclAllGather")
  EDIGPU_SYM(Send, "ncclSend")
  EDIGPU_SYM(Recv, "ncclRecv")
  EDIGPU_SYM(GroupStart, "ncclGroupStart")
  EDIGPU_SYM(GroupEnd, "ncclGroupEnd")
  EDIGPU_SYM(GetErrorString, "ncclGetErrorString")
#undef EDIGPU_SYM
  *(void**)(&api.CommCount) = dlsym(api.lib, "ncclCommCount");
  return &api;
}

#define EDIGPU_RCCL(call)                                                                              \
  do {                                                                                                 \
    ncclResult_t _r = (call);                                                                          \
    if (_r != ncclSuccess) {                                                                                     \
      set_error(std::string(#call) + " failed: " + (rccl() ? rccl()->GetErrorString(_r) : "no RCCL")); \
      return 1;                                                                                        \
    }                                                                                                  \
  } while (0)

}  // namespace edigpu

// ---------------------------------------------------------------------------------------------------------
// communicator
// ---------------------------------------------------------------------------------------------------------
struct ShmHeader {
  std::atomic<int> ready;       // set by rank 0 once the region is initialised
  std::atomic<int> arrived;     // barrier: arrivals of the current generation
  std::atomic<int> generation;  // barrier: bumped by the last arriver
  int world;
  int64_t slot_bytes;
  int64_t created_s;            // CLOCK_REALTIME seconds when rank 0 made the segment (stale leftovers are refused)
};

struct edigpu_comm_s {
  int rank = 0, world = 1;
  int kind = 0;  // 0 RCCL, 1 host-staged through shared memory
  int device = 0;
  ncclComm_t nccl = nullptr;
  hipStream_t side = nullptr;  // exchange stream: the all-to-all / all-gather of a product runs beside its local part
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;   // overlapped exchange: compute stream -> side, side -> compute stream
  hipEvent_t ev_in = nullptr, ev_out = nullptr;       // bracket of every other collective (side_begin / side_end)
  // shm transport
  std::string shm_name;
  void* shm = nullptr;
  size_t shm_bytes = 0;
  ShmHeader* hdr = nullptr;
  char* slots = nullptr;
  std::vector<char> stage;
  // sharded-recurrence workspace (grown on demand)
  // capacities in doubles: vin / vout / tmp, vfull (all-gather form only), the four exchange buffers.  Grow-only and
  // tracked one by one: a communicator serves sectors of different geometry (transposed and all-gather) in turn
  int64_t ws_chunk = 0, ws_full = 0, ws_x = 0, ws_bp = 0;
  double *vin = nullptr, *vout = nullptr, *tmp = nullptr, *vfull = nullptr;
  double *send = nullptr, *recv = nullptr, *hvc = nullptr, *back = nullptr;
  // transposed exchange on padded panels (ShardGeom::block); bp[4]: the work vector of the recurrence kept in that layout
  double* bp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double *hist = nullptr, *work = nullptr, *scr = nullptr;
  int64_t hist_cap = 0;
};

namespace edigpu {

// EDIGPU_FORCE_COLLECTIVES=1: a world of one still issues its collectives through RCCL (one-GPU rehearsal of the calls)
static bool force_collectives(const edigpu_comm_s* c) {
  static const bool f = getenv("EDIGPU_FORCE_COLLECTIVES") != nullptr;
  return f && c->kind == 0 && c->nccl != nullptr;
}

static int shm_barrier(edigpu_comm_s* c) {
  ShmHeader* h = c->hdr;
  const int gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.store(gen + 1, std::memory_order_release);
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (h->generation.load(std::memory_order_acquire) == gen) {
    std::this_thread::yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
      set_error("edigpu_comm (shm): barrier timed out after 120 s (a rank is missing)");
      return 1;
    }
  }
  return 0;
}

static char* shm_slot(edigpu_comm_s* c, int r) { return c->slots + (size_t)r * (size_t)c->hdr->slot_bytes; }

static int shm_fits(edigpu_comm_s* c, size_t bytes) {
  if ((int64_t)bytes > c->hdr->slot_bytes) {
    set_error("edigpu_comm (shm): message of " + std::to_string(bytes) + " bytes exceeds the slot size given at creation");
    return 1;
  }
  return 0;
}

// Every collective of a communicator is enqueued on the communicator's OWN stream (c->side), whatever stream the caller
// computes on: one communicator driven from two streams is legal only as long as every rank issues the same order, and
// that is exactly what a single in-order stream guarantees (the first multi-GPU run should not be the one to find out).
// side_begin: the collective waits for what `st` has enqueued so far; side_end: `st` waits for the collective.  A caller
// that overlaps the exchange with its own kernels passes c->side itself and places the two events where it needs them.
// EDIGPU_COMM_TWO_STREAMS=1 (measurement only): collectives on the caller's stream, as before round 3
static bool comm_two_streams() {
  static const bool f = getenv("EDIGPU_COMM_TWO_STREAMS") != nullptr;
  return f;
}
static int side_begin(edigpu_comm_s* c, hipStream_t st) {
  if (st == c->side) return 0;
  EDIGPU_HIP(hipEventRecord(c->ev_in, st));
  EDIGPU_HIP(hipStreamWaitEvent(c->side, c->ev_in, 0));
  return 0;
}
static int side_end(edigpu_comm_s* c, hipStream_t st) {
  if (st == c->side) return 0;
  EDIGPU_HIP(hipEventRecord(c->ev_out, c->side));
  EDIGPU_HIP(hipStreamWaitEvent(st, c->ev_out, 0));
  return 0;
}

// recv[s * n .. (s+1) * n) <- block `rank` of rank s's send buffer; n doubles per block (equal split)
static int comm_all_to_all(edigpu_comm_s* c, const double* send, double* recv, size_t n, hipStream_t caller) {
  const size_t bytes = n * sizeof(double);
  if (c->world == 1 && !force_collectives(c)) {
    EDIGPU_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, caller));
    return 0;
  }
  hipStream_t st = comm_two_streams() ? caller : c->side;
  if (side_begin(c, st == c->side ? caller : st)) return 1;
  struct End {
    edigpu_comm_s* c;
    hipStream_t s;
    ~End() { (void)side_end(c, s); }
  } end_{c, st == c->side ? caller : st};
  if (c->kind == 0) {
    RcclApi* r = rccl();
    EDIGPU_RCCL(r->GroupStart());
    for (int p = 0; p < c->world; p++) {
      EDIGPU_RCCL(r->Send(send + (size_t)p * n, n, RCCL_FLOAT64, p, c->nccl, st));
      EDIGPU_RCCL(r->Recv(recv + (size_t)p * n, n, RCCL_FLOAT64, p, c->nccl, st));
    }
    EDIGPU_RCCL(r->GroupEnd());
    return 0;
  }
  if (shm_fits(c, bytes * c->world)) return 1;
  EDIGPU_HIP(hipMemcpyAsync(shm_slot(c, c->rank), send, bytes * c->world, hipMemcpyDeviceToHost, st));
  EDIGPU_HIP(hipStreamSynchronize(st));
  if (shm_barrier(c)) return 1;
  c->stage.resize(bytes * c->world);
  for (int s = 0; s < c->world; s++) memcpy(c->stage.data() + (size_t)s * bytes, shm_slot(c, s) + (size_t)c->rank * bytes, bytes);
  if (shm_barrier(c)) return 1;  // every rank has read: the slots may be overwritten
  EDIGPU_HIP(hipMemcpyAsync(recv, c->stage.data(), bytes * c->world, hipMemcpyHostToDevice, st));
  EDIGPU_HIP(hipStreamSynchronize(st));
  return 0;
}

// recv[s * n .. (s+1) * n) <- send of rank s
static int comm_all_gather(edigpu_comm_s* c, const double* send, double* recv, size_t n, hipStream_t caller) {
  const size_t bytes = n * sizeof(double);
  if (c->world == 1 && !force_collectives(c)) {
    EDIGPU_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, caller));
    return 0;
  }
  hipStream_t st = comm_two_streams() ? caller : c->side;
  if (side_begin(c, st == c->side ? caller : st)) return 1;
  struct End {
    edigpu_comm_s* c;
    hipStream_t s;
    ~End() { (void)side_end(c, s); }
  } end_{c, st == c->side ? caller : st};
  if (c->kind == 0) {
    EDIGPU_RCCL(rccl()->AllGather(send, recv, n, RCCL_FLOAT64, c->nccl, st));
    return 0;
  }
  if (shm_fits(c, bytes)) return 1;
  EDIGPU_HIP(hipMemcpyAsync(shm_slot(c, c->rank), send, bytes, hipMemcpyDeviceToHost, st));
  EDIGPU_HIP(hipStreamSynchronize(st));
  if (shm_barrier(c)) return 1;
  c->stage.resize(bytes * c->world);
  for (int s = 0; s < c->world; s++) memcpy(c->stage.data() + (size_t)s * bytes, shm_slot(c, s), bytes);
  if (shm_barrier(c)) return 1;
  EDIGPU_HIP(hipMemcpyAsync(recv, c->stage.data(), bytes * c->world, hipMemcpyHostToDevice, st));
  EDIGPU_HIP(hipStreamSynchronize(st));
  return 0;
}

// buf[0..n) <- sum over the ranks (the same order on every rank: bit-identical results everywhere)
static int comm_all_reduce(edigpu_comm_s* c, double* buf, size_t n, hipStream_t caller) {
  if (c->world == 1 && !force_collectives(c)) return 0;
  hipStream_t st = comm_two_streams() ? caller : c->side;
  if (side_begin(c, st == c->side ? caller : st)) return 1;
  struct End {
    edigpu_comm_s* c;
    hipStream_t s;
    ~End() { (void)side_end(c, s); }
  } end_{c, st == c->side ? caller : st};
  if (c->kind == 0) {
    EDIGPU_RCCL(rccl()->AllReduce(buf, buf, n, RCCL_FLOAT64, RCCL_SUM, c->nccl, st));
    return 0;
  }
  const size_t bytes = n * sizeof(double);
  if (shm_fits(c, bytes)) return 1;
  EDIGPU_HIP(hipMemcpyAsync(shm_slot(c, c->rank), buf, bytes, hipMemcpyDeviceToHost, st));
  EDIGPU_HIP(hipStreamSynchronize(st));
  if (shm_barrier(c)) return 1;
  std::vector<double> acc(n, 0.0);
  for (int s = 0; s < c->world; s++) {
    const double* p = reinterpret_cast<const double*>(shm_slot(c, s));
    for (size_t i = 0; i < n; i++) acc[i] += p[i];
  }
  if (shm_barrier(c)) return 1;
  EDIGPU_HIP(hipMemcpyAsync(buf, acc.data(), bytes, hipMemcpyHostToDevice, st));
  EDIGPU_HIP(hipStreamSynchronize(st));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// vector kernels of the one-reduction recurrence on shards.  Per step the ranks reduce THREE sums,
//   T = (<v|w>, sum (w - sg v)^2, <v|v>),   sg = alpha of the previous step (0 on the first),
// from which alpha = T[0] and beta^2 = |w - alpha v|^2 = T[1] - 2 d (alpha - sg T[2]) + d^2 T[2], d = alpha - sg --
// an identity (no |v| = 1 assumed), well conditioned for spectra far from zero (see k_finalize_ab).
// ---------------------------------------------------------------------------------------------------------
constexpr int kShNT = 256;

__device__ inline void ab_from_sums(const double* __restrict__ t, const double* __restrict__ tprev, double& a, double& b) {
  const double sg = tprev ? tprev[0] : 0.0;
  a = t[0];
  const double d = a - sg;
  const double b2 = t[1] - 2.0 * d * (a - sg * t[2]) + d * d * t[2];
  b = sqrt(b2 > 0.0 ? b2 : 0.0);
}

__device__ inline void block_sum3(double s0, double s1, double s2, double* __restrict__ partial, int stride) {
  __shared__ double ws[3][kShNT / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off, 64);
    s1 += __shfl_down(s1, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    ws[0][w] = s0;
    ws[1][w] = s1;
    ws[2][w] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int i = 0; i < kShNT / 64; i++) {
      t0 += ws[0][i];
      t1 += ws[1][i];
      t2 += ws[2][i];
    }
    partial[blockIdx.x] = t0;
    partial[stride + blockIdx.x] = t1;
    partial[2 * stride + blockIdx.x] = t2;
  }
}

// w += tmp (+ back, the down half received from the column shards when dim_up > 0); partials of the three sums
__global__ void __launch_bounds__(kShNT)
    ks_add_dot3(int64_t n, int64_t dim_up, int64_t q, int64_t pcol, int halo, const double* __restrict__ vin,
                double* __restrict__ vout, const double* __restrict__ tmp, const double* __restrict__ back,
                const double* __restrict__ tprev, double* __restrict__ partial) {
  const double sg = tprev ? tprev[0] : 0.0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * kShNT + threadIdx.x; e < n; e += (int64_t)gridDim.x * kShNT) {
    double w = vout[e] + tmp[e];
    if (back) {
      const int64_t i = e / dim_up, col = e - i * dim_up;
      w += back[xch_back_slot(i, col, q, pcol, halo)];
    }
    vout[e] = w;
    const double v = vin[e], d = w - sg * v;
    s0 += v * w;
    s1 += d * d;
    s2 += v * v;
  }
  block_sum3(s0, s1, s2, partial, kRedBlocks);
}

// The recurrence on vectors that share one layout element by element (kept in the padded panel layout: the padding holds
// zeros in every buffer, so the three sums are those of the shard).  Two buffers that swap roles every step, eight vector
// passes per step:
//   ks_panel_rotate : x <- (x - alpha v) / beta   in place on the buffer that held w (v = the previous Lanczos vector)
//   ks_panel_add_dot: w = rowhalf + colhalf - beta v, written OVER v (no longer needed), and the three sums with x
__global__ void __launch_bounds__(kShNT)
    ks_panel_rotate(int64_t n2, double2* __restrict__ x, const double2* __restrict__ v, const double* __restrict__ t,
                    const double* __restrict__ tprev) {
  double a, b;
  ab_from_sums(t, tprev, a, b);
  const double ib = 1.0 / b;
  for (int64_t e = (int64_t)blockIdx.x * kShNT + threadIdx.x; e < n2; e += (int64_t)gridDim.x * kShNT) {
    const double2 w = x[e], p = v[e];
    x[e] = make_double2((w.x - a * p.x) * ib, (w.y - a * p.y) * ib);
  }
}

// t / tprev: the reduced sums of the two previous steps (null on the first step, where there is no v: dst = the other buffer)
__global__ void __launch_bounds__(kShNT)
    ks_panel_add_dot(int64_t n2, const double2* __restrict__ x, double2* vdst, const double2* __restrict__ t1,
                     const double2* __restrict__ t2, const double* __restrict__ t, const double* __restrict__ tprev,
                     double* __restrict__ partial) {
  double sg = 0.0, b = 0.0;
  if (t) ab_from_sums(t, tprev, sg, b);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * kShNT + threadIdx.x; e < n2; e += (int64_t)gridDim.x * kShNT) {
    const double2 r = t1[e], c = t2[e], v = x[e];
    double2 w = make_double2(r.x + c.x, r.y + c.y);
    if (t) {
      const double2 p = vdst[e];
      w.x -= b * p.x;
      w.y -= b * p.y;
    }
    vdst[e] = w;
    const double dx = w.x - sg * v.x, dy = w.y - sg * v.y;
    s0 += v.x * w.x + v.y * w.y;
    s1 += dx * dx + dy * dy;
    s2 += v.x * v.x + v.y * v.y;
  }
  block_sum3(s0, s1, s2, partial, kRedBlocks);
}

__global__ void __launch_bounds__(1024) ks_sum3(const double* __restrict__ partial, int np, double* __restrict__ out) {
  __shared__ double sh[3][1024];
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < np; i += 1024)
#pragma unroll
    for (int k = 0; k < 3; k++) s[k] += partial[k * kRedBlocks + i];
#pragma unroll
  for (int k = 0; k < 3; k++) sh[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off)
#pragma unroll
      for (int k = 0; k < 3; k++) sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < 3) out[threadIdx.x] = sh[threadIdx.x][0];
}

// (v, w) <- ((w - alpha v) / beta, -beta v) from the reduced sums of the previous step; with send != null the new v
// also goes into the all-to-all send buffer [world][q][pcol + 2 halo] (halo columns into every block that holds them)
__global__ void __launch_bounds__(kShNT)
    ks_rotate3(int first, int64_t n, int64_t dim_up, int64_t q, int world, int64_t pcol, int halo, double* __restrict__ vin,
               double* __restrict__ vout, const double* __restrict__ t, const double* __restrict__ tprev,
               double* __restrict__ send) {
  double a = 0.0, b = 1.0;
  if (!first) ab_from_sums(t, tprev, a, b);
  const double ib = 1.0 / b;
  for (int64_t e = (int64_t)blockIdx.x * kShNT + threadIdx.x; e < n; e += (int64_t)gridDim.x * kShNT) {
    double x = vin[e];
    if (!first) {
      const double p = x;
      x = (vout[e] - a * p) * ib;
      vin[e] = x;
      vout[e] = -b * p;
    }
    if (send) {
      const int64_t i = e / dim_up, col = e - i * dim_up;
      int64_t clo, chi;
      xch_blocks_of(col, pcol, halo, world, clo, chi);
      for (int64_t c = clo; c <= chi; c++) send[xch_send_slot(c, i, col, q, pcol, halo)] = x;
    }
  }
}

static inline dim3 sh_grid(int64_t n, int cap) {
  int64_t nb = (n + kShNT - 1) / kShNT;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return dim3((unsigned)nb);
}

// ---------------------------------------------------------------------------------------------------------
// the sharded product and recurrence
// ---------------------------------------------------------------------------------------------------------
struct ShardGeom {
  bool transposed = false;  // normal mode, whole-sector handle: two all-to-alls per product
  int w = 1;                // doubles per element
  int64_t units = 0, unit_len = 1, q = 0, first = 0, count = 0;
  int64_t nloc = 0, chunk = 0;  // elements of this rank's shard / of the padded chunk
  int nblk = 1;                 // phonon blocks per vector (Nph + 1): the local vector is nblk blocks of blk elements
  int64_t blk = 0;
  // transposed exchange
  int halo = 0;
  int64_t pcol = 0, col_first = 0, col_count = 0, pw = 0, xlen = 0;
  // the same exchange on the padded panel layout of the local-block kernels (kernels_sb.hip, "row shards"): no packing,
  // no halo -- the partner columns of an Hnd term sit in the same panel
  bool block = false;
  int npmax = 0;      // panels per rank
  int64_t bplen = 0;  // doubles of a vector in the shard form: world * npmax * q * 16
};

static int shard_geometry(const edigpu_sector* s, const edigpu_comm_s* c, ShardGeom& g) {
  g.w = s->is_complex ? 2 : 1;
  g.transposed = s->kind == 0 && s->nloc == s->dim && normal_transposable_el(s);
  g.nblk = s->nph > 0 ? s->nph + 1 : 1;
  if (s->kind == 0) {
    g.units = s->dim_dw;
    g.unit_len = s->dim_up;
  } else {
    g.units = s->dim / g.nblk;  // electronic rows: every phonon block is sharded alike
    g.unit_len = 1;
  }
  g.q = (g.units + c->world - 1) / c->world;
  g.first = std::min<int64_t>((int64_t)c->rank * g.q, g.units);
  g.count = std::max<int64_t>(0, std::min<int64_t>(g.q, g.units - g.first));
  g.blk = g.count * g.unit_len;
  g.nloc = g.blk * g.nblk;
  g.chunk = g.q * g.unit_len * g.nblk;
  if (g.transposed) {
    g.halo = s->col_halo;
    g.pcol = (s->dim_up + c->world - 1) / c->world;
    g.col_first = std::min<int64_t>((int64_t)c->rank * g.pcol, s->dim_up);
    g.col_count = std::max<int64_t>(0, std::min<int64_t>(g.pcol, s->dim_up - g.col_first));
    g.pw = g.pcol + 2 * g.halo;
    g.xlen = (int64_t)c->world * g.q * g.pw;
    if (g.pcol < 1 || g.halo > g.pcol) g.transposed = false;  // blocks narrower than the halo: all-gather form
  }
  if (g.transposed && g.nblk == 1 && sb_shardable(s) && g.q >= 1 && g.q <= 0xFFFF && !getenv("EDIGPU_SHARD_GENERIC")) {
    g.block = true;
    g.npmax = sb_shard_panels(s, c->world);
    g.bplen = (int64_t)c->world * g.npmax * g.q * 16;
  }
  if (g.nblk > 1 && (s->sub_a || (s->kind == 0 && !g.transposed))) {
    // like spMatVec_mpi_normal_main (ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:841-904) and spMatVec_mpi_superc_main /
    // _nonsu2_main: every phonon block is exchanged on its own and the phonon / electron-phonon pass is local, which
    // holds for density couplings only
    set_error("sharded call: phonon sectors shard with density couplings g_ph(a,a) only; normal mode through the "
              "transposed exchange (whole sector, factored Hnd, at least `halo` up columns per rank), superc / nonsu2 "
              "as row shards");
    return 1;
  }
  if (!g.transposed) {
    // all-gather form: the handle must BE this rank's shard
    if (s->nloc != g.nloc || s->row_first != g.first * g.unit_len) {
      set_error("sharded call: the handle does not hold this rank's shard (build it with the first / count of "
                "edigpu_shard_plan, or -- normal mode -- build the whole sector for the transposed exchange)");
      return 1;
    }
    if (s->kind == 4) {
      set_error("sharded call: four-product complex sectors are single-shard");
      return 1;
    }
  }
  return 0;
}

template <class T>
static int regrow(T*& p, size_t n) {
  if (p) (void)hipFree(p);
  p = nullptr;
  EDIGPU_HIP(hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)));
  EDIGPU_HIP(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)));
  // the memset runs on the null stream, which the handles' non-blocking streams do not wait for
  EDIGPU_HIP(hipDeviceSynchronize());
  return 0;
}

static int comm_workspace(edigpu_comm_s* c, const ShardGeom& g, int nlanc) {
  const int64_t chunk = g.chunk * g.w;
  if (chunk > c->ws_chunk) {
    if (regrow(c->vin, chunk) || regrow(c->vout, chunk) || regrow(c->tmp, chunk)) return 1;
    c->ws_chunk = chunk;
  }
  if (!g.transposed && (chunk * c->world > c->ws_full || !c->vfull)) {
    if (regrow(c->vfull, (size_t)chunk * c->world)) return 1;
    c->ws_full = chunk * c->world;
  }
  if (g.block && g.bplen > c->ws_bp) {
    for (double*& b : c->bp)
      if (regrow(b, g.bplen)) return 1;
    c->ws_bp = g.bplen;
  }
  if (g.transposed && g.xlen > c->ws_x) {
    if (regrow(c->send, g.xlen) || regrow(c->recv, g.xlen) || regrow(c->hvc, g.xlen) || regrow(c->back, g.xlen)) return 1;
    c->ws_x = g.xlen;
  }
  if (3 * (int64_t)nlanc + 8 > c->hist_cap) {
    if (regrow(c->hist, 3 * (size_t)nlanc + 8)) return 1;
    c->hist_cap = 3 * (int64_t)nlanc + 8;
  }
  if (!c->work && regrow(c->work, 3 * (size_t)kRedBlocks)) return 1;
  if (!c->scr && regrow(c->scr, 8)) return 1;
  return 0;
}

// Transposed exchange on padded panels (the local-block kernels' layout, shard form: kernels_sb.hip).  What a rank
// sends to rank d -- its rows of d's panels -- is one contiguous run of the vector in that form, and what comes back lands
// in that layout again: the two all-to-alls move the buffers as they are.  bp[0] = v in the shard form, bp[1] = what
// the first exchange delivers, bp[2] = the column half on it, bp[3] = the row half; the exchange back reuses bp[1].
// H v = *rowhalf + *colhalf, element by element in the shard form.
static int sharded_hv_panels(edigpu_sector* s, edigpu_comm_s* c, const ShardGeom& g, hipStream_t st, const double* v,
                             const double** rowhalf, const double** colhalf) {
  const bool alone = c->world == 1 && !force_collectives(c);
  const size_t per = (size_t)g.npmax * g.q * 16;
  if (!alone) {
    EDIGPU_HIP(hipEventRecord(c->ev_ready, st));
    EDIGPU_HIP(hipStreamWaitEvent(c->side, c->ev_ready, 0));
    if (comm_all_to_all(c, v, c->bp[1], per, c->side)) return 1;
    EDIGPU_HIP(hipEventRecord(c->ev_done, c->side));
  }
  if (launch_sb_rows_shard(s, g.first, g.count, g.q, v, c->bp[3], st)) return 1;  // beside the exchange
  if (!alone) EDIGPU_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
  const int p0 = c->rank * g.npmax, np = std::max(0, std::min(g.npmax, s->ib->npanels - p0));
  // (panels past the sector's last one are never computed: their slots must not hand stale numbers back)
  if (np < g.npmax) EDIGPU_HIP(hipMemsetAsync(c->bp[2], 0, (size_t)g.bplen * sizeof(double), st));
  if (launch_sb_cols_shard(s, p0, np, g.q, g.npmax, alone ? v : c->bp[1], c->bp[2], st)) return 1;
  *rowhalf = c->bp[3];
  *colhalf = c->bp[2];
  if (!alone) {
    if (comm_all_to_all(c, c->bp[2], c->bp[1], per, st)) return 1;
    *colhalf = c->bp[1];
  }
  return 0;
}

// tmp <- (H vin) on the local rows.  pre_packed: the send buffer already holds vin (fused rotate).  The exchange runs
// on the side stream beside the part of the product that needs no remote data.
static int sharded_hv(edigpu_sector* s, edigpu_comm_s* c, const ShardGeom& g, bool pre_packed, hipStream_t st,
                      const double** back_out = nullptr) {
  if (g.block) {
    const double *rowhalf = nullptr, *colhalf = nullptr;
    if (sb_shard_to_panels(s, c->vin, c->bp[0], g.count, g.q, c->world, st)) return 1;
    if (sharded_hv_panels(s, c, g, st, c->bp[0], &rowhalf, &colhalf)) return 1;
    if (back_out) *back_out = nullptr;  // nothing left for the caller to add
    return sb_shard_from_panels_add(s, rowhalf, colhalf, c->tmp, g.count, g.q, st);
  }
  if (g.transposed) {
    // a world of one exchanges nothing: the column half reads the packed buffer and the caller its result in place
    const bool alone = c->world == 1 && !force_collectives(c);
    for (int b = 0; b < g.nblk; b++) {
      const double* vb = c->vin + (size_t)b * g.blk;
      double* tb = c->tmp + (size_t)b * g.blk;
      if ((!pre_packed || g.nblk > 1) &&
          edigpu_transpose_pack(s->dim_up, g.count, g.q, c->world, g.pcol, g.halo, vb, c->send, st))
        return 1;
      if (!alone) {
        EDIGPU_HIP(hipEventRecord(c->ev_ready, st));
        EDIGPU_HIP(hipStreamWaitEvent(c->side, c->ev_ready, 0));
        if (comm_all_to_all(c, c->send, c->recv, (size_t)(g.q * g.pw), c->side)) return 1;
        EDIGPU_HIP(hipEventRecord(c->ev_done, c->side));
      }
      if (launch_normal_rows(s, g.first, g.count, vb, tb, st)) return 1;
      if (!alone) EDIGPU_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
      if (launch_normal_cols(s, g.col_first, g.col_count, g.pw, g.halo, alone ? c->send : c->recv, c->hvc, st)) return 1;
      const double* back = alone ? c->hvc : c->back;
      if (g.nblk == 1) {
        if (back_out) *back_out = back;
        if (alone) return 0;
        return comm_all_to_all(c, c->hvc, c->back, (size_t)(g.q * g.pw), st);  // consumed by the caller (unpack)
      }
      // phonon sectors: the buffers serve the next block, so the down half is added here
      if (!alone && comm_all_to_all(c, c->hvc, c->back, (size_t)(g.q * g.pw), st)) return 1;
      if (edigpu_transpose_unpack_add(s->dim_up, g.count, g.q, c->world, g.pcol, g.halo, back, tb, st)) return 1;
    }
    if (back_out) *back_out = nullptr;
    return launch_phonon_rows(s, g.first, g.count, c->vin, c->tmp, st);  // local: stored/H_ph.f90, H_e_ph.f90
  }
  if (g.nblk > 1) {
    // superc / nonsu2 phonon shards: one all-gather per phonon block beside that block's local part, then the phonon
    // and electron-phonon terms on the rows held here (stored/H_ph.f90, H_e_ph.f90: density couplings)
    const bool alone = c->world == 1 && !force_collectives(c);
    const size_t nq = (size_t)g.q * g.unit_len * g.w;  // doubles per rank and block (padded; tails are never addressed)
    for (int b = 0; b < g.nblk; b++) {
      const double* vb = c->vin + (size_t)b * g.blk * g.w;
      double* tb = c->tmp + (size_t)b * g.blk * g.w;
      if (!alone) {
        EDIGPU_HIP(hipEventRecord(c->ev_ready, st));
        EDIGPU_HIP(hipStreamWaitEvent(c->side, c->ev_ready, 0));
        if (comm_all_gather(c, vb, c->vfull, nq, c->side)) return 1;
        EDIGPU_HIP(hipEventRecord(c->ev_done, c->side));
      }
      if (apply_flat_block(s, vb, nullptr, tb, 1, st)) return 1;
      if (!alone) EDIGPU_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
      if (apply_flat_block(s, nullptr, alone ? vb : c->vfull, tb, 2, st)) return 1;
    }
    return launch_phonon(s, c->vin, c->tmp, st);
  }
  const size_t n = (size_t)g.chunk * g.w;
  if (c->world == 1 && !force_collectives(c)) {  // the chunk is the whole vector
    if (edigpu_apply_local_dev(s, c->vin, c->tmp, st)) return 1;
    return edigpu_apply_remote_dev(s, c->vin, c->tmp, st);
  }
  EDIGPU_HIP(hipEventRecord(c->ev_ready, st));
  EDIGPU_HIP(hipStreamWaitEvent(c->side, c->ev_ready, 0));
  if (comm_all_gather(c, c->vin, c->vfull, n, c->side)) return 1;
  EDIGPU_HIP(hipEventRecord(c->ev_done, c->side));
  if (edigpu_apply_local_dev(s, c->vin, c->tmp, st)) return 1;
  EDIGPU_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
  return edigpu_apply_remote_dev(s, c->vfull, c->tmp, st);
}

// one step of the one-reduction recurrence; T_it lands in hist[3 it .. 3 it + 3)
static int sharded_step(edigpu_sector* s, edigpu_comm_s* c, const ShardGeom& g, int it, hipStream_t st) {
  const int64_t n = g.nloc * g.w;
  double* t = c->hist + 3 * (size_t)it;
  const double* tp = it > 0 ? c->hist + 3 * (size_t)(it - 1) : nullptr;
  const double* tpp = it > 1 ? c->hist + 3 * (size_t)(it - 2) : nullptr;
  if (n > 0)
    hipLaunchKernelGGL(ks_rotate3, sh_grid(n, 256 * 16), dim3(kShNT), 0, st, it == 0 ? 1 : 0, n, s->dim_up, g.q, c->world,
                       g.pcol, g.halo, c->vin, c->vout, tp, tpp, g.transposed && g.nblk == 1 && !g.block ? c->send : nullptr);
  const double* back = nullptr;
  if (sharded_hv(s, c, g, true, st, &back)) return 1;
  const dim3 gr = sh_grid(std::max<int64_t>(n, 1), kRedBlocks);
  hipLaunchKernelGGL(ks_add_dot3, gr, dim3(kShNT), 0, st, n, s->dim_up, g.q, g.pcol, g.halo, c->vin, c->vout, c->tmp,
                     g.transposed ? back : nullptr, tp, c->work);  // back == NULL: phonon blocks, already added
  hipLaunchKernelGGL(ks_sum3, dim3(1), dim3(1024), 0, st, c->work, (int)gr.x, t);
  EDIGPU_HIP(hipGetLastError());
  return comm_all_reduce(c, t, 3, st);
}

// The recurrence with its two vectors KEPT in the padded panel layout (ShardGeom::block; EDIGPU_SHARD_PANEL_LOOP=0 falls back
// to the rows of the reference's layout with a conversion on either side of every product): the Lanczos vector is what the
// first all-to-all sends as it is; it and the work vector live in bp[0] and bp[4] and swap roles every step
// (sharded_step_panels); the padding of the layout holds zeros in every buffer and stays zero under the element-wise
// updates, so the three sums are those of the shard.
static bool panel_loop(const ShardGeom& g) {
  const char* e = getenv("EDIGPU_SHARD_PANEL_LOOP");
  return g.block && !(e && atoi(e) == 0);
}

// after load_seed: the normalised seed goes into the panel layout, every other buffer of the loop starts from zeros
// (the buffers serve sectors of different geometry in turn: what one leaves behind is not zero in another's padding)
static int panel_loop_begin(edigpu_sector* s, edigpu_comm_s* c, const ShardGeom& g, hipStream_t st) {
  for (int k = 1; k < 5; k++) EDIGPU_HIP(hipMemsetAsync(c->bp[k], 0, (size_t)g.bplen * sizeof(double), st));
  return sb_shard_to_panels(s, c->vin, c->bp[0], g.count, g.q, c->world, st);
}

static int sharded_step_panels(edigpu_sector* s, edigpu_comm_s* c, const ShardGeom& g, int it, hipStream_t st) {
  const int64_t n2 = g.bplen / 2;  // (a multiple of 8)
  double* t = c->hist + 3 * (size_t)it;
  const double* tp = it > 0 ? c->hist + 3 * (size_t)(it - 1) : nullptr;
  const double* tpp = it > 1 ? c->hist + 3 * (size_t)(it - 2) : nullptr;
  // the two buffers swap roles every step: x = the vector of this step (the seed in bp[0] on step 0), o = the other one,
  // which holds the previous vector and receives the new work vector
  double* x = (it & 1) ? c->bp[4] : c->bp[0];
  double* o = (it & 1) ? c->bp[0] : c->bp[4];
  if (it > 0)
    hipLaunchKernelGGL(ks_panel_rotate, sh_grid(n2, 256 * 16), dim3(kShNT), 0, st, n2, reinterpret_cast<double2*>(x),
                       reinterpret_cast<const double2*>(o), tp, tpp);
  const double *rowhalf = nullptr, *colhalf = nullptr;
  if (sharded_hv_panels(s, c, g, st, x, &rowhalf, &colhalf)) return 1;
  const dim3 gr = sh_grid(std::max<int64_t>(n2, 1), kRedBlocks);
  hipLaunchKernelGGL(ks_panel_add_dot, gr, dim3(kShNT), 0, st, n2, reinterpret_cast<const double2*>(x),
                     reinterpret_cast<double2*>(o), reinterpret_cast<const double2*>(rowhalf),
                     reinterpret_cast<const double2*>(colhalf), tp, tpp, c->work);
  hipLaunchKernelGGL(ks_sum3, dim3(1), dim3(1024), 0, st, c->work, (int)gr.x, t);
  EDIGPU_HIP(hipGetLastError());
  return comm_all_reduce(c, t, 3, st);
}

// literal two-reduction step (beta = |w - alpha v|): alpha_it -> hist[it], beta_it^2 -> hist[nlanc + it]
static int sharded_step_exact(edigpu_sector* s, edigpu_comm_s* c, const ShardGeom& g, int it, int nlanc, hipStream_t st) {
  const int64_t n = g.nloc * g.w;
  double* al = c->hist + it;
  double* b2 = c->hist + nlanc + it;
  if (it > 0 && vec_rotate(n, c->vin, c->vout, c->hist + nlanc + it - 1, st)) return 1;
  const double* back = nullptr;
  if (sharded_hv(s, c, g, false, st, &back)) return 1;
  if (g.transposed && back &&
      edigpu_transpose_unpack_add(s->dim_up, g.count, g.q, c->world, g.pcol, g.halo, back, c->tmp, st))
    return 1;
  if (vec_add_dot(n, c->vin, c->vout, c->tmp, al, c->work, st)) return 1;
  if (comm_all_reduce(c, al, 1, st)) return 1;
  if (vec_axpy_nrm2(n, c->vin, c->vout, al, b2, c->work, st)) return 1;
  return comm_all_reduce(c, b2, 1, st);
}

static int load_seed(edigpu_comm_s* c, const ShardGeom& g, const double* vin_shard, hipStream_t st) {
  const int64_t chunk = g.chunk * g.w, n = g.nloc * g.w;
  EDIGPU_HIP(hipMemsetAsync(c->vin, 0, (size_t)std::max<int64_t>(chunk, 1) * sizeof(double), st));
  EDIGPU_HIP(hipMemsetAsync(c->vout, 0, (size_t)std::max<int64_t>(chunk, 1) * sizeof(double), st));
  if (n > 0) EDIGPU_HIP(hipMemcpyAsync(c->vin, vin_shard, (size_t)n * sizeof(double), hipMemcpyDefault, st));
  // |v|^2 over all ranks, then scale (lanczos_iteration's first step)
  if (vec_axpy_nrm2(n, c->vin, c->vin, c->scr + 1, c->scr, c->work, st)) return 1;  // scr[1] = 0: v - 0 v
  if (comm_all_reduce(c, c->scr, 1, st)) return 1;
  return vec_scale(n, c->vin, c->scr, st);
}

static int sharded_tridiag(edigpu_sector* s, edigpu_comm_s* c, const double* vin_shard, int nlanc, double* alanc,
                           double* blanc, double threshold, int* niter_done, double* norm2) {
  ShardGeom g;
  if (shard_geometry(s, c, g)) return 1;
  EDIGPU_HIP(hipSetDevice(s->device));
  if (comm_workspace(c, g, 2 * nlanc)) return 1;
  hipStream_t st = s->stream;
  std::fill(alanc, alanc + nlanc, 0.0);
  std::fill(blanc, blanc + nlanc, 0.0);
  if (niter_done) *niter_done = 0;
  const bool force_exact = getenv("EDIGPU_LANCZOS_EXACTBETA") != nullptr;
  std::vector<double> h(3 * (size_t)nlanc + 8);
  for (int pass = force_exact ? 1 : 0; pass < 2; pass++) {
    EDIGPU_HIP(hipMemsetAsync(c->hist, 0, (size_t)c->hist_cap * sizeof(double), st));
    if (load_seed(c, g, vin_shard, st)) return 1;
    const bool panels = pass == 0 && panel_loop(g);
    if (panels && panel_loop_begin(s, c, g, st)) return 1;
    for (int it = 0; it < nlanc; it++)
      if (pass == 0 ? (panels ? sharded_step_panels(s, c, g, it, st) : sharded_step(s, c, g, it, st))
                    : sharded_step_exact(s, c, g, it, nlanc, st))
        return 1;
    EDIGPU_HIP(hipMemcpyAsync(h.data(), c->hist, (3 * (size_t)nlanc) * sizeof(double), hipMemcpyDeviceToHost, st));
    double n2 = 0.0;
    EDIGPU_HIP(hipMemcpyAsync(&n2, c->scr, sizeof(double), hipMemcpyDeviceToHost, st));
    EDIGPU_HIP(hipStreamSynchronize(st));
    if (norm2) *norm2 = n2;
    if (!(n2 > 0.0)) return 0;  // zero seed: nothing to do (the reference skips such channels)
    bool redo = false;
    int ndone = nlanc;
    for (int k = 0; k < nlanc; k++) {
      double a, b;
      if (pass == 0) {
        const double* t = &h[3 * (size_t)k];
        const double sg = k > 0 ? h[3 * (size_t)(k - 1)] : 0.0;
        a = t[0];
        const double d = a - sg;
        const double b2 = t[1] - 2.0 * d * (a - sg * t[2]) + d * d * t[2];
        // more than three digits lost in the difference (the same numbers on every rank): repeat literally
        if (!(b2 > 1e-3 * t[1]) && b2 > threshold * threshold) {
          redo = true;
          break;
        }
        b = sqrt(b2 > 0.0 ? b2 : 0.0);
      } else {
        a = h[k];
        b = sqrt(h[(size_t)nlanc + k] > 0.0 ? h[(size_t)nlanc + k] : 0.0);
      }
      alanc[k] = a;
      if (!(fabs(b) > 0.0) || fabs(b) < threshold) {
        ndone = k + 1;
        break;
      }
      if (k + 1 < nlanc) blanc[k + 1] = b;
    }
    if (redo) {
      std::fill(alanc, alanc + nlanc, 0.0);
      std::fill(blanc, blanc + nlanc, 0.0);
      continue;
    }
    if (niter_done) *niter_done = ndone;
    return 0;
  }
  set_error("edigpu_lanczos_tridiag_sharded: internal error");
  return 1;
}

}  // namespace edigpu

using namespace edigpu;

extern "C" {

int edigpu_shard_plan(int64_t units, int32_t world, int32_t rank, int64_t* first, int64_t* count, int64_t* q) {
  if (units < 0 || world < 1 || rank < 0 || rank >= world) {
    set_error("edigpu_shard_plan: bad argument");
    return 1;
  }
  const int64_t qq = (units + world - 1) / world;
  const int64_t f = std::min<int64_t>((int64_t)rank * qq, units);
  if (first) *first = f;
  if (count) *count = std::max<int64_t>(0, std::min<int64_t>(qq, units - f));
  if (q) *q = qq;
  return 0;
}

// Host-only (no GPU is touched): the index arithmetic of the transposed exchange as the kernels use it
// (exchange_index.hpp), for hosts that stage the exchange themselves and for the CPU suite.
int edigpu_exchange_send_map(int64_t dim_up, int64_t nrows, int64_t q, int32_t world, int64_t pcol, int32_t halo, int64_t* src) {
  if (!src || dim_up < 1 || nrows < 0 || q < nrows || world < 1 || pcol < 1 || halo < 0) {
    set_error("edigpu_exchange_send_map: bad argument");
    return 1;
  }
  const int64_t n = (int64_t)world * q * (pcol + 2 * halo);
  for (int64_t e = 0; e < n; e++) src[e] = xch_send_source(e, dim_up, nrows, q, pcol, halo);
  return 0;
}

int edigpu_exchange_back_map(int64_t dim_up, int64_t nrows, int64_t q, int32_t world, int64_t pcol, int32_t halo, int64_t* slot) {
  if (!slot || dim_up < 1 || nrows < 0 || q < nrows || world < 1 || pcol < 1 || halo < 0 || pcol * world < dim_up) {
    set_error("edigpu_exchange_back_map: bad argument");
    return 1;
  }
  for (int64_t i = 0; i < nrows; i++)
    for (int64_t col = 0; col < dim_up; col++) slot[i * dim_up + col] = xch_back_slot(i, col, q, pcol, halo);
  return 0;
}

int edigpu_comm_unique_id(void* id128) {
  if (!id128) {
    set_error("edigpu_comm_unique_id: NULL argument");
    return 1;
  }
  RcclApi* r = rccl();
  if (!r) {
    set_error("edigpu_comm_unique_id: RCCL (librccl.so) could not be loaded");
    return 1;
  }
  rccl_unique_id id;
  EDIGPU_RCCL(r->GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

static int comm_common(edigpu_comm_s* c) {
  EDIGPU_HIP(hipGetDevice(&c->device));
  EDIGPU_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  EDIGPU_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  EDIGPU_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  EDIGPU_HIP(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
  EDIGPU_HIP(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
  return 0;
}

int edigpu_comm_create(edigpu_comm* out, int32_t rank, int32_t world, const void* id128) {
  if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) {
    set_error("edigpu_comm_create: bad argument");
    return 1;
  }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    set_error("edigpu_comm_create: no usable HIP device; this library has no CPU fallback");
    return 1;
  }
  auto* c = new edigpu_comm_s();
  c->rank = rank;
  c->world = world;
  c->kind = 0;
  if (comm_common(c)) {
    delete c;
    return 1;
  }
  if (world > 1 || id128) {
    RcclApi* r = rccl();
    if (!r) {
      set_error("edigpu_comm_create: RCCL (librccl.so) could not be loaded");
      delete c;
      return 1;
    }
    rccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t rc = r->CommInitRank(&c->nccl, world, id, rank);
    if (rc != ncclSuccess) {
      set_error(std::string("edigpu_comm_create: ncclCommInitRank failed: ") + r->GetErrorString(rc));
      delete c;
      return 1;
    }
  }
  *out = c;
  return 0;
}

int edigpu_comm_create_shm(edigpu_comm* out, int32_t rank, int32_t world, const char* name, int64_t slot_bytes) {
  if (!out || !name || world < 1 || rank < 0 || rank >= world || slot_bytes < 64) {
    set_error("edigpu_comm_create_shm: bad argument");
    return 1;
  }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    set_error("edigpu_comm_create_shm: no usable HIP device; this library has no CPU fallback");
    return 1;
  }
  auto* c = new edigpu_comm_s();
  c->rank = rank;
  c->world = world;
  c->kind = 1;
  c->shm_name = name[0] == '/' ? name : std::string("/") + name;
  slot_bytes = (slot_bytes + 63) / 64 * 64;
  c->shm_bytes = 4096 + (size_t)world * (size_t)slot_bytes;
  int fd = -1;
  if (rank == 0) {
    shm_unlink(c->shm_name.c_str());
    fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0 && ftruncate(fd, (off_t)c->shm_bytes) != 0) {
      close(fd);
      fd = -1;
    }
  }
  const int64_t started_s = (int64_t)time(nullptr);
  const auto t_open = std::chrono::steady_clock::now();
  // Ranks > 0 may find a segment a crashed run left under the same name (right size, ready = 1) before rank 0 has
  // replaced it.  A segment is taken only if rank 0 made it no more than two minutes before this call and the name
  // still refers to it once it reads ready; otherwise it is dropped and the name is opened again.
  for (;;) {
    if (rank != 0) {
      fd = -1;
      while (fd < 0 && std::chrono::steady_clock::now() - t_open < std::chrono::seconds(60)) {
        fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
        struct stat sb;
        if (fd >= 0 && (fstat(fd, &sb) != 0 || (size_t)sb.st_size < c->shm_bytes)) {
          close(fd);
          fd = -1;
        }
        if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));
      }
    }
    if (fd < 0) {
      set_error("edigpu_comm_create_shm: cannot open the shared-memory segment " + c->shm_name);
      delete c;
      return 1;
    }
    struct stat mine;
    const bool have_ino = fstat(fd, &mine) == 0;
    c->shm = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->shm == MAP_FAILED) {
      set_error("edigpu_comm_create_shm: mmap failed");
      c->shm = nullptr;
      delete c;
      return 1;
    }
    c->hdr = reinterpret_cast<ShmHeader*>(c->shm);
    c->slots = reinterpret_cast<char*>(c->shm) + 4096;
    if (rank == 0) {
      c->hdr->arrived.store(0);
      c->hdr->generation.store(0);
      c->hdr->world = world;
      c->hdr->slot_bytes = slot_bytes;
      c->hdr->created_s = started_s;
      c->hdr->ready.store(1, std::memory_order_release);
      break;
    }
    bool ok = false, timeout = false;
    while (!ok && !timeout) {
      if (c->hdr->ready.load(std::memory_order_acquire) == 1) {
        struct stat now;
        const std::string path = "/dev/shm" + c->shm_name;
        const bool same = !have_ino || (stat(path.c_str(), &now) == 0 && now.st_ino == mine.st_ino);
        ok = same && c->hdr->created_s >= started_s - 120 && c->hdr->world == world && c->hdr->slot_bytes == slot_bytes;
        if (!ok) break;  // a leftover: map the name again
      } else {
        std::this_thread::yield();
      }
      timeout = std::chrono::steady_clock::now() - t_open > std::chrono::seconds(60);
    }
    if (ok) break;
    munmap(c->shm, c->shm_bytes);
    c->shm = nullptr;
    if (timeout || std::chrono::steady_clock::now() - t_open > std::chrono::seconds(60)) {
      set_error("edigpu_comm_create_shm: rank 0 never initialised the segment");
      delete c;
      return 1;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }
  if (comm_common(c) || shm_barrier(c)) {
    munmap(c->shm, c->shm_bytes);
    delete c;
    return 1;
  }
  *out = c;
  return 0;
}

int edigpu_comm_info(edigpu_comm c, int32_t* rank, int32_t* world, int32_t* kind) {
  if (!c) {
    set_error("edigpu_comm_info: NULL communicator");
    return 1;
  }
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (kind) *kind = c->kind;
  return 0;
}

int edigpu_comm_destroy(edigpu_comm c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  for (double** p : {&c->vin, &c->vout, &c->tmp, &c->vfull, &c->send, &c->recv, &c->hvc, &c->back, &c->hist, &c->work, &c->scr,
                     &c->bp[0], &c->bp[1], &c->bp[2], &c->bp[3], &c->bp[4]})
    if (*p) (void)hipFree(*p);
  if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->shm) {
    munmap(c->shm, c->shm_bytes);
    if (c->rank == 0) shm_unlink(c->shm_name.c_str());
  }
  delete c;
  return 0;
}

// (Nloc, v, Hv) on shards with host vectors: the contract of spMatVec_mpi_* / directMatVec_MPI_*
static int apply_sharded(edigpu_handle s, edigpu_comm c, int64_t nloc, const double* v, double* hv, int cplx) {
  if (!s || !c || (nloc > 0 && (!v || !hv))) {
    set_error("edigpu_apply_sharded: NULL argument");
    return 1;
  }
  if ((s->is_complex != 0) != (cplx != 0)) {
    set_error("edigpu_apply_sharded: real/complex mismatch between handle and entry point");
    return 1;
  }
  // _CMPLX_NORMAL held as one real sector on the doubled up index: its down rows shard like any real sector's (each row
  // 2 DimUp doubles = DimUp complex elements), so the transposed exchange serves complex sectors as well
  if (s->kind == 4 && s->sub_d) return apply_sharded(s->sub_d, c, 2 * nloc, v, hv, 0);
  ShardGeom g;
  if (shard_geometry(s, c, g)) return 1;
  if (nloc != g.nloc) {
    set_error("edigpu_apply_sharded: Nloc does not match this rank's shard (edigpu_shard_plan)");
    return 1;
  }
  EDIGPU_HIP(hipSetDevice(s->device));
  if (comm_workspace(c, g, 1)) return 1;
  hipStream_t st = s->stream;
  const int64_t n = g.nloc * g.w, chunk = g.chunk * g.w;
  EDIGPU_HIP(hipMemsetAsync(c->vin, 0, (size_t)std::max<int64_t>(chunk, 1) * sizeof(double), st));
  if (n > 0) EDIGPU_HIP(hipMemcpyAsync(c->vin, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  const double* back = nullptr;
  if (sharded_hv(s, c, g, false, st, &back)) return 1;
  if (g.transposed && back &&
      edigpu_transpose_unpack_add(s->dim_up, g.count, g.q, c->world, g.pcol, g.halo, back, c->tmp, st))
    return 1;
  if (n > 0) EDIGPU_HIP(hipMemcpyAsync(hv, c->tmp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  EDIGPU_HIP(hipStreamSynchronize(st));
  return 0;
}

int edigpu_apply_sharded_d(edigpu_handle h, edigpu_comm c, int64_t nloc, const double* v_shard_host, double* hv_shard_host) {
  return apply_sharded(h, c, nloc, v_shard_host, hv_shard_host, 0);
}

int edigpu_apply_sharded_z(edigpu_handle h, edigpu_comm c, int64_t nloc, const double* v_shard_host, double* hv_shard_host) {
  return apply_sharded(h, c, nloc, v_shard_host, hv_shard_host, 1);
}

int edigpu_lanczos_tridiag_sharded(edigpu_handle h, edigpu_comm c, const double* vin_shard, int nlanc, double* alanc,
                                   double* blanc, double threshold, int* niter_done, double* norm2) {
  if (!h || !c || !alanc || !blanc || nlanc < 1) {
    set_error("edigpu_lanczos_tridiag_sharded: bad argument");
    return 1;
  }
  // a world of one rank holding the whole sector: the fused single-GPU recurrence (half the vector traffic of the
  // exchange form: no send / receive buffers); EDIGPU_FORCE_COLLECTIVES=1 keeps the N > 1 code path (tests)
  if (h->kind == 4 && h->sub_d) h = h->sub_d;  // complex normal sector = its doubled real sector (see apply_sharded)
  if (c->world == 1 && h->nloc == h->dim && !getenv("EDIGPU_FORCE_COLLECTIVES") && vin_shard)
    return edigpu_lanczos_tridiag_dev(h, vin_shard, nlanc, alanc, blanc, threshold, niter_done, norm2);
  return sharded_tridiag(h, c, vin_shard, nlanc, alanc, blanc, threshold, niter_done, norm2);
}

// sp_eigh(MpiComm, spHtimesV_p, eval, evec, ...) / sp_lanc_eigh(MpiComm, ...) with every vector a device-resident shard
// (ED_NORMAL/ED_DIAG_NORMAL.f90:179-214 and the superc / nonsu2 twins): the thick-restart driver of the single-GPU
// solver (trl_solve, edigpu_capi.hip) on this rank's elements, the product through the sharded exchange, the Gram-
// Schmidt coefficients and norms through small all-reduces.  Nothing of vector length crosses PCIe.
int edigpu_lanczos_eigh_multi_sharded(edigpu_handle h, edigpu_comm c, int neigen, int ncv, double tol, int maxrestart,
                                      const double* v0_shard, double* evals, double* evecs_shard, int* nconv, int* nmatvec) {
  if (!h || !c || !evals || neigen <= 0) {
    set_error("edigpu_lanczos_eigh_multi_sharded: bad argument");
    return 1;
  }
  bool realified = false;
  if (h->kind == 4 && h->sub_d) {  // complex normal sector: the products run on its doubled real sector, whose real
    h = h->sub_d;                  // vectors ARE the interleaved complex ones -- the solver keeps the complex algebra
    realified = true;
  }
  // a world of one rank holding the whole sector: the single-GPU solver (no exchange buffers)
  if (c->world == 1 && h->nloc == h->dim && !getenv("EDIGPU_FORCE_COLLECTIVES") && !realified)
    return edigpu_lanczos_eigh_multi(h, neigen, ncv, tol, maxrestart, v0_shard, evals, evecs_shard, nconv, nmatvec);
  ShardGeom g;
  if (shard_geometry(h, c, g)) return 1;
  EDIGPU_HIP(hipSetDevice(h->device));
  if (comm_workspace(c, g, 8)) return 1;
  hipStream_t st = h->stream;
  const int64_t len = g.nloc * g.w, chunk = g.chunk * g.w;
  const int cplx = (g.w == 2 || realified) ? 1 : 0;
  const int64_t n = cplx && g.w == 1 ? g.nloc / 2 : g.nloc;
  const int64_t nglobal = (cplx && g.w == 1 ? h->dim / 2 : h->dim);
  // the padded tail of the exchange buffers stays zero for the whole solve
  EDIGPU_HIP(hipMemsetAsync(c->vin, 0, (size_t)std::max<int64_t>(chunk, 1) * sizeof(double), st));
  EDIGPU_HIP(hipMemsetAsync(c->tmp, 0, (size_t)std::max<int64_t>(chunk, 1) * sizeof(double), st));
  TrlOps ops;
  ops.apply = [&](const double* in, double* out, hipStream_t s2) -> int {
    if (g.block) {  // padded panels: the conversions read and write the solver's vectors themselves (no staging copies)
      const double *rowhalf = nullptr, *colhalf = nullptr;
      if (sb_shard_to_panels(h, in, c->bp[0], g.count, g.q, c->world, s2)) return 1;
      if (sharded_hv_panels(h, c, g, s2, c->bp[0], &rowhalf, &colhalf)) return 1;
      return sb_shard_from_panels_add(h, rowhalf, colhalf, out, g.count, g.q, s2);
    }
    if (len > 0) EDIGPU_HIP(hipMemcpyAsync(c->vin, in, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, s2));
    const double* back = nullptr;
    if (sharded_hv(h, c, g, false, s2, &back)) return 1;
    if (g.transposed && back &&
        edigpu_transpose_unpack_add(h->dim_up, g.count, g.q, c->world, g.pcol, g.halo, back, c->tmp, s2))
      return 1;
    if (len > 0) EDIGPU_HIP(hipMemcpyAsync(out, c->tmp, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, s2));
    return 0;
  };
  ops.allreduce = [&](double* dev, size_t cnt, hipStream_t s2) -> int { return comm_all_reduce(c, dev, cnt, s2); };
  return trl_solve(h->device, st, cplx, n, len, nglobal, ops, neigen, ncv, tol, maxrestart, v0_shard, (uint64_t)c->rank * 7919u,
                   evals, evecs_shard, nconv, nmatvec);
}

// lanc_method = "lanczos" (sp_lanc_eigh with MpiComm: the lowest pair): the same driver asked for one pair, its basis
// sized by nitermax (at most 128 vectors before a restart)
int edigpu_lanczos_eigh_sharded(edigpu_handle h, edigpu_comm c, int nitermax, double tol, const double* v0_shard, double* eval,
                                double* evec_shard, int* nmatvec) {
  if (!eval || nitermax <= 0) {
    set_error("edigpu_lanczos_eigh_sharded: bad argument");
    return 1;
  }
  int nconv = 0;
  const int ncv = std::max(8, std::min(nitermax, 48));
  const int maxrestart = std::max(1, nitermax / std::max(1, ncv / 2));
  return edigpu_lanczos_eigh_multi_sharded(h, c, 1, ncv, tol, maxrestart, v0_shard, eval, evec_shard, &nconv, nmatvec);
}

int edigpu_apply_cops_sharded(edigpu_handle src, edigpu_handle dst, edigpu_comm c, const double* v_src_shard,
                              double* v_dst_shard, int nops, const double* coef_re_im, const int32_t* create,
                              const int32_t* iorb, const int32_t* ispin) {
  if (!src || !dst || !c || nops <= 0 || !coef_re_im || !create || !iorb || !ispin) {
    set_error("edigpu_apply_cops_sharded: bad argument");
    return 1;
  }
  if (src->kind == 4 || dst->kind == 4 || src->nph > 0 || dst->nph > 0) {
    set_error("edigpu_apply_cops_sharded: complex normal-mode and phonon sectors are not supported");
    return 1;
  }
  ShardGeom gs, gd;
  if (shard_geometry(src, c, gs) || shard_geometry(dst, c, gd)) return 1;
  EDIGPU_HIP(hipSetDevice(src->device));
  hipStream_t st = src->stream;
  const size_t chunk = (size_t)std::max<int64_t>(gs.chunk * gs.w, 1), nsrc = (size_t)(gs.nloc * gs.w);
  const size_t ndst = (size_t)(gd.nloc * gd.w);
  if ((nsrc && !v_src_shard) || (ndst && !v_dst_shard)) {
    set_error("edigpu_apply_cops_sharded: NULL vector");
    return 1;
  }
  struct Bufs {
    double *send = nullptr, *full = nullptr, *out = nullptr;
    ~Bufs() { (void)hipFree(send); (void)hipFree(full); (void)hipFree(out); }
  } b;
  EDIGPU_HIP(hipMalloc((void**)&b.send, chunk * sizeof(double)));
  EDIGPU_HIP(hipMalloc((void**)&b.full, chunk * (size_t)c->world * sizeof(double)));
  EDIGPU_HIP(hipMalloc((void**)&b.out, std::max<size_t>(ndst, 1) * sizeof(double)));
  EDIGPU_HIP(hipMemsetAsync(b.send, 0, chunk * sizeof(double), st));
  if (nsrc) EDIGPU_HIP(hipMemcpyAsync(b.send, v_src_shard, nsrc * sizeof(double), hipMemcpyDefault, st));
  // equal padded chunks in rank order = the whole source vector in the reference's layout (+ a tail nobody reads)
  if (comm_all_gather(c, b.send, b.full, chunk, st)) return 1;
  if (ndst && apply_cops_rows(src, dst, b.full, b.out, gd.first, gd.count, nops, coef_re_im, create, iorb, ispin, st,
                              "edigpu_apply_cops_sharded"))
    return 1;
  if (ndst) EDIGPU_HIP(hipMemcpyAsync(v_dst_shard, b.out, ndst * sizeof(double), hipMemcpyDefault, st));
  EDIGPU_HIP(hipStreamSynchronize(st));
  return 0;
}

// bench.py --gpus N: what the transport itself reports and costs.  *rccl_ranks = ncclCommCount of the communicator (0:
// shared-memory transport, -1: RCCL without that symbol) -- the driver's scaling record can then show that RCCL saw N
// ranks; *ms_exchange = average time of the collectives of ONE product + step (the two all-to-alls or the all-gather, and
// the 3-double all-reduce) with nothing else running, HIP events on the communicator's stream.
int edigpu_exchange_bench(edigpu_handle s, edigpu_comm c, int steps, int32_t* rccl_ranks, double* ms_exchange) {
  if (!s || !c || steps < 1) {
    set_error("edigpu_exchange_bench: bad argument");
    return 1;
  }
  if (s->kind == 4 && s->sub_d) s = s->sub_d;
  if (rccl_ranks) {
    *rccl_ranks = 0;
    if (c->kind == 0) {
      int n = -1;
      if (c->nccl && rccl() && rccl()->CommCount && rccl()->CommCount(c->nccl, &n) != ncclSuccess) n = -1;
      *rccl_ranks = c->nccl ? n : 1;  // (a world of one makes no RCCL communicator)
    }
  }
  if (!ms_exchange) return 0;
  ShardGeom g;
  if (shard_geometry(s, c, g)) return 1;
  EDIGPU_HIP(hipSetDevice(s->device));
  if (comm_workspace(c, g, 8)) return 1;
  hipEvent_t e0, e1;
  EDIGPU_HIP(hipEventCreate(&e0));
  EDIGPU_HIP(hipEventCreate(&e1));
  int rc = 0;
  auto once = [&]() -> int {
    for (int b = 0; b < g.nblk; b++) {
      if (g.transposed) {
        if (comm_all_to_all(c, c->send, c->recv, (size_t)(g.q * g.pw), c->side)) return 1;
        if (comm_all_to_all(c, c->hvc, c->back, (size_t)(g.q * g.pw), c->side)) return 1;
      } else {
        const size_t n = g.nblk > 1 ? (size_t)g.q * g.unit_len * g.w : (size_t)g.chunk * g.w;
        if (comm_all_gather(c, c->vin, c->vfull, n, c->side)) return 1;
      }
    }
    return comm_all_reduce(c, c->scr, 3, c->side);
  };
  for (int k = 0; k < 2 && !rc; k++) rc |= once();
  rc |= (hipEventRecord(e0, c->side) != hipSuccess);
  for (int k = 0; k < steps && !rc; k++) rc |= once();
  rc |= (hipEventRecord(e1, c->side) != hipSuccess);
  rc |= (hipEventSynchronize(e1) != hipSuccess);
  float ms = 0.f;
  if (!rc) rc |= (hipEventElapsedTime(&ms, e0, e1) != hipSuccess);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) {
    set_error("edigpu_exchange_bench: failed");
    return 1;
  }
  *ms_exchange = (double)ms / steps;
  return 0;
}

int edigpu_lanczos_bench_sharded(edigpu_handle s, edigpu_comm c, int warmup, int steps, double* ms_per_step,
                                 int64_t* exchange_bytes) {
  if (!s || !c || steps < 1 || warmup < 0 || !ms_per_step) {
    set_error("edigpu_lanczos_bench_sharded: bad argument");
    return 1;
  }
  if (s->kind == 4 && s->sub_d) s = s->sub_d;
  ShardGeom g;
  if (shard_geometry(s, c, g)) return 1;
  EDIGPU_HIP(hipSetDevice(s->device));
  const int total = warmup + steps;
  if (comm_workspace(c, g, total)) return 1;
  hipStream_t st = s->stream;
  EDIGPU_HIP(hipMemsetAsync(c->hist, 0, (size_t)c->hist_cap * sizeof(double), st));
  // seeded random start vector of this shard
  const int64_t n = g.nloc * g.w, chunk = g.chunk * g.w;
  EDIGPU_HIP(hipMemsetAsync(c->tmp, 0, (size_t)std::max<int64_t>(chunk, 1) * sizeof(double), st));
  if (n > 0 && lz_fill_random(c->tmp, n, 12345ull + (uint64_t)c->rank, st)) return 1;
  if (load_seed(c, g, c->tmp, st)) return 1;
  const bool panels = panel_loop(g);
  if (panels && panel_loop_begin(s, c, g, st)) return 1;
  auto step = [&](int it) -> int { return panels ? sharded_step_panels(s, c, g, it, st) : sharded_step(s, c, g, it, st); };
  for (int it = 0; it < warmup; it++)
    if (step(it)) return 1;
  if (comm_all_reduce(c, c->scr + 2, 1, st)) return 1;  // barrier
  EDIGPU_HIP(hipStreamSynchronize(st));
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = warmup; it < total; it++)
    if (step(it)) return 1;
  EDIGPU_HIP(hipStreamSynchronize(st));
  if (comm_all_reduce(c, c->scr + 2, 1, st)) return 1;
  EDIGPU_HIP(hipStreamSynchronize(st));
  const auto t1 = std::chrono::steady_clock::now();
  *ms_per_step = std::chrono::duration<double, std::milli>(t1 - t0).count() / steps;
  if (exchange_bytes)
    *exchange_bytes = g.block        ? 2 * 8 * (int64_t)(c->world - 1) * g.npmax * g.q * 16
                      : g.transposed ? 2 * 8 * (int64_t)(c->world - 1) * g.q * g.pw * g.nblk
                                     : 8 * g.chunk * g.w * (int64_t)(c->world - 1);
  return 0;
}

int edigpu_shard_info(edigpu_handle s, edigpu_comm c, int32_t info[4]) {
  if (!s || !c || !info) {
    set_error("edigpu_shard_info: bad argument");
    return 1;
  }
  if (s->kind == 4 && s->sub_d) s = s->sub_d;
  ShardGeom g;
  if (shard_geometry(s, c, g)) return 1;
  info[0] = g.block ? 2 : g.transposed ? 1 : 0;
  info[1] = (int32_t)g.q;
  info[2] = g.block ? g.npmax : (int32_t)g.pcol;
  info[3] = g.block ? 0 : g.halo;
  return 0;
}

}  // extern "C"
