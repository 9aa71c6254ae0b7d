// kernels_sb_impl.hpp -- normal-mode H*v on LOCAL blocks (gfx950, round 4).  Included by kernels_sb<N>.hip, one
// translation unit per number of impurity levels (the instantiations compile in parallel).  host_sb.hpp explains the
// decomposition, sb_core.hpp holds the per-block arithmetic (shared with the CPU shim tests/host_sb.cpp).
//
//   Hv = Hd o v + (1 (x) Hup) v + (Hdw (x) 1) v + Hnd v      (spMatVec_normal_main,
//                                                             ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650)
// on vectors in the padded panel layout of the impurity-block image (IbDev).
//
//  sb_rows_kernel  Hd + 1 (x) Hup.  Persistent workgroups, one staged row of V at a time in the LDS, a lane owns whole
//                  blocks of up to C(5,2) = 10 columns.  The hops among a block's local levels are unrolled register
//                  arithmetic; per WALKED bath level one table look-up gives the partner block, whose words are read
//                  from the LDS and applied through the compile-time pattern of the class.  Two waves per SIMD with up
//                  to 256 registers each instead of four with 128: the next row is in flight (in registers) during the
//                  block updates, and a walk step carries ~30 independent multiply-adds, so the work of ONE wave covers
//                  the LDS round trips that four waves of the round-3 kernel waited for.
//  sb_cols_kernel  Hdw (x) 1 + Hnd.  A task = (panel, chunk of rows that share their high walked levels) staged in the
//                  LDS; a group of 8 lanes owns a block of up to 10 rows x 16 columns.  The rows kernel's part of the
//                  result and the partner rows of two high levels are requested before the LDS work starts.
// No MFMA: fp64 sparse.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "host_sb.hpp"
#include "kernels.hpp"
#include "sb_core.hpp"

namespace edigpu {

struct SbArgs {
  int npanels, plen, nterms;
  int64_t dim_dw, ps;
  // rows kernel
  int nbw_up, rimg_len;
  const uint16_t *urank, *ublist;
  const double* ebw;
  const int32_t* uslot;
  const uint32_t* rmap2;
  const double *up_vtab, *up_tloc, *e0, *xu, *ed;
  const uint32_t* up_korb;
  const uint8_t* impd;
  // rows staged in halves (host_sb.hpp SbUpHalf; TOP != 0): nbw_up, rimg_len, plen, uslot, rmap2, ebw are the half's, P / Q
  // point at its first panel; ublist32 = low word | skip flag | position of the partner block over the top level << 16,
  // ugap = where that block's columns skip positions, pfull = the whole vector the partner words are read from.  up_vtab
  // keeps all nbw_up + 1 rows.
  const uint32_t* ublist32;
  const uint8_t* ugap;
  const double* pfull;
  // columns kernel
  int nbw_dw, lowbits, nchunks, max_chunk_rows, max_chunk_slots;
  const int32_t *chunk_row, *chunk_slot, *cdesc_off;
  const uint8_t* cdesc;
  const double *dw_vtab, *dw_tloc, *ndcoef;
  const uint32_t* dw_korb;
  const uint32_t* nd_dw;
  const uint8_t* nd_up;
  // row shards (edigpu_shard.hip): the rows kernel works on the rank's rows [row0, row0 + dim_dw) of the sector; the columns
  // kernel (SH) on its panels [p0, p0 + npanels) with every rank's q rows of a panel in that rank's slot of the buffer:
  // row g of local panel pl at pl * q * 16 + g * 16 + (g / q) * kslot doubles (qmagic: g / q = (g * qmagic) >> 32)
  int64_t row0;
  int p0;
  uint32_t qmagic;
  int64_t kslot, q16;
  // fused Lanczos step
  const double* scal;
  double* partial;
  int lazy;
  const double* pold;  // columns kernel (ALPHA): the previous Lanczos vector, whose - beta multiple joins the rows kernel's part; or null
  long long* dbg;  // EDIGPU_SB_STAMP: per-wave cycle sums of the rows kernel's phases (workgroup 0), else null
};

// read-only tables through the constant address space: uniform loads become scalar loads (s_load) even in a kernel that
// stores to global memory (a plain global pointer is then read with vector loads: measured 6 M extra vector loads per
// product at Ns = 16, each a dependent L2 round trip in front of a block update)
template <class T>
__device__ inline const T __attribute__((address_space(4)))* sb_const(const T* p) {
  return (const T __attribute__((address_space(4)))*)(p);
}

#ifndef SB_V_LATE
#define SB_V_LATE 0  // pieces of the next row requested only after the block updates (768-thread geometry; tuning)
#endif

// 16-byte pieces of a row a thread moves, given its NBT blocks of at most MAXM columns (the set-up checks the row fits)
constexpr int sb_rows_nld(int nbt, int maxm) { return (3 * nbt * maxm + 9) / 10; }
constexpr int kSbVs = 10;  // doubles per level of the LDS copy of the amplitude table (sb_core.hpp rows_block)

// ---------------------------------------------------------------------------------------------------------
// rows kernel
// ---------------------------------------------------------------------------------------------------------
// FUSE 0: Q = (Hd + 1 (x) Hup) P                       (plain product; also the first Lanczos step)
// FUSE 1: x = (Q - alpha P) / beta; X <- x; Q <- (Hd + 1 (x) Hup) x - beta P   (alpha = 0 unless a.lazy).  The new
//         Lanczos vector goes to a THIRD buffer.  P's pieces are read a second time when the result leaves (the term
//         - beta P_old of the recurrence; the round-3 kernels left it to the columns kernel, which has no registers to
//         spare for it here).
// TOP (rows staged in halves, FUSE 0 only): 0 = the whole row is staged; 1 / 2 = the half with the top walked level empty /
//         occupied: the walk runs over the lower levels, the hop over the top one reads the partner block from a.pfull.
template <int NIMP, int NB0, int AMODE, int NT, int NBT, int CS, int FUSE, int TOP = 0>
__global__ void __launch_bounds__(NT, (NT <= 512 ? 2 : NT / 256)) sb_rows_kernel(SbArgs a, const double* __restrict__ P, double* __restrict__ Q, double* __restrict__ X) {
  static_assert(TOP == 0 || FUSE == 0, "rows staged in halves: plain product only");
  extern __shared__ double lds[];
  constexpr int NLOC = NIMP + NB0;
  constexpr int MAXM = sb::binom(NLOC, NLOC / 2);
  constexpr int NW = NT / 64;
  constexpr int NLD = sb_rows_nld(NBT, MAXM);
  const int nbw = a.nbw_up, plen2 = a.plen >> 1;
  double* row = lds;
  double* vtab = row + a.rimg_len;                                  // [nbw (+ 1: TOP)][kSbVs]
  double* ebath = vtab + (nbw + (TOP ? 1 : 0)) * kSbVs;             // [2^nbw]
  uint16_t* rank = reinterpret_cast<uint16_t*>(ebath + (1 << nbw)); // [2^nbw]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t dd = a.dim_dw, ps = a.ps;
  double ibeta = 1.0, alpha = 0.0, beta = 0.0;
  if (TOP != 0 && a.scal && sb_const(a.scal)[SC_STOP] != 0.0) return;  // (inside a recurrence that has terminated: uniform)
  if (FUSE) {
    const auto sc = sb_const(a.scal);  // (scalar loads: the three numbers stay in scalar registers)
    if (sc[SC_STOP] != 0.0) return;  // recurrence already terminated (uniform)
    beta = sc[SC_BETA];
    ibeta = 1.0 / beta;
    alpha = a.lazy ? sc[SC_ALPHA] : 0.0;
  }
  for (int i = tid; i < (1 << nbw); i += NT) {
    rank[i] = a.urank[i];
    ebath[i] = a.ebw[i];
  }
  for (int i = tid; i < (nbw + (TOP ? 1 : 0)) * 4; i += NT) vtab[(i >> 2) * kSbVs + (i & 3)] = a.up_vtab[i];
  for (int i = tid; i < a.rimg_len; i += NT) row[i] = 0.0;  // slack and the zero word stay zero
  sb::RowImage im;
  im.row = row;
  im.rank = rank;
  im.ebath = ebath;
  im.cs = CS;
  // this thread's blocks: the same for every row.  sd: class | first index inside the class << 8 (uniform; < 0: none)
  uint32_t bw[NBT];
  int sd[NBT];
  sb::sfor<0, NBT>([&](auto S) {
    constexpr int s = decltype(S)::value;
    sd[s] = __builtin_amdgcn_readfirstlane(a.uslot[s * NW + wave]);
    if constexpr (TOP != 0) bw[s] = a.ublist32[(size_t)(s * NW + wave) * 64 + (tid & 63)];
    else bw[s] = a.ublist[(size_t)(s * NW + wave) * 64 + (tid & 63)];
  });
  uint32_t ug = 0;  // TOP: the gap bytes of this thread's (at most four) partner blocks
  if constexpr (TOP != 0) {
    static_assert(NBT <= 4 || TOP == 0, "gap bytes: four blocks per thread");
    sb::sfor<0, (NBT < 4 ? NBT : 4)>([&](auto S) {
      constexpr int s = decltype(S)::value;
      ug |= (uint32_t)a.ugap[(size_t)(s * NW + wave) * 64 + (tid & 63)] << (8 * s);
    });
  }
  // A thread moves the 16-byte pieces tid, tid + NT, ... of a row: piece q2 is the column pair (q2 & 7) of panel q2 >> 3.
  const int64_t pstride0 = (int64_t)(NT / 8) * ps;
  int64_t pstride = pstride0;  // (+ an opaque zero inside the row loop: the per-piece addresses are not hoisted)
  int zt = 0;  // (an opaque zero, renewed per row: the 64-bit per-thread base below is recomputed, not kept in two registers)
  auto base_of = [&](int64_t r) -> int64_t { return (int64_t)((tid + zt) >> 3) * ps + r * 16 + ((tid & 7) << 1); };
  double2 pre[NLD];  // the next row on its way in (FUSE: Q)
  double2 pin[FUSE ? NLD : 1];
  int64_t gkeep = 0;  // (see the row loop: the prefetch's address register must stay untouched while the loads fly)
  // the pieces of row r: what (FUSE: Q) into pre; with FUSE P into pin
  auto issue = [&](int64_t r) {
    const int64_t g0 = base_of(r);
    gkeep = g0;
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      if (tid + i * NT < plen2) {
        if (FUSE) {
          pre[i] = *reinterpret_cast<const double2*>(Q + g0 + i * pstride);
          pin[i] = *reinterpret_cast<const double2*>(P + g0 + i * pstride);
        } else {
          pre[i] = *reinterpret_cast<const double2*>(P + g0 + i * pstride);
        }
      }
    }
  };
  // loaded pieces -> image (FUSE: x = (Q - alpha P) / beta, X <- x)
  auto land = [&](int64_t r, const uint32_t* mp) {
    const int64_t g0 = base_of(r);
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      if (tid + i * NT < plen2) {
        double2 x = pre[i];
        if (FUSE) {
          x.x = (x.x - alpha * pin[i].x) * ibeta;
          x.y = (x.y - alpha * pin[i].y) * ibeta;
          *reinterpret_cast<double2*>(X + g0 + i * pstride) = x;
        }
        row[mp[i] & 0xFFFFu] = x.x;
        row[mp[i] >> 16] = x.y;
      }
    }
  };
  auto load_map = [&](uint32_t* mp) {
#pragma unroll
    for (int i = 0; i < NLD; i++) mp[i] = tid + i * NT < plen2 ? a.rmap2[tid + i * NT] : 0u;
  };
  int64_t r = blockIdx.x;
  if (r >= dd) return;
  __syncthreads();  // the zeroed image before the first row lands
  {
    uint32_t mp[NLD];
    load_map(mp);
    issue(r);
    land(r, mp);
  }
  __syncthreads();
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  const bool stamp = a.dbg != nullptr && blockIdx.x == 0;  // EDIGPU_SB_STAMP=1: cycles per phase
  auto now = [&]() -> long long { return stamp ? (long long)__builtin_amdgcn_s_memtime() : 0; };
  for (; r < dd; r += gridDim.x) {
    const int64_t rn = r + gridDim.x;
    const bool more = rn < dd;
    const long long t0 = now();
    asm volatile("v_mov_b32 %0, 0" : "=v"(zt));
    constexpr bool EARLY = !FUSE && NT <= 768;  // registers for the next row beside the accumulators
    if (EARLY && more) issue(rn);  // in flight during the block updates
    // an opaque zero, new in every iteration: added to the block indices below so that the per-(slot, class) addresses are
    // recomputed where they are used (one add) instead of being hoisted out of the row loop into registers
    int zr, zs;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zr));
    asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
    pstride = pstride0 + zs;
    const int64_t rg = r + a.row0;  // the row's number in the sector (a.row0 = 0 unless the vector is a row shard)
    const double edr = sb_const(a.ed)[rg];
    const int ic = (int)((sb_const(reinterpret_cast<const uint32_t*>(a.impd))[rg >> 2] >> (8 * (int)(rg & 3))) & 0xFFu);
    const auto xuc = sb_const(a.xu) + ic * (1 << NIMP);
    double acc[NBT][MAXM];
    sb::sfor<0, NBT>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if (sd[s] >= 0) {  // uniform
        sb::for_class<NLOC>(sd[s] & 0xFF, [&](auto N) {
          constexpr int nn = decltype(N)::value;
          sb::rows_block<NIMP, NB0, AMODE, nn, CS>(im, (bw[s] + (uint32_t)zr) & 0x7FFFu, (uint32_t)((sd[s] >> 8) + lane + zr), nbw, vtab, kSbVs, sb_const(a.up_korb), sb_const(a.up_tloc), edr,
                                               xuc, sb_const(a.e0), acc[s]);
          if constexpr (TOP != 0) {
            // The partner block over the top level, word by word from the vector, requested after the walk.  Measured at
            // Ns = 17 (3.30 + 3.15 ms for the two halves; the round-3 kernel 2.77 + 2.58): the 8-byte gathers of 64 lanes land
            // in ~40 different lines per instruction and wait, on the in-order memory counter, behind the next row's pieces.
            // Requested BEFORE those pieces -- all blocks at once, or two at a time -- the kernel spills and takes 4.7-5.1 ms.
            constexpr int MPT = sb::rows_top_words<NLOC, nn, TOP == 2>();
            if constexpr (MPT > 0) {
              const uint32_t tp = (bw[s] >> 16) + (uint32_t)zr;
              const uint32_t gi = (ug >> (8 * (s & 3))) & 0xFFu, jg = gi & 15u, gg = gi >> 4;
              double xg[MPT];
              sb::sfor<0, MPT>([&](auto J) {
                constexpr uint32_t j = (uint32_t)decltype(J)::value;
                const uint32_t pj = tp + j + (j >= jg ? gg : 0u);
                xg[j] = tp != 0xFFFFu ? a.pfull[(int64_t)(pj >> 4) * ps + r * 16 + (pj & 15u)] : 0.0;
              });
              sb::rows_top<NIMP, NB0, nn, TOP == 2>(bw[s] & 0x7FFFu, vtab + nbw * kSbVs, xg, acc[s]);
            }
          }
        });
      }
    });
    // The compiler waits for a load to COMPLETE before it lets the load's address register be overwritten; reused as a
    // temporary of the block updates, the base address of the prefetch cost a vmcnt(0) at the head of every block -- the
    // next row never overlapped the updates.  Kept alive until here, it is not reused.
    asm volatile("" ::"v"(gkeep));
    const long long t1 = now();
    __syncthreads();  // every read of the row is done: the results take its place
    const long long t2 = now();
    sb::sfor<0, NBT>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if (sd[s] >= 0 && !(bw[s] & 0x8000u))
        sb::for_class<NLOC>(sd[s] & 0xFF, [&](auto N) {
          constexpr int nn = decltype(N)::value;
          double* own = row + sb::wbase(NLOC, nn) * CS + ((sd[s] >> 8) + lane + zr);
          sb::sfor<0, sb::binom(NLOC, nn)>([&](auto J) { own[decltype(J)::value * CS] = acc[s][decltype(J)::value]; });
        });
    });
    uint32_t mp[NLD];
    load_map(mp);
    double2 pold[FUSE ? NLD : 1];
    if (!EARLY && more) issue(rn);
    if (FUSE) {
      const int64_t g0 = base_of(r);
#pragma unroll
      for (int i = 0; i < NLD; i++)
        if (tid + i * NT < plen2) pold[i] = *reinterpret_cast<const double2*>(P + g0 + i * pstride);
    }
    const long long t3 = now();
    __syncthreads();
    const long long t4 = now();
    // the result leaves coalesced, the next row takes its place
    {
      const int64_t g0 = base_of(r);
#pragma unroll
      for (int i = 0; i < NLD; i++) {
        if (tid + i * NT < plen2) {
          double2 o;
          o.x = row[mp[i] & 0xFFFFu];
          o.y = row[mp[i] >> 16];
          if (FUSE) {
            o.x -= beta * pold[i].x;
            o.y -= beta * pold[i].y;
          }
          *reinterpret_cast<double2*>(Q + g0 + i * pstride) = o;
        }
      }
    }
    if (more) land(rn, mp);
    const long long t5 = now();
    __syncthreads();
    if (stamp) {
      const long long t6 = now();
      tacc[0] += t1 - t0;
      tacc[1] += t2 - t1;
      tacc[2] += t3 - t2;
      tacc[3] += t4 - t3;
      tacc[4] += t5 - t4;
      tacc[5] += t6 - t5;
    }
  }
  if (stamp && lane == 0)
    for (int i = 0; i < 6; i++) a.dbg[wave * 8 + i] = tacc[i];
}

// ---------------------------------------------------------------------------------------------------------
// columns kernel
// ---------------------------------------------------------------------------------------------------------
typedef double sb_d2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ inline T sb_nt_load(const double* p);
template <>
__device__ inline sb::Pair sb_nt_load<sb::Pair>(const double* p) {
  const sb_d2 t = __builtin_nontemporal_load(reinterpret_cast<const sb_d2*>(p));
  return sb::Pair{t.x, t.y};
}
template <>
__device__ inline double sb_nt_load<double>(const double* p) {
  return __builtin_nontemporal_load(p);
}
template <class T>
__device__ inline void sb_nt_store(double* p, const T& v);
template <>
__device__ inline void sb_nt_store<sb::Pair>(double* p, const sb::Pair& v) {
  sb_d2 t;
  t.x = v.x;
  t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<sb_d2*>(p));
}
template <>
__device__ inline void sb_nt_store<double>(double* p, const double& v) {
  __builtin_nontemporal_store(v, p);
}

// threads of the columns kernel with two columns per lane (tuning: 128 with chunks of <= 256 rows puts four workgroups on a CU)
#ifndef SB_COLS_NT2
#define SB_COLS_NT2 256
#endif

struct SbColsLds {  // byte offsets inside the dynamic LDS
  uint32_t chunk, desc, vtab, ndu, total;
};
// bytes of a chunk's packed descriptors (host_sb.hpp cdesc) for nsl wave-slots of gs blocks
__host__ __device__ inline uint32_t sb_desc_len(int nsl, int gs) {
  return (uint32_t)nsl * gs * 32 + (((uint32_t)nsl * gs * 2 + 15) & ~15u) + (((uint32_t)nsl * 4 + 15) & ~15u);
}
__host__ __device__ inline SbColsLds sb_cols_layout(int nbw, int nloc, int mcr, int mcs, int gs) {
  SbColsLds l;
  uint32_t at = 0;
  l.chunk = at;
  at += ((uint32_t)mcr * 128 + 1023) & ~1023u;  // whole 1 KiB pieces of the LDS-DMA staging
  l.desc = at;
  at += (sb_desc_len(mcs, gs) + 1023) & ~1023u;
  l.vtab = at;
  at += (uint32_t)nbw * kSbVs * 8;
  at = (at + 15) & ~15u;
  l.ndu = at;
  at += 16 * 16;
  l.total = at;
  return l;
}

// ALPHA: the fused Lanczos step -- the three sums <v|w>, sum (w - sigma v)^2, <v|v> are accumulated from the staged rows
// CW: columns per lane.  2: a group of 8 lanes x 16 bytes per block, 8 blocks per wave-slot, 256 threads (two waves per
// SIMD with up to 256 registers); 1: 16 lanes x 8 bytes, 4 blocks per wave-slot, 512 threads (four waves with 128)
// SH (row shards, edigpu_shard.hip): v = what the all-to-all delivered -- for each of this rank's panels every rank's rows in
// that rank's slot (SbArgs::kslot) -- and hv receives (Hdw (x) 1 + Hnd) v in the same form; nothing is read from hv (the
// rows kernel's part is added after the exchange back).
template <int NIMP, int NB0, int AMODE, int CW, bool DO_ND, bool ALPHA, bool SH = false>
__global__ void __launch_bounds__((CW == 2 ? SB_COLS_NT2 : 512), (CW == 2 ? 2 : 4)) sb_cols_kernel(SbArgs a, const double* __restrict__ v, double* __restrict__ hv) {
  static_assert(!(SH && ALPHA), "row shards: plain product only");
  constexpr int NT = CW == 2 ? SB_COLS_NT2 : 512;
  constexpr int GS = 4 * CW;                 // blocks per wave-slot
  constexpr int LPB = 64 / GS;               // lanes per block
  using T = std::conditional_t<CW == 2, sb::Pair, double>;
  __shared__ double red[3 * (NT / 64)];
  extern __shared__ double lds[];
  constexpr int NLOC = NIMP + NB0;
  constexpr int NW = NT / 64;
  char* base = reinterpret_cast<char*>(lds);
  const int nbw = a.nbw_dw;
  const SbColsLds L = sb_cols_layout(nbw, NLOC, a.max_chunk_rows, a.max_chunk_slots, GS);
  double* chunk = reinterpret_cast<double*>(base + L.chunk);
  double* vtab = reinterpret_cast<double*>(base + L.vtab);
  uint8_t* ndu = reinterpret_cast<uint8_t*>(base + L.ndu);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (ALPHA && sb_const(a.scal)[SC_STOP] != 0.0) {
    if (tid == 0) {
      a.partial[blockIdx.x] = 0.0;
      a.partial[gridDim.x + blockIdx.x] = 0.0;
      a.partial[2 * gridDim.x + blockIdx.x] = 0.0;
    }
    return;
  }
  const double sg = ALPHA ? sb_const(a.scal)[SC_ALPHA] : 0.0;  // <Q|Q> is accumulated about the previous alpha (k_finalize_ab)
  const double nbeta = (ALPHA && a.pold) ? -sb_const(a.scal)[SC_BETA] : 0.0;
  double asum = 0.0, qsum = 0.0, nsum = 0.0;
  for (int i = tid; i < nbw * 4; i += NT) vtab[(i >> 2) * kSbVs + (i & 3)] = a.dw_vtab[i];
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int nch = a.nchunks;
  const int panels_x = (a.npanels - x + 7) >> 3;
  const int col = (lane & (LPB - 1)) * CW;
  int cur_panel = -1;
  // A panel's tasks are consecutive: the workgroups with equal blockIdx % 8 (one XCD under the observed round-robin
  // dispatch; speed only) sweep the chunks of one or two panels at a time, whose V segments the chunks gather their
  // partners over the high levels from.
  for (int tt = slot; tt < panels_x * nch; tt += nslots) {
    const int pi = tt / nch, panel = pi * 8 + x + (SH ? a.p0 : 0);
    const int c = (tt - pi * nch + pi) % nch;  // rotated: a slot meets chunks of every size
    const int row0 = a.chunk_row[c], nrows = a.chunk_row[c + 1] - row0;
    const int slot0 = a.chunk_slot[c], nsl = a.chunk_slot[c + 1] - slot0;
    // doubles from the panel's base to row g (SH: + the slot of the rank that owns the row)
    auto roff = [&](int g) -> int64_t {
      if constexpr (SH) return (int64_t)g * 16 + (int64_t)__umulhi((uint32_t)g, a.qmagic) * a.kslot;
      else return (int64_t)g * 16;
    };
    const double* __restrict__ vp = v + (SH ? (int64_t)(panel - a.p0) * a.q16 : (int64_t)panel * a.ps);
    double* __restrict__ hp = hv + (SH ? (int64_t)(panel - a.p0) * a.q16 : (int64_t)panel * a.ps);
    const double* __restrict__ pp = (ALPHA && a.pold) ? a.pold + (int64_t)panel * a.ps : nullptr;
    // Staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land in 1 KiB of consecutive LDS bytes; no
    // registers, every piece of the task in flight at once): the chunk's rows (one contiguous run of the panel) and the
    // chunk's packed descriptors.  Staged through registers four pieces at a time, the 60 KB of a chunk took four
    // dependent HBM round trips per task (measured: the kernel ran 1.4 ms with every other global access compiled out).
    {
      typedef __attribute__((address_space(3))) void lds_void;
      typedef __attribute__((address_space(1))) const void glb_void;
      const char* src = reinterpret_cast<const char*>(vp + (int64_t)row0 * 16);
      const int n16 = nrows * 8;  // 16-byte units
      for (int u0 = wave * 64; u0 < n16; u0 += NW * 64) {
        const int u = u0 + lane < n16 ? u0 + lane : n16 - 1;  // the tail lanes re-read the last unit (their bytes are never used)
        const char* from = SH ? reinterpret_cast<const char*>(vp + roff(row0 + (u >> 3))) + (size_t)(u & 7) * 16 : src + (size_t)u * 16;
        __builtin_amdgcn_global_load_lds((glb_void*)from, (lds_void*)(base + L.chunk + (size_t)u0 * 16), 16, 0, 0);
      }
      // one word of every result row of the chunk: the line comes into the L2 beside the chunk, so that the block updates
      // below find the rows kernel's part of the result there instead of paying an HBM round trip per wave-slot
#if SB_V_TOUCH
      float touch = 0.f;
      for (int i = tid; i < nrows; i += NT) touch += reinterpret_cast<const float*>(hp + (int64_t)(row0 + i) * 16)[0];
      asm volatile("" ::"v"(touch));
#endif
      const char* dsrc = reinterpret_cast<const char*>(a.cdesc) + a.cdesc_off[c];
      const int d16 = (int)(sb_desc_len(nsl, GS) >> 4);
      for (int u0 = wave * 64; u0 < d16; u0 += NW * 64) {
        const int u = u0 + lane < d16 ? u0 + lane : d16 - 1;
        __builtin_amdgcn_global_load_lds((glb_void*)(dsrc + (size_t)u * 16), (lds_void*)(base + L.desc + (size_t)u0 * 16), 16, 0, 0);
      }
      if (DO_ND && panel != cur_panel) {
        for (int i = tid; i < a.nterms * 16; i += NT) ndu[i] = a.nd_up[(size_t)(i >> 4) * a.plen + panel * 16 + (i & 15)];
        cur_panel = panel;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA counts on vmcnt; a workgroup barrier alone does not wait for it
    }
    const uint16_t* lmeta = reinterpret_cast<const uint16_t*>(base + L.desc);
    const uint16_t* lbl = reinterpret_cast<const uint16_t*>(base + L.desc + (size_t)nsl * GS * 32);
    const int32_t* lslot = reinterpret_cast<const int32_t*>(base + L.desc + (size_t)nsl * GS * 32 + ((((size_t)nsl * GS * 2) + 15) & ~(size_t)15));
    __syncthreads();
    for (int q = wave; q < nsl; q += NW) {  // uniform per wave
      const int n = __builtin_amdgcn_readfirstlane(lslot[q]);
      if (n < 0) continue;
      const int bi = q * GS + lane / LPB;
      const uint32_t e = lbl[bi];
      const uint32_t w = e & 0x7FFFu;
      const uint32_t whigh = (uint32_t)__builtin_amdgcn_readfirstlane((int)w) >> a.lowbits;  // the same for the slot's blocks
      const uint16_t* meta = lmeta + (size_t)bi * 16;
      const int own = meta[14];
      sb::for_class<NLOC>(n, [&](auto N) {
        constexpr int nn = decltype(N)::value;
        constexpr int M = sb::binom(NLOC, nn);
        // the rows kernel's part of the result (its lines were pulled into the L2 when the chunk was staged): requested
        // first, added last
        T acc[M], h0[M];
        sb::sfor<0, M>([&](auto J) { acc[decltype(J)::value] = T{}; });
        auto get_h0 = [&]() {
          if constexpr (SH) {
            sb::sfor<0, M>([&](auto J) { h0[decltype(J)::value] = T{}; });
          } else {
            const double* hrow = hp + (int64_t)own * 16 + col;
            sb::sfor<0, M>([&](auto J) { h0[decltype(J)::value] = sb_nt_load<T>(hrow + decltype(J)::value * 16); });
          }
        };
#if SB_V_H0 == 1
        get_h0();
        // (ALPHA with P_old: its rows are requested beside the rows kernel's part and joined to it after the LDS work, before
        // the hops over the high levels take the registers)
        T pq[ALPHA ? M : 1];
        if constexpr (ALPHA)
          if (pp) {  // uniform
            const double* prow = pp + (int64_t)own * 16 + col;
            sb::sfor<0, M>([&](auto J) { pq[decltype(J)::value] = sb_nt_load<T>(prow + decltype(J)::value * 16); });
          }
        auto mid = [&]() {
          if constexpr (ALPHA)
            if (pp)
              sb::sfor<0, M>([&](auto J) {
                constexpr int j = decltype(J)::value;
                double* hf = reinterpret_cast<double*>(&h0[j]);
                const double* pf = reinterpret_cast<const double*>(&pq[j]);
                sb::sfor<0, CW>([&](auto C) { hf[decltype(C)::value] = __builtin_fma(nbeta, pf[decltype(C)::value], hf[decltype(C)::value]); });
              });
        };
#else
        auto mid = [&]() { get_h0(); };
#endif
        auto gload = [&](int grow) -> const double* { return vp + roff(grow) + col; };
        sb::cols_block<NIMP, NB0, AMODE, nn, T, !SH>(chunk, row0, w, whigh, own, meta, nbw, a.lowbits, vtab, kSbVs, sb_const(a.dw_korb), sb_const(a.dw_tloc), col, gload, acc, mid);
        if (DO_ND) sb::cols_block_nd<NIMP, NB0, nn, T>(chunk, own - row0, col, a.nterms, sb_const(a.ndcoef), sb_const(a.nd_dw), ndu, 16, acc);
        if (!(e & 0x8000u)) {
          double* orow = hp + (int64_t)own * 16 + col;
          sb::sfor<0, M>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const double* af = reinterpret_cast<const double*>(&acc[j]);
            const double* hf = reinterpret_cast<const double*>(&h0[j]);
            T res, ov;
            if constexpr (ALPHA) ov = *reinterpret_cast<const T*>(chunk + (own - row0 + j) * 16 + col);
            double* rf = reinterpret_cast<double*>(&res);
            const double* of = reinterpret_cast<const double*>(&ov);
            sb::sfor<0, CW>([&](auto C) {
              constexpr int cc = decltype(C)::value;
              rf[cc] = af[cc] + hf[cc];
              if constexpr (ALPHA) {  // (fused multiply-adds: half the instructions of the separate form)
                const double o = of[cc], dx = __builtin_fma(-sg, o, rf[cc]);
                asum = __builtin_fma(o, rf[cc], asum);
                qsum = __builtin_fma(dx, dx, qsum);
                nsum = __builtin_fma(o, o, nsum);
              }
            });
            if constexpr (SH) sb_nt_store<T>(hp + roff(own + j) + col, res);
            else sb_nt_store<T>(orow + j * 16, res);
          });
        }
      });
    }
    __syncthreads();  // the next task overwrites the staged data
  }
  if (ALPHA) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      asum += __shfl_down(asum, off, 64);
      qsum += __shfl_down(qsum, off, 64);
      nsum += __shfl_down(nsum, off, 64);
    }
    if (lane == 0) {
      red[wave] = asum;
      red[NW + wave] = qsum;
      red[2 * NW + wave] = nsum;
    }
    __syncthreads();
    if (tid == 0) {
      double t = 0.0, q = 0.0, n = 0.0;
#pragma unroll
      for (int i = 0; i < NW; i++) {
        t += red[i];
        q += red[NW + i];
        n += red[2 * NW + i];
      }
      a.partial[blockIdx.x] = t;
      a.partial[gridDim.x + blockIdx.x] = q;
      a.partial[2 * gridDim.x + blockIdx.x] = n;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// launchers of one (NIMP, NB0, AMODE)
// ---------------------------------------------------------------------------------------------------------

// (top: rows staged in halves -- nbw counts the walked levels below the top one, whose amplitudes take one more table row)
inline size_t sb_rows_lds_bytes(int nbw, int rimg_len, bool top = false) {
  return ((size_t)rimg_len + (size_t)(nbw + (top ? 1 : 0)) * kSbVs + ((size_t)1 << nbw)) * sizeof(double) + ((size_t)1 << nbw) * sizeof(uint16_t);
}

template <int NIMP, int NB0, int AMODE, int NT, int NBT, int CS>
int sb_launch_rows_t(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st) {
  const size_t lds = d->sb->rows_lds;
  if (fuse >= 2) {  // rows staged in halves: fuse = 2 / 3 = the half with the top walked level empty / occupied
    if constexpr (AMODE == 0 && NBT <= 4) {
      const void* kt = fuse == 2 ? (const void*)sb_rows_kernel<NIMP, NB0, 0, NT, NBT, CS, 0, 1> : (const void*)sb_rows_kernel<NIMP, NB0, 0, NT, NBT, CS, 0, 2>;
      if (ensure_dynamic_lds(kt, lds)) return 1;
      const int pc = resident_blocks(kt, NT, lds);
      if (pc < 1) {
        set_error("sb_rows_kernel: does not fit a CU");
        return 1;
      }
      const int64_t g = std::min<int64_t>(std::max<int64_t>(a.dim_dw, 1), (int64_t)pc * device_cu_count());
      if (fuse == 2)
        hipLaunchKernelGGL((sb_rows_kernel<NIMP, NB0, 0, NT, NBT, CS, 0, 1>), dim3((unsigned)g), dim3(NT), lds, st, a, P, Q, X);
      else
        hipLaunchKernelGGL((sb_rows_kernel<NIMP, NB0, 0, NT, NBT, CS, 0, 2>), dim3((unsigned)g), dim3(NT), lds, st, a, P, Q, X);
      EDIGPU_HIP(hipGetLastError());
      return 0;
    } else {
      set_error("sb_rows_kernel: rows staged in halves are built for the all-orbital walk and at most 4 blocks per thread");
      return 1;
    }
  }
  const void* k = fuse ? (const void*)sb_rows_kernel<NIMP, NB0, AMODE, NT, NBT, CS, 1> : (const void*)sb_rows_kernel<NIMP, NB0, AMODE, NT, NBT, CS, 0>;
  if (ensure_dynamic_lds(k, lds)) return 1;
  const int per_cu = resident_blocks(k, NT, lds);
  if (per_cu < 1) {
    set_error("sb_rows_kernel: does not fit a CU");
    return 1;
  }
  const int64_t grid = std::min<int64_t>(std::max<int64_t>(a.dim_dw, 1), (int64_t)per_cu * device_cu_count());
  if (fuse)
    hipLaunchKernelGGL((sb_rows_kernel<NIMP, NB0, AMODE, NT, NBT, CS, 1>), dim3((unsigned)grid), dim3(NT), lds, st, a, P, Q, X);
  else
    hipLaunchKernelGGL((sb_rows_kernel<NIMP, NB0, AMODE, NT, NBT, CS, 0>), dim3((unsigned)grid), dim3(NT), lds, st, a, P, Q, X);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// the geometries the rows kernel is built for: threads per workgroup, blocks per thread, class stride of the row image
// (kernels_sb.hip sb_rows_config chooses among them)
#define EDIGPU_SB_ROWS_GEOMETRIES(X) X(256, 3, 65) X(256, 3, 129) X(512, 3, 257) X(768, 3, 513) X(512, 5, 513)

template <int NIMP, int NB0, int AMODE>
int sb_launch_rows(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st) {
  const int nt = d->sb->rows_nt, nbt = d->sb->rows_nbt, cs = d->sb->rcs;
#define EDIGPU_SB_ONE(NT, NBT, CS) \
  if (nt == NT && nbt == NBT && cs == CS) return sb_launch_rows_t<NIMP, NB0, AMODE, NT, NBT, CS>(d, a, fuse, P, Q, X, st);
  EDIGPU_SB_ROWS_GEOMETRIES(EDIGPU_SB_ONE)
#undef EDIGPU_SB_ONE
  set_error("sb_rows_kernel: no instantiation for this geometry");
  return 1;
}

template <int NIMP, int NB0, int AMODE, int CW, bool DO_ND, bool ALPHA, bool SH>
int sb_launch_cols_t(const IbDev* d, const SbArgs& a, const double* v, double* hv, hipStream_t st, int* nblocks) {
  constexpr int NT = CW == 2 ? SB_COLS_NT2 : 512;
  const size_t lds = d->sb->cols_lds;
  const void* k = (const void*)sb_cols_kernel<NIMP, NB0, AMODE, CW, DO_ND, ALPHA, SH>;
  if (ensure_dynamic_lds(k, lds)) return 1;
  const int per_cu = resident_blocks(k, NT, lds);
  if (per_cu < 1) {
    set_error("sb_cols_kernel: does not fit a CU");
    return 1;
  }
  int64_t grid = (int64_t)per_cu * device_cu_count();
  const int64_t tasks = (int64_t)a.npanels * d->sb->nchunks;
  grid = std::min<int64_t>(grid, (tasks + 7) / 8 * 8);
  grid = std::max<int64_t>(8, grid / 8 * 8);
  if (ALPHA && 3 * grid > kMaxPartials) {
    set_error("sb_cols_kernel: partial buffer too small");
    return 1;
  }
  hipLaunchKernelGGL((sb_cols_kernel<NIMP, NB0, AMODE, CW, DO_ND, ALPHA, SH>), dim3((unsigned)grid), dim3(NT), lds, st, a, v, hv);
  EDIGPU_HIP(hipGetLastError());
  if (nblocks) *nblocks = (int)grid;
  return 0;
}

// mode: 0 plain product, 1 fused Lanczos step (the three sums), 2 row shards (write-only, rank slots)
template <int NIMP, int NB0, int AMODE, int CW>
int sb_launch_cols_w(const IbDev* d, const SbArgs& a, int mode, const double* v, double* hv, hipStream_t st, int* nblocks) {
  if (d->nterms > 0) {
    if (mode == 2) return sb_launch_cols_t<NIMP, NB0, AMODE, CW, true, false, true>(d, a, v, hv, st, nblocks);
    return mode ? sb_launch_cols_t<NIMP, NB0, AMODE, CW, true, true, false>(d, a, v, hv, st, nblocks)
                : sb_launch_cols_t<NIMP, NB0, AMODE, CW, true, false, false>(d, a, v, hv, st, nblocks);
  }
  if (mode == 2) return sb_launch_cols_t<NIMP, NB0, AMODE, CW, false, false, true>(d, a, v, hv, st, nblocks);
  return mode ? sb_launch_cols_t<NIMP, NB0, AMODE, CW, false, true, false>(d, a, v, hv, st, nblocks)
              : sb_launch_cols_t<NIMP, NB0, AMODE, CW, false, false, false>(d, a, v, hv, st, nblocks);
}

template <int NIMP, int NB0, int AMODE>
int sb_launch_cols(const IbDev* d, const SbArgs& a, int mode, const double* v, double* hv, hipStream_t st, int* nblocks) {
  if (d->sb->cols_gs == 4) return sb_launch_cols_w<NIMP, NB0, AMODE, 1>(d, a, mode, v, hv, st, nblocks);
  return sb_launch_cols_w<NIMP, NB0, AMODE, 2>(d, a, mode, v, hv, st, nblocks);
}

}  // namespace edigpu
