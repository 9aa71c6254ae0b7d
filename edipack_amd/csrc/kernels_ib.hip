// kernels_ib.hip -- normal-mode H*v on impurity blocks (gfx950).  host_ib.hpp explains the decomposition, ib_core.hpp
// holds the per-block arithmetic (shared with the CPU test shim tests/host_ib.cpp).
//
//   Hv = Hd o v + (1 (x) Hup) v + (Hdw (x) 1) v + Hnd v      (spMatVec_normal_main,
//                                                             ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650)
// on vectors in the padded panel layout (IbDev, edigpu_internal.hpp): 16-column panels, every (row, panel) segment one
// 128-byte line, blocks of columns never cut by a panel edge.
//
//  ib_rows_kernel  Hd + 1 (x) Hup.  Persistent workgroups, one staged row of V at a time in the LDS.  A lane owns
//                  whole BLOCKS of columns (the <= 3 adjacent states that share their bath word); per bath level one
//                  16-bit table look-up (LDS) gives the partner block, whose 1-3 adjacent words are read from the LDS
//                  and applied through the compile-time impurity pattern of the block's class.  No per-entry index
//                  data crosses the L2 (the typed ELL this replaces read 168 bytes of it per element and row); dead
//                  slots do not exist: a lane walks the set / clear bits of its own bath word.  Lanes of a wave hold
//                  blocks of one class, so the walk has the same length on all of them.  The result replaces the row
//                  in the LDS (all reads, barrier, all writes) and leaves coalesced, like it came in.
//  ib_cols_kernel  Hdw (x) 1 + Hnd.  A task = (panel, chunk of rows that share their high bath bits): the chunk's
//                  128-byte segments are staged in the LDS (<= 60 KB), a group of 8 lanes owns a block of rows x 16
//                  columns.  The hops to the low bath levels and the Hnd terms stay inside the chunk (LDS); the hops
//                  to the few high levels read the partner block's rows from the panel in the L2: once per block of
//                  rows, not once per row and hop.  Workgroups with equal blockIdx % 8 (one XCD under the round-robin
//                  dispatch; speed only) sweep the chunks of one panel together.
// No MFMA: fp64 sparse, bandwidth bound.
#include <algorithm>
#include <cstdlib>

#include "host_ib.hpp"
#include "ib_core.hpp"
#include "kernels.hpp"

namespace edigpu {

struct IbArgs {
  int nb_up, nb_dw, npanels, plen, nlist;
  int ucls[5];
  int lowbits, nchunks, max_chunk_rows, max_chunk_blocks, nterms;
  int nsub;  // workgroups that share a chunk of the columns kernel (each takes a part of its blocks)
  int64_t dim_dw, ps;
  const uint16_t *urank, *ublist;
  const uint32_t* rmap2;
  int rcb[5], rcs[5], rimg_len;
  const double *up_vtab, *up_timp, *up_ebath, *xu, *ed;
  const uint8_t* impd;
  const int32_t *chunk_row, *chunk_blk, *dcls;
  const uint16_t *dblist, *dmeta;
  const double *dw_vtab, *dw_timp, *ndcoef;
  const uint8_t *nd_dw, *nd_up;
  // bath-bath hops (replica / general baths; 0 otherwise)
  int up_np, dw_np;
  const uint32_t *up_pmask, *dw_pmask;
  const double *up_pt, *dw_pt;
  // split rows (host_ib.hpp IbUpHalf): the rows kernel sees ONE half -- nb_up, nlist, plen, ucls, rcb, rcs, rimg_len,
  // ublist, rmap2 and urank are that half's; panel0 = its first panel; utop = per list entry the position of the partner
  // block over the top bath level; top_eps = that level's energy when it is occupied in this half (else 0).  up_vtab
  // keeps all nb_up + 1 rows.
  int panel0;
  const uint16_t* utop;
  double top_eps;
  // fused Lanczos step
  const double* scal;
  double* partial;
  int lazy;
};

// ---------------------------------------------------------------------------------------------------------
// rows kernel
// ---------------------------------------------------------------------------------------------------------
// FUSE 0: Q = (Hd + 1 (x) Hup) P                       (plain product; also the first Lanczos step)
// FUSE 1: x = (Q - alpha P) / beta; X <- x; Q <- (Hd + 1 (x) Hup) x   (alpha = 0 unless a.lazy).  The new Lanczos
//         vector goes to a THIRD buffer and the term - beta P_old is left to the columns kernel, which reads P_old's
//         rows next to Q's: holding P_old in registers until the result leaves cost 28 of the 128 a lane has here and
//         put spill traffic into the hot loops (measured: 2.7 ms against 1.4 ms for the plain product).
// pieces of 16 bytes a thread moves per row, given its NBT blocks (a class-n block holds C(NORB, n) columns)
constexpr int ib_rows_nld(int norb, int nbt) { return norb == 1 ? nbt / 2 + 1 : norb == 2 ? nbt : nbt + 1; }

// TOP (split rows, FUSE 0 only): 0 = the whole row is staged; 1 / 2 = the half with the top bath level empty / occupied.
template <int NORB, int NT, int NBT, int FUSE, int TOP = 0>
__global__ void __launch_bounds__(NT, (NT == 768 ? 3 : (NT == 512 && NBT >= 12) ? 2 : 4)) ib_rows_kernel(IbArgs a, const double* __restrict__ P, double* __restrict__ Q, double* __restrict__ X) {
  static_assert(TOP == 0 || FUSE == 0, "split rows: plain product only");
  extern __shared__ double lds[];
  constexpr int MAXM = ib::binom(NORB, NORB / 2);
  constexpr int NIMP = 1 << NORB;
  constexpr int NLD = ib_rows_nld(NORB, NBT);  // (the set-up checks plen <= 2 NT NLD)
  const int nb = a.nb_up, plen2 = a.plen >> 1;
  double* row = lds;                                                      // the row image (ib_core.hpp RowImage)
  double* vtab = row + a.rimg_len;                                        // [nb + 1][4] (the last row: split rows)
  uint16_t* rank = reinterpret_cast<uint16_t*>(vtab + (nb + 1) * 4);      // [2^nb]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t dd = a.dim_dw, ps = a.ps;
  double ibeta = 1.0, alpha = 0.0;
  if (TOP != 0 && a.scal && a.scal[SC_STOP] != 0.0) return;  // (inside a recurrence that has terminated: uniform)
  if (FUSE) {
    if (a.scal[SC_STOP] != 0.0) return;  // recurrence already terminated (uniform)
    ibeta = 1.0 / a.scal[SC_BETA];
    alpha = a.lazy ? a.scal[SC_ALPHA] : 0.0;
  }
  for (int i = tid; i < (1 << nb); i += NT) rank[i] = a.urank[i];
  for (int i = tid; i < (nb + (TOP ? 1 : 0)) * 4; i += NT) vtab[i] = a.up_vtab[i];
  for (int i = tid; i < a.rimg_len; i += NT) row[i] = 0.0;  // slack and the zero word stay zero
  ib::RowImage im;
  im.row = row;
  im.rank = rank;
#pragma unroll
  for (int c = 0; c < 5; c++) {
    im.cb[c] = a.rcb[c];
    im.cs[c] = a.rcs[c];
  }
  // this thread's blocks: the same list entries for every row (two 16-bit entries per register: bath word | skip flag)
  uint32_t bw[(NBT + 1) / 2];
  ib::sfor<0, (NBT + 1) / 2>([&](auto S) {
    constexpr int s2 = decltype(S)::value;
    const int q0 = 2 * s2 * NT + tid, q1 = q0 + NT;
    bw[s2] = (q0 < a.nlist ? (uint32_t)a.ublist[q0] : 0u) | ((2 * s2 + 1 < NBT && q1 < a.nlist ? (uint32_t)a.ublist[q1] : 0u) << 16);
  });
  auto entry = [&](auto S) -> uint32_t {
    constexpr int s = decltype(S)::value;
    return (s & 1) ? bw[s / 2] >> 16 : bw[s / 2] & 0xFFFFu;
  };
  // split rows: where the partner block over the top level starts in the padded row (0xFFFF: no such block)
  uint32_t tw[TOP ? (NBT + 1) / 2 : 1];
  if constexpr (TOP != 0)
    ib::sfor<0, (NBT + 1) / 2>([&](auto S) {
      constexpr int s2 = decltype(S)::value;
      const int q0 = 2 * s2 * NT + tid, q1 = q0 + NT;
      tw[s2] = (q0 < a.nlist ? (uint32_t)a.utop[q0] : 0xFFFFu) | ((2 * s2 + 1 < NBT && q1 < a.nlist ? (uint32_t)a.utop[q1] : 0xFFFFu) << 16);
    });
  auto tentry = [&](auto S) -> uint32_t {
    constexpr int s = TOP ? decltype(S)::value : 0;
    return (s & 1) ? tw[s / 2] >> 16 : tw[s / 2] & 0xFFFFu;
  };
  // class of the 64 list entries a wave holds in slot s (uniform)
  auto cls_of = [&](int q0) -> int {
    int n = 0;
#pragma unroll
    for (int c = 1; c <= NORB; c++) n += q0 >= a.ucls[c] ? 1 : 0;
    return n;
  };
  // A thread moves the 16-byte pieces tid, tid + NT, ... of a row: piece q2 is the column pair (q2 & 7) of panel q2 >> 3,
  // so consecutive pieces of a thread lie (NT / 8) panels apart -- one base offset per row and a uniform stride (kept
  // that way on purpose: per-piece offsets would be hoisted out of the row loop into registers the blocks need).
  // a.rmap2[q2] names the two words of the image the piece's columns go to.
  const int64_t pstride0 = (int64_t)(NT / 8) * ps;
  int64_t pstride = pstride0;  // (+ an opaque zero inside the row loop, see zr there: the per-piece addresses are not hoisted)
  auto base_of = [&](int64_t r) -> int64_t { return (int64_t)((tid >> 3) + (TOP ? a.panel0 : 0)) * ps + r * 16 + ((tid & 7) << 1); };
  // The next row is requested when the blocks are done (its pieces land while the results are written back and leave):
  // requested before the block updates, the pieces in flight cost 28 registers the updates need.  EARLY where they fit.
  constexpr bool EARLY = !FUSE && NBT * MAXM <= 18;
  double2 pre[NLD];                    // the next row on its way in (FUSE: Q)
  double2 pin[FUSE ? NLD : 1];
  auto issue = [&](int64_t r) {
    const int64_t g0 = base_of(r);
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
  // loaded pieces -> image (FUSE: x = (Q - alpha P) / beta, P <- x, P_old kept)
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
  for (; r < dd; r += gridDim.x) {
    const int64_t rn = r + gridDim.x;
    const bool more = rn < dd;
    if (EARLY && more) issue(rn);  // in flight during the block updates
    // an opaque zero, new in every iteration: added to the per-(slot, class) block indices below so that they are
    // recomputed where they are used (one add) instead of being hoisted out of the row loop into 4 NBT registers
    int zr, zs;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zr));
    asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
    pstride = pstride0 + zs;
    const double edr = a.ed[r] + (TOP ? a.top_eps : 0.0);
    const double* xuc = a.xu + (int)a.impd[r] * NIMP;  // uniform: scalar loads (an LDS copy would sit in vector registers)
    double acc[NBT][MAXM];
    ib::sfor<0, NBT>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if (s * NT + wave * 64 < a.nlist) {  // uniform (the list is padded to whole waves)
        const int n = cls_of(s * NT + wave * 64);
        ib::for_class<NORB>(n, [&](auto N) {
          constexpr int nn = decltype(N)::value;
          constexpr int MPT = TOP ? ib::rows_top_words<NORB, nn, TOP == 2>() : 0;
          ib::rows_block<NORB, nn>(im, (entry(S) + (uint32_t)zr) & 0x7FFFu, (uint32_t)(s * NT + tid - a.ucls[nn] + zr), nb, vtab, a.up_timp, edr, xuc,
                                   acc[s], TOP ? 0 : a.up_np, a.up_pmask, a.up_pt);
          double xg[MPT > 0 ? MPT : 1];
          if constexpr (MPT > 0) {
            // the partner block over the top level, word by word (without Hnd terms the columns are not padded and a
            // block may cross a panel edge).  Requested AFTER the walk: in flight during it, the words cost registers
            // the walk spills for (measured at Ns = 17: 10.1 against 9.7 ms per product)
            const uint32_t tp = tentry(S) + (uint32_t)zr;
            ib::sfor<0, MPT>([&](auto J) {
              const uint32_t pj = tp + (uint32_t)decltype(J)::value;
              xg[decltype(J)::value] = tp != 0xFFFFu ? P[(int64_t)(pj >> 4) * ps + r * 16 + (pj & 15u)] : 0.0;
            });
          }
          if constexpr (MPT > 0) ib::rows_top<NORB, nn, TOP == 2>((entry(S) + (uint32_t)zr) & 0x7FFFu, vtab + nb * 4, xg, acc[s]);
        });
      }
    });
    __syncthreads();  // every read of the row is done: the results take its place
    ib::sfor<0, NBT>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if (s * NT + wave * 64 < a.nlist) {
        const int n = cls_of(s * NT + wave * 64);
        if (!(entry(S) & 0x8000u))
          ib::for_class<NORB>(n, [&](auto N) {
            constexpr int nn = decltype(N)::value;
            double* own = row + im.cb[nn + 1] + (s * NT + tid - a.ucls[nn] + zr);
            ib::sfor<0, ib::binom(NORB, nn)>([&](auto J) { own[decltype(J)::value * im.cs[nn + 1]] = acc[s][decltype(J)::value]; });
          });
      }
    });
    uint32_t mp[NLD];
    load_map(mp);
    if (!EARLY && more) issue(rn);
    __syncthreads();
    // the result leaves coalesced, the next row takes its place
    {
      const int64_t g0 = base_of(r);
#pragma unroll
      for (int i = 0; i < NLD; i++) {
        if (tid + i * NT < plen2) {
          double2 o;
          o.x = row[mp[i] & 0xFFFFu];
          o.y = row[mp[i] >> 16];
          *reinterpret_cast<double2*>(Q + g0 + i * pstride) = o;
        }
      }
    }
    if (more) land(rn, mp);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// columns kernel
// ---------------------------------------------------------------------------------------------------------
constexpr int kColsNT = 512;

// Streamed once: the result rows (read, updated, written) and P_old's rows.  Non-temporal, so that they do not push the
// panel's V segments -- gathered again and again by the neighbouring chunks -- out of the XCD's L2.
typedef double ib_d2 __attribute__((ext_vector_type(2)));
// compile-time ablations for scripts/r3_abl.sh (never set in the shipped build): 1 = no read of the rows kernel's
// result, 2 = partner rows outside the chunk read from the LDS instead, 4 = no stores
#ifndef IB_ABL
#define IB_ABL 0
#endif
__device__ inline ib::Pair nt_load(const double* p) {
  const ib_d2 t = __builtin_nontemporal_load(reinterpret_cast<const ib_d2*>(p));
  return ib::Pair{t.x, t.y};
}
__device__ inline void nt_store(double* p, const ib::Pair& v) {
  ib_d2 t;
  t.x = v.x;
  t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<ib_d2*>(p));
}

template <int NORB, bool DO_ND, bool ALPHA>
__global__ void __launch_bounds__(kColsNT, 4) ib_cols_kernel(IbArgs a, const double* __restrict__ v, double* __restrict__ hv, const double* __restrict__ pold) {
  __shared__ double red[3 * (kColsNT / 64)];
  extern __shared__ double lds[];
  const int nb = a.nb_dw;
  double* chunk = lds;
  double* vtab = chunk + (size_t)a.max_chunk_rows * 16;
  double* timp = vtab + nb * 4;
  double* ndc = timp + 16;
  uint16_t* lmeta = reinterpret_cast<uint16_t*>(ndc + 16);
  uint16_t* lbl = lmeta + (size_t)a.max_chunk_blocks * 16;
  uint8_t* nddw = reinterpret_cast<uint8_t*>(lbl + ((a.max_chunk_blocks + 7) & ~7));
  uint8_t* ndu = nddw + 16 * 4 * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (ALPHA && a.scal[SC_STOP] != 0.0) {
    if (tid == 0) {
      a.partial[blockIdx.x] = 0.0;
      a.partial[gridDim.x + blockIdx.x] = 0.0;
      a.partial[2 * gridDim.x + blockIdx.x] = 0.0;
    }
    return;
  }
  const double sg = ALPHA ? a.scal[SC_ALPHA] : 0.0;  // <Q|Q> is accumulated about the previous alpha (k_finalize_ab)
  // fused Lanczos step after the first: hv starts as (rows kernel's part) - beta * P_old (see ib_rows_kernel FUSE 1)
  const double nbeta = (ALPHA && pold) ? -a.scal[SC_BETA] : 0.0;
  double asum = 0.0, qsum = 0.0, nsum = 0.0;
  for (int i = tid; i < nb * 4; i += kColsNT) vtab[i] = a.dw_vtab[i];
  if (tid < 16) {
    timp[tid] = tid < NORB * NORB ? a.dw_timp[tid] : 0.0;
    ndc[tid] = DO_ND && tid < a.nterms ? a.ndcoef[tid] : 0.0;
  }
  if (DO_ND)
    for (int i = tid; i < a.nterms * (NORB + 1) * 4; i += kColsNT) nddw[i] = a.nd_dw[i];
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int nch = a.nchunks, nsub = a.nsub, tpp = nch * nsub;  // tasks per panel
  const int panels_x = (a.npanels - x + 7) >> 3;
  const int col = (lane & 7) << 1;
  int cur_panel = -1;
  // A panel's tasks are consecutive: the slots of an XCD sweep one or two panels at a time, whose V segments (1.6 MB per
  // panel at Ns = 16) the chunks gather their partners from.  With two panels in flight the L2 is over-subscribed
  // (measured: 4.4 GB fetched per product for 2.7 GB of V + result); nsub > 1 workgroups per chunk (each stages the
  // chunk and takes a part of its blocks) keep a single panel in flight, and measured slower (edigpu_capi.hip).
  for (int tt = slot; tt < panels_x * tpp; tt += nslots) {
    const int pi = tt / tpp, panel = pi * 8 + x;
    const int tc = tt - pi * tpp, sub = tc % nsub;
    const int c = (tc / nsub + pi) % nch;  // rotated: a slot meets chunks of every size
    const int row0 = a.chunk_row[c], nrows = a.chunk_row[c + 1] - row0;
    const int blk0 = a.chunk_blk[c], nblk = a.chunk_blk[c + 1] - blk0;
    const double* __restrict__ vp = v + (int64_t)panel * a.ps;
    double* __restrict__ hp = hv + (int64_t)panel * a.ps;
    const double* __restrict__ pp = (ALPHA && pold) ? pold + (int64_t)panel * a.ps : nullptr;
    {
      const double2* __restrict__ src = reinterpret_cast<const double2*>(vp + (int64_t)row0 * 16);
      double2* dst = reinterpret_cast<double2*>(chunk);
      const int n2 = nrows * 8;
      for (int i0 = 0; i0 < n2; i0 += 4 * kColsNT) {
        double2 t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + tid + u * kColsNT;
          t[u] = src[i < n2 ? i : n2 - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + tid + u * kColsNT;
          if (i < n2) dst[i] = t[u];
        }
      }
      for (int i = tid; i < nblk; i += kColsNT) lbl[i] = a.dblist[blk0 + i];
      for (int i = tid; i < nblk * 2; i += kColsNT) {
        const uint32_t b = a.dblist[blk0 + (i >> 1)] & 0x7FFFu;
        reinterpret_cast<uint4*>(lmeta)[i] = reinterpret_cast<const uint4*>(a.dmeta + (size_t)b * 16)[i & 1];
      }
      if (DO_ND && panel != cur_panel) {
        for (int i = tid; i < a.nterms * 16; i += kColsNT) ndu[i] = a.nd_up[(size_t)(i >> 4) * a.plen + panel * 16 + (i & 15)];
        cur_panel = panel;
      }
    }
    const int32_t* cl = a.dcls + (size_t)c * (kIbMaxNorb + 2);
    int cb[NORB + 1];
#pragma unroll
    for (int n = 1; n <= NORB; n++) cb[n] = cl[n];
    __syncthreads();
    const int ng = nblk >> 3, gb = (int)((int64_t)sub * ng / nsub), ge = (int)((int64_t)(sub + 1) * ng / nsub);
    for (int q0 = (gb + wave) * 8; q0 < ge * 8; q0 += kColsNT / 8) {  // uniform per wave (classes are padded to 8 blocks)
      int n = 0;
#pragma unroll
      for (int k = 1; k <= NORB; k++) n += q0 >= cb[k] ? 1 : 0;
      const int bi = q0 + (lane >> 3);
      const uint32_t e = lbl[bi];
      const uint32_t b = e & 0x7FFFu;
      const uint16_t* meta = lmeta + (size_t)bi * 16;
      const int own = meta[14];
      ib::for_class<NORB>(n, [&](auto N) {
        constexpr int nn = decltype(N)::value;
        constexpr int M = ib::binom(NORB, nn);
        // the rows kernel's part of the result: HBM misses, requested first and added last (plain product).  The fused
        // step has no registers to park them in (three accumulators + P_old's rows: spills measured slower): there the
        // accumulators start from them.
        ib::Pair acc[M], h0[ALPHA ? 1 : M];
        ib::sfor<0, M>([&](auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (ALPHA) {
            acc[j] = nt_load(hp + (int64_t)(own + j) * 16 + col);
            if (pp) {  // uniform
              const ib::Pair o = nt_load(pp + (int64_t)(own + j) * 16 + col);
              acc[j].x = __builtin_fma(nbeta, o.x, acc[j].x);
              acc[j].y = __builtin_fma(nbeta, o.y, acc[j].y);
            }
          } else {
            h0[j] = (IB_ABL & 1) ? ib::Pair{0.0, 0.0} : nt_load(hp + (int64_t)(own + j) * 16 + col);
            acc[j].x = acc[j].y = 0.0;
          }
        });
        auto gload = [&](int grow) -> ib::Pair {
            if (IB_ABL & 2) return *reinterpret_cast<const ib::Pair*>(chunk + (grow & 63) * 16 + col);
            return *reinterpret_cast<const ib::Pair*>(vp + (int64_t)grow * 16 + col);
          };
        ib::cols_block<NORB, nn>(chunk, row0, b, own, meta, nb, a.lowbits, vtab, timp, col, gload, acc, a.dw_np, a.dw_pmask, a.dw_pt, a.dmeta);
        if (DO_ND) ib::cols_block_nd<NORB, nn>(chunk, own - row0, col, a.nterms, ndc, nddw, ndu, 16, acc);
        if constexpr (!ALPHA)
          ib::sfor<0, M>([&](auto J) {
            constexpr int j = decltype(J)::value;
            acc[j].x += h0[j].x;
            acc[j].y += h0[j].y;
          });
        if (!(e & 0x8000u)) {
          ib::sfor<0, M>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if (!(IB_ABL & 4) || acc[j].x == 1.2345) nt_store(hp + (int64_t)(own + j) * 16 + col, acc[j]);
            if (ALPHA) {
              const ib::Pair o = *reinterpret_cast<const ib::Pair*>(chunk + (own - row0 + j) * 16 + col);
              const double dx = acc[j].x - sg * o.x, dy = acc[j].y - sg * o.y;
              asum += o.x * acc[j].x + o.y * acc[j].y;
              qsum += dx * dx + dy * dy;
              nsum += o.x * o.x + o.y * o.y;
            }
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
      red[kColsNT / 64 + wave] = qsum;
      red[2 * (kColsNT / 64) + wave] = nsum;
    }
    __syncthreads();
    if (tid == 0) {
      double t = 0.0, q = 0.0, n = 0.0;
#pragma unroll
      for (int i = 0; i < kColsNT / 64; i++) {
        t += red[i];
        q += red[kColsNT / 64 + i];
        n += red[2 * (kColsNT / 64) + i];
      }
      a.partial[blockIdx.x] = t;
      a.partial[gridDim.x + blockIdx.x] = q;
      a.partial[2 * gridDim.x + blockIdx.x] = n;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// columns kernel, pipelined form: ONE 1024-thread workgroup per CU, two LDS buffers, two LOADER waves
// ---------------------------------------------------------------------------------------------------------
// The form above alternates staging and block updates in every workgroup and relies on a second workgroup per CU to
// overlap them; its 64 workgroup slots per XCD keep two panels in flight, which over-subscribes the 4 MiB L2 (4.4 GB
// fetched per product at Ns = 16 for 2.7 GB of V + result).  Here a CU runs one workgroup whose last two waves do
// nothing but bring the NEXT task's chunk into the other LDS buffer -- by LDS-DMA (global_load_lds_dwordx4: no
// registers, no ds_write pass), plus the task's block records -- while the other fourteen update the blocks of the
// current task.  The loaders' counter (vmcnt) is their own, so the prefetch does not sit in front of the partner-row
// loads of the computing waves (within one wave it would: loads retire in order).  32 slots per XCD = one panel at a
// time at Ns = 16.
constexpr int kCols2NT = 1024, kCols2Loaders = 2, kCols2NCW = kCols2NT / 64 - kCols2Loaders;

struct Cols2Lds {  // byte offsets inside the dynamic LDS; buffer b of a pair sits at base + b * stride
  uint32_t chunk, chunk_stride, meta, meta_stride, lbl, lbl_stride, ndu, ndu_stride, vtab, timp, ndc, nddw, total;
};
__host__ __device__ inline Cols2Lds cols2_layout(int nb, int mcr, int mcb) {
  Cols2Lds l;
  uint32_t at = 0;
  l.chunk = at;
  l.chunk_stride = ((uint32_t)mcr * 128 + 1023) / 1024 * 1024;  // whole 1 KiB DMA pieces
  at += 2 * l.chunk_stride;
  l.meta = at;
  l.meta_stride = (uint32_t)mcb * 32;
  at += 2 * l.meta_stride;
  l.lbl = at;
  l.lbl_stride = (uint32_t)((mcb + 7) & ~7) * 2;
  at += 2 * l.lbl_stride;
  l.ndu = at;
  l.ndu_stride = 16 * 16;
  at += 2 * l.ndu_stride;
  at = (at + 15) & ~15u;
  l.vtab = at;
  at += (uint32_t)nb * 32;
  l.timp = at;
  at += 16 * 8;
  l.ndc = at;
  at += 16 * 8;
  l.nddw = at;
  at += 16 * 4 * 4;
  l.total = at;
  return l;
}

template <int NORB, bool DO_ND, bool ALPHA>
__global__ void __launch_bounds__(kCols2NT) ib_cols2_kernel(IbArgs a, const double* __restrict__ v, double* __restrict__ hv,
                                                            const double* __restrict__ pold) {
  __shared__ double red[3 * (kCols2NT / 64)];
  extern __shared__ double lds[];
  char* base = reinterpret_cast<char*>(lds);
  const int nb = a.nb_dw;
  const Cols2Lds L = cols2_layout(nb, a.max_chunk_rows, a.max_chunk_blocks);
  double* vtab = reinterpret_cast<double*>(base + L.vtab);
  double* timp = reinterpret_cast<double*>(base + L.timp);
  double* ndc = reinterpret_cast<double*>(base + L.ndc);
  uint8_t* nddw = reinterpret_cast<uint8_t*>(base + L.nddw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= kCols2NCW;
  if (ALPHA && a.scal[SC_STOP] != 0.0) {
    if (tid == 0) {
      a.partial[blockIdx.x] = 0.0;
      a.partial[gridDim.x + blockIdx.x] = 0.0;
      a.partial[2 * gridDim.x + blockIdx.x] = 0.0;
    }
    return;
  }
  const double sg = ALPHA ? a.scal[SC_ALPHA] : 0.0;
  const double nbeta = (ALPHA && pold) ? -a.scal[SC_BETA] : 0.0;
  double asum = 0.0, qsum = 0.0, nsum = 0.0;
  for (int i = tid; i < nb * 4; i += kCols2NT) vtab[i] = a.dw_vtab[i];
  if (tid < 16) {
    timp[tid] = tid < NORB * NORB ? a.dw_timp[tid] : 0.0;
    ndc[tid] = DO_ND && tid < a.nterms ? a.ndcoef[tid] : 0.0;
  }
  if (DO_ND)
    for (int i = tid; i < a.nterms * (NORB + 1) * 4; i += kCols2NT) nddw[i] = a.nd_dw[i];
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int nch = a.nchunks;
  const int panels_x = (a.npanels - x + 7) >> 3, ntask = panels_x * nch;
  const int col = (lane & 7) << 1;
  auto task_of = [&](int tt, int& panel, int& c) {
    const int pi = tt / nch;
    panel = pi * 8 + x;
    c = (tt - pi * nch + pi) % nch;  // rotated: a slot meets chunks of every size
  };
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  // loader waves: task tt into buffer bs
  auto stage = [&](int tt, int bs) {
    int panel, c;
    task_of(tt, panel, c);
    const int row0 = a.chunk_row[c], nrows = a.chunk_row[c + 1] - row0;
    const int blk0 = a.chunk_blk[c], nblk = a.chunk_blk[c + 1] - blk0;
    const int lw = wave - kCols2NCW, ll = lw * 64 + lane;  // 0 .. 64 * loaders - 1
    const double* src = v + (int64_t)panel * a.ps + (int64_t)row0 * 16;
    char* dstc = base + (L.chunk + bs * L.chunk_stride);
    const int n16 = nrows * 8;  // 16-byte units; a DMA instruction moves 64 of them (1 KiB) to consecutive LDS bytes
    for (int u0 = lw * 64; u0 < n16; u0 += kCols2Loaders * 64) {
      const int u = u0 + lane < n16 ? u0 + lane : n16 - 1;  // the tail lanes re-read the last unit (their bytes are never used)
      __builtin_amdgcn_global_load_lds((glb_void*)(src + (int64_t)u * 2), (lds_void*)(dstc + (size_t)u0 * 16), 16, 0, 0);
    }
    uint16_t* lbl = reinterpret_cast<uint16_t*>(base + (L.lbl + bs * L.lbl_stride));
    uint4* lmeta = reinterpret_cast<uint4*>(base + (L.meta + bs * L.meta_stride));
    for (int i = ll; i < nblk; i += kCols2Loaders * 64) lbl[i] = a.dblist[blk0 + i];
    for (int i = ll; i < nblk * 2; i += kCols2Loaders * 64) {
      const uint32_t b = a.dblist[blk0 + (i >> 1)] & 0x7FFFu;
      lmeta[i] = reinterpret_cast<const uint4*>(a.dmeta + (size_t)b * 16)[i & 1];
    }
    if (DO_ND) {
      uint8_t* ndu = reinterpret_cast<uint8_t*>(base + (L.ndu + bs * L.ndu_stride));
      for (int i = ll; i < a.nterms * 16; i += kCols2Loaders * 64) ndu[i] = a.nd_up[(size_t)(i >> 4) * a.plen + panel * 16 + (i & 15)];
    }
    // the DMA counts on vmcnt; a workgroup barrier alone does not wait for it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (loader && slot < ntask) stage(slot, 0);
  __syncthreads();
  int it = 0;
  for (int tt = slot; tt < ntask; tt += nslots, it++) {
    const int bs = it & 1;
    if (loader) {
      if (tt + nslots < ntask) stage(tt + nslots, bs ^ 1);
    } else {
      int panel, c;
      task_of(tt, panel, c);
      const int row0 = a.chunk_row[c];
      const int nblk = a.chunk_blk[c + 1] - a.chunk_blk[c];
      const double* chunk = reinterpret_cast<const double*>(base + (L.chunk + bs * L.chunk_stride));
      const uint16_t* lmeta = reinterpret_cast<const uint16_t*>(base + (L.meta + bs * L.meta_stride));
      const uint16_t* lbl = reinterpret_cast<const uint16_t*>(base + (L.lbl + bs * L.lbl_stride));
      const uint8_t* ndu = reinterpret_cast<const uint8_t*>(base + (L.ndu + bs * L.ndu_stride));
      const double* __restrict__ vp = v + (int64_t)panel * a.ps;
      double* __restrict__ hp = hv + (int64_t)panel * a.ps;
      const double* __restrict__ pp = (ALPHA && pold) ? pold + (int64_t)panel * a.ps : nullptr;
      const int32_t* cl = a.dcls + (size_t)c * (kIbMaxNorb + 2);
      int cb[NORB + 1];
#pragma unroll
      for (int n = 1; n <= NORB; n++) cb[n] = cl[n];
      for (int q0 = wave * 8; q0 < nblk; q0 += kCols2NCW * 8) {  // uniform per wave (classes are padded to 8 blocks)
        int n = 0;
#pragma unroll
        for (int k = 1; k <= NORB; k++) n += q0 >= cb[k] ? 1 : 0;
        const int bi = q0 + (lane >> 3);
        const uint32_t e = lbl[bi];
        const uint32_t b = e & 0x7FFFu;
        const uint16_t* meta = lmeta + (size_t)bi * 16;
        const int own = meta[14];
        ib::for_class<NORB>(n, [&](auto N) {
          constexpr int nn = decltype(N)::value;
          constexpr int M = ib::binom(NORB, nn);
          ib::Pair acc[M], h0[ALPHA ? 1 : M];
          ib::sfor<0, M>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (ALPHA) {
              acc[j] = nt_load(hp + (int64_t)(own + j) * 16 + col);
              if (pp) {  // uniform
                const ib::Pair o = nt_load(pp + (int64_t)(own + j) * 16 + col);
                acc[j].x = __builtin_fma(nbeta, o.x, acc[j].x);
                acc[j].y = __builtin_fma(nbeta, o.y, acc[j].y);
              }
            } else {
              h0[j] = (IB_ABL & 1) ? ib::Pair{0.0, 0.0} : nt_load(hp + (int64_t)(own + j) * 16 + col);
              acc[j].x = acc[j].y = 0.0;
            }
          });
          auto gload = [&](int grow) -> ib::Pair {
            if (IB_ABL & 2) return *reinterpret_cast<const ib::Pair*>(chunk + (grow & 63) * 16 + col);
            return *reinterpret_cast<const ib::Pair*>(vp + (int64_t)grow * 16 + col);
          };
          ib::cols_block<NORB, nn>(chunk, row0, b, own, meta, nb, a.lowbits, vtab, timp, col, gload, acc, a.dw_np, a.dw_pmask, a.dw_pt, a.dmeta);
          if (DO_ND) ib::cols_block_nd<NORB, nn>(chunk, own - row0, col, a.nterms, ndc, nddw, ndu, 16, acc);
          if constexpr (!ALPHA)
            ib::sfor<0, M>([&](auto J) {
              constexpr int j = decltype(J)::value;
              acc[j].x += h0[j].x;
              acc[j].y += h0[j].y;
            });
          if (!(e & 0x8000u)) {
            ib::sfor<0, M>([&](auto J) {
              constexpr int j = decltype(J)::value;
              if (!(IB_ABL & 4) || acc[j].x == 1.2345) nt_store(hp + (int64_t)(own + j) * 16 + col, acc[j]);
              if (ALPHA) {
                const ib::Pair o = *reinterpret_cast<const ib::Pair*>(chunk + (own - row0 + j) * 16 + col);
                const double dx = acc[j].x - sg * o.x, dy = acc[j].y - sg * o.y;
                asum += o.x * acc[j].x + o.y * acc[j].y;
                qsum += dx * dx + dy * dy;
                nsum += o.x * o.x + o.y * o.y;
              }
            });
          }
        });
      }
    }
    __syncthreads();  // the other buffer is complete (the loaders' DMA has retired), this one may be overwritten
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
      red[kCols2NT / 64 + wave] = qsum;
      red[2 * (kCols2NT / 64) + wave] = nsum;
    }
    __syncthreads();
    if (tid == 0) {
      double t = 0.0, q = 0.0, n = 0.0;
#pragma unroll
      for (int i = 0; i < kCols2NT / 64; i++) {
        t += red[i];
        q += red[kCols2NT / 64 + i];
        n += red[2 * (kCols2NT / 64) + i];
      }
      a.partial[blockIdx.x] = t;
      a.partial[gridDim.x + blockIdx.x] = q;
      a.partial[2 * gridDim.x + blockIdx.x] = n;
    }
  }
}

size_t ib_cols2_lds_bytes(int nb, int max_chunk_rows, int max_chunk_blocks) {
  return cols2_layout(nb, max_chunk_rows, max_chunk_blocks).total;
}

// ---------------------------------------------------------------------------------------------------------
// layout conversion: natural (idw * DimUp + iup) <-> padded panels
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_to_ib(const double* __restrict__ src, double* __restrict__ dst, const int32_t* __restrict__ colof,
                                               int64_t dim_up, int64_t dim_dw, int64_t ps, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t panel = i / ps, rem = i - panel * ps;
    const int64_t rowi = rem >> 4;
    const int c = colof[panel * 16 + (rem & 15)];
    dst[i] = (c >= 0 && rowi < dim_dw) ? src[rowi * dim_up + c] : 0.0;  // (rows past dim_dw: the padding of a padded panel stride)
  }
}

__global__ void __launch_bounds__(256) k_from_ib(const double* __restrict__ src, double* __restrict__ dst, const int32_t* __restrict__ pos,
                                                 int64_t dim_up, int64_t ps, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t rowi = i / dim_up, c = i - rowi * dim_up;
    const int p = pos[c];
    dst[i] = src[(int64_t)(p >> 4) * ps + (rowi << 4) + (p & 15)];
  }
}

int vec_to_ib(const IbDev* ib, const double* src, double* dst, hipStream_t st) {
  const unsigned g = (unsigned)std::min<int64_t>((ib->len + 255) / 256, 65536);
  hipLaunchKernelGGL(k_to_ib, dim3(g), dim3(256), 0, st, src, dst, ib->colof, ib->dim_up, ib->dim_dw, ib->ps, ib->len);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_from_ib(const IbDev* ib, const double* src, double* dst, hipStream_t st) {
  const int64_t n = ib->dim_up * ib->dim_dw;
  const unsigned g = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
  hipLaunchKernelGGL(k_from_ib, dim3(g), dim3(256), 0, st, src, dst, ib->pos, ib->dim_up, ib->ps, n);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
// half < 0: the whole row; 0 / 1: the rows kernel's view of one half of a split row
static void fill_ib_args(const IbDev* d, IbArgs& a, int half = -1) {
  a.nb_up = d->nb_up;
  a.nb_dw = d->nb_dw;
  a.npanels = d->npanels;
  a.plen = d->plen;
  a.nlist = d->nlist;
  for (int i = 0; i < 5; i++) a.ucls[i] = d->ucls[i];
  a.lowbits = d->lowbits;
  a.nchunks = d->nchunks;
  a.max_chunk_rows = d->max_chunk_rows;
  a.max_chunk_blocks = d->max_chunk_blocks;
  a.nterms = d->nterms;
  a.nsub = d->nsub;
  a.dim_dw = d->dim_dw;
  a.ps = d->ps;
  a.urank = d->urank;
  a.rmap2 = d->rmap2;
  for (int i = 0; i < 5; i++) {
    a.rcb[i] = d->rcb[i];
    a.rcs[i] = d->rcs[i];
  }
  a.rimg_len = d->rimg_len;
  a.ublist = d->ublist;
  a.up_vtab = d->up_vtab;
  a.up_timp = d->up_timp;
  a.up_ebath = d->up_ebath;
  a.xu = d->xu;
  a.ed = d->ed;
  a.impd = d->impd;
  a.chunk_row = d->chunk_row;
  a.chunk_blk = d->chunk_blk;
  a.dcls = d->dcls;
  a.dblist = d->dblist;
  a.dmeta = d->dmeta;
  a.dw_vtab = d->dw_vtab;
  a.dw_timp = d->dw_timp;
  a.ndcoef = d->ndcoef;
  a.nd_dw = d->nd_dw;
  a.nd_up = d->nd_up;
  a.up_np = d->up_np;
  a.dw_np = d->dw_np;
  a.up_pmask = d->up_pmask;
  a.dw_pmask = d->dw_pmask;
  a.up_pt = d->up_pt;
  a.dw_pt = d->dw_pt;
  a.scal = nullptr;
  a.partial = nullptr;
  a.lazy = 0;
  a.panel0 = 0;
  a.utop = nullptr;
  a.top_eps = 0.0;
  if (half >= 0) {
    const IbDevHalf& h = d->half[half];
    a.nb_up = d->nb_up - 1;
    a.nlist = h.nlist;
    a.plen = h.npanels * kIbPanel;
    a.panel0 = h.panel0;
    for (int i = 0; i < 5; i++) {
      a.ucls[i] = h.ucls[i];
      a.rcb[i] = h.rcb[i];
      a.rcs[i] = h.rcs[i];
    }
    a.rimg_len = h.rimg_len;
    a.ublist = h.ublist;
    a.rmap2 = h.rmap2;
    a.utop = h.utop;
    a.urank = d->urank_low;
    a.top_eps = half ? d->top_eps : 0.0;
  }
}

size_t ib_rows_lds_bytes(int nb, int rimg_len) {
  return ((size_t)rimg_len + (size_t)(nb + 1) * 4) * sizeof(double) + ((size_t)1 << nb) * sizeof(uint16_t);
}

size_t ib_cols_lds_bytes(int nb, int max_chunk_rows, int max_chunk_blocks) {
  return ((size_t)max_chunk_rows * 16 + (size_t)nb * 4 + 32) * sizeof(double) + (size_t)max_chunk_blocks * 32 +
         (size_t)((max_chunk_blocks + 7) & ~7) * 2 + 16 * 4 * 4 + 16 * 16;
}

// threads per workgroup / blocks per thread of the rows kernel for a list of nlist blocks and rows of plen columns;
// false: no instantiation fits (the caller keeps the generic kernels)
bool ib_rows_config(int norb, int nb, int nlist, int plen, int rimg_len, int* nt_out, int* nbt_out, bool split) {
  const size_t lds = ib_rows_lds_bytes(nb, rimg_len);
  if (lds > 158 * 1024) return false;
  // Candidates: threads per workgroup x blocks per thread the kernels are built for.  A lane has 128 registers when
  // 1024 threads share a CU; 8 blocks of three words do not fit them (spill traffic in the hot loops: measured), 6 just
  // do.  Rank: threads resident per CU, then no more than 6 blocks per thread, then workgroups per CU (one workgroup
  // alone has nothing to overlap its barriers and row moves with), then fewer blocks per thread.
  static const int opts[3][4] = {{8, 14, 0, 0}, {4, 6, 8, 0}, {4, 6, 8, 12}};
  int forced = 0;
  if (const char* e = getenv("EDIGPU_IB_NT")) forced = atoi(e);  // tuning
  long best = -1;
  for (int nt : {256, 512, 768, 1024}) {
    if (split ? nt != 1024 : (forced && nt != forced)) continue;  // (split rows: built for 1024 threads only)
    for (int nbt : opts[norb - 1]) {
      if (!nbt) continue;
      if (split && nbt > (norb == 1 ? 14 : 6)) continue;
      if (nbt == 8 && norb == 3 && nt == 1024) continue;
      if (nbt == 12 && nt != 512) continue;
      if ((int64_t)nbt * nt < nlist || (int64_t)2 * nt * ib_rows_nld(norb, nbt) < plen) continue;
      const int regs_threads = nbt == 12 ? 512 : nt == 768 ? 768 : 1024;  // lanes a CU holds at this register budget
      const int wgs = std::max(1, std::min<int>((int)((156 * 1024) / lds), regs_threads / nt));
      const long score = (long)(wgs * nt) * 1000000 + (nbt <= 6 ? 100000 : 0) + (long)std::min(wgs, 4) * 1000 + (100 - nbt);
      if (score > best) {
        best = score;
        *nt_out = nt;
        *nbt_out = nbt;
      }
    }
  }
  return best >= 0;
}

// split rows: the half named by top (1 / 2), plain product; built for the 1024-thread configurations only
template <int NORB, int NBT>
static int launch_rows_top(const IbDev* d, const IbArgs& a, int top, const double* P, double* Q, hipStream_t st) {
  constexpr int NT = 1024;
  const size_t lds = d->rows_lds;
  const void* k = top == 1 ? (const void*)ib_rows_kernel<NORB, NT, NBT, 0, 1> : (const void*)ib_rows_kernel<NORB, NT, NBT, 0, 2>;
  if (ensure_dynamic_lds(k, lds)) return 1;
  const int per_cu = resident_blocks(k, NT, lds);
  if (per_cu < 1) {
    set_error("ib_rows_kernel (split rows): does not fit a CU");
    return 1;
  }
  const int64_t grid = std::min<int64_t>(d->dim_dw, (int64_t)per_cu * device_cu_count());
  if (top == 1)
    hipLaunchKernelGGL((ib_rows_kernel<NORB, NT, NBT, 0, 1>), dim3((unsigned)grid), dim3(NT), lds, st, a, P, Q, nullptr);
  else
    hipLaunchKernelGGL((ib_rows_kernel<NORB, NT, NBT, 0, 2>), dim3((unsigned)grid), dim3(NT), lds, st, a, P, Q, nullptr);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

static int launch_ib_rows_top(const IbDev* d, const IbArgs& a, int top, const double* P, double* Q, hipStream_t st) {
  if (d->rows_nt == 1024) {
    if (d->norb == 1 && d->rows_nbt == 8) return launch_rows_top<1, 8>(d, a, top, P, Q, st);
    if (d->norb == 1 && d->rows_nbt == 14) return launch_rows_top<1, 14>(d, a, top, P, Q, st);
    if (d->norb == 2 && d->rows_nbt == 4) return launch_rows_top<2, 4>(d, a, top, P, Q, st);
    if (d->norb == 2 && d->rows_nbt == 6) return launch_rows_top<2, 6>(d, a, top, P, Q, st);
    if (d->norb == 3 && d->rows_nbt == 4) return launch_rows_top<3, 4>(d, a, top, P, Q, st);
    if (d->norb == 3 && d->rows_nbt == 6) return launch_rows_top<3, 6>(d, a, top, P, Q, st);
  }
  set_error("ib_rows_kernel (split rows): no instantiation for this sector");
  return 1;
}

template <int NORB, int NT, int NBT>
static int launch_rows_t(const IbDev* d, const IbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st) {
  const size_t lds = d->rows_lds;
  const void* k0 = (const void*)ib_rows_kernel<NORB, NT, NBT, 0>;
  const void* k1 = (const void*)ib_rows_kernel<NORB, NT, NBT, 1>;
  const void* k = fuse ? k1 : k0;
  if (ensure_dynamic_lds(k, lds)) return 1;
  const int per_cu = resident_blocks(k, NT, lds);
  if (per_cu < 1) {
    set_error("ib_rows_kernel: does not fit a CU");
    return 1;
  }
  const int64_t grid = std::min<int64_t>(d->dim_dw, (int64_t)per_cu * device_cu_count());
  if (fuse)
    hipLaunchKernelGGL((ib_rows_kernel<NORB, NT, NBT, 1>), dim3((unsigned)grid), dim3(NT), lds, st, a, P, Q, X);
  else
    hipLaunchKernelGGL((ib_rows_kernel<NORB, NT, NBT, 0>), dim3((unsigned)grid), dim3(NT), lds, st, a, P, Q, X);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

template <int NORB>
static int launch_rows_n(const IbDev* d, const IbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st) {
  constexpr int B0 = NORB == 1 ? 8 : 4, B1 = NORB == 1 ? 14 : 6;
#define EDIGPU_IB_ROWS(NT)                                                                                   \
  if (d->rows_nt == NT) {                                                                                    \
    if (d->rows_nbt == B0) return launch_rows_t<NORB, NT, B0>(d, a, fuse, P, Q, X, st);                         \
    if (d->rows_nbt == B1) return launch_rows_t<NORB, NT, B1>(d, a, fuse, P, Q, X, st);                         \
    if constexpr (NORB >= 2 && (NORB == 2 || NT != 1024))                                                    \
      if (d->rows_nbt == 8) return launch_rows_t<NORB, NT, 8>(d, a, fuse, P, Q, X, st);                         \
    if constexpr (NORB == 3 && NT == 512)                                                                    \
      if (d->rows_nbt == 12) return launch_rows_t<NORB, NT, 12>(d, a, fuse, P, Q, X, st);                       \
  }
  EDIGPU_IB_ROWS(256)
  EDIGPU_IB_ROWS(512)
  EDIGPU_IB_ROWS(768)
  EDIGPU_IB_ROWS(1024)
#undef EDIGPU_IB_ROWS
  set_error("ib_rows_kernel: no instantiation for this sector");
  return 1;
}

static int launch_ib_rows(const IbDev* d, const IbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st) {
  switch (d->norb) {
    case 1: return launch_rows_n<1>(d, a, fuse, P, Q, X, st);
    case 2: return launch_rows_n<2>(d, a, fuse, P, Q, X, st);
    case 3: return launch_rows_n<3>(d, a, fuse, P, Q, X, st);
  }
  set_error("ib_rows_kernel: norb");
  return 1;
}

template <int NORB, bool DO_ND, bool ALPHA>
static int launch_cols2_t(const IbDev* d, const IbArgs& a, const double* v, double* hv, const double* pold, hipStream_t st, int* nblocks) {
  const size_t lds = ib_cols2_lds_bytes(d->nb_dw, d->max_chunk_rows, d->max_chunk_blocks);
  const void* k = (const void*)ib_cols2_kernel<NORB, DO_ND, ALPHA>;
  if (ensure_dynamic_lds(k, lds)) return 1;
  if (resident_blocks(k, kCols2NT, lds) < 1) {
    set_error("ib_cols2_kernel: does not fit a CU");
    return 1;
  }
  int64_t grid = device_cu_count();  // one workgroup per CU
  const int64_t tasks = (int64_t)d->npanels * d->nchunks;
  grid = std::min<int64_t>(grid, (tasks + 7) / 8 * 8);
  grid = std::max<int64_t>(8, grid / 8 * 8);
  hipLaunchKernelGGL((ib_cols2_kernel<NORB, DO_ND, ALPHA>), dim3((unsigned)grid), dim3(kCols2NT), lds, st, a, v, hv, pold);
  EDIGPU_HIP(hipGetLastError());
  if (nblocks) *nblocks = (int)grid;
  return 0;
}

// which form of the columns kernel.  Measured (round 3): at Ns = 16 (panels of 1.6 MB) the pipelined form takes 2.33 ms
// per product against 2.36 -- inside the run-to-run spread, because the kernel is bound by the block updates themselves
// (with every per-element global access compiled out, -DIB_ABL=7, it still takes 0.94 ms).  At Ns = 17 a panel is 3.1 MB
// and the two panels per XCD that the plain form keeps in flight no longer fit its 4 MB L2 (23.8 GB fetched per product
// for 9.4 GB of V + result read-in): there the pipelined form, with one panel per XCD, measures 9.6 against 10.1 ms.
// Default: panels above 2 MiB.  EDIGPU_IB_COLS2=0 / 1 overrides.
static bool use_cols2(const IbDev* d) {
  static const char* e = getenv("EDIGPU_IB_COLS2");
  if (ib_cols2_lds_bytes(d->nb_dw, d->max_chunk_rows, d->max_chunk_blocks) > 156 * 1024) return false;
  if (e) return atoi(e) != 0;
  return d->dim_dw * (int64_t)(kIbPanel * sizeof(double)) > ((int64_t)2 << 20);
}

template <int NORB, bool DO_ND, bool ALPHA>
static int launch_cols_t(const IbDev* d, const IbArgs& a, const double* v, double* hv, const double* pold, hipStream_t st, int* nblocks) {
  if (use_cols2(d)) return launch_cols2_t<NORB, DO_ND, ALPHA>(d, a, v, hv, pold, st, nblocks);
  const size_t lds = d->cols_lds;
  const void* k = (const void*)ib_cols_kernel<NORB, DO_ND, ALPHA>;
  if (ensure_dynamic_lds(k, lds)) return 1;
  const int per_cu = resident_blocks(k, kColsNT, lds);
  if (per_cu < 1) {
    set_error("ib_cols_kernel: does not fit a CU");
    return 1;
  }
  int64_t grid = (int64_t)per_cu * device_cu_count();
  const int64_t tasks = (int64_t)d->npanels * d->nchunks * d->nsub;
  grid = std::min<int64_t>(grid, (tasks + 7) / 8 * 8);
  grid = std::max<int64_t>(8, grid / 8 * 8);
  if (ALPHA && 3 * grid > kMaxPartials) {
    set_error("ib_cols_kernel: partial buffer too small");
    return 1;
  }
  hipLaunchKernelGGL((ib_cols_kernel<NORB, DO_ND, ALPHA>), dim3((unsigned)grid), dim3(kColsNT), lds, st, a, v, hv, pold);
  EDIGPU_HIP(hipGetLastError());
  if (nblocks) *nblocks = (int)grid;
  return 0;
}

static int launch_ib_cols(const IbDev* d, const IbArgs& a, bool alpha, const double* v, double* hv, const double* pold, hipStream_t st, int* nblocks) {
  const bool nd = d->nterms > 0;
#define EDIGPU_IB_COLS(NORB)                                                                     \
  case NORB:                                                                                     \
    if (nd) return alpha ? launch_cols_t<NORB, true, true>(d, a, v, hv, pold, st, nblocks)             \
                         : launch_cols_t<NORB, true, false>(d, a, v, hv, pold, st, nblocks);           \
    return alpha ? launch_cols_t<NORB, false, true>(d, a, v, hv, pold, st, nblocks)                    \
                 : launch_cols_t<NORB, false, false>(d, a, v, hv, pold, st, nblocks);
  switch (d->norb) {
    EDIGPU_IB_COLS(1)
    EDIGPU_IB_COLS(2)
    EDIGPU_IB_COLS(3)
  }
#undef EDIGPU_IB_COLS
  set_error("ib_cols_kernel: norb");
  return 1;
}

// plain product on vectors in the padded panel layout
int launch_ib(const edigpu_sector* s, const double* v, double* hv, hipStream_t st) {
  if (s->ib->sb && s->ib->sb->nhalf == 1) return launch_sb(s, v, hv, st);
  IbArgs a;
  if (s->ib->nhalf == 2) {
    // rows longer than the LDS: one launch per half of the row (each reads the other half's words for the top level); the
    // local-block rows kernel where its tables exist, the columns kernel below either way
    for (int h = 0; h < 2; h++) {
      if (s->ib->sb) {
        if (launch_sb_rows_half(s, h, v, hv, nullptr, st)) return 1;
        continue;
      }
      fill_ib_args(s->ib, a, h);
      if (launch_ib_rows_top(s->ib, a, h + 1, v, hv, st)) return 1;
    }
    fill_ib_args(s->ib, a);
    return launch_ib_cols(s->ib, a, false, v, hv, nullptr, st, nullptr);
  }
  fill_ib_args(s->ib, a);
  if (launch_ib_rows(s->ib, a, 0, v, hv, nullptr, st)) return 1;
  return launch_ib_cols(s->ib, a, false, v, hv, nullptr, st, nullptr);
}

// One fused Lanczos step (launch_normal_lanczos, kernels_normal.hip, explains the protocol) on THREE buffers: P = the
// previous Lanczos vector, Q = the work vector, X = where the new Lanczos vector is written (steps after the first;
// the caller then takes X as the new P and the old P as the next X).
int launch_ib_lanczos(const edigpu_sector* s, const double* P, double* Q, double* X, const double* scal, double* partial,
                      int64_t partial_cap, bool first, bool lazy_axpy, hipStream_t st, int* npartial) {
  // The Lanczos step of the local-block kernels (launch_sb_lanczos chooses between its fused and its semi-fused form);
  // EDIGPU_SB_STEP=0 keeps the step on the kernels below while the plain product runs on the local blocks.
  if (s->ib->sb && s->ib->sb->nhalf == 1) {
    const char* es = getenv("EDIGPU_SB_STEP");
    const bool sb_step = (es ? atoi(es) != 0 : true) || s->ib->pr.on;  // (short rows: the two-buffer step of launch_sb_lanczos only)
    if (sb_step) return launch_sb_lanczos(s, P, Q, X, scal, partial, partial_cap, first, lazy_axpy, st, npartial);
  }
  IbArgs a;
  fill_ib_args(s->ib, a);
  a.scal = scal;
  a.partial = partial;
  a.lazy = lazy_axpy ? 1 : 0;
  (void)partial_cap;  // >= kMaxPartials (ensure_workspace); launch_cols_t checks its grid against that
  if (s->ib->nhalf == 2) {
    // rows staged in halves: the rotation the rows kernel does while it stages a WHOLE row is its own pass here; the
    // columns kernel with the - beta P_old term and the three sums is the one of the fused step
    auto rows2 = [&](const double* in, double* out) -> int {
      for (int h = 0; h < 2; h++) {
        if (s->ib->sb) {
          if (launch_sb_rows_half(s, h, in, out, scal, st)) return 1;
          continue;
        }
        IbArgs ah;
        fill_ib_args(s->ib, ah, h);
        ah.scal = scal;
        if (launch_ib_rows_top(s->ib, ah, h + 1, in, out, st)) return 1;
      }
      return 0;
    };
    if (first) {
      if (rows2(P, Q)) return 1;
      return launch_ib_cols(s->ib, a, true, P, Q, nullptr, st, npartial);
    }
    if (lz_next_vector(P, Q, X, s->ib->len, scal, lazy_axpy, st) || rows2(X, Q)) return 1;
    return launch_ib_cols(s->ib, a, true, X, Q, P, st, npartial);
  }
  if (first) {
    if (launch_ib_rows(s->ib, a, 0, P, Q, nullptr, st)) return 1;
    return launch_ib_cols(s->ib, a, true, P, Q, nullptr, st, npartial);
  }
  if (launch_ib_rows(s->ib, a, 1, P, Q, X, st)) return 1;
  return launch_ib_cols(s->ib, a, true, X, Q, P, st, npartial);
}

}  // namespace edigpu
