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
  int64_t dim_dw, ps;
  const uint16_t *upos, *ublist;
  const double *up_vtab, *up_timp, *up_ebath, *xu, *ed;
  const uint8_t* impd;
  const int32_t *chunk_row, *chunk_blk, *dcls;
  const uint16_t *dblist, *dmeta;
  const double *dw_vtab, *dw_timp, *ndcoef;
  const uint8_t *nd_dw, *nd_up;
  // fused Lanczos step
  const double* scal;
  double* partial;
  int lazy;
};

// ---------------------------------------------------------------------------------------------------------
// rows kernel
// ---------------------------------------------------------------------------------------------------------
// FUSE 0: hv = (Hd + 1 (x) Hup) v                       (also the first Lanczos step: v = P, hv = Q)
// FUSE 1: x = (Q - alpha P) / beta; P <- x; Q <- (Hd + 1 (x) Hup) x - beta P_old   (alpha = 0 unless a.lazy)
// pieces of 16 bytes a thread moves per row, given its NBT blocks (a class-n block holds C(NORB, n) columns)
constexpr int ib_rows_nld(int norb, int nbt) { return norb == 1 ? nbt / 2 + 1 : norb == 2 ? nbt : nbt + 1; }

template <int NORB, int NT, int NBT, int FUSE>
__global__ void __launch_bounds__(NT, 4) ib_rows_kernel(IbArgs a, double* __restrict__ P, double* __restrict__ Q) {
  extern __shared__ double lds[];
  constexpr int MAXM = ib::binom(NORB, NORB / 2);
  constexpr int NIMP = 1 << NORB;
  constexpr int NLD = ib_rows_nld(NORB, NBT);  // (the set-up checks plen <= 2 NT NLD)
  const int nb = a.nb_up, plen2 = a.plen >> 1;
  double* row = lds;
  double* vtab = row + a.plen + 8;
  double* timp = vtab + nb * 4;
  double* xu = timp + 16;
  uint16_t* upos = reinterpret_cast<uint16_t*>(xu + NIMP * NIMP);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t dd = a.dim_dw, ps = a.ps;
  double beta = 0.0, ibeta = 1.0, alpha = 0.0;
  if (FUSE) {
    if (a.scal[SC_STOP] != 0.0) return;  // recurrence already terminated (uniform)
    beta = a.scal[SC_BETA];
    ibeta = 1.0 / beta;
    alpha = a.lazy ? a.scal[SC_ALPHA] : 0.0;
  }
  for (int i = tid; i < (1 << nb); i += NT) upos[i] = a.upos[i];
  for (int i = tid; i < nb * 4; i += NT) vtab[i] = a.up_vtab[i];
  if (tid < 16) timp[tid] = tid < NORB * NORB ? a.up_timp[tid] : 0.0;
  for (int i = tid; i < NIMP * NIMP; i += NT) xu[i] = a.xu[i];
  if (tid < 8) row[a.plen + tid] = 0.0;
  // this thread's blocks: the same list entries for every row
  uint32_t bw[NBT];  // bath word | skip flag (bit 15) | position of the block's first column << 16
  double eb[NBT];
  ib::sfor<0, NBT>([&](auto S) {
    constexpr int s = decltype(S)::value;
    const int q = s * NT + tid;
    bw[s] = q < a.nlist ? (uint32_t)a.ublist[q] : 0u;
    eb[s] = q < a.nlist ? a.up_ebath[bw[s] & 0x7FFFu] : 0.0;
  });
  __syncthreads();
  ib::sfor<0, NBT>([&](auto S) {
    constexpr int s = decltype(S)::value;
    bw[s] |= (uint32_t)upos[bw[s] & 0x7FFFu] << 16;
  });
  // class of the 64 list entries a wave holds in slot s (uniform)
  auto cls_of = [&](int q0) -> int {
    int n = 0;
#pragma unroll
    for (int c = 1; c <= NORB; c++) n += q0 >= a.ucls[c] ? 1 : 0;
    return n;
  };
  // piece i of a row: double2 index, global offset
  auto piece = [&](int i, int& q2, int64_t& g, int64_t r) -> bool {
    q2 = tid + i * NT;
    const int qc = q2 < plen2 ? q2 : plen2 - 1;  // clamped: always a valid address
    g = (int64_t)(qc >> 3) * ps + r * 16 + ((qc & 7) << 1);
    return q2 < plen2;
  };
  double2 pre[NLD];                    // the next row on its way in (FUSE: Q)
  double2 pold[FUSE ? NLD : 1];        // FUSE: P of the staged row, needed again when the result leaves
  double2 pin[FUSE ? NLD : 1];
  auto issue = [&](int64_t r) {
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      int q2;
      int64_t g;
      piece(i, q2, g, r);
      if (FUSE) {
        pre[i] = *reinterpret_cast<const double2*>(Q + g);
        pin[i] = *reinterpret_cast<const double2*>(P + g);
      } else {
        pre[i] = *reinterpret_cast<const double2*>(P + g);
      }
    }
  };
  // loaded pieces -> LDS (FUSE: x = (Q - alpha P) / beta, P <- x, P_old kept)
  auto land = [&](int64_t r) {
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      int q2;
      int64_t g;
      if (!piece(i, q2, g, r)) continue;
      double2 x = pre[i];
      if (FUSE) {
        x.x = (x.x - alpha * pin[i].x) * ibeta;
        x.y = (x.y - alpha * pin[i].y) * ibeta;
        pold[i] = pin[i];
        *reinterpret_cast<double2*>(P + g) = x;
      }
      reinterpret_cast<double2*>(row)[q2] = x;
    }
  };
  int64_t r = blockIdx.x;
  if (r >= dd) return;
  issue(r);
  land(r);
  __syncthreads();
  for (; r < dd; r += gridDim.x) {
    const int64_t rn = r + gridDim.x;
    const bool more = rn < dd;
    if (!FUSE && more) issue(rn);  // in flight during the block updates
    const double edr = a.ed[r];
    const double* xuc = xu + (int)a.impd[r] * NIMP;
    double acc[NBT][MAXM];
    ib::sfor<0, NBT>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if (s * NT + wave * 64 < a.nlist) {  // uniform (the list is padded to whole waves)
        const int n = cls_of(s * NT + wave * 64);
        ib::for_class<NORB>(n, [&](auto N) {
          ib::rows_block<NORB, decltype(N)::value>(row, bw[s] & 0x7FFFu, bw[s] >> 16, nb, upos, vtab, timp, eb[s] + edr, xuc, acc[s]);
        });
      }
    });
    __syncthreads();  // every read of the row is done: the results take its place
    ib::sfor<0, NBT>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if (s * NT + wave * 64 < a.nlist) {
        const int n = cls_of(s * NT + wave * 64);
        if (!(bw[s] & 0x8000u))
          ib::for_class<NORB>(n, [&](auto N) {
            ib::sfor<0, ib::binom(NORB, decltype(N)::value)>([&](auto J) { row[(bw[s] >> 16) + decltype(J)::value] = acc[s][decltype(J)::value]; });
          });
      }
    });
    if (FUSE && more) issue(rn);
    __syncthreads();
    // the result leaves coalesced, the next row takes its place
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      int q2;
      int64_t g;
      if (!piece(i, q2, g, r)) continue;
      double2 o = reinterpret_cast<const double2*>(row)[q2];
      if (FUSE) {
        o.x -= beta * pold[i].x;
        o.y -= beta * pold[i].y;
      }
      *reinterpret_cast<double2*>(Q + g) = o;
    }
    if (more) land(rn);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// columns kernel
// ---------------------------------------------------------------------------------------------------------
constexpr int kColsNT = 512;

template <int NORB, bool DO_ND, bool ALPHA>
__global__ void __launch_bounds__(kColsNT, 4) ib_cols_kernel(IbArgs a, const double* __restrict__ v, double* __restrict__ hv) {
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
  double asum = 0.0, qsum = 0.0, nsum = 0.0;
  for (int i = tid; i < nb * 4; i += kColsNT) vtab[i] = a.dw_vtab[i];
  if (tid < 16) {
    timp[tid] = tid < NORB * NORB ? a.dw_timp[tid] : 0.0;
    ndc[tid] = DO_ND && tid < a.nterms ? a.ndcoef[tid] : 0.0;
  }
  if (DO_ND)
    for (int i = tid; i < a.nterms * (NORB + 1) * 4; i += kColsNT) nddw[i] = a.nd_dw[i];
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int nch = a.nchunks;
  const int panels_x = (a.npanels - x + 7) >> 3;
  const int col = (lane & 7) << 1;
  int cur_panel = -1;
  for (int tt = slot; tt < panels_x * nch; tt += nslots) {
    const int pi = tt / nch, panel = pi * 8 + x;
    const int c = (tt - pi * nch + pi) % nch;  // rotated: a slot meets chunks of every size
    const int row0 = a.chunk_row[c], nrows = a.chunk_row[c + 1] - row0;
    const int blk0 = a.chunk_blk[c], nblk = a.chunk_blk[c + 1] - blk0;
    const double* __restrict__ vp = v + (int64_t)panel * a.ps;
    double* __restrict__ hp = hv + (int64_t)panel * a.ps;
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
    for (int q0 = wave * 8; q0 < nblk; q0 += kColsNT / 8) {  // uniform per wave (classes are padded to 8 blocks)
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
        ib::Pair acc[M];
        ib::sfor<0, M>([&](auto J) {
          acc[decltype(J)::value] = *reinterpret_cast<const ib::Pair*>(hp + (int64_t)(own + decltype(J)::value) * 16 + col);
        });
        auto gload = [&](int grow) -> ib::Pair { return *reinterpret_cast<const ib::Pair*>(vp + (int64_t)grow * 16 + col); };
        ib::cols_block<NORB, nn>(chunk, row0, b, own, meta, nb, a.lowbits, vtab, timp, col, gload, acc);
        if (DO_ND) ib::cols_block_nd<NORB, nn>(chunk, own - row0, col, a.nterms, ndc, nddw, ndu, 16, acc);
        if (!(e & 0x8000u)) {
          ib::sfor<0, M>([&](auto J) {
            constexpr int j = decltype(J)::value;
            *reinterpret_cast<ib::Pair*>(hp + (int64_t)(own + j) * 16 + col) = acc[j];
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
// layout conversion: natural (idw * DimUp + iup) <-> padded panels
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_to_ib(const double* __restrict__ src, double* __restrict__ dst, const int32_t* __restrict__ colof,
                                               int64_t dim_up, int64_t ps, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t panel = i / ps, rem = i - panel * ps;
    const int64_t rowi = rem >> 4;
    const int c = colof[panel * 16 + (rem & 15)];
    dst[i] = c >= 0 ? src[rowi * dim_up + c] : 0.0;
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
  hipLaunchKernelGGL(k_to_ib, dim3(g), dim3(256), 0, st, src, dst, ib->colof, ib->dim_up, ib->ps, ib->len);
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
static void fill_ib_args(const IbDev* d, IbArgs& a) {
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
  a.dim_dw = d->dim_dw;
  a.ps = d->ps;
  a.upos = d->upos;
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
  a.scal = nullptr;
  a.partial = nullptr;
  a.lazy = 0;
}

size_t ib_rows_lds_bytes(int norb, int nb, int plen) {
  return ((size_t)plen + 8 + (size_t)nb * 4 + 16 + ((size_t)1 << (2 * norb))) * sizeof(double) + ((size_t)1 << nb) * sizeof(uint16_t);
}

size_t ib_cols_lds_bytes(int nb, int max_chunk_rows, int max_chunk_blocks) {
  return ((size_t)max_chunk_rows * 16 + (size_t)nb * 4 + 32) * sizeof(double) + (size_t)max_chunk_blocks * 32 +
         (size_t)((max_chunk_blocks + 7) & ~7) * 2 + 16 * 4 * 4 + 16 * 16;
}

// threads per workgroup / blocks per thread of the rows kernel for a list of nlist blocks and rows of plen columns;
// false: no instantiation fits (the caller keeps the generic kernels)
bool ib_rows_config(int norb, int nb, int nlist, int plen, int* nt_out, int* nbt_out) {
  const size_t lds = ib_rows_lds_bytes(norb, nb, plen);
  if (lds > 158 * 1024) return false;
  // small rows: several 256-thread workgroups per CU; rows beyond half the LDS: one 1024-thread workgroup
  static const int opts[3][3] = {{8, 14, 0}, {4, 6, 8}, {4, 6, 0}};  // blocks per thread the kernels are built for
  for (int nt : {256, 512, 1024}) {
    if (nt == 256 && lds > 40 * 1024) continue;
    if (nt == 512 && lds > 80 * 1024) continue;
    for (int nbt : opts[norb - 1]) {
      if (nbt && (int64_t)nbt * nt >= nlist && (int64_t)2 * nt * ib_rows_nld(norb, nbt) >= plen) {
        *nt_out = nt;
        *nbt_out = nbt;
        return true;
      }
    }
  }
  return false;
}

template <int NORB, int NT, int NBT>
static int launch_rows_t(const IbDev* d, const IbArgs& a, int fuse, double* P, double* Q, hipStream_t st) {
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
    hipLaunchKernelGGL((ib_rows_kernel<NORB, NT, NBT, 1>), dim3((unsigned)grid), dim3(NT), lds, st, a, P, Q);
  else
    hipLaunchKernelGGL((ib_rows_kernel<NORB, NT, NBT, 0>), dim3((unsigned)grid), dim3(NT), lds, st, a, P, Q);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

template <int NORB>
static int launch_rows_n(const IbDev* d, const IbArgs& a, int fuse, double* P, double* Q, hipStream_t st) {
  constexpr int B0 = NORB == 1 ? 8 : 4, B1 = NORB == 1 ? 14 : 6;
#define EDIGPU_IB_ROWS(NT)                                                                                   \
  if (d->rows_nt == NT) {                                                                                    \
    if (d->rows_nbt == B0) return launch_rows_t<NORB, NT, B0>(d, a, fuse, P, Q, st);                         \
    if (d->rows_nbt == B1) return launch_rows_t<NORB, NT, B1>(d, a, fuse, P, Q, st);                         \
    if constexpr (NORB == 2)                                                                                 \
      if (d->rows_nbt == 8) return launch_rows_t<NORB, NT, 8>(d, a, fuse, P, Q, st);                         \
  }
  EDIGPU_IB_ROWS(256)
  EDIGPU_IB_ROWS(512)
  EDIGPU_IB_ROWS(1024)
#undef EDIGPU_IB_ROWS
  set_error("ib_rows_kernel: no instantiation for this sector");
  return 1;
}

static int launch_ib_rows(const IbDev* d, const IbArgs& a, int fuse, double* P, double* Q, hipStream_t st) {
  switch (d->norb) {
    case 1: return launch_rows_n<1>(d, a, fuse, P, Q, st);
    case 2: return launch_rows_n<2>(d, a, fuse, P, Q, st);
    case 3: return launch_rows_n<3>(d, a, fuse, P, Q, st);
  }
  set_error("ib_rows_kernel: norb");
  return 1;
}

template <int NORB, bool DO_ND, bool ALPHA>
static int launch_cols_t(const IbDev* d, const IbArgs& a, const double* v, double* hv, hipStream_t st, int* nblocks) {
  const size_t lds = d->cols_lds;
  const void* k = (const void*)ib_cols_kernel<NORB, DO_ND, ALPHA>;
  if (ensure_dynamic_lds(k, lds)) return 1;
  const int per_cu = resident_blocks(k, kColsNT, lds);
  if (per_cu < 1) {
    set_error("ib_cols_kernel: does not fit a CU");
    return 1;
  }
  int64_t grid = (int64_t)per_cu * device_cu_count();
  const int64_t tasks = (int64_t)d->npanels * d->nchunks;
  grid = std::min<int64_t>(grid, (tasks + 7) / 8 * 8);
  grid = std::max<int64_t>(8, grid / 8 * 8);
  if (ALPHA && 3 * grid > kMaxPartials) {
    set_error("ib_cols_kernel: partial buffer too small");
    return 1;
  }
  hipLaunchKernelGGL((ib_cols_kernel<NORB, DO_ND, ALPHA>), dim3((unsigned)grid), dim3(kColsNT), lds, st, a, v, hv);
  EDIGPU_HIP(hipGetLastError());
  if (nblocks) *nblocks = (int)grid;
  return 0;
}

static int launch_ib_cols(const IbDev* d, const IbArgs& a, bool alpha, const double* v, double* hv, hipStream_t st, int* nblocks) {
  const bool nd = d->nterms > 0;
#define EDIGPU_IB_COLS(NORB)                                                                     \
  case NORB:                                                                                     \
    if (nd) return alpha ? launch_cols_t<NORB, true, true>(d, a, v, hv, st, nblocks)             \
                         : launch_cols_t<NORB, true, false>(d, a, v, hv, st, nblocks);           \
    return alpha ? launch_cols_t<NORB, false, true>(d, a, v, hv, st, nblocks)                    \
                 : launch_cols_t<NORB, false, false>(d, a, v, hv, st, nblocks);
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
  IbArgs a;
  fill_ib_args(s->ib, a);
  if (launch_ib_rows(s->ib, a, 0, const_cast<double*>(v), hv, st)) return 1;
  return launch_ib_cols(s->ib, a, false, v, hv, st, nullptr);
}

// one fused Lanczos step (launch_normal_lanczos, kernels_normal.hip, explains the protocol)
int launch_ib_lanczos(const edigpu_sector* s, double* P, double* Q, const double* scal, double* partial, int64_t partial_cap,
                      bool first, bool lazy_axpy, hipStream_t st, int* npartial) {
  IbArgs a;
  fill_ib_args(s->ib, a);
  a.scal = scal;
  a.partial = partial;
  a.lazy = lazy_axpy ? 1 : 0;
  (void)partial_cap;  // >= kMaxPartials (ensure_workspace); launch_cols_t checks its grid against that
  if (launch_ib_rows(s->ib, a, first ? 0 : 1, P, Q, st)) return 1;
  return launch_ib_cols(s->ib, a, true, P, Q, st, npartial);
}

}  // namespace edigpu
