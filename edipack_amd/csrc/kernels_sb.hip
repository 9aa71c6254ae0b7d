// kernels_sb.hip -- launchers of the local-block kernels (kernels_sb_impl.hpp; one translation unit per orbital count in
// kernels_sb1/2/3.hip) and the choice of their geometry.
#include "kernels_sb_impl.hpp"

namespace edigpu {

int sb_rows_1(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st);
int sb_rows_2(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st);
int sb_rows_3(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st);
int sb_cols_1(const IbDev* d, const SbArgs& a, int mode, const double* v, double* hv, hipStream_t st, int* nblocks);
int sb_cols_2(const IbDev* d, const SbArgs& a, int mode, const double* v, double* hv, hipStream_t st, int* nblocks);
int sb_cols_3(const IbDev* d, const SbArgs& a, int mode, const double* v, double* hv, hipStream_t st, int* nblocks);

// low bath levels folded into the blocks, per orbital count: the kernels are built for 5 local levels
int sb_nb0(int norb) { return norb >= 1 && norb <= 3 ? 5 - norb : 0; }

// columns kernel: blocks per wave-slot (8: two columns per lane, 256 threads; 4: one column per lane, 512 threads) and waves
// per workgroup.  EDIGPU_SB_CW = 1 / 2 chooses (tuning)
int sb_cols_gs() {
  const char* e = getenv("EDIGPU_SB_CW");  // (read per sector set-up)
  return e && atoi(e) == 1 ? 4 : 8;
}
int sb_cols_waves() { return sb_cols_gs() == 4 ? 8 : SB_COLS_NT2 / 64; }

size_t sb_rows_lds(int nbw, int rimg_len, bool top) { return sb_rows_lds_bytes(nbw, rimg_len, top); }

size_t sb_cols_lds(int nbw, int nloc, int max_chunk_rows, int max_chunk_slots, int gs) {
  return sb_cols_layout(nbw, nloc, max_chunk_rows, max_chunk_slots, gs).total;
}

// threads per workgroup / blocks per thread of the rows kernel for `slots` wave-slots, rows of plen columns (padded) and a
// row image of class stride cs; false: no instantiation fits (the caller keeps the impurity-block kernels).
// EDIGPU_SB_NT / EDIGPU_SB_NBT force one.
bool sb_rows_config(int norb, int slots, int plen, int cs, int* nt_out, int* nbt_out) {
  const int nloc = norb + sb_nb0(norb);
  const int maxm = sb::binom(nloc, nloc / 2);
  int fnt = 0, fnbt = 0;
  if (const char* e = getenv("EDIGPU_SB_NT")) fnt = atoi(e);
  if (const char* e = getenv("EDIGPU_SB_NBT")) fnbt = atoi(e);
#define EDIGPU_SB_ONE(NT, NBT, CS)                                                                              \
  if (cs == CS && (!fnt || fnt == NT) && (!fnbt || fnbt == NBT) && slots <= (NT / 64) * NBT &&                    \
      plen / 2 <= NT * sb_rows_nld(NBT, maxm)) {                                                                \
    *nt_out = NT;                                                                                               \
    *nbt_out = NBT;                                                                                             \
    return true;                                                                                                \
  }
  EDIGPU_SB_ROWS_GEOMETRIES(EDIGPU_SB_ONE)
#undef EDIGPU_SB_ONE
  return false;
}

static void fill_sb_args(const IbDev* d, SbArgs& a) {
  const DevSb* s = d->sb;
  a.npanels = d->npanels;
  a.plen = d->plen;
  a.nterms = d->nterms;
  a.dim_dw = d->dim_dw;
  a.ps = d->ps;
  a.nbw_up = s->nbw_up;
  a.rimg_len = s->rimg_len;
  a.urank = s->urank;
  a.ublist = s->ublist;
  a.uslot = s->uslot;
  a.ebw = s->ebw;
  a.rmap2 = s->rmap2;
  a.up_vtab = s->up_vtab;
  a.up_tloc = s->up_tloc;
  a.e0 = s->e0;
  a.xu = d->xu;
  a.ed = d->ed;
  a.up_korb = s->up_korb;
  a.impd = d->impd;
  a.nbw_dw = s->nbw_dw;
  a.lowbits = s->lowbits;
  a.nchunks = s->nchunks;
  a.max_chunk_rows = s->max_chunk_rows;
  a.max_chunk_slots = s->max_chunk_slots;
  a.chunk_row = s->chunk_row;
  a.chunk_slot = s->chunk_slot;
  a.cdesc_off = s->cdesc_off;
  a.cdesc = s->cdesc;
  a.dw_vtab = s->dw_vtab;
  a.dw_tloc = s->dw_tloc;
  a.ndcoef = d->ndcoef;
  a.dw_korb = s->dw_korb;
  a.nd_dw = s->nd_dw;
  a.nd_up = d->nd_up;
  a.ublist32 = nullptr;
  a.ugap = nullptr;
  a.pfull = nullptr;
  a.row0 = 0;
  a.p0 = 0;
  a.qmagic = 0;
  a.kslot = 0;
  a.q16 = d->ps;
  a.scal = nullptr;
  a.partial = nullptr;
  a.lazy = 0;
  a.pold = nullptr;
  a.dbg = nullptr;
}

static int rows(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st) {
  switch (d->norb) {
    case 1: return sb_rows_1(d, a, fuse, P, Q, X, st);
    case 2: return sb_rows_2(d, a, fuse, P, Q, X, st);
    case 3: return sb_rows_3(d, a, fuse, P, Q, X, st);
  }
  set_error("sb_rows_kernel: norb");
  return 1;
}

static int cols(const IbDev* d, const SbArgs& a, int mode, const double* v, double* hv, hipStream_t st, int* nblocks) {
  switch (d->norb) {
    case 1: return sb_cols_1(d, a, mode, v, hv, st, nblocks);
    case 2: return sb_cols_2(d, a, mode, v, hv, st, nblocks);
    case 3: return sb_cols_3(d, a, mode, v, hv, st, nblocks);
  }
  set_error("sb_cols_kernel: norb");
  return 1;
}

// rows staged in halves: (Hd + 1 (x) Hup) v for the columns of half h (the other half's part of the result is untouched)
int launch_sb_rows_half(const edigpu_sector* s, int h, const double* v, double* hv, const double* scal, hipStream_t st) {
  const IbDev* d = s->ib;
  const DevSb::Half& hf = d->sb->half[h];
  SbArgs a;
  fill_sb_args(d, a);
  a.nbw_up = d->sb->nbw_up - 1;
  a.plen = hf.npanels * kIbPanel;
  a.uslot = hf.uslot;
  a.rmap2 = hf.rmap2;
  a.ebw = hf.ebw;
  a.ublist32 = hf.ublist32;
  a.ugap = hf.ugap;
  a.pfull = v;
  a.scal = scal;
  const int64_t off = (int64_t)hf.panel0 * d->ps;
  return rows(d, a, 2 + h, v + off, hv + off, nullptr, st);
}

// plain product on vectors in the padded panel layout
int launch_sb(const edigpu_sector* s, const double* v, double* hv, hipStream_t st) {
  SbArgs a;
  fill_sb_args(s->ib, a);
  static const bool stamp = getenv("EDIGPU_SB_STAMP") != nullptr;
  if (stamp) {  // measurement aid: cycles per phase of the rows kernel, workgroup 0, printed per launch
    static long long* dbg = nullptr;
    if (!dbg) EDIGPU_HIP(hipMalloc((void**)&dbg, 32 * 8 * sizeof(long long)));
    EDIGPU_HIP(hipMemsetAsync(dbg, 0, 32 * 8 * sizeof(long long), st));
    a.dbg = dbg;
    if (rows(s->ib, a, 0, v, hv, nullptr, st)) return 1;
    long long h[32 * 8];
    EDIGPU_HIP(hipMemcpyAsync(h, dbg, sizeof(h), hipMemcpyDeviceToHost, st));
    EDIGPU_HIP(hipStreamSynchronize(st));
    for (int w = 0; w < s->ib->sb->rows_nt / 64; w++)
      fprintf(stderr, "sb_rows stamps wave %2d: compute %9lld bar1 %9lld writeback %9lld bar2 %9lld out+land %9lld bar3 %9lld\n", w, h[w * 8], h[w * 8 + 1],
              h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5]);
    a.dbg = nullptr;
    return cols(s->ib, a, 0, v, hv, st, nullptr);
  }
  // short rows: the generic LDS row kernel in position order takes the rows half (IbDev::pr)
  if (s->ib->pr.on ? launch_normal_rows_pos(s, v, hv, nullptr, 1, nullptr, st) : rows(s->ib, a, 0, v, hv, nullptr, st)) return 1;
  return cols(s->ib, a, 0, v, hv, st, nullptr);
}

// One fused Lanczos step on THREE buffers (launch_ib_lanczos, kernels_ib.hip, has the protocol): here the rows kernel also
// subtracts beta * P_old, so the columns kernel reads two vectors, not three.
int launch_sb_lanczos(const edigpu_sector* s, const double* P, double* Q, double* X, const double* scal, double* partial,
                      int64_t partial_cap, bool first, bool lazy_axpy, hipStream_t st, int* npartial) {
  SbArgs a;
  fill_sb_args(s->ib, a);
  a.scal = scal;
  a.partial = partial;
  a.lazy = lazy_axpy ? 1 : 0;
  (void)partial_cap;  // >= kMaxPartials (ensure_workspace); sb_launch_cols_t checks its grid against that
  if (s->ib->pr.on) {
    // short rows: the generic LDS row kernel in position order with its fused forms -- x = (Q - alpha P) / beta written OVER
    // P, Q <- (Hd + 1 (x) Hup) x - beta P_old -- then the columns kernel with the three sums.  Two buffers, not three (X is
    // not used and launch_normal_lanczos tells the caller so): with a third vector config 2's working set (3 x 98 MB)
    // leaves the 256 MB Infinity Cache and the fused row kernel takes 122 us instead of ~80.
    if (launch_normal_rows_pos(s, P, Q, nullptr, first ? 101 : (lazy_axpy ? 103 : 102), scal, st)) return 1;
    return cols(s->ib, a, 1, P, Q, st, npartial);
  }
  if (first) {
    if (rows(s->ib, a, 0, P, Q, nullptr, st)) return 1;
    return cols(s->ib, a, 1, P, Q, st, npartial);
  }
  // The fused rows kernel (x formed while the row is staged, - beta P_old subtracted when the result leaves: three more
  // streams of pieces) fits its registers in the 256- and 512-thread geometries.  In the 768-thread geometry of the longest
  // rows it spills (2.4 ms per launch at Ns = 16 against 0.83 plain): there the new vector is its own pass, the plain rows
  // kernel follows, and the columns kernel joins - beta P_old to the rows kernel's part.  EDIGPU_SB_STEP=2 forces that form.
  const char* es = getenv("EDIGPU_SB_STEP");
  if (s->ib->sb->rows_nt == 768 || (es && atoi(es) == 2)) {
    if (lz_next_vector(P, Q, X, s->ib->len, scal, lazy_axpy, st) || rows(s->ib, a, 0, X, Q, nullptr, st)) return 1;
    a.pold = P;
    return cols(s->ib, a, 1, X, Q, st, npartial);
  }
  if (rows(s->ib, a, 1, P, Q, X, st)) return 1;
  return cols(s->ib, a, 1, X, Q, st, npartial);
}

// ---------------------------------------------------------------------------------------------------------
// row shards (edigpu_shard.hip): the two halves of the product on the rank's part of the padded panel layout
// ---------------------------------------------------------------------------------------------------------
// Shard form of the layout: P = world * npmax panels (npmax = ceil(npanels / world): rank d owns the panels [d npmax,
// (d + 1) npmax) for the column half), every panel q = ceil(DimDw / world) rows of 16 doubles -- the rank's own rows;
// rows past its count and panels past npanels hold zeros.  What rank d needs from rank r for the column half, r's q rows
// of d's npmax panels, is ONE contiguous run of npmax * q * 16 doubles: the all-to-all needs no packing, and what it
// delivers -- for every source rank its rows of my panels -- is what sb_cols_kernel<SH> reads through SbArgs::kslot.
// The way back is the same exchange and lands in this layout again.
__global__ void __launch_bounds__(256) k_shard_to_panels(const double* __restrict__ src, double* __restrict__ dst, const int32_t* __restrict__ colof,
                                                         int64_t dim_up, int64_t count, int64_t q, int npanels, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t p = e / (q * 16), rem = e - p * q * 16, i = rem >> 4;
    const int c = p < npanels ? colof[p * 16 + (rem & 15)] : -1;
    dst[e] = (c >= 0 && i < count) ? src[i * dim_up + c] : 0.0;
  }
}

// dst (the rank's rows in the reference's layout) = a + b, both in the shard form
__global__ void __launch_bounds__(256) k_shard_from_panels_add(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ dst,
                                                               const int32_t* __restrict__ pos, int64_t dim_up, int64_t q, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / dim_up, c = e - i * dim_up;
    const int p = pos[c];
    const int64_t at = ((int64_t)(p >> 4) * q + i) * 16 + (p & 15);
    dst[e] = a[at] + b[at];
  }
}

bool sb_shardable(const edigpu_sector* s) { return s->kind == 0 && s->ib && s->ib->sb && s->ib->nhalf == 1; }

int sb_shard_panels(const edigpu_sector* s, int world) { return (s->ib->npanels + world - 1) / world; }

int sb_shard_to_panels(const edigpu_sector* s, const double* rows, double* dst, int64_t count, int64_t q, int world, hipStream_t st) {
  const int64_t n = (int64_t)world * sb_shard_panels(s, world) * q * 16;
  const unsigned g = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
  hipLaunchKernelGGL(k_shard_to_panels, dim3(std::max(g, 1u)), dim3(256), 0, st, rows, dst, s->ib->colof, s->ib->dim_up, count, q, s->ib->npanels, n);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int sb_shard_from_panels_add(const edigpu_sector* s, const double* a, const double* b, double* rows, int64_t count, int64_t q, hipStream_t st) {
  const int64_t n = count * s->ib->dim_up;
  if (n == 0) return 0;
  const unsigned g = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
  hipLaunchKernelGGL(k_shard_from_panels_add, dim3(g), dim3(256), 0, st, a, b, rows, s->ib->pos, s->ib->dim_up, q, n);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// hv = (Hd + 1 (x) Hup) v on the rank's rows [row0, row0 + count), both in the shard form
int launch_sb_rows_shard(const edigpu_sector* s, int64_t row0, int64_t count, int64_t q, const double* v, double* hv, hipStream_t st) {
  if (count == 0) return 0;
  SbArgs a;
  fill_sb_args(s->ib, a);
  a.dim_dw = count;
  a.ps = q * 16;
  a.row0 = row0;
  return rows(s->ib, a, 0, v, hv, nullptr, st);
}

// out = (Hdw (x) 1 + Hnd) v on the panels [p0, p0 + np) this rank owns, all rows; v, out: what the all-to-all delivers /
// returns (every rank's rows in its slot of npmax * q * 16 doubles)
int launch_sb_cols_shard(const edigpu_sector* s, int p0, int np, int64_t q, int npmax, const double* v, double* out, hipStream_t st) {
  if (np <= 0) return 0;
  if (q < 1 || q > 0xFFFF) {
    set_error("launch_sb_cols_shard: rows per rank");
    return 1;
  }
  SbArgs a;
  fill_sb_args(s->ib, a);
  a.npanels = np;
  a.p0 = p0;
  a.q16 = q * 16;
  a.kslot = (int64_t)npmax * q * 16 - q * 16;
  a.qmagic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)q + 1);
  return cols(s->ib, a, 2, v, out, st, nullptr);
}

}  // namespace edigpu
