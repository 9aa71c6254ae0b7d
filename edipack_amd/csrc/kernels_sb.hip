// kernels_sb.hip -- launchers of the local-block kernels (kernels_sb_impl.hpp; one translation unit per orbital count in
// kernels_sb1/2/3.hip) and the choice of their geometry.
#include "kernels_sb_impl.hpp"

namespace edigpu {

int sb_rows_1(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st);
int sb_rows_2(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st);
int sb_rows_3(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st);
int sb_cols_1(const IbDev* d, const SbArgs& a, bool alpha, const double* v, double* hv, hipStream_t st, int* nblocks);
int sb_cols_2(const IbDev* d, const SbArgs& a, bool alpha, const double* v, double* hv, hipStream_t st, int* nblocks);
int sb_cols_3(const IbDev* d, const SbArgs& a, bool alpha, const double* v, double* hv, hipStream_t st, int* nblocks);

// low bath levels folded into the blocks, per orbital count: the kernels are built for 5 local levels
int sb_nb0(int norb) { return norb >= 1 && norb <= 3 ? 5 - norb : 0; }

// columns kernel: blocks per wave-slot (8: two columns per lane, 256 threads; 4: one column per lane, 512 threads) and waves
// per workgroup.  EDIGPU_SB_CW = 1 / 2 chooses (tuning)
int sb_cols_gs() {
  const char* e = getenv("EDIGPU_SB_CW");  // (read per sector set-up)
  return e && atoi(e) == 1 ? 4 : 8;
}
int sb_cols_waves() { return sb_cols_gs() == 4 ? 8 : 4; }

size_t sb_rows_lds(int nbw, int rimg_len) { return sb_rows_lds_bytes(nbw, rimg_len); }

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
  a.scal = nullptr;
  a.partial = nullptr;
  a.lazy = 0;
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

static int cols(const IbDev* d, const SbArgs& a, bool alpha, const double* v, double* hv, hipStream_t st, int* nblocks) {
  switch (d->norb) {
    case 1: return sb_cols_1(d, a, alpha, v, hv, st, nblocks);
    case 2: return sb_cols_2(d, a, alpha, v, hv, st, nblocks);
    case 3: return sb_cols_3(d, a, alpha, v, hv, st, nblocks);
  }
  set_error("sb_cols_kernel: norb");
  return 1;
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
    return cols(s->ib, a, false, v, hv, st, nullptr);
  }
  if (rows(s->ib, a, 0, v, hv, nullptr, st)) return 1;
  return cols(s->ib, a, false, v, hv, st, nullptr);
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
  if (first) {
    if (rows(s->ib, a, 0, P, Q, nullptr, st)) return 1;
    return cols(s->ib, a, true, P, Q, st, npartial);
  }
  if (rows(s->ib, a, 1, P, Q, X, st)) return 1;
  return cols(s->ib, a, true, X, Q, st, npartial);
}

}  // namespace edigpu
