// normal_args.hpp -- kernel argument block shared by the normal-mode kernels
// (kernels_normal.hip: row kernel A, kernels_panel.hip: column-panel kernel B).
#pragma once
#include <cstdint>

#include "kernels.hpp"

namespace edigpu {

struct NormalArgs {
  int64_t dim_up, dim_dw, dw_first, dw_count;
  // diagonal: explicit (hd) or factored (eux[impd[g]][iup] + ed[g])
  const double* hd;
  const double* eux;
  const double* ed;
  const uint8_t* impd;
  // Hup as ELL: packed (pk + coef table) or plain (col,val)
  const uint32_t* ell_pk;
  const double* ell_coef;  // 128 entries
  const int32_t* ell_col;
  const double* ell_val;
  int ell_w;
  int ell_typed;  // packed ELL whose slot k holds the hop with amplitude ell_coef[k]
  int64_t split_first, split_count;  // SPLIT row kernel: the columns staged by this launch
  int64_t ell_pitch;
  const int32_t* dw_rowptr;
  const int32_t* dw_col;
  const double* dw_val;
  // Hnd: CSR over the local rows (explicit) or factored terms
  const int32_t* nd_rp32;
  const int64_t* nd_rp64;
  const int32_t* nd_col;
  const double* nd_val;
  int dw_maxrow;  // longest row of Hdw
  int has_nd;
  int nterms;
  const double* nd_coef;
  const uint32_t* jup;  // nterms * dim_up
  const uint32_t* jdw;  // nterms * dim_dw
  // per LOCAL row: Hdw entries + applicable Hnd terms in one list (panel kernel)
  // fused Lanczos step (device scalars, see kernels.hpp SC_*) and per-workgroup alpha partials
  const double* scal;
  double* partial;
  double* xout;  // fused rows kernel (FUSE >= 2): where the new Lanczos vector is written; null: over the previous one (v_local)
  int64_t partial_cap;  // doubles in `partial`; the panel launch checks its grid against it before enqueueing
  // finalize of the fused step inside the sweep (lz_finalize.hpp): arrival counter (zero at launch; nullptr = a separate
  // finalize kernel follows), recurrence length, doubles per vector
  unsigned int* lz_counter;
  int lz_nlanc;
  int64_t lz_len;
  // panel sweep variant chosen when the sector was set up (environment switches are read there):
  // 0 one column per lane, 1 two columns per lane, 2 two columns + LDS-staged row chunks (tile_chunks: nchunks + 1
  // local row offsets, tile_rows: the longest chunk)
  int panel_mode;
  int tile_nchunks, tile_rows;
  const int32_t* tile_chunks;
  const int32_t* tile_lbeg;  // nchunks + 1: first list entry of a chunk's rows (tl_col / tl_val)
  int tile_list_cap;         // list entries of the fullest chunk
  // per local row (tile form): tl_meta = (first entry, hops inside the row's chunk, hops leaving it, Hnd terms);
  // tl_col: staged row index / global row / (partner row | tag << 24); tl_val: weights; both padded by 8 entries
  const int4* tl_meta;
  const int32_t* tl_col;
  const double* tl_val;
  int tl_has_nd;  // the lists carry the factored Hnd terms
  // Panel-major ("blocked") vector layout of the device-resident Lanczos loop: element (idw, iup) lives at
  // (iup >> blk_shift) * blk_ps + (idw << blk_shift) + (iup & (W - 1)), W = 1 << blk_shift columns per panel, blk_ps =
  // DimDw * W; the last panel is padded with zeros.  blk_shift = 0: natural layout (idw * DimUp + iup).
  int blk_shift;
  int64_t blk_ps;
  // lists of the blocked sweep (normal_dw_blk_kernel): per row bl_meta = (first entry, hops inside the row's LDS block,
  // hops leaving it -- both padded to x4 --, Hnd terms); entry = row (16 bit: index inside the block / global row) |
  // weight index << 16 | (Hnd term + 1) << 24; bl_wtab: 256 weights
  const int4* bl_meta;
  const uint32_t* bl_ent;
  int blk_rows;          // rows of an LDS block of the blocked sweep
  int blk_list_cap;      // list entries of the fullest block (multiple of 4)
  const int32_t* bl_lend;  // per block: one past its last list entry
  const double* bl_wtab;
  const int32_t* mx_rowptr;  // merged list: ptr[dw_count+1]
  const int32_t* mx_col;     // partner row (24 bit) | tag << 24
  const double* mx_val;
};

// B: (Hdw (x) 1) + factored Hnd as an L2-blocked column-panel sweep (kernels_panel.hip).
// alpha: also write the per-workgroup partials of <v|hv> and <hv|hv> (fused Lanczos step).
int launch_dw_panels(const NormalArgs& a, bool do_dw, bool do_nd, const double* v_full, double* hv,
                     hipStream_t st, bool alpha = false, int* nblocks = nullptr);

// the same sweep on vectors in the panel-major layout (a.blk_shift > 0)
int launch_dw_blocked(const NormalArgs& a, bool do_nd, const double* v, double* hv, hipStream_t st, bool alpha,
                      int* nblocks);

int launch_dw_panel_cols(const NormalArgs& a, bool do_nd, int64_t col_first, int64_t ncol, int64_t stride, int halo,
                         const double* w, double* hv, hipStream_t st);

}  // namespace edigpu
