// edigpu_internal.hpp -- internal types of the gfx950 H*v engine (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/edigpu.h"
#include "host_build.hpp"

namespace edigpu {

void set_error(const std::string& msg);

#define EDIGPU_HIP(call)                                                                     \
  do {                                                                                       \
    hipError_t _e = (call);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      ::edigpu::set_error(std::string(#call) + " failed: " + hipGetErrorString(_e) + " (" +  \
                          __FILE__ + ":" + std::to_string(__LINE__) + ")");                  \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

// device CSR with 32-bit or 64-bit row pointers (64-bit only when nnz >= 2^31)
struct DevCsr {
  int64_t nrow = 0, nnz = 0;
  int wide = 0;             // 1: rowptr64 used
  int32_t* rowptr32 = nullptr;
  int64_t* rowptr64 = nullptr;
  int32_t* col = nullptr;
  double* val = nullptr;    // nnz (real) or 2*nnz (complex)
  double avg_row = 0.0;
  // SELL-64 image (sliced ELL, 64 rows per slice, entries sorted by column inside a row):
  // entry (slice s, slot k, lane l) at sell_ptr[s]*64 + k*64 + l; padding = own row, value 0
  int sell = 0;
  int64_t nslice = 0;
  int32_t* sell_ptr = nullptr;   // nslice+1, in units of 64 entries
  int32_t* sell_col = nullptr;
  double* sell_val = nullptr;
  // value-dictionary form of the SELL image: 4 B/entry = column (24 bit) | id (8 bit), values in a
  // <= 256-entry table (id 0 = 0.0), diagonal kept apart (one value per row, loc block only)
  int sell_packed = 0;
  uint32_t* sell_pk = nullptr;
  double* sell_dict = nullptr;
  double* sell_diag = nullptr;
};

// ed_total_ud = F ("orbs") sector: kernel argument block (kernels_orbs.hip); all pointers are device memory
constexpr int kOrbsMaxAxes = 2 * EDIGPU_MAXORB;
struct OrbsArgs {
  int naxes = 0;
  int64_t dim = 0;
  int64_t dims[kOrbsMaxAxes] = {0};
  int64_t stride[kOrbsMaxAxes] = {0};
  int off[kOrbsMaxAxes] = {0};        // offset of axis k inside eax / impbit
  int width[kOrbsMaxAxes] = {0};      // ELL width of factor k
  int64_t elloff[kOrbsMaxAxes] = {0};  // offset of factor k inside ell_col / ell_val ([slot][row])
  const int32_t* ell_col = nullptr;
  const double* ell_val = nullptr;
  const double* hd = nullptr;         // explicit diagonal (hand-over) or null: factored tables below
  const double* eax = nullptr;
  const uint8_t* impbit = nullptr;
  const double* xtab = nullptr;
};

// ELL (column-major [slot][row]) image of a small square factor matrix
struct DevEll {
  int64_t nrow = 0, pitch = 0;
  int width = 0;
  int typed = 0;            // packed image with one hop amplitude per slot (coef[k])
  // packed image (preferred): 24-bit column | 7-bit coefficient id | sign; coef[128] table
  uint32_t* pk = nullptr;
  double* coef = nullptr;
  // plain image (fallback when > 127 distinct |values| or nrow >= 2^24)
  int32_t* col = nullptr;   // width*pitch, padding: col=row, val=0
  double* val = nullptr;
};

// Impurity-block image of a whole normal-mode sector on the device (host_ib.hpp; kernels_ib.hip).  Vectors of the
// device-resident Lanczos loops live in the padded panel layout: element (idw, iup) at
// (pos[iup] / 16) * ps + idw * 16 + pos[iup] % 16, ps = dim_dw * 16, len = npanels * ps doubles, padding = zeros.
struct IbDevHalf {  // rows kernel's tables of one half of a split row (host_ib.hpp IbUpHalf)
  int panel0 = 0, npanels = 0, nlist = 0, rimg_len = 0;
  int ucls[5] = {0, 0, 0, 0, 0}, rcb[5] = {0, 0, 0, 0, 0}, rcs[5] = {0, 0, 0, 0, 0};
  uint16_t *ublist = nullptr, *utop = nullptr;
  uint32_t* rmap2 = nullptr;
};

// local-block tables on the same vector layout (host_sb.hpp, kernels_sb.hip; round 4): when present, the product and
// the fused Lanczos step of the impurity-block image run on these
struct DevSb {
  int nb0 = 0, nloc = 0, amode = 0, nbw_up = 0, nbw_dw = 0;
  int rows_nt = 0, rows_nbt = 0, rimg_len = 0, rcs = 0;
  size_t rows_lds = 0, cols_lds = 0;
  uint16_t *urank = nullptr, *ublist = nullptr;
  double* ebw = nullptr;
  int32_t* uslot = nullptr;
  uint32_t* rmap2 = nullptr;
  double *up_vtab = nullptr, *up_tloc = nullptr, *e0 = nullptr;
  uint32_t* up_korb = nullptr;
  int lowbits = 0, nchunks = 0, max_chunk_rows = 0, max_chunk_slots = 0, cols_gs = 8;
  int32_t *chunk_row = nullptr, *chunk_slot = nullptr, *cdesc_off = nullptr;
  uint8_t* cdesc = nullptr;
  double *dw_vtab = nullptr, *dw_tloc = nullptr;
  uint32_t* dw_korb = nullptr;
  uint32_t* nd_dw = nullptr;
  // rows staged in halves (host_sb.hpp SbUpHalf): nhalf = 2; urank then ranks the LOW words, ublist / uslot / rmap2 / ebw of the
  // whole row are absent and rows_lds, rimg_len, rows_nt / rows_nbt serve both halves; the columns kernel of such a sector is
  // the impurity-block one
  int nhalf = 1;
  struct Half {
    int panel0 = 0, npanels = 0;
    uint32_t* ublist32 = nullptr;  // low word | skip | partner position over the top level << 16
    uint8_t* ugap = nullptr;
    int32_t* uslot = nullptr;
    uint32_t* rmap2 = nullptr;
    double* ebw = nullptr;
  } half[2];
};

struct IbDev {
  DevSb* sb = nullptr;
  int nhalf = 1;                 // 2: rows longer than the LDS, staged one half (value of the top bath bit) at a time
  IbDevHalf half[2];
  uint16_t* urank_low = nullptr; // [2^(nb_up - 1)]
  double top_eps = 0.0;          // energy of the top bath level of the up species
  int norb = 0, nb_up = 0, nb_dw = 0, npanels = 0, plen = 0, nlist = 0;
  int ucls[5] = {0, 0, 0, 0, 0};
  int lowbits = 0, nchunks = 0, max_chunk_rows = 0, max_chunk_blocks = 0, nterms = 0, nsub = 1;
  int64_t dim_up = 0, dim_dw = 0, ps = 0, len = 0;
  int rows_nt = 0, rows_nbt = 0;      // rows kernel: threads per workgroup, blocks per thread
  size_t rows_lds = 0, cols_lds = 0;  // dynamic LDS of the two kernels
  uint16_t *urank = nullptr, *ublist = nullptr;  // rows kernel: rank of a bath word inside its class; block list
  uint32_t* rmap2 = nullptr;                     // [plen / 2]: image words of a pair of adjacent positions (lo | hi << 16)
  int rcb[5] = {0, 0, 0, 0, 0}, rcs[5] = {0, 0, 0, 0, 0}, rimg_len = 0;
  double *up_vtab = nullptr, *up_timp = nullptr, *up_ebath = nullptr, *xu = nullptr, *ed = nullptr;
  uint8_t* impd = nullptr;
  int32_t *pos = nullptr, *colof = nullptr;  // column -> position, position -> column (-1: padding)
  int32_t *chunk_row = nullptr, *chunk_blk = nullptr, *dcls = nullptr;
  uint16_t *dblist = nullptr, *dmeta = nullptr;
  double *dw_vtab = nullptr, *dw_timp = nullptr, *ndcoef = nullptr;
  uint8_t *nd_dw = nullptr, *nd_up = nullptr;
  // short rows: the generic LDS row kernel in position order (kernels_normal.hip launch_normal_rows_pos) takes the rows half of
  // the product and of the fused step; Hup as ELL over positions, the diagonal table [impurity pattern of the down word][plen]
  struct PosRows {
    bool on = false;
    DevEll ell;
    double* eux = nullptr;
    int td = 0;  // rows per workgroup
  } pr;
  // bath-bath hops (host_ib.hpp IbSide::pmask, pt): replica / general baths
  int up_np = 0, dw_np = 0;
  uint32_t *up_pmask = nullptr, *dw_pmask = nullptr;
  double *up_pt = nullptr, *dw_pt = nullptr;
};

}  // namespace edigpu

struct edigpu_sector {
  int kind = 0;        // 0 normal (Kronecker), 1 flat CSR, 2 direct (on-the-fly superc/nonsu2), 3 orbs (ed_total_ud=F)
  int is_complex = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t dim = 0, nloc = 0, row_first = 0;
  // ---- normal ----
  int64_t dim_up = 0, dim_dw = 0, dw_first = 0, dw_count = 0;
  double* d_hd = nullptr;
  edigpu::DevEll up_ell;
  edigpu::DevCsr dw;          // DimDw rows
  int dw_maxrow = 0;
  edigpu::DevCsr nd;          // local rows, global columns
  int has_nd = 0;
  edigpu::HostCsr h_up, h_dw; // kept for export (tiny)
  int rows_per_block = 1;     // TD of the LDS row-block kernel (0: generic kernel)
  // factored diagonal / non-local block (library-built sectors; see HostFactored)
  int factored = 0;
  int fac_nimp = 0, fac_nterms = 0;
  double* d_eux = nullptr;      // nimp * dim_up
  double* d_ed = nullptr;       // dim_dw
  uint8_t* d_impd = nullptr;    // dim_dw
  double* d_ndcoef = nullptr;   // nterms
  uint32_t* d_jup = nullptr;    // nterms * dim_up
  // kind 4 (_CMPLX_NORMAL): H = S + iA as two real normal-mode handles + planar work vectors (6 * dim doubles)
  edigpu_sector* sub_s = nullptr;
  edigpu_sector* sub_a = nullptr;
  // kind 4, preferred: the same operator as ONE real sector on the doubled up index 2 iup + (re | im)
  // (host_build.cpp: build_normal_doubled); the interleaved complex vectors are its real vectors
  edigpu_sector* sub_d = nullptr;
  double* d_cz = nullptr;
  int panel_mode = 0;           // panel sweep variant (NormalArgs::panel_mode), fixed at set-up
  int tile_nchunks = 0, tile_rows = 0;
  int32_t* d_tile_chunks = nullptr;
  int32_t* d_tile_lbeg = nullptr;
  int tile_list_cap = 0;
  int4* d_tl_meta = nullptr;    // tile form of the panel sweep: per-row lists split by chunk membership
  int32_t* d_tl_col = nullptr;
  double* d_tl_val = nullptr;
  int tl_has_nd = 0;
  // panel-major vector layout of the fused Lanczos loop (NormalArgs::blk_shift): chosen at set-up for large factored
  // whole sectors, 0 = not available; lz_blocked: the current recurrence runs on it
  int blk_shift = 0;
  int64_t blk_ps = 0, blk_len = 0;
  int4* d_bl_meta = nullptr;
  int blk_rows = 0;             // rows of an LDS block of the blocked sweep
  int blk_list_cap = 0;
  int32_t* d_bl_lend = nullptr;
  uint32_t* d_bl_ent = nullptr;
  double* d_bl_wtab = nullptr;
  bool lz_blocked = false;
  int64_t lz_len = 0;           // doubles per vector of the current recurrence (blk_len, ib->len or ws_len)
  // impurity-block image (large factored whole sectors built from a model; takes precedence over the panel-major image
  // above: lz_blocked then means "the recurrence runs on ib's padded panel layout")
  edigpu::IbDev* ib = nullptr;
  int row_split = 1;  // rows longer than the LDS: number of column parts the row kernel stages them in (SPLIT)
  int col_halo = 0;             // max |partner column - column| over the factored Hnd terms (transposed exchange)
  uint32_t* d_jdw = nullptr;    // nterms * dim_dw
  int32_t* d_mx_rowptr = nullptr;  // per local row: Hdw entries + applicable Hnd terms
  int32_t* d_mx_col = nullptr;
  double* d_mx_val = nullptr;
  // phonon branches (normal mode, whole sectors): vectors hold (nph + 1) electronic blocks of dim_el elements
  int nph = 0;
  int64_t dim_el = 0;
  double w0_ph = 0.0, a_ph = 0.0;
  double* d_gu = nullptr;       // [dim_up]  sum_a g_aa n_a,up
  double* d_gd = nullptr;       // [dim_dw]
  int64_t nd_nnz = 0;           // nnz of the (possibly host-only) CSR image of Hnd
  std::vector<double> h_hd;     // host copies for export when the device holds the factored form;
  edigpu::HostCsr h_nd;         // built on the first export (lazy_export) from the stored model
  bool lazy_export = false;
  edigpu_model model;           // library-built sectors: what edigpu_*_build was given
  int sec_a = 0, sec_b = 0;
  bool jz = false;                 // nonsu2 sector of Jz_basis=T: (Ntot, twoJz) = (sec_a, sec_b)
  bool built_by_library = false;   // made by edigpu_*_build: model, sec_a, sec_b are valid
  bool from_model() const { return built_by_library; }
  // ---- flat ----
  edigpu::DevCsr loc, nonloc; // local rows; loc columns are shard-relative, nonloc global
  // ---- orbs (ed_total_ud = F) ----
  edigpu::OrbsArgs orbs;
  std::vector<edigpu::HostCsr> h_orbs_fac;  // kept for export (tiny)
  std::vector<double> h_orbs_hd;            // explicit diagonal when handed over
  // ---- direct (on-the-fly) ----
  int dir_ns = 0, dir_norb = 0, dir_nterms = 0;
  int32_t* d_dir_states = nullptr;
  int32_t* d_dir_offdw = nullptr;
  int32_t* d_dir_rkup = nullptr;
  edigpu::DirectTerm* d_dir_terms = nullptr;
  uint2* d_dir_tests = nullptr;   // (need_set, need_set | need_clear) per term, padded to a multiple of 4
  double* d_dir_dtab = nullptr;
  double* d_dir_xtab = nullptr;
  // ---- Lanczos workspace (lazily allocated) ----
  int64_t partial_cap = 0;      // doubles in d_partial
  double* d_vin = nullptr;
  double* d_vout = nullptr;
  double* d_tmp = nullptr;
  double* d_partial = nullptr;  // reduction partials
  double* d_scal = nullptr;     // alpha/beta/flags on device
  bool lz_exactbeta = false;    // EDIGPU_LANCZOS_EXACTBETA at the start of the current recurrence
  // launch-bound sectors replay the steps of a recurrence from a captured hipGraph (lanczos_run): the executable and
  // what it was captured for
  hipGraphExec_t lz_graph = nullptr;
  int lz_graph_k = 0, lz_graph_nlanc = 0;
  const void* lz_graph_scal = nullptr;
  bool lz_graph_blocked = false, lz_graph_failed = false;
  size_t scal_cap = 0;          // doubles allocated in d_scal
  unsigned int* d_lzcnt = nullptr;  // arrival counter of the in-kernel finalize (lz_finalize.hpp)
  int64_t ws_len = 0;           // in doubles per vector
};
