// kernels_normal.hip -- normal-mode (Kronecker) H*v for gfx950.
//
//   Hv = Hd o v + (1 (x) Hup) v + (Hdw (x) 1) v + Hnd v
//
// takes the place of spMatVec_normal_main (reference ED_NORMAL/
// ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650) and, in its two-phase form, of
// spMatVec_mpi_normal_main (:765-929).  The state vector is viewed as the matrix
// V[idw][iup] (iup contiguous, ED_SECTOR.f90:1705-1717).
//
// Mapping to the hardware (one workgroup = TD consecutive idw rows, all iup):
//  * the TD rows of V are staged once into LDS (coalesced HBM read); the Hup term is
//    a gather inside a row and is served from LDS, not from L1/L2;
//  * Hup is held as ELL, column-major, so that the 64 lanes of a wave read
//    consecutive slots -- coalesced, L2-resident, shared by the TD rows; a slot is
//    one packed 32-bit word (24-bit column, 7-bit coefficient id, sign) when the
//    matrix has <= 127 distinct |values| (always true for bath hybridisations);
//  * the Hdw term reads whole neighbour rows V[jdw][:], contiguous along iup:
//    coalesced; the neighbour list of each row is broadcast from LDS;
//  * every thread owns E columns per pass so that E independent loads are in flight
//    per neighbour row / ELL slot (memory-level parallelism instead of occupancy);
//  * Hd is streamed, Hnd (0.3 nnz/row) is a short per-row CSR gather.
// No MFMA: ~0.25 flop/byte, HBM/L2 bound.
#include <cstdlib>
#include <string>

#include "kernels.hpp"

namespace edigpu {

struct NormalArgs {
  int64_t dim_up, dw_first, dw_count;
  const double* hd;
  // Hup as ELL: packed (pk + coef table) or plain (col,val)
  const uint32_t* ell_pk;
  const double* ell_coef;  // 128 entries
  const int32_t* ell_col;
  const double* ell_val;
  int ell_w;
  int64_t ell_pitch;
  const int32_t* dw_rowptr;
  const int32_t* dw_col;
  const double* dw_val;
  const int32_t* nd_rp32;
  const int64_t* nd_rp64;
  const int32_t* nd_col;
  const double* nd_val;
  int has_nd;
};

constexpr int kNT = 512;
constexpr int kMaxNbr = 64;  // neighbour-list slots per row kept in LDS

template <int TD, int E, bool USE_LDS, bool LOCAL, bool DW, bool ND, bool PACKED>
__global__ void __launch_bounds__(kNT)
    normal_rows_kernel(NormalArgs a, const double* __restrict__ v_local,
                       const double* __restrict__ v_full, double* __restrict__ hv) {
  extern __shared__ double vs[];  // TD*DimUp staged rows (LOCAL && USE_LDS)
  __shared__ int32_t nb_col[TD][kMaxNbr];
  __shared__ double nb_val[TD][kMaxNbr];
  __shared__ int nb_cnt[TD];
  __shared__ double coef_s[128];

  const int64_t DimUp = a.dim_up;
  const int64_t r0 = (int64_t)blockIdx.x * TD;  // first local row of this block
  const int tid = threadIdx.x;
  int nr = TD;
  if (r0 + nr > a.dw_count) nr = (int)(a.dw_count - r0);

  if (LOCAL && PACKED && tid < 128) coef_s[tid] = a.ell_coef[tid];
  if (DW) {
    // neighbour lists of the TD rows -> LDS (rows longer than kMaxNbr keep cnt = -1: slow path)
    for (int r = 0; r < nr; r++) {
      const int64_t g = a.dw_first + r0 + r;
      const int32_t b = a.dw_rowptr[g], n = a.dw_rowptr[g + 1] - b;
      if (n <= kMaxNbr) {
        if (tid < n) {
          nb_col[r][tid] = a.dw_col[b + tid];
          nb_val[r][tid] = a.dw_val[b + tid];
        }
        if (tid == 0) nb_cnt[r] = n;
      } else if (tid == 0) {
        nb_cnt[r] = -1;
      }
    }
  }
  if (LOCAL && USE_LDS) {
    for (int r = 0; r < nr; r++) {
      const double* src = v_local + (r0 + r) * DimUp;
      for (int64_t iup = tid; iup < DimUp; iup += kNT) vs[r * DimUp + iup] = src[iup];
    }
  }
  __syncthreads();

  for (int64_t c0 = 0; c0 < DimUp; c0 += (int64_t)kNT * E) {
    double acc[TD][E];
    int64_t col[E];
    bool ok[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      col[e] = c0 + tid + (int64_t)e * kNT;
      ok[e] = col[e] < DimUp;
      if (!ok[e]) col[e] = DimUp - 1;  // clamp: loads stay in bounds, result discarded
#pragma unroll
      for (int r = 0; r < TD; r++) acc[r][e] = 0.0;
    }

    if (LOCAL) {
      // ---- diagonal ----
#pragma unroll
      for (int r = 0; r < TD; r++)
        if (r < nr) {
#pragma unroll
          for (int e = 0; e < E; e++) {
            const int64_t i = (r0 + r) * DimUp + col[e];
            const double x = USE_LDS ? vs[r * DimUp + col[e]] : v_local[i];
            acc[r][e] = a.hd[i] * x;
          }
        }
      // ---- (1 (x) Hup): gather inside the row ----
      for (int k = 0; k < a.ell_w; k++) {
        int32_t cc[E];
        double ww[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
          if (PACKED) {
            const uint32_t p = a.ell_pk[(int64_t)k * a.ell_pitch + col[e]];
            cc[e] = (int32_t)(p & 0xFFFFFFu);
            const double m = coef_s[(p >> 24) & 0x7Fu];
            ww[e] = (p >> 31) ? -m : m;
          } else {
            cc[e] = a.ell_col[(int64_t)k * a.ell_pitch + col[e]];
            ww[e] = a.ell_val[(int64_t)k * a.ell_pitch + col[e]];
          }
        }
#pragma unroll
        for (int r = 0; r < TD; r++)
          if (r < nr) {
#pragma unroll
            for (int e = 0; e < E; e++) {
              const double x = USE_LDS ? vs[r * DimUp + cc[e]] : v_local[(r0 + r) * DimUp + cc[e]];
              acc[r][e] += ww[e] * x;
            }
          }
      }
    }
    if (DW) {
      // ---- (Hdw (x) 1): whole neighbour rows, contiguous in iup ----
#pragma unroll
      for (int r = 0; r < TD; r++)
        if (r < nr) {
          const int n = nb_cnt[r];
          if (n >= 0) {
#pragma unroll 4
            for (int jj = 0; jj < n; jj++) {
              const double w = nb_val[r][jj];
              const double* row = v_full + (int64_t)nb_col[r][jj] * DimUp;
#pragma unroll
              for (int e = 0; e < E; e++) acc[r][e] += w * row[col[e]];
            }
          } else {
            const int64_t g = a.dw_first + r0 + r;
            for (int32_t jj = a.dw_rowptr[g]; jj < a.dw_rowptr[g + 1]; jj++) {
              const double w = a.dw_val[jj];
              const double* row = v_full + (int64_t)a.dw_col[jj] * DimUp;
#pragma unroll
              for (int e = 0; e < E; e++) acc[r][e] += w * row[col[e]];
            }
          }
        }
    }
    if (ND) {
      // ---- Hnd: short CSR rows with global columns ----
      if (a.has_nd) {
#pragma unroll
        for (int r = 0; r < TD; r++)
          if (r < nr) {
            int64_t b[E], en[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
              const int64_t i = (r0 + r) * DimUp + col[e];
              if (a.nd_rp64) {
                b[e] = a.nd_rp64[i];
                en[e] = a.nd_rp64[i + 1];
              } else {
                b[e] = a.nd_rp32[i];
                en[e] = a.nd_rp32[i + 1];
              }
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
              double s = 0.0;
              for (int64_t jj = b[e]; jj < en[e]; jj++) s += a.nd_val[jj] * v_full[a.nd_col[jj]];
              acc[r][e] += s;
            }
          }
      }
    }
#pragma unroll
    for (int r = 0; r < TD; r++)
      if (r < nr) {
#pragma unroll
        for (int e = 0; e < E; e++)
          if (ok[e]) {
            const int64_t i = (r0 + r) * DimUp + col[e];
            if (LOCAL)
              hv[i] = acc[r][e];
            else
              hv[i] += acc[r][e];
          }
      }
  }
}

// ------------------------------------------------------------------------------------------
// (Hdw (x) 1) as a column-panel sweep:  hv[:, panel] += Hdw * V[:, panel]
//
// The down term touches ~6.5 other rows of V per output row.  V (94 MB for config 2) does not
// fit the 8 x 4 MiB L2s, but one panel of W <= 64 columns over all DimDw rows does (1.7 MB for
// config 2).  All workgroups that share an XCD (blockIdx % 8, the observed round-robin dispatch;
// a different placement only costs speed) sweep the same panel at the same time, so every
// neighbour-row read after the first touch is an L2 hit and HBM sees V exactly once.
// A wave owns one output row at a time (64 lanes = 64 panel columns, one 512-B segment);
// the row's neighbour list is wave-uniform and is fetched with scalar loads.
// ------------------------------------------------------------------------------------------
struct PanelArgs {
  int64_t dim_up, dw_first, dw_count;
  int npanels, width, blocks_per_panel, rows_per_block;
  const int32_t* dw_rowptr;
  const int32_t* dw_col;
  const double* dw_val;
};

constexpr int kPanelNT = 512;

__global__ void __launch_bounds__(kPanelNT, 2)
    normal_dw_panel_kernel(PanelArgs a, const double* __restrict__ v_full, double* __restrict__ hv) {
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int panel = (k / a.blocks_per_panel) * 8 + x;
  if (panel >= a.npanels) return;
  const int chunk = k % a.blocks_per_panel;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t DimUp = a.dim_up;
  const int64_t c = (int64_t)panel * a.width + lane;
  const bool ok = lane < a.width && c < DimUp;
  const int64_t cc = ok ? c : DimUp - 1;
  int64_t rend = (int64_t)(chunk + 1) * a.rows_per_block;
  if (rend > a.dw_count) rend = a.dw_count;
  constexpr int NW = kPanelNT / 64;
  for (int64_t r = (int64_t)chunk * a.rows_per_block + wave; r < rend; r += 2 * NW) {
    // two rows per iteration: independent load streams in flight
    const int64_t r2 = r + NW;
    const bool two = r2 < rend;
    const int64_t g = a.dw_first + r;
    const int32_t b0 = a.dw_rowptr[g], e0 = a.dw_rowptr[g + 1];
    int32_t b1 = 0, e1 = 0;
    if (two) {
      b1 = a.dw_rowptr[g + NW];
      e1 = a.dw_rowptr[g + NW + 1];
    }
    double acc0 = hv[r * DimUp + cc];
    double acc1 = two ? hv[r2 * DimUp + cc] : 0.0;
#pragma unroll 4
    for (int32_t jj = b0; jj < e0; jj++)
      acc0 += a.dw_val[jj] * v_full[(int64_t)a.dw_col[jj] * DimUp + cc];
#pragma unroll 4
    for (int32_t jj = b1; jj < e1; jj++)
      acc1 += a.dw_val[jj] * v_full[(int64_t)a.dw_col[jj] * DimUp + cc];
    if (ok) {
      hv[r * DimUp + c] = acc0;
      if (two) hv[r2 * DimUp + c] = acc1;
    }
  }
}

static int launch_dw_panels(const NormalArgs& a, const double* v_full, double* hv, hipStream_t st) {
  PanelArgs p;
  p.dim_up = a.dim_up;
  p.dw_first = a.dw_first;
  p.dw_count = a.dw_count;
  p.dw_rowptr = a.dw_rowptr;
  p.dw_col = a.dw_col;
  p.dw_val = a.dw_val;
  // panels: a multiple of 8 (one stream of panels per XCD), at most 64 columns wide
  int np = (int)((a.dim_up + 511) / 512) * 8;
  if (np < 8) np = 8;
  p.width = (int)((a.dim_up + np - 1) / np);
  if (p.width < 1) p.width = 1;
  p.npanels = (int)((a.dim_up + p.width - 1) / p.width);
  // ~64 workgroups per panel = what one XCD keeps resident (32 CUs x 2): one panel in flight per XCD
  int bpp = 64;
  p.rows_per_block = (int)((a.dw_count + bpp - 1) / bpp);
  if (p.rows_per_block < 16) p.rows_per_block = 16;
  bpp = (int)((a.dw_count + p.rows_per_block - 1) / p.rows_per_block);
  p.blocks_per_panel = bpp;
  const int panel_groups = (p.npanels + 7) / 8;
  const int64_t nblk = (int64_t)panel_groups * bpp * 8;
  hipLaunchKernelGGL(normal_dw_panel_kernel, dim3((unsigned)nblk), dim3(kPanelNT), 0, st, p, v_full, hv);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// TD rows of V (8 B/elem) must fit the LDS budget of one workgroup.  Keep two
// workgroups per CU when possible (<= 64 KiB each of the 160 KiB LDS).
int normal_pick_rows_per_block(int64_t dim_up, int64_t dw_count) {
  const int64_t row_bytes = dim_up * 8;
  if (row_bytes > 150 * 1024) return 0;  // generic (no LDS) kernel
  int td = 1;
  for (int cand : {2, 4, 8}) {
    if (cand * row_bytes > 64 * 1024) break;
    if ((dw_count + cand - 1) / cand < 1024) break;  // keep >= 4 workgroups per CU in the grid
    td = cand;
  }
  return td;
}

// what: bit0 = diagonal+up (overwrite), bit1 = down term inside the row kernel, bit2 = Hnd
template <int TD, int E, bool USE_LDS, bool PACKED>
static int launch_te(const NormalArgs& a, const double* vl, const double* vf, double* hv,
                     int what, hipStream_t st) {
  const int64_t nblk = (a.dw_count + TD - 1) / TD;
  const size_t lds = USE_LDS ? (size_t)TD * a.dim_up * sizeof(double) : 0;
  dim3 grid((unsigned)nblk), block(kNT);
#define EDIGPU_LAUNCH_ROWS(LOC, DWF, NDF, LDSB, UL, PK)                                          \
  do {                                                                                           \
    auto kern = normal_rows_kernel<TD, E, UL, LOC, DWF, NDF, PK>;                                \
    if ((LDSB) > 48 * 1024)                                                                      \
      EDIGPU_HIP(hipFuncSetAttribute((const void*)kern,                                          \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSB)));  \
    hipLaunchKernelGGL(kern, grid, block, (LDSB), st, a, vl, vf, hv);                            \
  } while (0)
  switch (what) {
    case 1: EDIGPU_LAUNCH_ROWS(true, false, false, lds, USE_LDS, PACKED); break;
    case 5: EDIGPU_LAUNCH_ROWS(true, false, true, lds, USE_LDS, PACKED); break;
    case 7: EDIGPU_LAUNCH_ROWS(true, true, true, lds, USE_LDS, PACKED); break;
    case 4: EDIGPU_LAUNCH_ROWS(false, false, true, (size_t)0, false, false); break;
    case 6: EDIGPU_LAUNCH_ROWS(false, true, true, (size_t)0, false, false); break;
    case 2: EDIGPU_LAUNCH_ROWS(false, true, false, (size_t)0, false, false); break;
    default: set_error("launch_normal: bad term mask"); return 1;
  }
#undef EDIGPU_LAUNCH_ROWS
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

template <int TD, bool USE_LDS>
static int launch_td(const NormalArgs& a, bool packed, const double* vl, const double* vf,
                     double* hv, int what, hipStream_t st) {
  // E columns per thread and pass: enough to cover the row in few passes, capped by registers
  const int64_t per_thread = (a.dim_up + kNT - 1) / kNT;
  constexpr int EMAX = (TD >= 4) ? 2 : 4;
  if (per_thread >= 3 && EMAX >= 4)
    return packed ? launch_te<TD, 4, USE_LDS, true>(a, vl, vf, hv, what, st)
                  : launch_te<TD, 4, USE_LDS, false>(a, vl, vf, hv, what, st);
  if (per_thread >= 2)
    return packed ? launch_te<TD, 2, USE_LDS, true>(a, vl, vf, hv, what, st)
                  : launch_te<TD, 2, USE_LDS, false>(a, vl, vf, hv, what, st);
  return packed ? launch_te<TD, 1, USE_LDS, true>(a, vl, vf, hv, what, st)
                : launch_te<TD, 1, USE_LDS, false>(a, vl, vf, hv, what, st);
}

static int launch_rows(const edigpu_sector* s, const NormalArgs& a, const double* vl,
                       const double* vf, double* hv, int what, hipStream_t st) {
  const bool packed = s->up_ell.pk != nullptr;
  switch (s->rows_per_block) {
    case 0: return launch_td<1, false>(a, packed, vl, vf, hv, what, st);
    case 1: return launch_td<1, true>(a, packed, vl, vf, hv, what, st);
    case 2: return launch_td<2, true>(a, packed, vl, vf, hv, what, st);
    case 4: return launch_td<4, true>(a, packed, vl, vf, hv, what, st);
    case 8: return launch_td<8, true>(a, packed, vl, vf, hv, what, st);
    default: set_error("launch_normal: bad rows_per_block"); return 1;
  }
}

// EDIGPU_NORMAL_DW=rows keeps the down term inside the row kernel (single pass, neighbour rows
// through L2/Infinity Cache); the default "panels" runs it as the L2-blocked column-panel sweep.
static bool dw_in_rows() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("EDIGPU_NORMAL_DW");
    mode = (e && std::string(e) == "rows") ? 1 : 0;
  }
  return mode == 1;
}

int launch_normal(const edigpu_sector* s, const double* v_local, const double* v_full, double* hv,
                  int phase, hipStream_t st) {
  NormalArgs a;
  a.dim_up = s->dim_up;
  a.dw_first = s->dw_first;
  a.dw_count = s->dw_count;
  a.hd = s->d_hd;
  a.ell_pk = s->up_ell.pk;
  a.ell_coef = s->up_ell.coef;
  a.ell_col = s->up_ell.col;
  a.ell_val = s->up_ell.val;
  a.ell_w = s->up_ell.width;
  a.ell_pitch = s->up_ell.pitch;
  a.dw_rowptr = s->dw.rowptr32;
  a.dw_col = s->dw.col;
  a.dw_val = s->dw.val;
  a.nd_rp32 = s->nd.rowptr32;
  a.nd_rp64 = s->nd.wide ? s->nd.rowptr64 : nullptr;
  a.nd_col = s->nd.col;
  a.nd_val = s->nd.val;
  a.has_nd = s->has_nd;
  if (s->dw_count == 0) return 0;
  const bool rows = dw_in_rows();
  if (phase == 1) return launch_rows(s, a, v_local, v_full, hv, 1, st);
  if (phase == 3) {
    if (rows) return launch_rows(s, a, v_local, v_full, hv, 7, st);
    if (launch_rows(s, a, v_local, v_full, hv, s->has_nd ? 5 : 1, st)) return 1;
    return launch_dw_panels(a, v_full, hv, st);
  }
  // phase 2: the terms that need the gathered vector, accumulated into hv
  if (rows) return launch_rows(s, a, v_local, v_full, hv, s->has_nd ? 6 : 2, st);
  if (s->has_nd && launch_rows(s, a, v_local, v_full, hv, 4, st)) return 1;
  return launch_dw_panels(a, v_full, hv, st);
}

}  // namespace edigpu
