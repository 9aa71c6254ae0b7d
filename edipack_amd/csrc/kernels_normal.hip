// kernels_normal.hip -- normal-mode (Kronecker) H*v for gfx950.
//
//   Hv = Hd o v + (1 (x) Hup) v + (Hdw (x) 1) v + Hnd v
//
// takes the place of spMatVec_normal_main (reference ED_NORMAL/
// ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650) and, in its two-phase form, of
// spMatVec_mpi_normal_main (:765-929).  The state vector is viewed as the matrix
// V[idw][iup] (iup contiguous, ED_SECTOR.f90:1705-1717).
//
// Mapping to the hardware (one workgroup = TD consecutive idw rows):
//  * the TD rows of V are staged once into LDS (coalesced HBM read); the Hup term is
//    a gather inside a row and is served from LDS, not from L1/L2;
//  * Hup is held as ELL, column-major, so that the 64 lanes of a wave read
//    consecutive (col,val) slots -- coalesced, L2-resident, shared by the TD rows;
//  * the Hdw term reads whole neighbour rows V[jdw][:], contiguous along iup:
//    coalesced; the row list of idw is wave-uniform and lands in SGPRs;
//  * Hd is streamed, Hnd (0.3 nnz/row) is a short per-row CSR gather.
// No MFMA: ~0.25 flop/byte, HBM/L2 bound.
#include "kernels.hpp"

namespace edigpu {

struct NormalArgs {
  int64_t dim_up, dw_first, dw_count;
  const double* hd;
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

template <int TD, bool USE_LDS, bool LOCAL, bool REMOTE>
__global__ void __launch_bounds__(kNT)
    normal_rows_kernel(NormalArgs a, const double* __restrict__ v_local,
                       const double* __restrict__ v_full, double* __restrict__ hv) {
  extern __shared__ double vs[];
  const int64_t DimUp = a.dim_up;
  const int64_t r0 = (int64_t)blockIdx.x * TD;  // first local row of this block
  const int tid = threadIdx.x;
  int nr = TD;
  if (r0 + nr > a.dw_count) nr = (int)(a.dw_count - r0);

  if (LOCAL && USE_LDS) {
    for (int r = 0; r < nr; r++) {
      const double* src = v_local + (r0 + r) * DimUp;
      for (int64_t iup = tid; iup < DimUp; iup += kNT) vs[r * DimUp + iup] = src[iup];
    }
    __syncthreads();
  }

  for (int64_t iup = tid; iup < DimUp; iup += kNT) {
    double acc[TD];
#pragma unroll
    for (int r = 0; r < TD; r++) acc[r] = 0.0;

    if (LOCAL) {
      // diagonal
#pragma unroll
      for (int r = 0; r < TD; r++)
        if (r < nr) {
          const int64_t i = (r0 + r) * DimUp + iup;
          const double x = USE_LDS ? vs[r * DimUp + iup] : v_local[i];
          acc[r] = a.hd[i] * x;
        }
      // (1 (x) Hup): gather inside the row
      for (int k = 0; k < a.ell_w; k++) {
        const int32_t c = a.ell_col[(int64_t)k * a.ell_pitch + iup];
        const double w = a.ell_val[(int64_t)k * a.ell_pitch + iup];
#pragma unroll
        for (int r = 0; r < TD; r++)
          if (r < nr) {
            const double x = USE_LDS ? vs[r * DimUp + c] : v_local[(r0 + r) * DimUp + c];
            acc[r] += w * x;
          }
      }
    }
    if (REMOTE) {
      // (Hdw (x) 1): whole neighbour rows, contiguous in iup
#pragma unroll
      for (int r = 0; r < TD; r++)
        if (r < nr) {
          const int64_t g = a.dw_first + r0 + r;
          const int32_t b = a.dw_rowptr[g], e = a.dw_rowptr[g + 1];
          double s = 0.0;
          for (int32_t jj = b; jj < e; jj++)
            s += a.dw_val[jj] * v_full[(int64_t)a.dw_col[jj] * DimUp + iup];
          acc[r] += s;
        }
      // Hnd: short CSR rows with global columns
      if (a.has_nd) {
#pragma unroll
        for (int r = 0; r < TD; r++)
          if (r < nr) {
            const int64_t i = (r0 + r) * DimUp + iup;
            int64_t b, e;
            if (a.nd_rp64) {
              b = a.nd_rp64[i];
              e = a.nd_rp64[i + 1];
            } else {
              b = a.nd_rp32[i];
              e = a.nd_rp32[i + 1];
            }
            double s = 0.0;
            for (int64_t jj = b; jj < e; jj++) s += a.nd_val[jj] * v_full[a.nd_col[jj]];
            acc[r] += s;
          }
      }
    }
#pragma unroll
    for (int r = 0; r < TD; r++)
      if (r < nr) {
        const int64_t i = (r0 + r) * DimUp + iup;
        if (LOCAL)
          hv[i] = acc[r];
        else
          hv[i] += acc[r];
      }
  }
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

template <int TD, bool USE_LDS>
static int launch_td(const NormalArgs& a, const double* vl, const double* vf, double* hv,
                     int phase, hipStream_t st) {
  const int64_t nblk = (a.dw_count + TD - 1) / TD;
  const size_t lds = USE_LDS ? (size_t)TD * a.dim_up * sizeof(double) : 0;
  dim3 grid((unsigned)nblk), block(kNT);
  if (phase == 3) {
    if (lds > 64 * 1024)
      EDIGPU_HIP(hipFuncSetAttribute((const void*)normal_rows_kernel<TD, USE_LDS, true, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((normal_rows_kernel<TD, USE_LDS, true, true>), grid, block, lds, st, a, vl,
                       vf, hv);
  } else if (phase == 1) {
    if (lds > 64 * 1024)
      EDIGPU_HIP(hipFuncSetAttribute((const void*)normal_rows_kernel<TD, USE_LDS, true, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((normal_rows_kernel<TD, USE_LDS, true, false>), grid, block, lds, st, a,
                       vl, vf, hv);
  } else {
    hipLaunchKernelGGL((normal_rows_kernel<TD, false, false, true>), grid, block, 0, st, a, vl,
                       vf, hv);
  }
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_normal(const edigpu_sector* s, const double* v_local, const double* v_full, double* hv,
                  int phase, hipStream_t st) {
  NormalArgs a;
  a.dim_up = s->dim_up;
  a.dw_first = s->dw_first;
  a.dw_count = s->dw_count;
  a.hd = s->d_hd;
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
  switch (s->rows_per_block) {
    case 0: return launch_td<1, false>(a, v_local, v_full, hv, phase, st);
    case 1: return launch_td<1, true>(a, v_local, v_full, hv, phase, st);
    case 2: return launch_td<2, true>(a, v_local, v_full, hv, phase, st);
    case 4: return launch_td<4, true>(a, v_local, v_full, hv, phase, st);
    case 8: return launch_td<8, true>(a, v_local, v_full, hv, phase, st);
    default: set_error("launch_normal: bad rows_per_block"); return 1;
  }
}

}  // namespace edigpu
