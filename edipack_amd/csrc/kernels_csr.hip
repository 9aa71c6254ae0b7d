// kernels_csr.hip -- flat row-CSR SpMV (real and complex) for gfx950.
//
// Takes the place of spMatVec_superc_main / spMatVec_mpi_superc_main (reference
// ED_SUPERC/ED_HAMILTONIAN_SUPERC_STORED_HxV.f90:312-432), spMatVec_nonsu2_main /
// spMatVec_mpi_nonsu2_main (ED_NONSU2/ED_HAMILTONIAN_NONSU2_STORED_HxV.f90:194-267)
// and of the real sp_matvec (ED_SPARSE_MATRIX.f90:778-793).
//
// A group of LPR lanes (power of two, 2..64, chosen from the average row length)
// owns one row: its (col,val) pairs are read coalesced, the x gathers go through
// L2/Infinity Cache, and the partial sums are combined with wavefront shuffles
// (64-wide waves: 64/LPR rows per wave).  No MFMA -- bandwidth bound.
#include "kernels.hpp"

namespace edigpu {

constexpr int kCsrNT = 256;

template <int LPR, bool CPLX, bool ACC, typename RP>
__global__ void __launch_bounds__(kCsrNT)
    csr_rows_kernel(int64_t nrow, const RP* __restrict__ rowptr, const int32_t* __restrict__ col,
                    const double* __restrict__ val, const double* __restrict__ x,
                    double* __restrict__ y) {
  constexpr int RPB = kCsrNT / LPR;
  const int lane = threadIdx.x % LPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  double sr = 0.0, si = 0.0;
  if (row < nrow) {
    const int64_t b = rowptr[row], e = rowptr[row + 1];
    for (int64_t k = b + lane; k < e; k += LPR) {
      const int64_t c = col[k];
      if (CPLX) {
        const double2 a = reinterpret_cast<const double2*>(val)[k];
        const double2 xv = reinterpret_cast<const double2*>(x)[c];
        sr += a.x * xv.x - a.y * xv.y;
        si += a.x * xv.y + a.y * xv.x;
      } else {
        sr += val[k] * x[c];
      }
    }
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) {
    sr += __shfl_down(sr, off, LPR);
    if (CPLX) si += __shfl_down(si, off, LPR);
  }
  if (row < nrow && lane == 0) {
    if (CPLX) {
      double2* yy = reinterpret_cast<double2*>(y) + row;
      double2 o = ACC ? *yy : make_double2(0.0, 0.0);
      o.x += sr;
      o.y += si;
      *yy = o;
    } else {
      y[row] = (ACC ? y[row] : 0.0) + sr;
    }
  }
}

// DOT epilogue (fused Lanczos step, y = Q accumulates H*x on top of -beta*v_prev): per-workgroup
// partials of <x|y_new> and <y_new|y_new> over the real view, written to partial[blockIdx] and
// partial[gridDim + blockIdx], and <x|x> to partial[2 gridDim + blockIdx] (deterministic two-stage reduction,
// finalised by k_finalize_ab).
__device__ inline void block_dot_partials(double a, double q, double n, double* __restrict__ partial) {
  __shared__ double red_a[kCsrNT / 64], red_q[kCsrNT / 64], red_n[kCsrNT / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    q += __shfl_down(q, off, 64);
    n += __shfl_down(n, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red_a[threadIdx.x >> 6] = a;
    red_q[threadIdx.x >> 6] = q;
    red_n[threadIdx.x >> 6] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tq = 0.0, tn = 0.0;
#pragma unroll
    for (int i = 0; i < kCsrNT / 64; i++) {
      ta += red_a[i];
      tq += red_q[i];
      tn += red_n[i];
    }
    partial[blockIdx.x] = ta;
    partial[gridDim.x + blockIdx.x] = tq;
    partial[2 * gridDim.x + blockIdx.x] = tn;
  }
}

// SELL-64: one lane = one row of a 64-row slice, slot k of all 64 rows is contiguous.  Rows are sorted by
// column, so the 64 gathers of a slot land on few cache lines (adjacent rows of these Hamiltonians
// reach adjacent columns through the same hop): ~3x fewer L1 accesses than the lane-group CSR kernel,
// whose slot k mixes unrelated hops of neighbouring rows.  No cross-lane reduction.
template <bool CPLX, bool ACC, bool DOT>
__global__ void __launch_bounds__(kCsrNT)
    sell_rows_kernel(int64_t nrow, int64_t nslice, const int32_t* __restrict__ sptr,
                     const int32_t* __restrict__ col, const double* __restrict__ val,
                     const double* __restrict__ x, double* __restrict__ y, double* __restrict__ partial,
                     const double* __restrict__ sig) {
  const double sg = (DOT && sig) ? sig[0] : 0.0;  // <y|y> accumulated about the previous alpha, see k_finalize_ab
  const int lane = threadIdx.x & 63;
  const int64_t slice = (int64_t)blockIdx.x * (kCsrNT / 64) + (threadIdx.x >> 6);
  if (!DOT && slice >= nslice) return;
  const int64_t row = slice * 64 + lane;
  const bool in = slice < nslice;
  const int32_t b = in ? sptr[slice] : 0, e = in ? sptr[slice + 1] : 0;
  double sr = 0.0, si = 0.0, da = 0.0, dq = 0.0, dn = 0.0;
#pragma unroll 8
  for (int32_t k = b; k < e; k++) {
    const int64_t o = (int64_t)k * 64 + lane;
    const int64_t c = col[o];
    if (CPLX) {
      const double2 a = reinterpret_cast<const double2*>(val)[o];
      const double2 xv = reinterpret_cast<const double2*>(x)[c];
      sr += a.x * xv.x - a.y * xv.y;
      si += a.x * xv.y + a.y * xv.x;
    } else {
      sr += val[o] * x[c];
    }
  }
  if (in && row < nrow) {
    if (CPLX) {
      double2* yy = reinterpret_cast<double2*>(y) + row;
      double2 o2 = ACC ? *yy : make_double2(0.0, 0.0);
      o2.x += sr;
      o2.y += si;
      *yy = o2;
      if (DOT) {
        const double2 xo = reinterpret_cast<const double2*>(x)[row];
        da = xo.x * o2.x + xo.y * o2.y;
        dq = (o2.x - sg * xo.x) * (o2.x - sg * xo.x) + (o2.y - sg * xo.y) * (o2.y - sg * xo.y);
        dn = xo.x * xo.x + xo.y * xo.y;
      }
    } else {
      const double o = (ACC ? y[row] : 0.0) + sr;
      y[row] = o;
      if (DOT) {
        da = x[row] * o;
        dq = (o - sg * x[row]) * (o - sg * x[row]);
        dn = x[row] * x[row];
      }
    }
  }
  if (DOT) block_dot_partials(da, dq, dn, partial);
}

// SELL-64 with the value dictionary: 4 bytes per entry, values from a <= 256-entry LDS table, the
// diagonal of the loc block as a separate stream.
template <bool CPLX, bool ACC, bool DOT>
__global__ void __launch_bounds__(kCsrNT)
    sell_rows_packed_kernel(int64_t nrow, int64_t nslice, const int32_t* __restrict__ sptr,
                            const uint32_t* __restrict__ pk, const double* __restrict__ dict,
                            const double* __restrict__ diag, const double* __restrict__ x,
                            double* __restrict__ y, double* __restrict__ partial, const double* __restrict__ sig) {
  const double sg = (DOT && sig) ? sig[0] : 0.0;
  __shared__ double dict_s[CPLX ? 512 : 256];
  for (int i = threadIdx.x; i < (CPLX ? 512 : 256); i += kCsrNT) dict_s[i] = dict[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t slice = (int64_t)blockIdx.x * (kCsrNT / 64) + (threadIdx.x >> 6);
  if (!DOT && slice >= nslice) return;
  const int64_t row = slice * 64 + lane;
  const bool in = slice < nslice;
  const int32_t b = in ? sptr[slice] : 0, e = in ? sptr[slice + 1] : 0;
  double sr = 0.0, si = 0.0, da = 0.0, dq = 0.0, dn = 0.0;
#pragma unroll 8
  for (int32_t k = b; k < e; k++) {
    const uint32_t p = pk[(int64_t)k * 64 + lane];
    const int64_t c = p & 0xFFFFFFu;
    const int id = p >> 24;
    if (CPLX) {
      const double ar = dict_s[2 * id], ai = dict_s[2 * id + 1];
      const double2 xv = reinterpret_cast<const double2*>(x)[c];
      sr += ar * xv.x - ai * xv.y;
      si += ar * xv.y + ai * xv.x;
    } else {
      sr += dict_s[id] * x[c];
    }
  }
  if (in && row < nrow) {
    if (diag != nullptr) {
      if (CPLX) {
        const double2 d = reinterpret_cast<const double2*>(diag)[row];
        const double2 xv = reinterpret_cast<const double2*>(x)[row];
        sr += d.x * xv.x - d.y * xv.y;
        si += d.x * xv.y + d.y * xv.x;
      } else {
        sr += diag[row] * x[row];
      }
    }
    if (CPLX) {
      double2* yy = reinterpret_cast<double2*>(y) + row;
      double2 o2 = ACC ? *yy : make_double2(0.0, 0.0);
      o2.x += sr;
      o2.y += si;
      *yy = o2;
      if (DOT) {
        const double2 xo = reinterpret_cast<const double2*>(x)[row];
        da = xo.x * o2.x + xo.y * o2.y;
        dq = (o2.x - sg * xo.x) * (o2.x - sg * xo.x) + (o2.y - sg * xo.y) * (o2.y - sg * xo.y);
        dn = xo.x * xo.x + xo.y * xo.y;
      }
    } else {
      const double o = (ACC ? y[row] : 0.0) + sr;
      y[row] = o;
      if (DOT) {
        da = x[row] * o;
        dq = (o - sg * x[row]) * (o - sg * x[row]);
        dn = x[row] * x[row];
      }
    }
  }
  if (DOT) block_dot_partials(da, dq, dn, partial);
}

__global__ void zero_kernel(double* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = 0.0;
}

int launch_zero(double* y, int64_t n, hipStream_t st) {
  if (n <= 0) return 0;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)nb), dim3(256), 0, st, y, n);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

template <int LPR, bool CPLX, bool ACC>
static int launch_lpr(const DevCsr& a, const double* x, double* y, hipStream_t st) {
  constexpr int RPB = kCsrNT / LPR;
  const int64_t nb = (a.nrow + RPB - 1) / RPB;
  if (a.wide)
    hipLaunchKernelGGL((csr_rows_kernel<LPR, CPLX, ACC, int64_t>), dim3((unsigned)nb),
                       dim3(kCsrNT), 0, st, a.nrow, a.rowptr64, a.col, a.val, x, y);
  else
    hipLaunchKernelGGL((csr_rows_kernel<LPR, CPLX, ACC, int32_t>), dim3((unsigned)nb),
                       dim3(kCsrNT), 0, st, a.nrow, a.rowptr32, a.col, a.val, x, y);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

template <bool CPLX, bool ACC>
static int launch_pick(const DevCsr& a, const double* x, double* y, hipStream_t st) {
  const double r = a.avg_row;
  if (a.sell) {
    const int64_t nb = (a.nslice + kCsrNT / 64 - 1) / (kCsrNT / 64);
    if (a.sell_packed) {
      hipLaunchKernelGGL((sell_rows_packed_kernel<CPLX, ACC, false>), dim3((unsigned)nb), dim3(kCsrNT), 0, st,
                         a.nrow, a.nslice, a.sell_ptr, a.sell_pk, a.sell_dict, a.sell_diag, x, y, (double*)nullptr, (const double*)nullptr);
      EDIGPU_HIP(hipGetLastError());
      return 0;
    }
    hipLaunchKernelGGL((sell_rows_kernel<CPLX, ACC, false>), dim3((unsigned)nb), dim3(kCsrNT), 0, st, a.nrow,
                       a.nslice, a.sell_ptr, a.sell_col, a.sell_val, x, y, (double*)nullptr, (const double*)nullptr);
    EDIGPU_HIP(hipGetLastError());
    return 0;
  }
  if (r <= 3.0) return launch_lpr<2, CPLX, ACC>(a, x, y, st);
  if (r <= 6.0) return launch_lpr<4, CPLX, ACC>(a, x, y, st);
  if (r <= 12.0) return launch_lpr<8, CPLX, ACC>(a, x, y, st);
  if (r <= 24.0) return launch_lpr<16, CPLX, ACC>(a, x, y, st);
  if (r <= 48.0) return launch_lpr<32, CPLX, ACC>(a, x, y, st);
  return launch_lpr<64, CPLX, ACC>(a, x, y, st);
}

int launch_csr(const DevCsr& a, int cplx, const double* x, double* y, int accumulate,
               hipStream_t st) {
  if (a.nrow == 0) return 0;
  if (a.nnz == 0) {
    if (!accumulate) return launch_zero(y, a.nrow * (cplx ? 2 : 1), st);
    return 0;
  }
  if (cplx) return accumulate ? launch_pick<true, true>(a, x, y, st) : launch_pick<true, false>(a, x, y, st);
  return accumulate ? launch_pick<false, true>(a, x, y, st) : launch_pick<false, false>(a, x, y, st);
}

// Fused Lanczos step on a SELL block that holds whole rows of a square matrix (one shard):
// y += A x with the <x|y>, <y|y> partials.  Returns the number of partial pairs in *np.
bool csr_lanczos_fusable(const DevCsr& a) { return a.sell != 0 && a.nrow > 0 && a.nnz > 0; }

int launch_csr_lanczos(const DevCsr& a, int cplx, const double* x, double* y, double* partial, int64_t cap, int* np,
                       const double* sig, hipStream_t st) {
  const int64_t nb = (a.nslice + kCsrNT / 64 - 1) / (kCsrNT / 64);
  if (3 * nb > cap) {
    set_error("launch_csr_lanczos: partial buffer too small");
    return 1;
  }
  *np = (int)nb;
  // the caller's partial buffer holds 2 * nb doubles (ensure_workspace sizes it from the row count)
  const dim3 g((unsigned)nb), blk(kCsrNT);
  if (a.sell_packed) {
    if (cplx)
      hipLaunchKernelGGL((sell_rows_packed_kernel<true, true, true>), g, blk, 0, st, a.nrow, a.nslice, a.sell_ptr,
                         a.sell_pk, a.sell_dict, a.sell_diag, x, y, partial, sig);
    else
      hipLaunchKernelGGL((sell_rows_packed_kernel<false, true, true>), g, blk, 0, st, a.nrow, a.nslice, a.sell_ptr,
                         a.sell_pk, a.sell_dict, a.sell_diag, x, y, partial, sig);
  } else {
    if (cplx)
      hipLaunchKernelGGL((sell_rows_kernel<true, true, true>), g, blk, 0, st, a.nrow, a.nslice, a.sell_ptr,
                         a.sell_col, a.sell_val, x, y, partial, sig);
    else
      hipLaunchKernelGGL((sell_rows_kernel<false, true, true>), g, blk, 0, st, a.nrow, a.nslice, a.sell_ptr,
                         a.sell_col, a.sell_val, x, y, partial, sig);
  }
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

}  // namespace edigpu
