// kernels_ops.hip -- c / c^+ on a normal-mode sector vector, device to device.
//
// Takes the place of apply_op_C / apply_op_CDG (reference ED_SECTOR.f90:465-536, 654-839) for the
// ed_total_ud=T normal mode: the step between the diagonalisation and the Green's-function
// tridiagonalisation (ED_NORMAL/ED_GF_NORMAL.f90:141-175).  In the reference it runs on the master rank and is
// followed by a scatter; here the eigenvector stays on the device and the result is the seed of
// edigpu_lanczos_tridiag_dev.  The operator changes one spin species only, so it is a signed partial
// permutation of the columns (up) or of the rows (down) of V[idw][iup]:
//     up  : dst[jdw][jup] = sgn(jup) * src[jdw][part(jup)]
//     down: dst[jdw][jup] = sgn(jdw) * src[part(jdw)][jup]
// part = index in the source sector | sign << 31, 0xFFFFFFFF = no preimage (host table, O(DimUp|DimDw)).
#include "exchange_index.hpp"
#include "kernels.hpp"

namespace edigpu {

__global__ void __launch_bounds__(256)
    apply_op_normal_kernel(int64_t dst_dimup, int64_t dst_dimdw, int64_t src_dimup, int spin_down,
                           const uint32_t* __restrict__ part, const double* __restrict__ src,
                           double* __restrict__ dst, double coef, int accumulate) {
  const int64_t n = dst_dimup * dst_dimdw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t jdw = i / dst_dimup, jup = i - jdw * dst_dimup;
    const uint32_t p = part[spin_down ? jdw : jup];
    double x = 0.0;
    if (p != 0xFFFFFFFFu) {
      const int64_t k = (int64_t)(p & 0x7FFFFFFFu);
      x = spin_down ? src[k * src_dimup + jup] : src[jdw * src_dimup + k];
      if (p >> 31) x = -x;
    }
    // apply_Cops: a coefficient per operator, the operators after the first add to the result
    dst[i] = accumulate ? fma(coef, x, dst[i]) : coef * x;
  }
}

// flat (superc / nonsu2) sectors: one map over the 2*Ns-bit states; the sign counts every occupied level
// below the operator's level (ED_AUX_FUNX.f90:334-384), complex vectors
__global__ void __launch_bounds__(256)
    apply_op_flat_kernel(int64_t ndst, int ns, uint32_t bit, int create, const int32_t* __restrict__ dst_states,
                         const int32_t* __restrict__ src_offdw, const int32_t* __restrict__ src_rkup,
                         const double2* __restrict__ src, double2* __restrict__ dst, double cre, double cim, int accumulate) {
  const uint32_t lomask = (1u << ns) - 1u;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < ndst; j += (int64_t)gridDim.x * 256) {
    const uint32_t t = (uint32_t)dst_states[j];
    double2 x = make_double2(0.0, 0.0);
    if (create ? (t & bit) != 0u : (t & bit) == 0u) {
      const uint32_t sst = t ^ bit;
      const int64_t i = (int64_t)src_offdw[sst >> ns] + src_rkup[sst & lomask];
      x = src[i];
      if (__popc(sst & (bit - 1u)) & 1) {
        x.x = -x.x;
        x.y = -x.y;
      }
    }
    // apply_Cops: a complex coefficient per operator, the operators after the first add to the result
    double2 y = make_double2(cre * x.x - cim * x.y, cre * x.y + cim * x.x);
    if (accumulate) {
      const double2 o = dst[j];
      y.x += o.x;
      y.y += o.y;
    }
    dst[j] = y;
  }
}

int launch_apply_op_flat(int64_t ndst, int ns, uint32_t bit, int create, const int32_t* dst_states,
                         const int32_t* src_offdw, const int32_t* src_rkup, const double* src, double* dst,
                         hipStream_t st, double cre, double cim, int accumulate) {
  if (ndst == 0) return 0;
  int64_t nb = (ndst + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(apply_op_flat_kernel, dim3((unsigned)nb), dim3(256), 0, st, ndst, ns, bit, create, dst_states,
                     src_offdw, src_rkup, reinterpret_cast<const double2*>(src), reinterpret_cast<double2*>(dst), cre, cim, accumulate);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// Phonon branches of spMatVec_normal_main (reference ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:597-629):
//   hv(i_el, n) += w0 n v(i_el, n) + (A + g_el(i_el)) (sqrt(n) v(i_el, n-1) + sqrt(n+1) v(i_el, n+1)),
// g_el = sum_a g_aa (n_a,up + n_a,dw) = gu[iup] + gd[idw] (density couplings; stored/H_ph.f90, H_e_ph.f90).
// The electronic part has already been applied to every phonon block; this is one streaming pass.
__global__ void __launch_bounds__(256)
    phonon_kernel(int64_t dim_el, int dimph, int64_t dim_up, double w0, double a_ph, const double* __restrict__ gu,
                  const double* __restrict__ gd, const double* __restrict__ v, double* __restrict__ hv) {
  const int64_t n = dim_el * dimph;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t iph = i / dim_el, iel = i - iph * dim_el;
    const int64_t idw = iel / dim_up, iup = iel - idw * dim_up;
    const double g = a_ph + gu[iup] + gd[idw];
    double acc = hv[i] + w0 * (double)iph * v[i];
    if (iph > 0) acc = fma(g * sqrt((double)iph), v[i - dim_el], acc);
    if (iph + 1 < dimph) acc = fma(g * sqrt((double)(iph + 1)), v[i + dim_el], acc);
    hv[i] = acc;
  }
}

// ---- electron-phonon coupling with a general g_ph(a,b): t = O v[jph] goes to the blocks jph +- 1 ----
__global__ void __launch_bounds__(256)
    eph_scatter_kernel(int64_t n, const double* __restrict__ t, double* __restrict__ up, double c_up,
                       double* __restrict__ dn, double c_dn) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double x = t[i];
    if (up) up[i] = fma(c_up, x, up[i]);
    if (dn) dn[i] = fma(c_dn, x, dn[i]);
  }
}

int launch_eph_scatter(int64_t n, const double* t, double* up, double c_up, double* dn, double c_dn, hipStream_t st) {
  if (n <= 0 || (!up && !dn)) return 0;
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(eph_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, t, up, c_up, dn, c_dn);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// ---- _CMPLX_NORMAL (complex algebra in normal mode): planar work vectors around four real products ----
__global__ void __launch_bounds__(256)
    deinterleave_kernel(int64_t n, const double2* __restrict__ z, double* __restrict__ re, double* __restrict__ im) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double2 x = z[i];
    re[i] = x.x;
    im[i] = x.y;
  }
}

__global__ void __launch_bounds__(256)
    combine_interleave_kernel(int64_t n, const double* __restrict__ yr, const double* __restrict__ yi,
                              const double* __restrict__ t1, const double* __restrict__ t2, double2* __restrict__ z) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double a = yr[i], b = yi[i];
    if (t1) {
      a -= t1[i];
      b += t2[i];
    }
    z[i] = make_double2(a, b);
  }
}

int launch_deinterleave(int64_t n, const double* z, double* re, double* im, hipStream_t st) {
  if (n <= 0) return 0;
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)nb), dim3(256), 0, st, n,
                     reinterpret_cast<const double2*>(z), re, im);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_combine_interleave(int64_t n, const double* yr, const double* yi, const double* t1, const double* t2,
                              double* z, hipStream_t st) {
  if (n <= 0) return 0;
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(combine_interleave_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, yr, yi, t1, t2,
                     reinterpret_cast<double2*>(z));
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// ---- transposed exchange (reference vector_transpose_MPI, ED_NORMAL/ED_HAMILTONIAN_NORMAL_COMMON.f90:66-167) ----
// A rank owns q down rows of V[idw][iup] (row shard).  For the down half of the product every rank needs ALL rows of
// a block of pcol up columns (+ halo columns on both sides for Hnd): block (r -> c) = rows of rank r, columns
// [c*pcol - halo, (c+1)*pcol + halo).  send[c][i][j] is laid out so that one equal-split all-to-all delivers
// recv[r][i][j] = rows r*q + i in order, i.e. the column shard with row stride pcol + 2*halo and no unpacking.
// Rows >= nrows (the tail rank's padding) and columns outside [0, DimUp) are sent as zeros.
__global__ void __launch_bounds__(256)
    transpose_pack_kernel(int64_t dim_up, int64_t nrows, int64_t q, int world, int64_t pcol, int halo,
                          const double* __restrict__ v, double* __restrict__ send) {
  const int64_t pw = pcol + 2 * halo;
  const int64_t n = (int64_t)world * q * pw;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t src = xch_send_source(e, dim_up, nrows, q, pcol, halo);
    send[e] = src >= 0 ? v[src] : 0.0;
  }
}

// the way back: recv[c][i][halo + j] = the down half of H*v for (row i of this rank, column c*pcol + j)
__global__ void __launch_bounds__(256)
    transpose_unpack_add_kernel(int64_t dim_up, int64_t nrows, int64_t q, int64_t pcol, int halo,
                                const double* __restrict__ recv, double* __restrict__ hv) {
  const int64_t n = nrows * dim_up;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / dim_up, col = e - i * dim_up;
    hv[e] += recv[xch_back_slot(i, col, q, pcol, halo)];
  }
}

int launch_transpose_pack(int64_t dim_up, int64_t nrows, int64_t q, int world, int64_t pcol, int halo,
                          const double* v_rows, double* send, hipStream_t st) {
  const int64_t n = (int64_t)world * q * (pcol + 2 * halo);
  if (n == 0) return 0;
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(transpose_pack_kernel, dim3((unsigned)nb), dim3(256), 0, st, dim_up, nrows, q, world, pcol, halo,
                     v_rows, send);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_transpose_unpack_add(int64_t dim_up, int64_t nrows, int64_t q, int world, int64_t pcol, int halo,
                                const double* recv, double* hv_rows, hipStream_t st) {
  (void)world;
  const int64_t n = nrows * dim_up;
  if (n == 0) return 0;
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(transpose_unpack_add_kernel, dim3((unsigned)nb), dim3(256), 0, st, dim_up, nrows, q, pcol, halo,
                     recv, hv_rows);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// the same pass for the complex (superc / nonsu2) sectors: g_el(i_el) comes as one table over the sector rows
__global__ void __launch_bounds__(256)
    phonon_flat_kernel(int64_t dim_el, int dimph, double w0, double a_ph, const double* __restrict__ gel,
                       const double2* __restrict__ v, double2* __restrict__ hv) {
  const int64_t n = dim_el * dimph;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t iph = i / dim_el, iel = i - iph * dim_el;
    const double g = a_ph + gel[iel];
    double2 acc = hv[i];
    const double2 x = v[i];
    acc.x = fma(w0 * (double)iph, x.x, acc.x);
    acc.y = fma(w0 * (double)iph, x.y, acc.y);
    if (iph > 0) {
      const double2 y = v[i - dim_el];
      const double c = g * sqrt((double)iph);
      acc.x = fma(c, y.x, acc.x);
      acc.y = fma(c, y.y, acc.y);
    }
    if (iph + 1 < dimph) {
      const double2 y = v[i + dim_el];
      const double c = g * sqrt((double)(iph + 1));
      acc.x = fma(c, y.x, acc.x);
      acc.y = fma(c, y.y, acc.y);
    }
    hv[i] = acc;
  }
}

int launch_phonon(const edigpu_sector* s, const double* v, double* hv, hipStream_t st) {
  if (s->is_complex) {
    const int64_t n = s->dim_el * (s->nph + 1);
    int64_t nb = (n + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(phonon_flat_kernel, dim3((unsigned)nb), dim3(256), 0, st, s->dim_el, s->nph + 1, s->w0_ph,
                       s->a_ph, s->d_gu, reinterpret_cast<const double2*>(v), reinterpret_cast<double2*>(hv));
    EDIGPU_HIP(hipGetLastError());
    return 0;
  }
  const int64_t n = s->dim_el * (s->nph + 1);
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(phonon_kernel, dim3((unsigned)nb), dim3(256), 0, st, s->dim_el, s->nph + 1, s->dim_up, s->w0_ph,
                     s->a_ph, s->d_gu, s->d_gd, v, hv);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_phonon_rows(const edigpu_sector* s, int64_t dw_first, int64_t dw_count, const double* v, double* hv,
                       hipStream_t st) {
  const int64_t blk = dw_count * s->dim_up, n = blk * (s->nph + 1);
  if (n <= 0) return 0;
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(phonon_kernel, dim3((unsigned)nb), dim3(256), 0, st, blk, s->nph + 1, s->dim_up, s->w0_ph, s->a_ph,
                     s->d_gu, s->d_gd + dw_first, v, hv);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_apply_op_normal(int64_t dst_dimup, int64_t dst_dimdw, int64_t src_dimup, int spin_down,
                           const uint32_t* part, const double* src, double* dst, hipStream_t st, double coef,
                           int accumulate) {
  const int64_t n = dst_dimup * dst_dimdw;
  if (n == 0) return 0;
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(apply_op_normal_kernel, dim3((unsigned)nb), dim3(256), 0, st, dst_dimup, dst_dimdw, src_dimup,
                     spin_down, part, src, dst, coef, accumulate);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

}  // namespace edigpu
