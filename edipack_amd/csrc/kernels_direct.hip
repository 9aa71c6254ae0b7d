// kernels_direct.hip -- on-the-fly ("direct", ed_sparse_H=F) H*v for the superc / nonsu2 sectors on gfx950.
//
// Takes the place of directMatVec_nonsu2_main / directMatVec_MPI_nonsu2_main (reference
// ED_NONSU2/ED_HAMILTONIAN_NONSU2_DIRECT_HxV.f90:22-252 + direct/HxV*.f90) and of
// directMatVec_[MPI_]superc_main (ED_SUPERC/ED_HAMILTONIAN_SUPERC_DIRECT_HxV.f90:22-311): the matrix
// is never stored; every product regenerates the elements from the sector map.
//
// What the reference does per element -- bdecomp into an int array, O(pos) btest loops for the
// fermionic sign, a recursive binary search for the column -- becomes bit arithmetic in registers:
//   applicability : (s & need_set) == need_set && (s & need_clear) == 0
//   sign          : parity of popcount(s & sign_mask) (+ a per-term constant)
//   column        : off_dw[w >> Ns] + rk_up[w & (2^Ns-1)]   (two table lookups, w = s ^ flip)
// One lane owns one row; the term list is wave-uniform (scalar loads).  The gathers of v go through
// L2 / Infinity Cache (consecutive rows map to nearby columns: the rank maps are monotone).
// Integer/latency bound, not a bandwidth kernel: judged by iterations/s.
#include "kernels.hpp"

namespace edigpu {

constexpr int kDirNT = 256;

__global__ void __launch_bounds__(kDirNT)
    direct_rows_kernel(int64_t nrow, int64_t row_first, int ns, int norb, int nterms,
                       const int32_t* __restrict__ states, const int32_t* __restrict__ off_dw,
                       const int32_t* __restrict__ rk_up, const DirectTerm* __restrict__ terms,
                       const double* __restrict__ dtab, const double* __restrict__ xtab,
                       const double2* __restrict__ v_full, double2* __restrict__ hv) {
  const uint32_t lomask = (1u << ns) - 1u, impmask = (1u << norb) - 1u;
  for (int64_t r = (int64_t)blockIdx.x * kDirNT + threadIdx.x; r < nrow; r += (int64_t)gridDim.x * kDirNT) {
    const uint32_t s = (uint32_t)states[r];
    // diagonal: one-body energies byte by byte + impurity interaction table
    const double dg = dtab[s & 255u] + dtab[256 + ((s >> 8) & 255u)] + dtab[512 + ((s >> 16) & 255u)] +
                      dtab[768 + (s >> 24)] + xtab[(((s >> ns) & impmask) << norb) | (s & impmask)];
    const double2 x0 = v_full[row_first + r];
    double ar = dg * x0.x, ai = dg * x0.y;
    for (int t = 0; t < nterms; t++) {
      const DirectTerm tm = terms[t];  // wave-uniform
      if ((s & tm.need_set) == tm.need_set && (s & tm.need_clear) == 0u) {
        const uint32_t w = s ^ tm.flip;
        const int64_t j = (int64_t)off_dw[w >> ns] + rk_up[w & lomask];
        const bool neg = ((__popc(s & tm.sign_mask) + tm.csign) & 1) != 0;
        const double cr = neg ? -tm.cre : tm.cre, ci = neg ? -tm.cim : tm.cim;
        const double2 x = v_full[j];
        ar += cr * x.x - ci * x.y;
        ai += cr * x.y + ci * x.x;
      }
    }
    hv[r] = make_double2(ar, ai);
  }
}

int launch_direct(const edigpu_sector* s, const double* v_full, double* hv, hipStream_t st) {
  if (s->nloc == 0) return 0;
  int64_t nb = (s->nloc + kDirNT - 1) / kDirNT;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(direct_rows_kernel, dim3((unsigned)nb), dim3(kDirNT), 0, st, s->nloc, s->row_first,
                     s->dir_ns, s->dir_norb, s->dir_nterms, s->d_dir_states, s->d_dir_offdw, s->d_dir_rkup,
                     s->d_dir_terms, s->d_dir_dtab, s->d_dir_xtab, reinterpret_cast<const double2*>(v_full),
                     reinterpret_cast<double2*>(hv));
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

}  // namespace edigpu
