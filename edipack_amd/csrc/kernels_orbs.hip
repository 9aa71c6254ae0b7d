// kernels_orbs.hip -- H*v for the ed_total_ud = F ("orbs") normal-mode sectors on gfx950.
//
// Takes the place of spMatVec_normal_orbs (reference ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:652-761,
// electronic part): the vector is the rank-2*Norb tensor [iup_1..iup_Norb, idw_1..idw_Norb] (first index
// fastest, state2indices / indices2state) and
//     Hv = Hd o v + sum_k (1 (x) .. (x) F_k (x) .. (x) 1) v ,   F_k = (1+Nbath)-level chain of one orbital and spin.
// Where the reference converts every element to an index tuple and back (integer div/mod chains inside
// state2indices / indices2state for every matrix element), a lane decodes its own tuple once and reaches a
// neighbour by adding (col - idx_k) * stride_k.  One lane = one element; the tiny factor matrices (ELL,
// <= Nbath entries per row) and the diagonal tables are read through the caches.  Not the BASELINE path
// (SURVEY.md 8a row a9): a plain memory-bound kernel, no LDS staging.
#include "kernels.hpp"

namespace edigpu {

constexpr int kOrbsNT = 256;

__global__ void __launch_bounds__(kOrbsNT)
    orbs_rows_kernel(OrbsArgs a, int64_t row_first, int64_t row_count, const double* __restrict__ v,
                     double* __restrict__ hv) {
  // rows [row_first, row_first + row_count) of the sector (a shard: spMatVec_mpi_normal_orbs, :932-1082, as the
  // all-gather form); v is the whole vector, hv holds the shard's rows
  for (int64_t il = (int64_t)blockIdx.x * kOrbsNT + threadIdx.x; il < row_count; il += (int64_t)gridDim.x * kOrbsNT) {
    const int64_t i = row_first + il;
    uint32_t rem = (uint32_t)i;
    int idx[kOrbsMaxAxes];
    uint32_t bits = 0;
    double dg = 0.0;
#pragma unroll
    for (int k = 0; k < kOrbsMaxAxes; k++) {
      if (k < a.naxes) {
        const uint32_t d = (uint32_t)a.dims[k];
        const uint32_t q = rem / d;
        idx[k] = (int)(rem - q * d);
        rem = q;
        if (a.hd == nullptr) {
          dg += a.eax[a.off[k] + idx[k]];
          bits |= (uint32_t)a.impbit[a.off[k] + idx[k]] << k;
        }
      } else {
        idx[k] = 0;
      }
    }
    dg = a.hd != nullptr ? a.hd[il] : dg + a.xtab[bits];
    double acc = dg * v[i];
#pragma unroll
    for (int k = 0; k < kOrbsMaxAxes; k++) {
      if (k < a.naxes) {
        const int w = a.width[k];
        const int64_t base = a.elloff[k];
        const int d = (int)a.dims[k];
        for (int s = 0; s < w; s++) {
          const int32_t c = a.ell_col[base + (int64_t)s * d + idx[k]];
          const double x = a.ell_val[base + (int64_t)s * d + idx[k]];  // 0 for padding (col = own index)
          acc = fma(x, v[i + (int64_t)(c - idx[k]) * a.stride[k]], acc);
        }
      }
    }
    hv[il] = acc;
  }
}

int launch_orbs(const edigpu_sector* s, const double* v, double* hv, hipStream_t st) {
  if (s->nloc == 0) return 0;
  int64_t nb = (s->nloc + kOrbsNT - 1) / kOrbsNT;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(orbs_rows_kernel, dim3((unsigned)nb), dim3(kOrbsNT), 0, st, s->orbs, s->row_first, s->nloc, v, hv);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

}  // namespace edigpu
