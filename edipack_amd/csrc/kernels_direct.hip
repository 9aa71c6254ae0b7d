// kernels_direct.hip -- on-the-fly ("direct", ed_sparse_H=F) H*v for the superc / nonsu2 sectors on gfx950.
//
// Takes the place of directMatVec_nonsu2_main / directMatVec_MPI_nonsu2_main (reference
// ED_NONSU2/ED_HAMILTONIAN_NONSU2_DIRECT_HxV.f90:22-252 + direct/HxV*.f90) and of
// directMatVec_[MPI_]superc_main (ED_SUPERC/ED_HAMILTONIAN_SUPERC_DIRECT_HxV.f90:22-311): the matrix
// is never stored; every product regenerates the elements from the sector map.
//
// What the reference does per element -- bdecomp into an int array, O(pos) btest loops for the
// fermionic sign, a recursive binary search for the column -- becomes bit arithmetic in registers:
//   applicability : (s & (need_set | need_clear)) == need_set      (two VALU instructions)
//   sign          : parity of popcount(s & sign_mask) (+ a per-term constant)
//   column        : off_dw[w >> Ns] + rk_up[w & (2^Ns-1)]   (two table lookups, w = s ^ flip)
// One lane owns one row; the term list is wave-uniform (scalar loads).  The gathers of v go through
// L2 / Infinity Cache (consecutive rows map to nearby columns: the rank maps are monotone).
// Integer/latency bound, not a bandwidth kernel: judged by iterations/s.
#include "kernels.hpp"

namespace edigpu {

constexpr int kDirNT = 1024;  // 2 workgroups/CU with the 2 x 2^Ns-entry rank tables staged in LDS (Ns <= 13)

// LZ (fused Lanczos step): hv = Q accumulates H*v on top of -beta*v_prev, and the workgroup writes its
// partials of <v|Q_new> and <Q_new|Q_new> (real view) to partial[blockIdx], partial[gridDim + blockIdx].
template <bool LDS_TABLES, bool LZ>
__global__ void __launch_bounds__(kDirNT)
    direct_rows_kernel(int64_t nrow, int64_t row_first, int ns, int norb, int nterms,
                       const int32_t* __restrict__ states, const int32_t* __restrict__ off_dw,
                       const int32_t* __restrict__ rk_up, const DirectTerm* __restrict__ terms,
                       const uint2* __restrict__ tests, const double* __restrict__ dtab,
                       const double* __restrict__ xtab, const double2* __restrict__ v_full,
                       double2* __restrict__ hv, double* __restrict__ partial) {
  extern __shared__ int32_t tabs[];  // [off_dw | rk_up] when LDS_TABLES
  __shared__ double red_a[kDirNT / 64], red_q[kDirNT / 64];
  double da = 0.0, dq = 0.0;
  const uint32_t lomask = (1u << ns) - 1u, impmask = (1u << norb) - 1u;
  if (LDS_TABLES) {
    const int n = 1 << ns;
    for (int i = threadIdx.x; i < n; i += kDirNT) {
      tabs[i] = off_dw[i];
      tabs[n + i] = rk_up[i];
    }
    __syncthreads();
  }
  const int32_t* __restrict__ t_off = LDS_TABLES ? tabs : off_dw;
  const int32_t* __restrict__ t_rk = LDS_TABLES ? tabs + (1 << ns) : rk_up;
  for (int64_t r = (int64_t)blockIdx.x * kDirNT + threadIdx.x; r < nrow; r += (int64_t)gridDim.x * kDirNT) {
    const uint32_t s = (uint32_t)states[r];
    // diagonal: one-body energies byte by byte + impurity interaction table
    const double dg = dtab[s & 255u] + dtab[256 + ((s >> 8) & 255u)] + dtab[512 + ((s >> 16) & 255u)] +
                      dtab[768 + (s >> 24)] + xtab[(((s >> ns) & impmask) << norb) | (s & impmask)];
    const double2 x0 = v_full[row_first + r];
    double ar = dg * x0.x, ai = dg * x0.y;
    // tests[t] = (need_set, need_set | need_clear), padded to a multiple of 4 with never-matching entries:
    // four applicability tests per batch of scalar loads; the rest of a term is fetched only where it applies
    for (int t0 = 0; t0 < nterms; t0 += 4) {
      uint2 tq[4];
#pragma unroll
      for (int u = 0; u < 4; u++) tq[u] = tests[t0 + u];      // wave-uniform: scalar loads
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if ((s & tq[u].y) == tq[u].x) {                       // v_and + v_cmp: the whole applicability test
          const DirectTerm tm = terms[t0 + u];
          const uint32_t w = s ^ tm.flip;
          const int64_t j = (int64_t)t_off[w >> ns] + t_rk[w & lomask];
          // sign = parity of the occupied levels the operators cross (+ a per-term constant), folded into x
          const int sg = (int)((uint32_t)((__popc(s & tm.sign_mask) + tm.csign) & 1) << 31);
          double2 x = v_full[j];
          x.x = __hiloint2double(__double2hiint(x.x) ^ sg, __double2loint(x.x));
          x.y = __hiloint2double(__double2hiint(x.y) ^ sg, __double2loint(x.y));
          ar = fma(tm.cre, x.x, ar);
          ar = fma(-tm.cim, x.y, ar);
          ai = fma(tm.cre, x.y, ai);
          ai = fma(tm.cim, x.x, ai);
        }
      }
    }
    if (LZ) {
      const double2 q = hv[r];
      ar += q.x;
      ai += q.y;
      da += x0.x * ar + x0.y * ai;
      dq += ar * ar + ai * ai;
    }
    hv[r] = make_double2(ar, ai);
  }
  if (LZ) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      da += __shfl_down(da, off, 64);
      dq += __shfl_down(dq, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      red_a[threadIdx.x >> 6] = da;
      red_q[threadIdx.x >> 6] = dq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double ta = 0.0, tq = 0.0;
#pragma unroll
      for (int i = 0; i < kDirNT / 64; i++) {
        ta += red_a[i];
        tq += red_q[i];
      }
      partial[blockIdx.x] = ta;
      partial[gridDim.x + blockIdx.x] = tq;
    }
  }
}

template <bool LZ>
static int launch_direct_t(const edigpu_sector* s, const double* v_full, double* hv, double* partial, int* np,
                           hipStream_t st) {
  int64_t nb = (s->nloc + kDirNT - 1) / kDirNT;
  if (nb > 256 * 2) nb = 256 * 2;  // persistent: two workgroups per CU sweep the rows
  if (np) *np = (int)nb;
  const size_t tab_bytes = (size_t)2 * sizeof(int32_t) << s->dir_ns;
  const double2* v2 = reinterpret_cast<const double2*>(v_full);
  double2* h2 = reinterpret_cast<double2*>(hv);
  if (tab_bytes <= 64 * 1024) {
    auto kern = direct_rows_kernel<true, LZ>;
    if (tab_bytes > 48 * 1024)
      EDIGPU_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(kDirNT), tab_bytes, st, s->nloc, s->row_first, s->dir_ns,
                       s->dir_norb, s->dir_nterms, s->d_dir_states, s->d_dir_offdw, s->d_dir_rkup, s->d_dir_terms,
                       s->d_dir_tests, s->d_dir_dtab, s->d_dir_xtab, v2, h2, partial);
  } else {
    hipLaunchKernelGGL((direct_rows_kernel<false, LZ>), dim3((unsigned)nb), dim3(kDirNT), 0, st, s->nloc,
                       s->row_first, s->dir_ns, s->dir_norb, s->dir_nterms, s->d_dir_states, s->d_dir_offdw,
                       s->d_dir_rkup, s->d_dir_terms, s->d_dir_tests, s->d_dir_dtab, s->d_dir_xtab, v2, h2, partial);
  }
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_direct(const edigpu_sector* s, const double* v_full, double* hv, hipStream_t st) {
  if (s->nloc == 0) return 0;
  return launch_direct_t<false>(s, v_full, hv, nullptr, nullptr, st);
}

// fused Lanczos step (single shard): Q += H*v with the alpha / <Q|Q> partials
int launch_direct_lanczos(const edigpu_sector* s, const double* v_full, double* q, double* partial, int* np,
                          hipStream_t st) {
  return launch_direct_t<true>(s, v_full, q, partial, np, st);
}

}  // namespace edigpu
