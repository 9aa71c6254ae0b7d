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
#include <cstdlib>

#include "kernels.hpp"

namespace edigpu {

constexpr int kDirMaxTerms = 512;  // COMPACT variant: term table kept in LDS (14 KiB)
constexpr int kDirNT = 1024;  // 2 workgroups/CU with the 2 x 2^Ns-entry rank tables staged in LDS (Ns <= 13)

// LZ (fused Lanczos step): hv = Q accumulates H*v on top of -beta*v_prev, and the workgroup writes its
// partials of <v|Q_new> and <Q_new|Q_new> (real view) to partial[blockIdx], partial[gridDim + blockIdx].
// COMPACT: instead of visiting the terms in order with a quarter of the lanes active per gather, a lane first
// collects the applicability of 32 terms in a bit mask and then pops its own set bits, two per round: every
// gather instruction has all lanes active and two are in flight per lane -- about 4x fewer dependent gather
// rounds per row.  The per-term data then differ from lane to lane and come from an LDS copy of the term list.
// (two workgroups of 16 waves per CU need 8 waves per SIMD, i.e. at most 64 VGPRs: the fused-step variant would take 68
// and run at half the occupancy -- 1.70 against 1.39 ms on config 5 -- without the second launch bound)
template <bool LDS_TABLES, bool LZ, bool COMPACT>
__global__ void __launch_bounds__(kDirNT, 8)
    direct_rows_kernel(int64_t nrow, int64_t row_first, int ns, int norb, int nterms,
                       const int32_t* __restrict__ states, const int32_t* __restrict__ off_dw,
                       const int32_t* __restrict__ rk_up, const DirectTerm* __restrict__ terms,
                       const uint2* __restrict__ tests, const double* __restrict__ dtab,
                       const double* __restrict__ xtab, const double2* __restrict__ v_full,
                       double2* __restrict__ hv, double* __restrict__ partial, const double* __restrict__ sig) {
  const double sgm = (LZ && sig) ? sig[0] : 0.0;  // <Q|Q> accumulated about the previous alpha, see k_finalize_ab
  extern __shared__ int32_t tabs[];  // [off_dw | rk_up] when LDS_TABLES
  __shared__ double red_a[kDirNT / 64], red_q[kDirNT / 64], red_n[kDirNT / 64];
  double da = 0.0, dq = 0.0, dn = 0.0;
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
  // COMPACT: term table in LDS behind the rank tables: flip | sign_mask | csign (256 words each) | coef (256 x 16 B)
  uint32_t* t_flip = reinterpret_cast<uint32_t*>(tabs + (LDS_TABLES ? (2 << ns) : 0));
  uint32_t* t_smask = t_flip + kDirMaxTerms;
  uint32_t* t_csign = t_smask + kDirMaxTerms;
  double2* t_coef = reinterpret_cast<double2*>(t_csign + kDirMaxTerms);
  if (COMPACT) {
    for (int t = threadIdx.x; t < nterms; t += kDirNT) {
      const DirectTerm tm = terms[t];
      t_flip[t] = tm.flip;
      t_smask[t] = tm.sign_mask;
      t_csign[t] = (uint32_t)tm.csign;
      t_coef[t] = make_double2(tm.cre, tm.cim);
    }
    __syncthreads();
  }
  for (int64_t r = (int64_t)blockIdx.x * kDirNT + threadIdx.x; r < nrow; r += (int64_t)gridDim.x * kDirNT) {
    const uint32_t s = (uint32_t)states[r];
    // diagonal: one-body energies byte by byte + impurity interaction table
    const double dg = dtab[s & 255u] + dtab[256 + ((s >> 8) & 255u)] + dtab[512 + ((s >> 16) & 255u)] +
                      dtab[768 + (s >> 24)] + xtab[(((s >> ns) & impmask) << norb) | (s & impmask)];
    const double2 x0 = v_full[row_first + r];
    double ar = dg * x0.x, ai = dg * x0.y;
    if (COMPACT) {
      for (int g0 = 0; g0 < nterms; g0 += 32) {
        uint32_t mw = 0;
#pragma unroll
        for (int b4 = 0; b4 < 32; b4 += 4) {
          uint2 tq[4];
#pragma unroll
          for (int u = 0; u < 4; u++) tq[u] = tests[g0 + b4 + u];  // padded to a multiple of 32
#pragma unroll
          for (int u = 0; u < 4; u++) mw |= ((s & tq[u].y) == tq[u].x) ? (1u << (b4 + u)) : 0u;
        }
        while (__any(mw != 0u)) {
          // two applicable terms of this lane (the second may be missing: weight 0, own row)
          const bool h1 = mw != 0u;
          const int b1 = h1 ? __ffs(mw) - 1 : 0;
          mw &= mw - 1u;
          const bool h2 = mw != 0u;
          const int b2 = h2 ? __ffs(mw) - 1 : 0;
          mw &= mw - 1u;
          const int t1 = g0 + b1, t2 = g0 + b2;
          const uint32_t w1 = s ^ t_flip[t1], w2 = s ^ t_flip[t2];
          const int64_t j1 = h1 ? (int64_t)t_off[w1 >> ns] + t_rk[w1 & lomask] : row_first + r;
          const int64_t j2 = h2 ? (int64_t)t_off[w2 >> ns] + t_rk[w2 & lomask] : row_first + r;
          double2 x1 = v_full[j1];
          double2 x2 = v_full[j2];
          const int sg1 = (int)((uint32_t)((__popc(s & t_smask[t1]) + (int)t_csign[t1]) & 1) << 31);
          const int sg2 = (int)((uint32_t)((__popc(s & t_smask[t2]) + (int)t_csign[t2]) & 1) << 31);
          double2 c1 = t_coef[t1], c2 = t_coef[t2];
          if (!h1) c1 = make_double2(0.0, 0.0);
          if (!h2) c2 = make_double2(0.0, 0.0);
          x1.x = __hiloint2double(__double2hiint(x1.x) ^ sg1, __double2loint(x1.x));
          x1.y = __hiloint2double(__double2hiint(x1.y) ^ sg1, __double2loint(x1.y));
          x2.x = __hiloint2double(__double2hiint(x2.x) ^ sg2, __double2loint(x2.x));
          x2.y = __hiloint2double(__double2hiint(x2.y) ^ sg2, __double2loint(x2.y));
          ar = fma(c1.x, x1.x, ar);
          ar = fma(-c1.y, x1.y, ar);
          ai = fma(c1.x, x1.y, ai);
          ai = fma(c1.y, x1.x, ai);
          ar = fma(c2.x, x2.x, ar);
          ar = fma(-c2.y, x2.y, ar);
          ai = fma(c2.x, x2.y, ai);
          ai = fma(c2.y, x2.x, ai);
        }
      }
    }
    // tests[t] = (need_set, need_set | need_clear), padded with never-matching entries:
    // four applicability tests per batch of scalar loads; the rest of a term is fetched only where it applies
    for (int t0 = COMPACT ? nterms : 0; t0 < nterms; t0 += 4) {
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
      dq += (ar - sgm * x0.x) * (ar - sgm * x0.x) + (ai - sgm * x0.y) * (ai - sgm * x0.y);
      dn += x0.x * x0.x + x0.y * x0.y;
    }
    hv[r] = make_double2(ar, ai);
  }
  if (LZ) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      da += __shfl_down(da, off, 64);
      dq += __shfl_down(dq, off, 64);
      dn += __shfl_down(dn, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      red_a[threadIdx.x >> 6] = da;
      red_q[threadIdx.x >> 6] = dq;
      red_n[threadIdx.x >> 6] = dn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double ta = 0.0, tq = 0.0, tn = 0.0;
#pragma unroll
      for (int i = 0; i < kDirNT / 64; i++) {
        ta += red_a[i];
        tq += red_q[i];
        tn += red_n[i];
      }
      partial[blockIdx.x] = ta;
      partial[gridDim.x + blockIdx.x] = tq;
      partial[2 * gridDim.x + blockIdx.x] = tn;
    }
  }
}

template <bool LZ>
static int launch_direct_t(const edigpu_sector* s, const double* v_full, double* hv, double* partial, int64_t cap,
                           int* np, const double* sig, hipStream_t st) {
  const int64_t nrow = s->nph > 0 ? s->dim_el : s->nloc;  // phonon sectors: one electronic block per launch
  int64_t nb = (nrow + kDirNT - 1) / kDirNT;
  static const int wgs_per_cu = getenv("EDIGPU_DIRECT_WGS") ? atoi(getenv("EDIGPU_DIRECT_WGS")) : 2;
  if (nb > 256 * wgs_per_cu) nb = 256 * wgs_per_cu;  // persistent workgroups sweep the rows
  if (LZ && 3 * nb > cap) {
    set_error("launch_direct_lanczos: partial buffer too small");
    return 1;
  }
  if (np) *np = (int)nb;
  const size_t tab_bytes = (size_t)2 * sizeof(int32_t) << s->dir_ns;
  const size_t term_bytes = (size_t)kDirMaxTerms * (3 * sizeof(uint32_t) + sizeof(double2));
  static const bool no_compact = getenv("EDIGPU_DIRECT_TERMORDER") != nullptr;
  const bool compact = !no_compact && s->dir_nterms <= kDirMaxTerms;
  const double2* v2 = reinterpret_cast<const double2*>(v_full);
  double2* h2 = reinterpret_cast<double2*>(hv);
#define EDIGPU_LAUNCH_DIRECT(LT, CP, LDSB)                                                                       \
  do {                                                                                                           \
    auto kern = direct_rows_kernel<LT, LZ, CP>;                                                                  \
    if (ensure_dynamic_lds((const void*)kern, (LDSB))) return 1; \
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(kDirNT), (LDSB), st, nrow, s->row_first, s->dir_ns,        \
                       s->dir_norb, s->dir_nterms, s->d_dir_states, s->d_dir_offdw, s->d_dir_rkup,              \
                       s->d_dir_terms, s->d_dir_tests, s->d_dir_dtab, s->d_dir_xtab, v2, h2, partial, sig);     \
  } while (0)
  if (tab_bytes <= 64 * 1024) {
    if (compact)
      EDIGPU_LAUNCH_DIRECT(true, true, tab_bytes + term_bytes);
    else
      EDIGPU_LAUNCH_DIRECT(true, false, tab_bytes);
  } else {
    if (compact)
      EDIGPU_LAUNCH_DIRECT(false, true, term_bytes);
    else
      EDIGPU_LAUNCH_DIRECT(false, false, (size_t)0);
  }
#undef EDIGPU_LAUNCH_DIRECT
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_direct(const edigpu_sector* s, const double* v_full, double* hv, hipStream_t st) {
  if (s->nloc == 0) return 0;
  return launch_direct_t<false>(s, v_full, hv, nullptr, 0, nullptr, nullptr, st);
}

// fused Lanczos step (single shard): Q += H*v with the alpha / <Q|Q> partials
int launch_direct_lanczos(const edigpu_sector* s, const double* v_full, double* q, double* partial, int64_t cap, int* np,
                          const double* sig, hipStream_t st) {
  return launch_direct_t<true>(s, v_full, q, partial, cap, np, sig, st);
}

}  // namespace edigpu
