// kernels_build.hip -- on-device construction of the stored (SELL-64 + value dictionary) image of a
// superc / nonsu2 sector Hamiltonian from its on-the-fly description.
//
// Takes the place of the insert-and-search construction of spH0 (reference ed_buildH_superc_main /
// ed_buildH_nonsu2_main, ED_SUPERC/ED_HAMILTONIAN_SUPERC_STORED_HxV.f90:29-293,
// ED_NONSU2/ED_HAMILTONIAN_NONSU2_STORED_HxV.f90:29-175, through sp_insert_element,
// ED_SPARSE_MATRIX.f90:328-491), which the reference repeats in every tridiag_Hv_sector_* call and which
// dominates once H*v is fast.  The matrix elements are generated exactly as the
// direct kernel generates them (kernels_direct.hip: applicability masks, popcount signs, two-table
// ranking); instead of being multiplied into a vector they are written out as packed SELL words.
//
// Three passes, one lane = one row:
//   count : entries per row that fall inside / outside the shard's column window, slice widths by
//           atomicMax (64 rows per slice)
//   (host): exclusive scan of the slice widths (a few 10^4 integers)
//   fill  : regenerate the row, sort its entries by column in LDS (the SpMV's gathers of one slot then
//           land on few cache lines), write `column | id << 24` words column-major inside the slice;
//           the diagonal goes to its own stream.
// The value of an entry is +-(coefficient of its term): ids index a <= 256-entry dictionary built on the
// host from the term list (id 0 = padding).
#include "kernels.hpp"

namespace edigpu {

constexpr int kBuildNT = 256;


// f(global column, dictionary id) for every off-diagonal element of the row with basis state s
template <typename F>
__device__ inline void for_each_element(const BuildArgs& a, uint32_t s, F&& f) {
  const uint32_t lomask = (1u << a.ns) - 1u;
  for (int t = 0; t < a.nterms; t++) {
    const DirectTerm tm = a.terms[t];  // wave-uniform
    bool on, fwd = true;
    int cs;
    if (tm.pair) {
      fwd = (s & tm.need_set) != 0u;
      on = fwd != ((s & tm.need_clear) != 0u);
      cs = fwd ? (tm.csign & 1) : ((tm.csign >> 16) & 1);
    } else {
      on = (s & tm.need_set) == tm.need_set && (s & tm.need_clear) == 0u;
      cs = tm.csign & 1;
    }
    if (on) {
      const uint32_t w = s ^ tm.flip;
      const int64_t j = (int64_t)a.off_dw[w >> a.ns] + a.rk_up[w & lomask];
      const uint32_t neg = (uint32_t)((__popc(s & tm.sign_mask) + cs) & 1);
      f(j, (uint32_t)a.vid[2 * t + (fwd ? 0 : 1)] ^ neg);
    }
  }
}

// which = 0: loc block (columns inside [lo, hi)), 1: non-local block
__global__ void __launch_bounds__(kBuildNT)
    build_count_kernel(BuildArgs a, int32_t* __restrict__ width_loc, int32_t* __restrict__ width_non,
                       unsigned long long* __restrict__ totals) {
  const int64_t r = (int64_t)blockIdx.x * kBuildNT + threadIdx.x;
  int nl = 0, nn = 0;
  if (r < a.nrow) {
    const uint32_t s = (uint32_t)a.states[r];
    for_each_element(a, s, [&](int64_t j, uint32_t) {
      if (j >= a.lo && j < a.hi)
        nl++;
      else
        nn++;
    });
  }
  // slice = 64 consecutive rows = one wave
  int ml = nl, mn = nn;
  long long sl = nl, sn = nn;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ml = max(ml, __shfl_down(ml, off, 64));
    mn = max(mn, __shfl_down(mn, off, 64));
    sl += __shfl_down(sl, off, 64);
    sn += __shfl_down(sn, off, 64);
  }
  if ((threadIdx.x & 63) == 0 && r < a.nrow) {
    const int64_t slice = r >> 6;
    width_loc[slice] = ml;
    width_non[slice] = mn;
    atomicAdd(&totals[0], (unsigned long long)sl);
    atomicAdd(&totals[1], (unsigned long long)sn);
    atomicMax((unsigned int*)&totals[2], (unsigned int)ml);
    atomicMax((unsigned int*)&totals[3], (unsigned int)mn);
  }
}

template <int WHICH>
__global__ void __launch_bounds__(kBuildNT)
    build_fill_kernel(BuildArgs a, int maxlen, const int32_t* __restrict__ sptr, uint32_t* __restrict__ pk,
                      double2* __restrict__ diag) {
  extern __shared__ uint32_t buf[];  // [maxlen][kBuildNT]: thread-private columns, conflict-free
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t r = (int64_t)blockIdx.x * kBuildNT + tid;
  const int64_t slice = r >> 6;
  if (slice * 64 >= a.nrow) return;  // whole wave past the end (no barrier in this kernel)
  int n = 0;
  if (r < a.nrow) {
    const uint32_t s = (uint32_t)a.states[r];
    for_each_element(a, s, [&](int64_t j, uint32_t id) {
      const bool in = j >= a.lo && j < a.hi;
      if (in == (WHICH == 0)) {
        const uint32_t c = (uint32_t)(WHICH == 0 ? j - a.lo : j);
        if (n < maxlen) buf[n * kBuildNT + tid] = (c << 8) | id;  // sort key: column, then id
        n++;
      }
    });
    if (n > maxlen) n = maxlen;  // cannot happen: maxlen is the measured maximum
    // insertion sort by column
    for (int i = 1; i < n; i++) {
      const uint32_t key = buf[i * kBuildNT + tid];
      int j = i - 1;
      while (j >= 0 && buf[j * kBuildNT + tid] > key) {
        buf[(j + 1) * kBuildNT + tid] = buf[j * kBuildNT + tid];
        j--;
      }
      buf[(j + 1) * kBuildNT + tid] = key;
    }
    if (WHICH == 0 && diag != nullptr) {
      const uint32_t impmask = (1u << a.norb) - 1u;
      const double dg = a.dtab[s & 255u] + a.dtab[256 + ((s >> 8) & 255u)] + a.dtab[512 + ((s >> 16) & 255u)] +
                        a.dtab[768 + (s >> 24)] + a.xtab[(((s >> a.ns) & impmask) << a.norb) | (s & impmask)];
      diag[r] = make_double2(dg, 0.0);
    }
  }
  const int32_t b = sptr[slice], w = sptr[slice + 1] - b;
  // padding repeats the row's last column (same cache line) with id 0 = 0.0
  const uint32_t padc = n > 0 ? (buf[(n - 1) * kBuildNT + tid] >> 8) : 0u;
  for (int k = 0; k < w; k++) {
    uint32_t word = padc;
    if (k < n) {
      const uint32_t x = buf[k * kBuildNT + tid];
      word = (x >> 8) | (x << 24);
    }
    pk[((int64_t)b + k) * 64 + lane] = word;
  }
}

int launch_build_count(const BuildArgs& a, int32_t* width_loc, int32_t* width_non, unsigned long long* totals,
                       hipStream_t st) {
  const int64_t nb = (a.nrow + kBuildNT - 1) / kBuildNT;
  hipLaunchKernelGGL(build_count_kernel, dim3((unsigned)nb), dim3(kBuildNT), 0, st, a, width_loc, width_non, totals);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_build_fill(const BuildArgs& a, int which, int maxlen, const int32_t* sptr, uint32_t* pk, double* diag,
                      hipStream_t st) {
  if (maxlen < 1) maxlen = 1;
  const size_t lds = (size_t)maxlen * kBuildNT * sizeof(uint32_t);
  const int64_t nb = (a.nrow + kBuildNT - 1) / kBuildNT;
  if (which == 0) {
    auto kern = build_fill_kernel<0>;
    if (ensure_dynamic_lds((const void*)kern, lds)) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(kBuildNT), lds, st, a, maxlen, sptr, pk,
                       reinterpret_cast<double2*>(diag));
  } else {
    auto kern = build_fill_kernel<1>;
    if (ensure_dynamic_lds((const void*)kern, lds)) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(kBuildNT), lds, st, a, maxlen, sptr, pk,
                       reinterpret_cast<double2*>(diag));
  }
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

}  // namespace edigpu
