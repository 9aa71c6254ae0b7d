// kernels_panel.hip -- normal mode, kernel B: hv[:, panel] += (Hdw (x) 1) V[:, panel] + factored Hnd.
//
// Takes the place of the "down" and "non-local" loops of spMatVec_normal_main (reference
// ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650; MPI form :765-929, where the reference
// transposes the vector with MPI_Alltoallv to make this term contiguous).
//
// The down term couples whole rows of V[idw][iup]: output row idw reads ~7 neighbour rows at the
// same columns.  A panel of <= 64 columns over all DimDw rows fits one XCD's 4 MiB L2, so all
// workgroups of an XCD (blockIdx % 8, the observed round-robin dispatch; another placement only
// costs speed) sweep the same panel at the same time: the neighbour reads are L2 hits and the
// fabric sees V once.  A wave owns one output row at a time (64 lanes = 64 panel columns, one
// 512-byte segment); its neighbour list is wave-uniform (scalar loads).
//
// Measured (round 1, cfg2): issuing the whole list of two rows as one batch of independent loads
// (6..8 in flight per row) is 10-20 % SLOWER than this two-entries-at-a-time loop: the kernel is
// bound by the fabric/L2 traffic of the sweep, not by load latency, and a deeper queue only spreads
// the workgroups of a panel further apart.  Ablation of this kernel on config 2 (99 us under rocprofv3): no
// result store 94, no result load 94, neither 82, no neighbour gathers 40, nothing but the list handling 18
// -- i.e. ~60 us are the L2 gathers.  A variant with two columns per lane (16-byte gathers, a half-wave per
// row, lists staged in LDS: half as many gather instructions) measured 1.8x SLOWER (175 us).  What does pay is
// two columns per lane with a FULL wave per row (normal_dw_panel2_kernel below: 1024-byte segments, 16-byte
// gathers, panels twice as wide): 95 -> 81 us.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lz_finalize.hpp"
#include "normal_args.hpp"

namespace edigpu {

struct PanelArgs {
  int npanels, width, blocks_per_panel, rows_per_block;
  // COLS (column shard of the transposed exchange, see launch_dw_panel_cols): the buffers hold the columns
  // [col_first - halo, col_first + ncol + halo) of every row at the given row stride (>= ncol + 2 * halo)
  int64_t col_first, ncol, stride;
  int halo;
  int tile_rows;  // LDS-tiled form: rows of the largest chunk
  int list_cap;   // LDS-tiled form: list entries of the fullest chunk (multiple of 4)
};

constexpr int kPanelNT = 512;
constexpr int kMaxNdTerms = 16;

// ALPHA: also accumulate <v|hv_new> over the local rows (v = v_full rows of this shard) and write
// one partial per workgroup (deterministic two-stage reduction, see kernels_lanczos.hip)
// COLS: column-shard form -- hv is written (not accumulated), columns are shard-local (see PanelArgs)
template <bool DO_DW, bool DO_ND, bool ALPHA, bool COLS = false>
__global__ void __launch_bounds__(kPanelNT)
    normal_dw_panel_kernel(NormalArgs a, PanelArgs p, const double* __restrict__ v_full,
                           double* __restrict__ hv) {
  __shared__ double red[3 * (kPanelNT / 64)];
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int panel = (k / p.blocks_per_panel) * 8 + x;
  __shared__ double lzsh[ALPHA ? 3 * kPanelNT + 2 : 1];  // in-kernel finalize of the fused step (lz_finalize.hpp)
  if (ALPHA) {
    if (panel >= p.npanels || a.scal[SC_STOP] != 0.0) {
      if (threadIdx.x == 0) {
        a.partial[blockIdx.x] = 0.0;
        a.partial[gridDim.x + blockIdx.x] = 0.0;
        a.partial[2 * gridDim.x + blockIdx.x] = 0.0;
      }
      // (a workgroup of the grid's padding still counts as arrived)
      if (a.lz_counter)
        lz_finalize_if_last<kPanelNT>(a.lz_counter, a.partial, 0.0, 0.0, 0.0, v_full, hv, a.lz_len, const_cast<double*>(a.scal),
                                      a.lz_nlanc, lzsh);
      return;
    }
  }
  if (panel >= p.npanels) return;
  double asum = 0.0, qsum = 0.0, nsum = 0.0;  // <v|w>, sum (w - sg v)^2, <v|v> (k_finalize_ab)
  // <Q|Q> is accumulated about the previous alpha (sg): beta^2 = sum (Q - sg v)^2 - (alpha - sg)^2 stays well
  // conditioned when the spectrum sits far from zero (|alpha| >> beta), see k_finalize_ab
  const double sg = ALPHA ? a.scal[SC_ALPHA] : 0.0;
  const int chunk = k % p.blocks_per_panel;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // DimUp doubles as the row stride of v_full / hv; c, cc index into a row
  const int64_t NCol = COLS ? p.ncol : a.dim_up;
  const int64_t DimUp = COLS ? p.stride : a.dim_up;
  const int64_t c0 = (int64_t)panel * p.width + lane;
  const bool ok = lane < p.width && c0 < NCol;
  const int64_t c = c0 + (COLS ? p.halo : 0);
  const int64_t cc = ok ? c : NCol - 1 + (COLS ? p.halo : 0);
  int64_t rend = (int64_t)(chunk + 1) * p.rows_per_block;
  if (rend > a.dw_count) rend = a.dw_count;
  constexpr int NW = kPanelNT / 64;
  // the lane's partner column of every Hnd term (depends on the column only) in LDS; a thread reads
  // back its own words only, so no barrier is needed
  __shared__ uint32_t ju_s[(DO_ND ? kMaxNdTerms : 1) * kPanelNT];
  if (DO_ND)
    for (int t = 0; t < a.nterms; t++) {
      if (COLS) {
        // global partner column -> position in the shard's rows (inside the halo by construction)
        const uint32_t jt = a.jup[(int64_t)t * a.dim_up + (p.col_first + cc - p.halo)];
        ju_s[t * kPanelNT + threadIdx.x] =
            jt == 0xFFFFFFFFu ? jt
                              : ((jt & 0x80000000u) | (uint32_t)((int64_t)(jt & 0x7FFFFFFFu) - p.col_first + p.halo));
      } else {
        ju_s[t * kPanelNT + threadIdx.x] = a.jup[(int64_t)t * DimUp + cc];
      }
    }
  for (int64_t r = (int64_t)chunk * p.rows_per_block + wave; r < rend; r += 2 * NW) {
    // two rows per iteration: independent load streams in flight
    const int64_t r2 = r + NW;
    const bool two = r2 < rend;
    const int64_t g = a.dw_first + r;
    double acc0 = COLS ? 0.0 : hv[r * DimUp + cc];
    double acc1 = (!COLS && two) ? hv[r2 * DimUp + cc] : 0.0;
    // ALPHA: the lane's own elements of v, issued with the result loads instead of after the gather chain
    double own0 = 0.0, own1 = 0.0;
    if (ALPHA) {
      own0 = v_full[(a.dw_first + r) * DimUp + cc];
      own1 = two ? v_full[(a.dw_first + r2) * DimUp + cc] : 0.0;
    }
    if (DO_DW && !DO_ND) {
      const int32_t b0 = a.dw_rowptr[g], e0 = a.dw_rowptr[g + 1];
      int32_t b1 = 0, e1 = 0;
      if (two) {
        b1 = a.dw_rowptr[g + NW];
        e1 = a.dw_rowptr[g + NW + 1];
      }
#pragma unroll 4
      for (int32_t jj = b0; jj < e0; jj++)
        acc0 += a.dw_val[jj] * v_full[(int64_t)a.dw_col[jj] * DimUp + cc];
#pragma unroll 4
      for (int32_t jj = b1; jj < e1; jj++)
        acc1 += a.dw_val[jj] * v_full[(int64_t)a.dw_col[jj] * DimUp + cc];
    }
    if (DO_ND) {
      // merged per-LOCAL-row list: down hops (tag 0) followed by the applicable Hnd terms
      // (tag = term id + 1 in bits 24..30 of the column word, weight = +-coef of the down side)
      const int32_t b0 = a.mx_rowptr[r], e0 = a.mx_rowptr[r + 1];
      int32_t b1 = 0, e1 = 0;
      if (two) {
        b1 = a.mx_rowptr[r2];
        e1 = a.mx_rowptr[r2 + 1];
      }
#pragma unroll 2
      for (int32_t jj = b0; jj < e0; jj++) {
        const uint32_t cw = (uint32_t)a.mx_col[jj];
        const int tag = (int)(cw >> 24);
        double w = a.mx_val[jj];
        int64_t col = cc;
        if (tag) {
          const uint32_t jt = ju_s[(tag - 1) * kPanelNT + threadIdx.x];
          const bool v = jt != 0xFFFFFFFFu;
          w = v ? ((jt >> 31) ? -w : w) : 0.0;
          col = v ? (int64_t)(jt & 0x7FFFFFFFu) : cc;
        } else if (!DO_DW) {
          w = 0.0;
        }
        acc0 += w * v_full[(int64_t)(cw & 0xFFFFFFu) * DimUp + col];
      }
#pragma unroll 2
      for (int32_t jj = b1; jj < e1; jj++) {
        const uint32_t cw = (uint32_t)a.mx_col[jj];
        const int tag = (int)(cw >> 24);
        double w = a.mx_val[jj];
        int64_t col = cc;
        if (tag) {
          const uint32_t jt = ju_s[(tag - 1) * kPanelNT + threadIdx.x];
          const bool v = jt != 0xFFFFFFFFu;
          w = v ? ((jt >> 31) ? -w : w) : 0.0;
          col = v ? (int64_t)(jt & 0x7FFFFFFFu) : cc;
        } else if (!DO_DW) {
          w = 0.0;
        }
        acc1 += w * v_full[(int64_t)(cw & 0xFFFFFFu) * DimUp + col];
      }
    }
    if (ok) {
      hv[r * DimUp + c] = acc0;
      if (two) hv[r2 * DimUp + c] = acc1;
      if (ALPHA) {
        asum += own0 * acc0;
        qsum += (acc0 - sg * own0) * (acc0 - sg * own0);
        nsum += own0 * own0;
        if (two) {
          asum += own1 * acc1;
          qsum += (acc1 - sg * own1) * (acc1 - sg * own1);
          nsum += own1 * own1;
        }
      }
    }
  }
  if (ALPHA) {
    // per-workgroup partials of <v|Q> and <Q|Q> (the latter gives beta^2 = <Q|Q> - alpha^2)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      asum += __shfl_down(asum, off, 64);
      qsum += __shfl_down(qsum, off, 64);
      nsum += __shfl_down(nsum, off, 64);
    }
    if (lane == 0) {
      red[wave] = asum;
      red[kPanelNT / 64 + wave] = qsum;
      red[2 * (kPanelNT / 64) + wave] = nsum;
    }
    __syncthreads();
    double t = 0.0, q = 0.0, n = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < kPanelNT / 64; i++) {
        t += red[i];
        q += red[kPanelNT / 64 + i];
        n += red[2 * (kPanelNT / 64) + i];
      }
      if (!a.lz_counter) {
        a.partial[blockIdx.x] = t;
        a.partial[gridDim.x + blockIdx.x] = q;
        a.partial[2 * gridDim.x + blockIdx.x] = n;
      }
    }
    if (a.lz_counter)
      lz_finalize_if_last<kPanelNT>(a.lz_counter, a.partial, t, q, n, v_full, hv, a.lz_len, const_cast<double*>(a.scal),
                                    a.lz_nlanc, lzsh);
  }
}

// Two adjacent columns per lane (default for large sectors with an even DimUp; EDIGPU_PANEL_VEC2=0 switches it
// off): a wave still owns one output row, now as a
// 1024-byte segment of up to 128 columns, and the down-hop gathers are 16-byte loads -- the L2 serves 16-byte
// accesses at ~1.5-1.8x the rate of 8-byte ones (MI355X_MICROARCH.md, scope/cache-policy table).  The panel is
// twice as wide, i.e. twice the L2 footprint.  Needs an even DimUp (16-byte aligned rows).
// EDGE: odd DimUp -- rows are only 8-byte aligned (d2u below: the hardware takes 16-byte loads at 4-byte alignment)
// and the last column has no right-hand partner: that one lane falls back to 8-byte accesses.
struct alignas(8) d2u {
  double x, y;
};

template <bool DO_ND, bool ALPHA, bool EDGE>
__global__ void __launch_bounds__(kPanelNT)
    normal_dw_panel2_kernel(NormalArgs a, PanelArgs p, const double* __restrict__ v_full, double* __restrict__ hv) {
  __shared__ double red[3 * (kPanelNT / 64)];
  extern __shared__ uint32_t ju2[];  // [2 * nterms][kPanelNT]: partner columns of the lane's two columns
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int panel = (k / p.blocks_per_panel) * 8 + x;
  __shared__ double lzsh[ALPHA ? 3 * kPanelNT + 2 : 1];  // in-kernel finalize of the fused step (lz_finalize.hpp)
  if (ALPHA) {
    if (panel >= p.npanels || a.scal[SC_STOP] != 0.0) {
      if (threadIdx.x == 0) {
        a.partial[blockIdx.x] = 0.0;
        a.partial[gridDim.x + blockIdx.x] = 0.0;
        a.partial[2 * gridDim.x + blockIdx.x] = 0.0;
      }
      // (a workgroup of the grid's padding still counts as arrived)
      if (a.lz_counter)
        lz_finalize_if_last<kPanelNT>(a.lz_counter, a.partial, 0.0, 0.0, 0.0, v_full, hv, a.lz_len, const_cast<double*>(a.scal),
                                      a.lz_nlanc, lzsh);
      return;
    }
  }
  if (panel >= p.npanels) return;
  double asum = 0.0, qsum = 0.0, nsum = 0.0;  // <v|w>, sum (w - sg v)^2, <v|v> (k_finalize_ab)
  const double sg = ALPHA ? a.scal[SC_ALPHA] : 0.0;  // see normal_dw_panel_kernel
  const int chunk = k % p.blocks_per_panel;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t DimUp = a.dim_up;
  const int64_t c = (int64_t)panel * p.width + 2 * lane;  // even (the width is even)
  const bool ok = 2 * lane < p.width && c < DimUp;
  const int64_t cc = ok ? c : (EDGE ? DimUp - 1 : DimUp - 2);
  const bool pair = !EDGE || cc + 1 < DimUp;  // false only on the lane that owns the last column of an odd DimUp
  auto ld2 = [&](const double* q) -> double2 {
    if (!EDGE) return *reinterpret_cast<const double2*>(q);
    if (pair) {
      const d2u t = *reinterpret_cast<const d2u*>(q);
      return make_double2(t.x, t.y);
    }
    return make_double2(q[0], 0.0);
  };
  auto st2 = [&](double* q, double2 t) {
    if (!EDGE) {
      *reinterpret_cast<double2*>(q) = t;
    } else if (pair) {
      d2u u;
      u.x = t.x;
      u.y = t.y;
      *reinterpret_cast<d2u*>(q) = u;
    } else {
      q[0] = t.x;
    }
  };
  int64_t rend = (int64_t)(chunk + 1) * p.rows_per_block;
  if (rend > a.dw_count) rend = a.dw_count;
  constexpr int NW = kPanelNT / 64;
  // Hnd partner of a column, per term: 0xFFFFFFFF = the term does not apply to this column; bit 31 = sign;
  // inside this wave's segment: bits 0..6 = (lane << 1) | component of the lane that holds the partner column
  // (the partner row's segment is loaded once, coalesced, and the element fetched with a cross-lane read);
  // bit 30 = outside the segment (panel edge): bits 0..29 = the column, fetched with its own 8-byte load
  const int64_t pbase = (int64_t)panel * p.width;
  auto encode = [&](uint32_t jt) -> uint32_t {
    if (jt == 0xFFFFFFFFu) return jt;
    const int64_t rel = (int64_t)(jt & 0x7FFFFFFFu) - pbase;
    if (rel >= 0 && rel < p.width) return (jt & 0x80000000u) | (uint32_t)rel;
    return (jt & 0x80000000u) | 0x40000000u | (jt & 0x3FFFFFFFu);
  };
  if (DO_ND)
    for (int t = 0; t < a.nterms; t++) {
      ju2[(2 * t) * kPanelNT + threadIdx.x] = encode(a.jup[(int64_t)t * DimUp + cc]);
      ju2[(2 * t + 1) * kPanelNT + threadIdx.x] = pair ? encode(a.jup[(int64_t)t * DimUp + cc + 1]) : 0xFFFFFFFFu;
    }
  auto row_sum = [&](int64_t r, double2 acc) -> double2 {
    if (!DO_ND) {
      const int64_t g = a.dw_first + r;
      const int32_t b0 = a.dw_rowptr[g], e0 = a.dw_rowptr[g + 1];
#pragma unroll 4
      for (int32_t jj = b0; jj < e0; jj++) {
        const double w = a.dw_val[jj];
        const double2 y = ld2(&v_full[(int64_t)a.dw_col[jj] * DimUp + cc]);
        acc.x += w * y.x;
        acc.y += w * y.y;
      }
    } else {
      // merged list of the row: its down hops first (as many as the row of Hdw holds), then the applicable Hnd terms
      const int32_t b0 = a.mx_rowptr[r], e0 = a.mx_rowptr[r + 1];
      const int64_t g = a.dw_first + r;
      const int32_t mid = b0 + (a.dw_rowptr[g + 1] - a.dw_rowptr[g]);
#pragma unroll 4
      for (int32_t jj = b0; jj < mid; jj++) {
        const double w = a.mx_val[jj];
        const double2 y = ld2(&v_full[(int64_t)((uint32_t)a.mx_col[jj] & 0xFFFFFFu) * DimUp + cc]);
        acc.x += w * y.x;
        acc.y += w * y.y;
      }
      for (int32_t jj = mid; jj < e0; jj++) {
        const uint32_t cw = (uint32_t)a.mx_col[jj];
        const int tag = (int)(cw >> 24);  // wave-uniform, >= 1
        const double w = a.mx_val[jj];
        const int64_t base = (int64_t)(cw & 0xFFFFFFu) * DimUp;
        const uint32_t j0 = ju2[(2 * (tag - 1)) * kPanelNT + threadIdx.x];
        const uint32_t j1 = ju2[(2 * (tag - 1) + 1) * kPanelNT + threadIdx.x];
        const bool v0 = j0 != 0xFFFFFFFFu, v1 = j1 != 0xFFFFFFFFu;
        const double w0 = v0 ? ((j0 >> 31) ? -w : w) : 0.0, w1 = v1 ? ((j1 >> 31) ? -w : w) : 0.0;
        // the partner row's segment, coalesced like a down hop; the partner columns sit a few lanes away
        const double2 y = ld2(&v_full[base + cc]);
        const int l0 = (int)((j0 >> 1) & 63u), l1 = (int)((j1 >> 1) & 63u);
        const double s0x = __shfl(y.x, l0, 64), s0y = __shfl(y.y, l0, 64);
        const double s1x = __shfl(y.x, l1, 64), s1y = __shfl(y.y, l1, 64);
        double p0 = (j0 & 1u) ? s0y : s0x, p1 = (j1 & 1u) ? s1y : s1x;
        if (v0 && (j0 & 0x40000000u)) p0 = v_full[base + (int64_t)(j0 & 0x3FFFFFFFu)];  // panel edge
        if (v1 && (j1 & 0x40000000u)) p1 = v_full[base + (int64_t)(j1 & 0x3FFFFFFFu)];
        acc.x += w0 * p0;
        acc.y += w1 * p1;
      }
    }
    return acc;
  };
  for (int64_t r = (int64_t)chunk * p.rows_per_block + wave; r < rend; r += 2 * NW) {
    const int64_t r2 = r + NW;
    const bool two = r2 < rend;
    double2 acc0 = ld2(&hv[r * DimUp + cc]);
    double2 acc1 = two ? ld2(&hv[r2 * DimUp + cc]) : make_double2(0.0, 0.0);
    double2 own0 = make_double2(0.0, 0.0), own1 = own0;
    if (ALPHA) {
      own0 = ld2(&v_full[(a.dw_first + r) * DimUp + cc]);
      if (two) own1 = ld2(&v_full[(a.dw_first + r2) * DimUp + cc]);
    }
    acc0 = row_sum(r, acc0);
    if (two) acc1 = row_sum(r2, acc1);
    if (ok) {
      st2(&hv[r * DimUp + c], acc0);
      if (two) st2(&hv[r2 * DimUp + c], acc1);
      if (ALPHA) {
        const double d0x = acc0.x - sg * own0.x, d0y = acc0.y - sg * own0.y;
        asum += own0.x * acc0.x + own0.y * acc0.y;
        qsum += d0x * d0x + d0y * d0y;
        nsum += own0.x * own0.x + own0.y * own0.y;
        if (two) {
          const double d1x = acc1.x - sg * own1.x, d1y = acc1.y - sg * own1.y;
          asum += own1.x * acc1.x + own1.y * acc1.y;
          qsum += d1x * d1x + d1y * d1y;
          nsum += own1.x * own1.x + own1.y * own1.y;
        }
      }
    }
  }
  if (ALPHA) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      asum += __shfl_down(asum, off, 64);
      qsum += __shfl_down(qsum, off, 64);
      nsum += __shfl_down(nsum, off, 64);
    }
    if (lane == 0) {
      red[wave] = asum;
      red[kPanelNT / 64 + wave] = qsum;
      red[2 * (kPanelNT / 64) + wave] = nsum;
    }
    __syncthreads();
    double t = 0.0, q = 0.0, n = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < kPanelNT / 64; i++) {
        t += red[i];
        q += red[kPanelNT / 64 + i];
        n += red[2 * (kPanelNT / 64) + i];
      }
      if (!a.lz_counter) {
        a.partial[blockIdx.x] = t;
        a.partial[gridDim.x + blockIdx.x] = q;
        a.partial[2 * gridDim.x + blockIdx.x] = n;
      }
    }
    if (a.lz_counter)
      lz_finalize_if_last<kPanelNT>(a.lz_counter, a.partial, t, q, n, v_full, hv, a.lz_len, const_cast<double*>(a.scal),
                                    a.lz_nlanc, lzsh);
  }
}

// LDS-tiled, persistent form of the two-column sweep (default for large sectors; EDIGPU_PANEL_TILE=0 switches it off).
//
// What the profiles of the sweep above showed (config 2, r02): its waves spend two thirds of their cycles waiting
// (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.71) and it moves V + result once (FETCH = 155 MB) but at 3.2 TB/s.  In-kernel
// time stamps and the ISA explain it:
//  * the launch is bound by the WAVE LAUNCH RATE: 8192 workgroups x 8 waves at the ~0.5-0.8 waves/ns the dispatcher
//    sustains for 512-thread groups IS the 86 us the sweep takes (the row kernel: 3432 x 8 waves, 52 us) -- so the
//    grid here is persistent: one launch of as many workgroups as stay resident, each looping over its tasks;
//  * a row's neighbour list was walked entry by entry -- s_load, wait, gather -- paying a scalar-cache miss (~0.5 us
//    under load) per dependent step; here the lists of a task are copied to LDS once, coalesced, next to the data;
//  * the result row (an HBM miss) was waited for together with the first gather (vmcnt retires in order) and two
//    rows per wave were all the memory parallelism there was; here every HBM-latency access of a task is issued up
//    front: each wave loads the result segments of ALL its rows into registers and its share of the task's own V
//    segments, which go to LDS (the alpha partial needs them anyway).
// A task = a CHUNK of <= 40 consecutive down rows of one panel (host-planned so that most hops stay inside it: rows
// that share their high bath bits are contiguous and closed under the hops among the low levels).  The host has split
// a row's hops into those that stay inside the chunk (entry = staged row index) and those that leave it (entry =
// global row), NormalArgs::tl_*; both lists are padded to whole batches of four with (own row, weight 0) entries, so a
// batch is four accesses issued back to back behind one wait, never branched over entry by entry (hipcc guards every
// load behind a branch with a full vmcnt(0)): from LDS a contiguous 1 KiB ds_read_b128 sweep without bank
// conflicts, from L2 a 1 KiB segment.  Workgroups with equal blockIdx % 8 (one XCD under the observed round-robin
// placement; speed only) walk the tasks of the same panel together.
constexpr int kTileSeg = 64;       // double2 per staged row segment (128 columns)
constexpr int kTileRowsPerWave = 4;
constexpr int kTileMaxRows = kTileRowsPerWave * 16;  // 1024-thread workgroups; 512-thread ones take half as many
constexpr int kTileBatch = 4;      // list entries per batch (a row's two hop lists are padded to whole batches)

// BLK: vectors in the panel-major layout with 128-column panels (NormalArgs::blk_shift == 7): a panel is one
// contiguous array of DimDw segments of 1024 bytes, the segment of row r starts at r * 128
template <int NT, bool DO_ND, bool ALPHA, bool EDGE, bool BLK = false>
// (forcing 64 VGPRs -- a fourth 512-thread workgroup per CU -- spills ten registers in the Hnd variants and measured
// 3-5 % slower than three workgroups per CU at 66-76 VGPRs)
__global__ void __launch_bounds__(NT)
    normal_dw_tile_kernel(NormalArgs a, PanelArgs p, const double* __restrict__ v_full, double* __restrict__ hv) {
  __shared__ double red[3 * (NT / 64)];
  // [tile_rows][kTileSeg] double2 | list weights [list_cap] | list columns [list_cap] | row meta [tile_rows] int4 |
  // Hnd partner table [2 * nterms][64]
  extern __shared__ double2 tile[];
  double* lval = reinterpret_cast<double*>(tile + (size_t)p.tile_rows * kTileSeg);
  int32_t* lcol = reinterpret_cast<int32_t*>(lval + p.list_cap);
  int4* lmeta = reinterpret_cast<int4*>(lcol + p.list_cap);
  uint32_t* ju2 = reinterpret_cast<uint32_t*>(lmeta + p.tile_rows);
  if (ALPHA && a.scal[SC_STOP] != 0.0) {
    if (threadIdx.x == 0) {
      a.partial[blockIdx.x] = 0.0;
      a.partial[gridDim.x + blockIdx.x] = 0.0;
    }
    return;
  }
  double asum = 0.0, qsum = 0.0, nsum = 0.0;  // <v|w>, sum (w - sg v)^2, <v|v> (k_finalize_ab)
  const double sg = ALPHA ? a.scal[SC_ALPHA] : 0.0;  // see normal_dw_panel_kernel
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NW = NT / 64;
  const int64_t DimUp = a.dim_up;
  int cur_panel = -1;
  for (int task = slot;; task += nslots) {
    const int panel = (task / p.blocks_per_panel) * 8 + x;
    if (panel >= p.npanels) break;  // uniform: the whole workgroup leaves together
    const int chunk = task % p.blocks_per_panel;
    const int rb = a.tile_chunks[chunk], nrows = a.tile_chunks[chunk + 1] - rb;
    const int lb = a.tile_lbeg[chunk], ln = a.tile_lbeg[chunk + 1] - lb;
    const int64_t pbase = (int64_t)panel * p.width;
    // BLK: row stride 128 inside the panel's own array, every lane owns a (possibly zero-padded) column pair
    const int64_t RS = BLK ? 128 : DimUp;
    const double* __restrict__ vb = BLK ? v_full + (int64_t)panel * a.blk_ps : v_full;
    double* __restrict__ hb = BLK ? hv + (int64_t)panel * a.blk_ps : hv;
    auto col_of = [&](int l, bool& okl, bool& pairl) -> int64_t {
      const int64_t cl = pbase + 2 * l;
      okl = 2 * l < p.width && cl < DimUp;
      const int64_t ccl = okl ? cl : (EDGE ? DimUp - 1 : DimUp - 2);
      pairl = !EDGE || ccl + 1 < DimUp;
      return ccl;
    };
    bool ok, pair;
    int64_t cc = col_of(lane, ok, pair);
    int64_t c = pbase + 2 * lane;
    if (BLK) {
      cc = c = 2 * lane;
      ok = pair = true;  // the padding columns hold zeros and receive zeros
    }
    auto ld2 = [&](const double* q) -> double2 {
      if (!EDGE) return *reinterpret_cast<const double2*>(q);
      if (pair) {
        const d2u t = *reinterpret_cast<const d2u*>(q);
        return make_double2(t.x, t.y);
      }
      return make_double2(q[0], 0.0);
    };
    auto st2 = [&](double* q, double2 t) {
      if (!EDGE) {
        *reinterpret_cast<double2*>(q) = t;
      } else if (pair) {
        d2u u;
        u.x = t.x;
        u.y = t.y;
        *reinterpret_cast<d2u*>(q) = u;
      } else {
        q[0] = t.x;
      }
    };
    // ---- every HBM-latency load first: the result segments of this wave's rows (kept in registers) and the
    // chunk's own V segments (to LDS); rows wave, wave + NW, ... in both cases; then the task's lists ----
    const int64_t g0 = a.dw_first + rb;  // global index of the first staged row
    double2 acc[kTileRowsPerWave];
    {
      double2 own[kTileRowsPerWave];
#pragma unroll
      for (int j = 0; j < kTileRowsPerWave; j++) {
        const int r = wave + j * NW;
        const int rr = r < nrows ? r : nrows - 1;  // clamped: a valid address
        acc[j] = ld2(&hb[(int64_t)(rb + rr) * RS + cc]);
      }
#pragma unroll
      for (int j = 0; j < kTileRowsPerWave; j++) {
        const int r = wave + j * NW;
        const int rr = r < nrows ? r : nrows - 1;
        own[j] = ld2(&vb[(g0 + rr) * RS + cc]);
      }
      for (int i = threadIdx.x; i < ln; i += NT) {
        lcol[i] = a.tl_col[lb + i];
        lval[i] = a.tl_val[lb + i];
      }
      if ((int)threadIdx.x < nrows) {
        int4 m = a.tl_meta[rb + threadIdx.x];
        m.x -= lb;
        lmeta[threadIdx.x] = m;
      }
#pragma unroll
      for (int j = 0; j < kTileRowsPerWave; j++) {
        const int r = wave + j * NW;
        const int rr = r < nrows ? r : nrows - 1;  // the clamped rows rewrite the last row with its own data
        tile[rr * kTileSeg + lane] = own[j];
      }
    }
    // Hnd partner of a column, per term and component (encoding of normal_dw_panel2_kernel), one word per LANE
    if (DO_ND && panel != cur_panel) {
      auto encode = [&](uint32_t jt) -> uint32_t {
        if (jt == 0xFFFFFFFFu) return jt;
        const int64_t rel = (int64_t)(jt & 0x7FFFFFFFu) - pbase;
        if (rel >= 0 && rel < p.width) return (jt & 0x80000000u) | (uint32_t)rel;
        return (jt & 0x80000000u) | 0x40000000u | (jt & 0x3FFFFFFFu);
      };
      for (int i = threadIdx.x; i < 2 * a.nterms * 64; i += NT) {
        const int t2 = i >> 6, l = i & 63;
        bool okl, pairl;
        const int64_t ccl = col_of(l, okl, pairl);
        const int64_t src = (int64_t)(t2 >> 1) * DimUp + ccl;
        if (BLK) {  // the lane's own columns, whether they exist or are padding
          const int64_t cb = pbase + 2 * l + (t2 & 1);
          ju2[i] = cb < DimUp ? encode(a.jup[(int64_t)(t2 >> 1) * DimUp + cb]) : 0xFFFFFFFFu;
          continue;
        }
        ju2[i] = (t2 & 1) ? (pairl ? encode(a.jup[src + 1]) : 0xFFFFFFFFu) : encode(a.jup[src]);
      }
      cur_panel = panel;
    }
    __syncthreads();
    auto row_sum = [&](int r, double2 s) -> double2 {
      const int4 m = lmeta[r];  // first entry, hops inside the chunk, hops leaving it, Hnd terms (LDS broadcast)
      const int mb = __builtin_amdgcn_readfirstlane(m.x), ni = __builtin_amdgcn_readfirstlane(m.y),
                no = __builtin_amdgcn_readfirstlane(m.z), nn = __builtin_amdgcn_readfirstlane(m.w);
      // hops that leave the chunk first (L2 latency), then the staged ones
      const int ob = mb + ni, oe = ob + no;
      for (int jb = ob; jb < oe; jb += kTileBatch) {
        const int4 cw = *reinterpret_cast<const int4*>(lcol + jb);
        const double2 wa = *reinterpret_cast<const double2*>(lval + jb);
        const double2 wb = *reinterpret_cast<const double2*>(lval + jb + 2);
        const double2 y0 = ld2(&vb[(int64_t)__builtin_amdgcn_readfirstlane(cw.x) * RS + cc]);
        const double2 y1 = ld2(&vb[(int64_t)__builtin_amdgcn_readfirstlane(cw.y) * RS + cc]);
        const double2 y2 = ld2(&vb[(int64_t)__builtin_amdgcn_readfirstlane(cw.z) * RS + cc]);
        const double2 y3 = ld2(&vb[(int64_t)__builtin_amdgcn_readfirstlane(cw.w) * RS + cc]);
        s.x += wa.x * y0.x;
        s.y += wa.x * y0.y;
        s.x += wa.y * y1.x;
        s.y += wa.y * y1.y;
        s.x += wb.x * y2.x;
        s.y += wb.x * y2.y;
        s.x += wb.y * y3.x;
        s.y += wb.y * y3.y;
      }
      for (int jb = mb; jb < ob; jb += kTileBatch) {
        const int4 cw = *reinterpret_cast<const int4*>(lcol + jb);
        const double2 wa = *reinterpret_cast<const double2*>(lval + jb);
        const double2 wb = *reinterpret_cast<const double2*>(lval + jb + 2);
        const double2 y0 = tile[__builtin_amdgcn_readfirstlane(cw.x) * kTileSeg + lane];
        const double2 y1 = tile[__builtin_amdgcn_readfirstlane(cw.y) * kTileSeg + lane];
        const double2 y2 = tile[__builtin_amdgcn_readfirstlane(cw.z) * kTileSeg + lane];
        const double2 y3 = tile[__builtin_amdgcn_readfirstlane(cw.w) * kTileSeg + lane];
        s.x += wa.x * y0.x;
        s.y += wa.x * y0.y;
        s.x += wa.y * y1.x;
        s.y += wa.y * y1.y;
        s.x += wb.x * y2.x;
        s.y += wb.x * y2.y;
        s.x += wb.y * y3.x;
        s.y += wb.y * y3.y;
      }
      if (DO_ND) {
        for (int jj = oe; jj < oe + nn; jj++) {
          const uint32_t cw = (uint32_t)__builtin_amdgcn_readfirstlane(lcol[jj]);
          const int tag = (int)(cw >> 24);  // wave-uniform, >= 1
          const double w = lval[jj];
          const int64_t prow = (int64_t)(cw & 0xFFFFFFu);
          const uint32_t j0 = ju2[(2 * (tag - 1)) * 64 + lane];
          const uint32_t j1 = ju2[(2 * (tag - 1) + 1) * 64 + lane];
          const bool v0 = j0 != 0xFFFFFFFFu, v1 = j1 != 0xFFFFFFFFu;
          const double w0 = v0 ? ((j0 >> 31) ? -w : w) : 0.0, w1 = v1 ? ((j1 >> 31) ? -w : w) : 0.0;
          // the partner row's segment, coalesced like a down hop; the partner columns sit a few lanes away
          const double2 y = ld2(&vb[prow * RS + cc]);
          const int l0 = (int)((j0 >> 1) & 63u), l1 = (int)((j1 >> 1) & 63u);
          const double s0x = __shfl(y.x, l0, 64), s0y = __shfl(y.y, l0, 64);
          const double s1x = __shfl(y.x, l1, 64), s1y = __shfl(y.y, l1, 64);
          double p0 = (j0 & 1u) ? s0y : s0x, p1 = (j1 & 1u) ? s1y : s1x;
          auto edge = [&](uint32_t jw) -> double {  // a partner column in a neighbouring panel
            const int64_t jc = (int64_t)(jw & 0x3FFFFFFFu);
            return BLK ? v_full[(jc >> 7) * a.blk_ps + prow * 128 + (jc & 127)] : v_full[prow * DimUp + jc];
          };
          if (v0 && (j0 & 0x40000000u)) p0 = edge(j0);
          if (v1 && (j1 & 0x40000000u)) p1 = edge(j1);
          s.x += w0 * p0;
          s.y += w1 * p1;
        }
      }
      return s;
    };
#pragma unroll
    for (int j = 0; j < kTileRowsPerWave; j++) {
      const int r = wave + j * NW;
      if (r < nrows) {  // wave-uniform
        acc[j] = row_sum(r, acc[j]);
        if (ok) {
          st2(&hb[(int64_t)(rb + r) * RS + c], acc[j]);
          if (ALPHA) {
            const double2 o = tile[r * kTileSeg + lane];  // the row's own segment of v
            const double dx = acc[j].x - sg * o.x, dy = acc[j].y - sg * o.y;
            asum += o.x * acc[j].x + o.y * acc[j].y;
            qsum += dx * dx + dy * dy;
            nsum += o.x * o.x + o.y * o.y;
          }
        }
      }
    }
    __syncthreads();  // the next task overwrites the staged data
  }
  if (ALPHA) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      asum += __shfl_down(asum, off, 64);
      qsum += __shfl_down(qsum, off, 64);
      nsum += __shfl_down(nsum, off, 64);
    }
    if (lane == 0) {
      red[wave] = asum;
      red[NT / 64 + wave] = qsum;
      red[2 * (NT / 64) + wave] = nsum;
    }
    __syncthreads();
    double t = 0.0, q = 0.0, n = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NT / 64; i++) {
        t += red[i];
        q += red[NT / 64 + i];
        n += red[2 * (NT / 64) + i];
      }
      if (!a.lz_counter) {
        a.partial[blockIdx.x] = t;
        a.partial[gridDim.x + blockIdx.x] = q;
        a.partial[2 * gridDim.x + blockIdx.x] = n;
      }
    }
    // the staged tile is no longer needed: its LDS serves the in-kernel finalize (the launcher sizes it for that)
    if (a.lz_counter)
      lz_finalize_if_last<NT>(a.lz_counter, a.partial, t, q, n, v_full, hv, a.lz_len, const_cast<double*>(a.scal), a.lz_nlanc,
                              reinterpret_cast<double*>(tile));
  }
}


// ---- the sweep on the panel-major layout (NormalArgs::blk_shift) ----
// A panel of V is one contiguous array of DimDw segments of W doubles (whole 128-byte lines), sized to stay in the L2 of
// the XCD that sweeps it, so the fabric sees V and the result once (in the natural layout a segment of row r starts
// at byte r * DimUp * 8 and straddles lines: a panel narrow enough for the L2 of a long sector occupies twice its size
// in lines, and narrower panels only RAISED the fetch traffic there -- measured).  But L2 hits are not free either: a
// sweep that gathers every hop from the L2 moves (hops per row + 2) * 8 bytes per element through it, 34 GB per
// product on the Ns = 16 ladder, which is the whole 2.1 ms of the natural-layout kernel at the ~16 TB/s the L2
// delivers.  Narrow panels make tall LDS tiles affordable -- 512 rows x 16 columns are 64 KiB -- and with rows in
// ascending order of the down word (impurity bits lowest) a contiguous block of 512 rows contains 59 % of its own
// hop partners (128 rows: 43 %, 32 rows: 28 %): a workgroup stages its block once and takes those gathers from the
// LDS, only the rest from the L2.
// A segment is served by L = W / 2 lanes (16 bytes each), a wave works on 64 / L rows at once, each lane group on
// kBlkJ rows interleaved (4 * kBlkJ independent gathers in flight per lane); the hop lists are 4-byte entries (staged
// row / global row, weight index, Hnd term) read 16 bytes at a time, the weights come from a 256-entry table in the
// LDS.  Workgroups with the same blockIdx & 7 (one XCD under the round-robin dispatch) walk a contiguous range of
// panels, one panel at a time.
constexpr int kBlkNT = 512;
constexpr int kBlkJ = 2;

struct BlkArgs {
  int npanels, panels_per_xcd, nchunks, rows_per_task, list_cap;
};

template <int SHIFT, bool DO_ND, bool ALPHA>
__global__ void __launch_bounds__(kBlkNT)
    normal_dw_blk_kernel(NormalArgs a, BlkArgs p, const double* __restrict__ v, double* __restrict__ hv) {
  constexpr int W = 1 << SHIFT, L = W / 2, RW = 64 / L, NW = kBlkNT / 64, RB = NW * RW;
  // dynamic LDS: the block's own segments of V [rows_per_task][L] double2 | its list entries [list_cap] | row meta
  extern __shared__ double2 stile[];
  __shared__ double wtab[256];
  __shared__ double red[3 * NW];
  __shared__ uint32_t ju2[DO_ND ? 2 * kMaxNdTerms * L : 1];
  __shared__ double lzsh[ALPHA ? 3 * kBlkNT + 2 : 1];  // in-kernel finalize of the fused step (lz_finalize.hpp)
  const int R = p.rows_per_task;
  uint32_t* lent = reinterpret_cast<uint32_t*>(stile + (size_t)R * L);
  int4* lmeta = reinterpret_cast<int4*>(lent + p.list_cap);
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / L, lg = lane % L;
  double asum = 0.0, qsum = 0.0, nsum = 0.0;
  const bool stop = ALPHA && a.scal[SC_STOP] != 0.0;
  const double sg = ALPHA ? a.scal[SC_ALPHA] : 0.0;
  if (threadIdx.x < 256) wtab[threadIdx.x] = a.bl_wtab[threadIdx.x];
  const int64_t DimDw = a.dim_dw, DimUp = a.dim_up, PS = a.blk_ps;
  const int pfirst = x * p.panels_per_xcd;
  int plast = pfirst + p.panels_per_xcd;
  if (plast > p.npanels) plast = p.npanels;
  const int ntasks = stop ? 0 : (plast - pfirst) * p.nchunks;
  int cur_panel = -1, cur_chunk = -1;
  for (int task = slot; task < ntasks; task += nslots) {
    // chunk-major inside a panel group would re-stage lists less often, but the L2 wants one panel at a time
    const int panel = pfirst + task / p.nchunks, chunk = task % p.nchunks;
    const double* __restrict__ vp = v + (int64_t)panel * PS;
    double* __restrict__ hp = hv + (int64_t)panel * PS;
    const int64_t rb = (int64_t)chunk * R;
    const int nrows = (int)(DimDw - rb < R ? DimDw - rb : R);
    __syncthreads();  // the previous task's readers of the tile, the lists and ju2 are done
    // ---- stage the block: nrows * L contiguous double2, and (when the chunk changed) its lists ----
    {
      const double2* __restrict__ src = reinterpret_cast<const double2*>(vp + (rb << SHIFT));
      const int n2 = nrows * L;
      for (int i0 = 0; i0 < n2; i0 += 4 * kBlkNT) {
        double2 t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + u * kBlkNT + threadIdx.x;
          t[u] = src[i < n2 ? i : n2 - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + u * kBlkNT + threadIdx.x;
          if (i < n2) stile[i] = t[u];
        }
      }
      if (chunk != cur_chunk) {
        const int lbeg = a.bl_meta[rb].x;  // multiple of 4
        const int lend = a.bl_lend[chunk];  // one past the chunk's last entry, multiple of 4
        const uint4* __restrict__ ls = reinterpret_cast<const uint4*>(a.bl_ent + lbeg);
        uint4* ld = reinterpret_cast<uint4*>(lent);
        for (int i = threadIdx.x; i < (lend - lbeg) / 4; i += kBlkNT) ld[i] = ls[i];
        for (int i = threadIdx.x; i < nrows; i += kBlkNT) {
          int4 m = a.bl_meta[rb + i];
          m.x -= lbeg;
          lmeta[i] = m;
        }
        cur_chunk = chunk;
      }
    }
    if (DO_ND && panel != cur_panel) {  // partner columns of this panel's columns, per term and component
      const int64_t pbase = (int64_t)panel << SHIFT;
      for (int i = threadIdx.x; i < 2 * a.nterms * L; i += kBlkNT) {
        const int t2 = i / L, l = i % L;
        const int64_t col = pbase + 2 * l + (t2 & 1);
        uint32_t w = 0xFFFFFFFFu;
        if (col < DimUp) {
          const uint32_t jt = a.jup[(int64_t)(t2 >> 1) * DimUp + col];
          if (jt != 0xFFFFFFFFu) {
            const int64_t rel = (int64_t)(jt & 0x7FFFFFFFu) - pbase;
            w = (rel >= 0 && rel < W) ? ((jt & 0x80000000u) | (uint32_t)rel)
                                      : ((jt & 0x80000000u) | 0x40000000u | (jt & 0x3FFFFFFFu));
          }
        }
        ju2[i] = w;
      }
    }
    cur_panel = panel;
    __syncthreads();
    for (int pass = 0; pass < nrows; pass += RB * kBlkJ) {
      const int lr0 = pass + wave * RW + grp;  // staged index of this lane group's first row
      double2 acc[kBlkJ];
      int4 meta[kBlkJ];
      bool live[kBlkJ];
#pragma unroll
      for (int j = 0; j < kBlkJ; j++) {
        const int lr = lr0 + j * RB;
        live[j] = lr < nrows;
        const int lc = live[j] ? lr : nrows - 1;  // clamped: a valid address, masked at the store
        acc[j] = *reinterpret_cast<const double2*>(hp + ((rb + lc) << SHIFT) + 2 * lg);
        meta[j] = lmeta[lc];
      }
      // ---- hops that leave the block: L2 gathers, 4 * kBlkJ in flight per lane; entries come from the LDS ----
      int nomax = 0;
#pragma unroll
      for (int j = 0; j < kBlkJ; j++) {
        if (!live[j]) meta[j].z = 0;
        nomax = meta[j].z > nomax ? meta[j].z : nomax;
      }
      for (int b = 0; b < nomax; b += 4) {
        uint4 e[kBlkJ];
        double2 y[kBlkJ][4];
#pragma unroll
        for (int j = 0; j < kBlkJ; j++) {
          // a lane group whose list is exhausted keeps issuing loads with the others: of its own row, weight 0 (never
          // of whatever follows its list in the LDS)
          const bool on = b < meta[j].z;
          e[j] = *reinterpret_cast<const uint4*>(lent + meta[j].x + meta[j].y + (on ? b : 0));
          const uint32_t own = (uint32_t)(rb + (live[j] ? lr0 + j * RB : nrows - 1));
          const uint32_t r0 = on ? (e[j].x & 0xFFFFu) : own, r1 = on ? (e[j].y & 0xFFFFu) : own,
                         r2 = on ? (e[j].z & 0xFFFFu) : own, r3 = on ? (e[j].w & 0xFFFFu) : own;
          y[j][0] = *reinterpret_cast<const double2*>(vp + ((int64_t)r0 << SHIFT) + 2 * lg);
          y[j][1] = *reinterpret_cast<const double2*>(vp + ((int64_t)r1 << SHIFT) + 2 * lg);
          y[j][2] = *reinterpret_cast<const double2*>(vp + ((int64_t)r2 << SHIFT) + 2 * lg);
          y[j][3] = *reinterpret_cast<const double2*>(vp + ((int64_t)r3 << SHIFT) + 2 * lg);
        }
#pragma unroll
        for (int j = 0; j < kBlkJ; j++) {
          const bool on = b < meta[j].z;
          const double w0 = on ? wtab[(e[j].x >> 16) & 255u] : 0.0, w1 = on ? wtab[(e[j].y >> 16) & 255u] : 0.0,
                       w2 = on ? wtab[(e[j].z >> 16) & 255u] : 0.0, w3 = on ? wtab[(e[j].w >> 16) & 255u] : 0.0;
          acc[j].x += w0 * y[j][0].x;
          acc[j].y += w0 * y[j][0].y;
          acc[j].x += w1 * y[j][1].x;
          acc[j].y += w1 * y[j][1].y;
          acc[j].x += w2 * y[j][2].x;
          acc[j].y += w2 * y[j][2].y;
          acc[j].x += w3 * y[j][3].x;
          acc[j].y += w3 * y[j][3].y;
        }
      }
      // ---- hops inside the block: entries and data from the LDS ----
#pragma unroll
      for (int j = 0; j < kBlkJ; j++) {
        if (!live[j]) continue;
        const uint32_t* li = lent + meta[j].x;
        for (int b = 0; b < meta[j].y; b += 4) {
          const uint4 ei = *reinterpret_cast<const uint4*>(li + b);
          const double2 y0 = stile[(int)(ei.x & 0xFFFFu) * L + lg], y1 = stile[(int)(ei.y & 0xFFFFu) * L + lg],
                        y2 = stile[(int)(ei.z & 0xFFFFu) * L + lg], y3 = stile[(int)(ei.w & 0xFFFFu) * L + lg];
          const double w0 = wtab[(ei.x >> 16) & 255u], w1 = wtab[(ei.y >> 16) & 255u], w2 = wtab[(ei.z >> 16) & 255u],
                       w3 = wtab[(ei.w >> 16) & 255u];
          acc[j].x += w0 * y0.x;
          acc[j].y += w0 * y0.y;
          acc[j].x += w1 * y1.x;
          acc[j].y += w1 * y1.y;
          acc[j].x += w2 * y2.x;
          acc[j].y += w2 * y2.y;
          acc[j].x += w3 * y3.x;
          acc[j].y += w3 * y3.y;
        }
      }
#pragma unroll
      for (int j = 0; j < kBlkJ; j++) {
        const int lr = lr0 + j * RB;
        if (!live[j]) continue;
        const int64_t r = rb + lr;
        double2 s = acc[j];
        if (DO_ND) {
          const uint32_t* ln = lent + meta[j].x + meta[j].y + meta[j].z;
          for (int q = 0; q < meta[j].w; q++) {
            const uint32_t ee = ln[q];
            const int term = (int)(ee >> 24) - 1;
            const double w = wtab[(ee >> 16) & 255u];
            const int64_t prow = (int64_t)(ee & 0xFFFFu);
            const uint32_t j0 = ju2[(2 * term) * L + lg], j1 = ju2[(2 * term + 1) * L + lg];
            const bool v0 = j0 != 0xFFFFFFFFu, v1 = j1 != 0xFFFFFFFFu;
            const double w0 = v0 ? ((j0 >> 31) ? -w : w) : 0.0, w1 = v1 ? ((j1 >> 31) ? -w : w) : 0.0;
            // the partner row's segment (an L2 hit like a hop); the partner columns sit a few lanes away in the group
            const double2 y = *reinterpret_cast<const double2*>(vp + (prow << SHIFT) + 2 * lg);
            const int l0 = grp * L + (int)((j0 >> 1) & (L - 1)), l1 = grp * L + (int)((j1 >> 1) & (L - 1));
            const double s0x = __shfl(y.x, l0, 64), s0y = __shfl(y.y, l0, 64);
            const double s1x = __shfl(y.x, l1, 64), s1y = __shfl(y.y, l1, 64);
            double p0 = (j0 & 1u) ? s0y : s0x, p1 = (j1 & 1u) ? s1y : s1x;
            if (v0 && (j0 & 0x40000000u)) {  // the partner column lies in a neighbouring panel
              const int64_t jc = (int64_t)(j0 & 0x3FFFFFFFu);
              p0 = v[(jc >> SHIFT) * PS + (prow << SHIFT) + (jc & (W - 1))];
            }
            if (v1 && (j1 & 0x40000000u)) {
              const int64_t jc = (int64_t)(j1 & 0x3FFFFFFFu);
              p1 = v[(jc >> SHIFT) * PS + (prow << SHIFT) + (jc & (W - 1))];
            }
            s.x += w0 * p0;
            s.y += w1 * p1;
          }
        }
        *reinterpret_cast<double2*>(hp + (r << SHIFT) + 2 * lg) = s;
        if (ALPHA) {
          const double2 o = stile[lr * L + lg];  // the row's own segment of v (zero in the padding columns)
          const double dx = s.x - sg * o.x, dy = s.y - sg * o.y;
          asum += o.x * s.x + o.y * s.y;
          qsum += dx * dx + dy * dy;
          nsum += o.x * o.x + o.y * o.y;
        }
      }
    }
  }
  if (ALPHA) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      asum += __shfl_down(asum, off, 64);
      qsum += __shfl_down(qsum, off, 64);
      nsum += __shfl_down(nsum, off, 64);
    }
    if (lane == 0) {
      red[wave] = asum;
      red[NW + wave] = qsum;
      red[2 * NW + wave] = nsum;
    }
    __syncthreads();
    double t = 0.0, q = 0.0, n = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NW; i++) {
        t += red[i];
        q += red[NW + i];
        n += red[2 * NW + i];
      }
      if (!a.lz_counter) {
        a.partial[blockIdx.x] = t;
        a.partial[gridDim.x + blockIdx.x] = q;
        a.partial[2 * gridDim.x + blockIdx.x] = n;
      }
    }
    if (a.lz_counter)
      lz_finalize_if_last<kBlkNT>(a.lz_counter, a.partial, t, q, n, v, hv, a.lz_len, const_cast<double*>(a.scal), a.lz_nlanc,
                                  lzsh);
  }
}

// 128-column panels of the panel-major layout: the LDS-tiled sweep of the natural layout (normal_dw_tile_kernel, its
// chunk plan and lists) on contiguous, line-aligned panels
static int launch_dw_blocked_tiles(const NormalArgs& a, bool do_nd, const double* v, double* hv, hipStream_t st, bool alpha,
                                   int* nblocks) {
  if (!a.tile_chunks || !a.tl_meta || a.tile_rows > kTileMaxRows || (do_nd && !a.tl_has_nd) || a.dw_first != 0 ||
      a.dw_count != a.dim_dw) {
    set_error("launch_dw_blocked: 128-column panels need the tiled sweep's row lists");
    return 1;
  }
  PanelArgs p;
  p.width = 128;
  p.npanels = (int)((a.dim_up + 127) / 128);
  p.col_first = 0;
  p.ncol = a.dim_up;
  p.stride = a.dim_up;
  p.halo = 0;
  p.rows_per_block = 0;
  p.tile_rows = a.tile_rows;
  p.blocks_per_panel = a.tile_nchunks;
  p.list_cap = a.tile_list_cap;
  const int panel_groups = (p.npanels + 7) / 8;
  const int64_t g = (int64_t)panel_groups * p.blocks_per_panel * 8;
  if (nblocks) *nblocks = (int)g;
  if (alpha && 3 * g > a.partial_cap) {
    set_error("launch_dw_blocked: partial buffer too small for this grid");
    return 1;
  }
  size_t lds = (size_t)p.tile_rows * kTileSeg * sizeof(double2) + (size_t)p.list_cap * (sizeof(double) + sizeof(int32_t)) +
               (size_t)p.tile_rows * sizeof(int4) + (do_nd ? (size_t)2 * a.nterms * 64 * sizeof(uint32_t) : 0);
  if (alpha) lds = std::max<size_t>(lds, 3 * 1024 * sizeof(double) + 16);  // the in-kernel finalize reuses the tile's LDS
  const bool big = p.tile_rows > kTileRowsPerWave * (kPanelNT / 64);
#define EDIGPU_LAUNCH_BT(NTV, ND, AL)                                                   \
  do {                                                                                  \
    auto kern = normal_dw_tile_kernel<NTV, ND, AL, false, true>;                        \
    if (ensure_dynamic_lds((const void*)kern, lds)) return 1;                           \
    hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(NTV), lds, st, a, p, v, hv);       \
  } while (0)
#define EDIGPU_LAUNCH_BT2(ND, AL)          \
  do {                                     \
    if (big)                               \
      EDIGPU_LAUNCH_BT(1024, ND, AL);      \
    else                                   \
      EDIGPU_LAUNCH_BT(kPanelNT, ND, AL);  \
  } while (0)
  if (do_nd && alpha)
    EDIGPU_LAUNCH_BT2(true, true);
  else if (do_nd)
    EDIGPU_LAUNCH_BT2(true, false);
  else if (alpha)
    EDIGPU_LAUNCH_BT2(false, true);
  else
    EDIGPU_LAUNCH_BT2(false, false);
#undef EDIGPU_LAUNCH_BT2
#undef EDIGPU_LAUNCH_BT
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_dw_blocked(const NormalArgs& a, bool do_nd, const double* v, double* hv, hipStream_t st, bool alpha,
                      int* nblocks) {
  if (a.blk_shift == 7) return launch_dw_blocked_tiles(a, do_nd, v, hv, st, alpha, nblocks);
  if (a.blk_shift < 4 || a.blk_shift > 6 || a.blk_rows < 32 || !a.bl_meta || !a.bl_ent || !a.bl_wtab || !a.bl_lend || a.blk_list_cap < 4 || a.dw_first != 0 ||
      a.dw_count != a.dim_dw) {
    set_error("launch_dw_blocked: the sector has no panel-major image");
    return 1;
  }
  if (do_nd && a.nterms > kMaxNdTerms) {
    set_error("launch_dw_blocked: too many factored Hnd terms");
    return 1;
  }
  BlkArgs p;
  const int W = 1 << a.blk_shift;
  p.npanels = (int)((a.dim_up + W - 1) >> a.blk_shift);
  p.panels_per_xcd = (p.npanels + 7) / 8;
  p.rows_per_task = a.blk_rows;  // rows of an LDS block: the lists were split by it at set-up
  p.nchunks = (int)((a.dim_dw + p.rows_per_task - 1) / p.rows_per_task);
  p.list_cap = a.blk_list_cap;
  const size_t lds = (size_t)p.rows_per_task * (W / 2) * sizeof(double2) + (size_t)p.list_cap * sizeof(uint32_t) +
                     (size_t)p.rows_per_task * sizeof(int4);
  // a persistent grid: as many workgroups as stay resident (EDIGPU_BLOCKED_WGS caps them per CU), each walking its
  // XCD's tasks in order, so that an XCD works on one panel (at a boundary: two) at a time
  static const int cap_per_cu = [] {
    const char* e = getenv("EDIGPU_BLOCKED_WGS");
    const int n = e ? atoi(e) : 8;
    return n >= 1 && n <= 8 ? n : 8;
  }();
  const int64_t most = (int64_t)p.panels_per_xcd * p.nchunks * 8;
#define EDIGPU_LAUNCH_BLK2(SH, ND, AL)                                                                  \
  do {                                                                                                  \
    auto kern = normal_dw_blk_kernel<SH, ND, AL>;                                                       \
    if (ensure_dynamic_lds((const void*)kern, lds)) return 1;                                           \
    int per_cu = resident_blocks((const void*)kern, kBlkNT, lds);                                       \
    if (per_cu < 1) return 1;                                                                           \
    if (per_cu > cap_per_cu) per_cu = cap_per_cu;                                                       \
    int64_t g = (int64_t)per_cu * device_cu_count();                                                    \
    g -= g % 8;                                                                                         \
    if (g > most) g = most;                                                                             \
    if (g < 8) g = 8;                                                                                   \
    if (nblocks) *nblocks = (int)g;                                                                     \
    if (alpha && 3 * g > a.partial_cap) {                                                               \
      set_error("launch_dw_blocked: partial buffer too small for this grid");                           \
      return 1;                                                                                         \
    }                                                                                                   \
    hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(kBlkNT), lds, st, a, p, v, hv);                    \
  } while (0)
#define EDIGPU_LAUNCH_BLK(SH)              \
  do {                                     \
    if (do_nd && alpha)                    \
      EDIGPU_LAUNCH_BLK2(SH, true, true);  \
    else if (do_nd)                        \
      EDIGPU_LAUNCH_BLK2(SH, true, false); \
    else if (alpha)                        \
      EDIGPU_LAUNCH_BLK2(SH, false, true); \
    else                                   \
      EDIGPU_LAUNCH_BLK2(SH, false, false);\
  } while (0)
  if (a.blk_shift == 4)
    EDIGPU_LAUNCH_BLK(4);
  else if (a.blk_shift == 5)
    EDIGPU_LAUNCH_BLK(5);
  else
    EDIGPU_LAUNCH_BLK(6);
#undef EDIGPU_LAUNCH_BLK2
#undef EDIGPU_LAUNCH_BLK
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

static int panel_resident_blocks() {
  // EDIGPU_PANEL_BPP: workgroups per panel (tuning knob); default = what one XCD keeps resident
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("EDIGPU_PANEL_BPP");
    v = e ? atoi(e) : 128;
    if (v < 1) v = 128;
  }
  return v;
}

// panels over ncol columns: a multiple of 8 (one stream of panels per XCD), at most 64 columns wide
static void plan_panels(PanelArgs& p, int64_t ncol) {
  int wmax = 64;
  if (const char* e = getenv("EDIGPU_PANEL_W")) {
    wmax = atoi(e);
    if (wmax < 1 || wmax > 64) wmax = 64;
  }
  int np = (int)((ncol + 8 * wmax - 1) / (8 * wmax)) * 8;
  if (np < 8) np = 8;
  p.width = (int)((ncol + np - 1) / np);
  if (p.width < 1) p.width = 1;
  p.npanels = (int)((ncol + p.width - 1) / p.width);
}

// Column-shard form (transposed exchange, reference spMatVec_mpi_normal_main :834-866 +
// vector_transpose_MPI): w holds the columns [col_first - halo, col_first + ncol + halo) of ALL DimDw rows
// (row stride `stride`: the full column block of the exchange, which the last rank only partly owns); hv (same layout) receives (Hdw (x) 1 + Hnd) v for the ncol owned columns.
// a must describe the whole sector (dw_first = 0, dw_count = DimDw).
int launch_dw_panel_cols(const NormalArgs& a, bool do_nd, int64_t col_first, int64_t ncol, int64_t stride, int halo,
                         const double* w, double* hv, hipStream_t st) {
  if (ncol <= 0) return 0;
  if (do_nd && a.nterms > kMaxNdTerms) {
    set_error("launch_dw_panel_cols: too many factored Hnd terms");
    return 1;
  }
  PanelArgs p;
  plan_panels(p, ncol);
  p.col_first = col_first;
  p.ncol = ncol;
  p.stride = stride;
  p.halo = halo;
  int bpp = panel_resident_blocks();
  p.rows_per_block = (int)((a.dw_count + bpp - 1) / bpp);
  if (p.rows_per_block < 16) p.rows_per_block = 16;
  bpp = (int)((a.dw_count + p.rows_per_block - 1) / p.rows_per_block);
  p.blocks_per_panel = bpp;
  const int panel_groups = (p.npanels + 7) / 8;
  const dim3 grid((unsigned)((int64_t)panel_groups * bpp * 8)), block(kPanelNT);
  if (do_nd)
    hipLaunchKernelGGL((normal_dw_panel_kernel<true, true, false, true>), grid, block, 0, st, a, p, w, hv);
  else
    hipLaunchKernelGGL((normal_dw_panel_kernel<true, false, false, true>), grid, block, 0, st, a, p, w, hv);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int launch_dw_panels(const NormalArgs& a, bool do_dw, bool do_nd, const double* v_full, double* hv,
                     hipStream_t st, bool alpha, int* nblocks) {
  if (!do_dw && !do_nd && !alpha) return 0;
  if (do_nd && a.nterms > kMaxNdTerms) {
    set_error("launch_dw_panels: too many factored Hnd terms");
    return 1;
  }
  const bool edge = (a.dim_up % 2) != 0;  // odd DimUp: 8-byte aligned rows, see d2u
  // the variant was fixed when the sector was set up (NormalArgs::panel_mode); the buffers must be aligned for it
  const bool vec2 = a.panel_mode >= 1 && do_dw && (((uintptr_t)v_full | (uintptr_t)hv) & (edge ? 7 : 15)) == 0;
  const bool tiled = vec2 && a.panel_mode == 2 && a.tile_chunks != nullptr && a.tl_meta != nullptr && a.tile_rows <= kTileMaxRows &&
                     (!do_nd || a.tl_has_nd);
  PanelArgs p;
  if (vec2) {
    int wmax = 128;
    if (const char* e = getenv("EDIGPU_PANEL_W")) {
      wmax = atoi(e) & ~1;
      if (wmax < 2 || wmax > 128) wmax = 128;
    }
    int np = (int)((a.dim_up + 8 * wmax - 1) / (8 * wmax)) * 8;
    if (np < 8) np = 8;
    p.width = (int)((a.dim_up + np - 1) / np);
    p.width += p.width & 1;
    p.npanels = (int)((a.dim_up + p.width - 1) / p.width);
  } else {
    plan_panels(p, a.dim_up);
  }
  p.col_first = 0;
  p.ncol = a.dim_up;
  p.stride = a.dim_up;
  p.halo = 0;
  p.tile_rows = 0;
  int bpp = panel_resident_blocks();
  if (vec2 && !getenv("EDIGPU_PANEL_BPP")) bpp = 256;  // half as many panels: twice the workgroups on each (measured)
  p.rows_per_block = (int)((a.dw_count + bpp - 1) / bpp);
  if (p.rows_per_block < 16) p.rows_per_block = 16;
  bpp = (int)((a.dw_count + p.rows_per_block - 1) / p.rows_per_block);
  if (tiled) {  // one workgroup per planned row chunk
    bpp = a.tile_nchunks;
    p.tile_rows = a.tile_rows;
  }
  p.blocks_per_panel = bpp;
  const int panel_groups = (p.npanels + 7) / 8;
  const dim3 grid((unsigned)((int64_t)panel_groups * bpp * 8)), block(kPanelNT);
  if (nblocks) *nblocks = (int)grid.x;
  if (!tiled && alpha && 3 * (int64_t)grid.x > a.partial_cap) {  // before anything that writes the partials is enqueued
    set_error("launch_dw_panels: partial buffer too small for this grid");
    return 1;
  }
  if (tiled) {
    p.list_cap = a.tile_list_cap;
    size_t lds = (size_t)p.tile_rows * kTileSeg * sizeof(double2) + (size_t)p.list_cap * (sizeof(double) + sizeof(int32_t)) +
                 (size_t)p.tile_rows * sizeof(int4) + (do_nd ? (size_t)2 * a.nterms * 64 * sizeof(uint32_t) : 0);
    if (alpha) lds = std::max<size_t>(lds, 3 * 1024 * sizeof(double) + 16);  // the in-kernel finalize reuses the tile's LDS
    // persistent grid: as many workgroups as stay resident (a multiple of 8: one stream of tasks per XCD), never
    // more than there are tasks
    const int64_t ntasks = (int64_t)panel_groups * bpp * 8;
    // workgroup size: 512 threads (4 per CU) while a chunk fits their registers, else 1024 (2 per CU, twice the rows);
    // EDIGPU_TILE_PERSIST=1: a persistent grid (as many workgroups as stay resident, each looping over its tasks)
    // instead of one task per workgroup -- measured 3-5 % slower on config 2 and the Ns=15 ladder
    static const bool persist = getenv("EDIGPU_TILE_PERSIST") && atoi(getenv("EDIGPU_TILE_PERSIST")) != 0;
    const bool big = p.tile_rows > kTileRowsPerWave * (kPanelNT / 64);
#define EDIGPU_LAUNCH_P3T(NTV, ND, AL)                                                                         \
  do {                                                                                                         \
    auto kern = edge ? normal_dw_tile_kernel<NTV, ND, AL, true> : normal_dw_tile_kernel<NTV, ND, AL, false>;   \
    if (ensure_dynamic_lds((const void*)kern, lds)) return 1;                                                  \
    const int per_cu = resident_blocks((const void*)kern, NTV, lds);                                           \
    if (per_cu < 1) return 1;                                                                                  \
    int64_t g = (int64_t)per_cu * device_cu_count();                                                           \
    g -= g % 8;                                                                                                \
    if (g > ntasks || !persist) g = ntasks;                                                                    \
    if (g < 8) g = 8;                                                                                          \
    pgrid = dim3((unsigned)g);                                                                                 \
    if (nblocks) *nblocks = (int)g;                                                                            \
    if (alpha && 3 * g > a.partial_cap) {                                                                      \
      set_error("launch_dw_panels: partial buffer too small for this grid");                                   \
      return 1;                                                                                                \
    }                                                                                                          \
    hipLaunchKernelGGL(kern, pgrid, dim3(NTV), lds, st, a, p, v_full, hv);                                     \
  } while (0)
#define EDIGPU_LAUNCH_P3(ND, AL)             \
  do {                                       \
    if (big)                                 \
      EDIGPU_LAUNCH_P3T(1024, ND, AL);       \
    else                                     \
      EDIGPU_LAUNCH_P3T(kPanelNT, ND, AL);   \
  } while (0)
    dim3 pgrid;
    if (do_nd && alpha)
      EDIGPU_LAUNCH_P3(true, true);
    else if (do_nd)
      EDIGPU_LAUNCH_P3(true, false);
    else if (alpha)
      EDIGPU_LAUNCH_P3(false, true);
    else
      EDIGPU_LAUNCH_P3(false, false);
#undef EDIGPU_LAUNCH_P3
#undef EDIGPU_LAUNCH_P3T
    EDIGPU_HIP(hipGetLastError());
    return 0;
  }
  if (vec2) {
    const size_t lds = do_nd ? (size_t)2 * a.nterms * kPanelNT * sizeof(uint32_t) : 0;
#define EDIGPU_LAUNCH_P2(ND, AL)                                                                               \
  do {                                                                                                         \
    auto kern = edge ? normal_dw_panel2_kernel<ND, AL, true> : normal_dw_panel2_kernel<ND, AL, false>;         \
    if (ensure_dynamic_lds((const void*)kern, lds)) return 1; \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, p, v_full, hv);                                          \
  } while (0)
    if (do_nd && alpha)
      EDIGPU_LAUNCH_P2(true, true);
    else if (do_nd)
      EDIGPU_LAUNCH_P2(true, false);
    else if (alpha)
      EDIGPU_LAUNCH_P2(false, true);
    else
      EDIGPU_LAUNCH_P2(false, false);
#undef EDIGPU_LAUNCH_P2
    EDIGPU_HIP(hipGetLastError());
    return 0;
  }
  if (alpha) {
    if (do_nd)
      hipLaunchKernelGGL((normal_dw_panel_kernel<true, true, true>), grid, block, 0, st, a, p, v_full, hv);
    else
      hipLaunchKernelGGL((normal_dw_panel_kernel<true, false, true>), grid, block, 0, st, a, p, v_full, hv);
  } else if (do_dw && do_nd)
    hipLaunchKernelGGL((normal_dw_panel_kernel<true, true, false>), grid, block, 0, st, a, p, v_full, hv);
  else if (do_dw)
    hipLaunchKernelGGL((normal_dw_panel_kernel<true, false, false>), grid, block, 0, st, a, p, v_full, hv);
  else
    hipLaunchKernelGGL((normal_dw_panel_kernel<false, true, false>), grid, block, 0, st, a, p, v_full, hv);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

}  // namespace edigpu
