// kernels_normal.hip -- normal-mode (Kronecker) H*v for gfx950.
//
//   Hv = Hd o v + (1 (x) Hup) v + (Hdw (x) 1) v + Hnd v
//
// takes the place of spMatVec_normal_main (reference ED_NORMAL/
// ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650) and, in its two-phase form, of
// spMatVec_mpi_normal_main (:765-929).  The state vector is viewed as the matrix
// V[idw][iup] (iup contiguous, ED_SECTOR.f90:1705-1717).
//
// Two kernels, each with its irregular accesses served on-chip:
//
//  A. normal_rows_kernel -- one workgroup = TD consecutive idw rows, all iup.
//     The TD rows of V are staged once into LDS (coalesced 16-byte HBM reads); the Hup term
//     is a gather inside a row and is served from LDS.  Hup is held as ELL, column-major,
//     one packed 32-bit word per slot (24-bit column, 7-bit coefficient id, sign) so that a
//     lane fetches the slots of its 4 adjacent columns with one 16-byte load, shared by the
//     TD rows.  The diagonal is streamed (explicit Hd) or regenerated from three small tables
//     (sectors built by the library).
//
//  B. normal_dw_panel_kernel -- (Hdw (x) 1) + Hnd as a column-panel sweep.  A panel of <= 64
//     columns over all DimDw rows fits one XCD's 4 MiB L2; all workgroups of an XCD sweep the
//     same panel, so the ~6.5 neighbour-row reads per output row are L2 hits and the fabric
//     sees V once.  Hnd is applied in factored form (signed partial permutations of the
//     impurity bits: partners are adjacent rows / columns) or as CSR in kernel A.
//
// No MFMA: ~0.25 flop/byte, bandwidth bound.
#include <cstdlib>
#include <string>

#include "kernels.hpp"
#include "normal_args.hpp"

namespace edigpu {



constexpr int kE = 4;        // adjacent columns owned by one thread per pass (16/32-byte accesses)

// VEC: DimUp even => every (row, 2-column pair) is 16-byte aligned and never straddles a row end
template <bool VEC>
__device__ inline void load4(const double* __restrict__ base, int64_t i0, const bool* ok, double* out) {
  if (VEC) {
    const double2 a = ok[0] ? *reinterpret_cast<const double2*>(base + i0) : make_double2(0.0, 0.0);
    const double2 b = ok[2] ? *reinterpret_cast<const double2*>(base + i0 + 2) : make_double2(0.0, 0.0);
    out[0] = a.x;
    out[1] = a.y;
    out[2] = b.x;
    out[3] = b.y;
  } else {
#pragma unroll
    for (int e = 0; e < kE; e++) out[e] = ok[e] ? base[i0 + e] : 0.0;
  }
}

// LDS read at an absolute byte address: and + ds_read_b64, no base add.  Valid for the staged rows
// because the row kernel's dynamic LDS starts at address 0 (no static LDS in those instantiations;
// checked once per workgroup).
typedef const double __attribute__((address_space(3))) lds_cdouble;
__device__ inline double lds_abs_read(uint32_t byte_addr) { return *reinterpret_cast<lds_cdouble*>(byte_addr); }

// HDF: diagonal from the factored tables instead of the explicit hd array
// FUSE (Lanczos step fused into the row kernel, v_local = P = previous Lanczos vector, hv = Q):
//   0: plain H*v.   1: first step: x = P, Q <- (Hd+Hup) x.
//   2: x = Q/beta (the new Lanczos vector), P <- x, Q <- (Hd+Hup) x - beta*P_old.
//   3: as 2 with the pending axpy folded in: x = (Q - alpha*P)/beta  (no separate beta kernel).
// NT: threads per workgroup (512; 1024 when the staged rows leave room for one workgroup per CU only)
// SPLIT (rows too long for the LDS, e.g. the Ns=17 ladder: 194 KB): the row is staged in column parts, one launch
//   per part [split_first, split_first + split_count); a launch adds the hops whose SOURCE column lies in its part
//   to every output column (entries pointing elsewhere are redirected to the zero slot), the first launch also
//   the diagonal, the later ones accumulate into hv.  All gathers come from the LDS instead of global memory.
template <int NT, int TD, bool USE_LDS, bool LOCAL, bool ND, bool PACKED, bool VEC, bool HDF, int FUSE,
          bool SPLIT = false>
__global__ void __launch_bounds__(NT)
    normal_rows_kernel(NormalArgs a, const double* v_local, const double* __restrict__ v_full,
                       double* hv) {
  // dynamic LDS: TD staged rows (LOCAL && USE_LDS) at offset 0 -- the packed ELL holds byte offsets
  // into them, so no base has to be added per gather -- followed by the 128 hop amplitudes
  extern __shared__ double vs[];

  const int64_t DimUp = a.dim_up;
  // element (row, col) of a vector: natural layout, or the panel-major one of the Lanczos loop (normal_args.hpp);
  // 4 adjacent columns starting at a multiple of 4 are contiguous in both
  const int bsh = a.blk_shift;
  const int64_t bps = a.blk_ps;
  auto vix = [&](int64_t row, int64_t col) -> int64_t {
    return bsh ? (col >> bsh) * bps + (row << bsh) + (col & (((int64_t)1 << bsh) - 1)) : row * DimUp + col;
  };
  // staged part of a row: all of it, or the columns [sf, sf + sc) of this launch (SPLIT)
  const int64_t sf = SPLIT ? a.split_first : 0, sc = SPLIT ? a.split_count : DimUp;
  // staged row stride: sc columns + a zero slot at index sc (target of the dead ELL slots), even
  const int S = ((int)sc + 2) & ~1;
  const int rowB = S * 8;
  double* coef_s = vs + (USE_LDS ? TD * S : 0);
  const int64_t r0 = (int64_t)blockIdx.x * TD;  // first local row of this block
  const int tid = threadIdx.x;
  int nr = TD;
  if (r0 + nr > a.dw_count) nr = (int)(a.dw_count - r0);
  double beta = 0.0, ibeta = 1.0;
  if (FUSE != 0) {
    if (a.scal[SC_STOP] != 0.0) return;  // recurrence already terminated (uniform)
    if (FUSE >= 2) {
      beta = a.scal[SC_BETA];
      ibeta = 1.0 / beta;
    }
  }
  const double alpha = (FUSE == 3) ? a.scal[SC_ALPHA] : 0.0;
  // source of the staged rows: the vector itself, or Q (scaled by 1/beta) in the fused rotate
  const double* __restrict__ v_src = (FUSE >= 2) ? hv : v_local;

  if (LOCAL && PACKED && tid < 128) coef_s[tid] = a.ell_coef[tid];
  if (LOCAL && USE_LDS) {
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)vs != 0u) __builtin_trap();
    if (tid < TD) vs[tid * S + sc] = 0.0;
    // stage the TD rows: 4 independent loads in flight per thread and row
    if (VEC) {
      const int64_t n2 = sc >> 1;  // SPLIT: sf and sc are even
      for (int64_t j0 = 0; j0 < n2; j0 += 4 * NT) {
        double2 t[TD][4];
#pragma unroll
        for (int r = 0; r < TD; r++)
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int64_t j = j0 + tid + u * NT;
            const int rr = r < nr ? r : 0;  // clamped: always a valid address
            const int64_t jc = 2 * (j < n2 ? j : n2 - 1);
            t[r][u] = *reinterpret_cast<const double2*>(v_src + vix(r0 + rr, sf + jc));
            if (FUSE == 3) {
              const double2 pp = *reinterpret_cast<const double2*>(v_local + vix(r0 + rr, jc));
              t[r][u].x -= alpha * pp.x;
              t[r][u].y -= alpha * pp.y;
            }
            if (FUSE >= 2) {
              t[r][u].x *= ibeta;
              t[r][u].y *= ibeta;
            }
          }
#pragma unroll
        for (int r = 0; r < TD; r++)
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int64_t j = j0 + tid + u * NT;
            if (r < nr && j < n2) reinterpret_cast<double2*>(vs + r * S)[j] = t[r][u];
          }
      }
    } else {
      for (int64_t j0 = 0; j0 < sc; j0 += 4 * NT) {
        double t[TD][4];
#pragma unroll
        for (int r = 0; r < TD; r++)
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int64_t j = j0 + tid + u * NT;
            const int rr = r < nr ? r : 0;
            t[r][u] = v_src[vix(r0 + rr, sf + (j < sc ? j : sc - 1))];
            if (FUSE == 3) t[r][u] -= alpha * v_local[vix(r0 + rr, j < DimUp ? j : DimUp - 1)];
            if (FUSE >= 2) t[r][u] *= ibeta;
          }
#pragma unroll
        for (int r = 0; r < TD; r++)
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int64_t j = j0 + tid + u * NT;
            if (r < nr && j < sc) vs[r * S + j] = t[r][u];
          }
      }
    }
  }
  __syncthreads();

  for (int64_t c0 = 0; c0 < DimUp; c0 += (int64_t)NT * kE) {
    const int64_t col0 = c0 + (int64_t)tid * kE;
    if (col0 >= DimUp) break;  // no barrier below: safe to leave
    double acc[TD][kE];
    bool ok[kE];
#pragma unroll
    for (int e = 0; e < kE; e++) {
      ok[e] = col0 + e < DimUp;
#pragma unroll
      for (int r = 0; r < TD; r++) acc[r][e] = 0.0;
    }

    if (LOCAL) {
      // ---- diagonal ----
#pragma unroll
      for (int r = 0; r < TD; r++)
        if (r < nr) {
          double h[kE], x[kE];
          if (HDF) {
            const int64_t g = a.dw_first + r0 + r;
            const double edr = a.ed[g];
            load4<VEC>(a.eux, (int64_t)a.impd[g] * DimUp + col0, ok, h);
#pragma unroll
            for (int e = 0; e < kE; e++) h[e] += edr;
          } else {
            load4<VEC>(a.hd, (r0 + r) * DimUp + col0, ok, h);
          }
          if (USE_LDS && !SPLIT) {
#pragma unroll
            for (int e = 0; e < kE; e++) x[e] = ok[e] ? vs[r * S + col0 + e] : 0.0;
          } else {
            load4<VEC>(v_local, vix(r0 + r, col0), ok, x);
          }
#pragma unroll
          for (int e = 0; e < kE; e++) acc[r][e] = (SPLIT && sf != 0) ? 0.0 : h[e] * x[e];  // the diagonal: first part only
        }
      // ---- (1 (x) Hup): gather inside the row; one 16-byte load brings 4 packed slots.
      // Slots are fetched KU at a time into registers first (KU = 4 measured 2 % faster than 6 on config 2 and the
      // Ns=16 ladder, 12 is 7 % slower: registers, not load latency, are the limit), so that KU independent L2 loads are
      // in flight per lane before the first LDS gather is issued. ----
      constexpr int KU = 4;
      // Typed ELL of an LDS sector (the normal case): slot k = one hop type with a wave-uniform
      // amplitude; an entry is (byte offset in the staged row) | sign << 31, dead entries name the
      // row's zero slot.  Per gather: v_and, ds_read_b64, v_bitop3 (sign), v_fma_f64.
      const bool fast = PACKED && USE_LDS && a.ell_typed != 0;
      if (fast) {
        const uint32_t* __restrict__ epk = a.ell_pk + col0;
        const int epitch = (int)a.ell_pitch;
        for (int k0 = 0; k0 < a.ell_w; k0 += KU) {
          uint4 pk4[KU];
          double tk[KU];
#pragma unroll
          for (int u = 0; u < KU; u++) {
            // slots past the width: clamped address, amplitude 0 (the table is zero from ell_w on)
            const int k = k0 + u < a.ell_w ? k0 + u : a.ell_w - 1;
            pk4[u] = *reinterpret_cast<const uint4*>(epk + k * epitch);  // 32-bit: staged rows are short
            tk[u] = coef_s[k0 + u < 127 ? k0 + u : 127];                 // LDS broadcast
          }
#pragma unroll
          for (int u = 0; u < KU; u++) {
            const uint32_t p[kE] = {pk4[u].x, pk4[u].y, pk4[u].z, pk4[u].w};
#pragma unroll
            for (int r = 0; r < TD; r++)
              if (TD == 1 || r < nr) {
#pragma unroll
                for (int e = 0; e < kE; e++) {
                  uint32_t off = p[e] & 0xFFFFFFu;
                  if (SPLIT) {
                    // byte offset in the full row -> in the staged part, or the zero slot when the source column
                    // is not in this part (unsigned compare also catches the columns below sf)
                    const uint32_t rel = off - (uint32_t)(sf * 8);
                    off = rel < (uint32_t)(sc * 8) ? rel : (uint32_t)(sc * 8);
                  }
                  double x = lds_abs_read((uint32_t)(r * rowB) + off);
                  x = __hiloint2double(__double2hiint(x) ^ (int)(p[e] & 0x80000000u), __double2loint(x));
                  acc[r][e] = fma(tk[u], x, acc[r][e]);
                }
              }
          }
        }
      }
      for (int k0 = fast ? a.ell_w : 0; k0 < a.ell_w; k0 += KU) {
        uint4 pk4[KU];
        int4 pc4[KU];
        double2 pw0[KU], pw1[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
          const int k = k0 + u < a.ell_w ? k0 + u : a.ell_w - 1;       // clamped, weight zeroed below
          const int64_t o = (int64_t)k * a.ell_pitch + col0;           // multiple of 4, inside the pitch
          if (PACKED) {
            pk4[u] = *reinterpret_cast<const uint4*>(a.ell_pk + o);
          } else {
            pc4[u] = *reinterpret_cast<const int4*>(a.ell_col + o);
            pw0[u] = *reinterpret_cast<const double2*>(a.ell_val + o);
            pw1[u] = *reinterpret_cast<const double2*>(a.ell_val + o + 2);
          }
        }
#pragma unroll
        for (int u = 0; u < KU; u++) {
          const bool live = k0 + u < a.ell_w;
          int32_t cc[kE];
          double ww[kE];
          if (PACKED) {
            const uint32_t p[kE] = {pk4[u].x, pk4[u].y, pk4[u].z, pk4[u].w};
            if (a.ell_typed) {
              // slot k = one hop type: the amplitude is wave-uniform, no table lookup
              const double tk = live ? coef_s[k0 + u] : 0.0;  // LDS broadcast (a global load here stalls on vmcnt)
              if (USE_LDS) {
                // low 24 bits = byte offset inside the staged row (dead slots name the zero slot, so
                // there is no live test), bit 31 = sign: and + and + xor + fma per gather
#pragma unroll
                for (int e = 0; e < kE; e++) {
                  cc[e] = (int32_t)(p[e] & 0xFFFFFFu);
                  ww[e] = __hiloint2double(__double2hiint(tk) ^ (int)(p[e] & 0x80000000u), __double2loint(tk));
                }
              } else {
#pragma unroll
                for (int e = 0; e < kE; e++) {
                  cc[e] = (int32_t)(p[e] & 0xFFFFFFu);
                  const double m = ((p[e] >> 24) & 0x7Fu) ? tk : 0.0;
                  ww[e] = (p[e] >> 31) ? -m : m;
                }
              }
            } else {
#pragma unroll
              for (int e = 0; e < kE; e++) {
                cc[e] = (int32_t)(p[e] & 0xFFFFFFu);
                const double m = live ? coef_s[(p[e] >> 24) & 0x7Fu] : 0.0;
                ww[e] = (p[e] >> 31) ? -m : m;
              }
            }
          } else {
            cc[0] = pc4[u].x; cc[1] = pc4[u].y; cc[2] = pc4[u].z; cc[3] = pc4[u].w;
            ww[0] = live ? pw0[u].x : 0.0; ww[1] = live ? pw0[u].y : 0.0;
            ww[2] = live ? pw1[u].x : 0.0; ww[3] = live ? pw1[u].y : 0.0;
          }
#pragma unroll
          for (int r = 0; r < TD; r++)
            if (r < nr) {
#pragma unroll
              for (int e = 0; e < kE; e++) {
                // packed ELL of an LDS sector holds byte offsets (upload_ell), the plain one columns
                const double x = USE_LDS ? (PACKED ? *reinterpret_cast<const double*>(
                                                         reinterpret_cast<const char*>(vs) + r * rowB + cc[e])
                                                   : vs[r * S + cc[e]])
                                         : v_local[vix(r0 + r, cc[e])];
                acc[r][e] = fma(ww[e], x, acc[r][e]);
              }
            }
        }
      }
    }
    if (ND) {
      // ---- Hnd: short CSR rows with global columns ----
      if (a.has_nd) {
        int64_t rp[TD][kE + 1];
#pragma unroll
        for (int r = 0; r < TD; r++)
          if (r < nr) {
            const int64_t i0 = (r0 + r) * DimUp + col0;
#pragma unroll
            for (int e = 0; e <= kE; e++) {
              const int64_t i = (e == 0 || ok[e - 1]) ? i0 + e : i0;  // stays inside the row pointers
              rp[r][e] = a.nd_rp64 ? a.nd_rp64[i] : (int64_t)a.nd_rp32[i];
            }
          }
#pragma unroll
        for (int r = 0; r < TD; r++)
          if (r < nr) {
#pragma unroll
            for (int e = 0; e < kE; e++)
              if (ok[e]) {
                double s = 0.0;
                for (int64_t jj = rp[r][e]; jj < rp[r][e + 1]; jj++)
                  s += a.nd_val[jj] * v_full[a.nd_col[jj]];
                acc[r][e] += s;
              }
          }
      }
    }
#pragma unroll
    for (int r = 0; r < TD; r++)
      if (r < nr) {
        double* dst = hv + vix(r0 + r, col0);
        if (!LOCAL || (SPLIT && sf != 0)) {
          double old[kE];
          load4<VEC>(hv, vix(r0 + r, col0), ok, old);
#pragma unroll
          for (int e = 0; e < kE; e++) acc[r][e] += old[e];
        }
        if (FUSE >= 2) {
          // Q <- acc - beta * P_old ; P <- x (the staged, normalised vector)
          double pold[kE];
          load4<VEC>(v_local, vix(r0 + r, col0), ok, pold);
          double* pdst = (a.xout ? a.xout : const_cast<double*>(v_local)) + vix(r0 + r, col0);
#pragma unroll
          for (int e = 0; e < kE; e++) acc[r][e] -= beta * pold[e];
          if (VEC) {
            if (ok[0]) *reinterpret_cast<double2*>(pdst) = make_double2(vs[r * S + col0], vs[r * S + col0 + 1]);
            if (ok[2]) *reinterpret_cast<double2*>(pdst + 2) = make_double2(vs[r * S + col0 + 2], vs[r * S + col0 + 3]);
          } else {
#pragma unroll
            for (int e = 0; e < kE; e++)
              if (ok[e]) pdst[e] = vs[r * S + col0 + e];
          }
        }
        if (VEC) {
          if (ok[0]) *reinterpret_cast<double2*>(dst) = make_double2(acc[r][0], acc[r][1]);
          if (ok[2]) *reinterpret_cast<double2*>(dst + 2) = make_double2(acc[r][2], acc[r][3]);
        } else {
#pragma unroll
          for (int e = 0; e < kE; e++)
            if (ok[e]) dst[e] = acc[r][e];
        }
      }
  }
}

// TD rows of V (8 B/elem) must fit the LDS budget of one workgroup.  Keep two
// workgroups per CU when possible (<= 64 KiB each of the 160 KiB LDS).
int normal_pick_rows_per_block(int64_t dim_up, int64_t dw_count) {
  const int64_t row_bytes = ((dim_up + 2) & ~(int64_t)1) * 8;  // staged row: columns + zero slot
  if (const char* e = getenv("EDIGPU_ROWS_TD")) {  // tuning override
    const int td = atoi(e);
    if ((td == 1 || td == 2 || td == 4 || td == 8) && td * row_bytes <= 150 * 1024) return td;
  }
  if (row_bytes > 150 * 1024) return 0;  // generic (no LDS) kernel
  int td = 1;
  for (int cand : {2, 4, 8}) {
    if (cand * row_bytes > 24 * 1024) break;  // measured (config 2): 4 workgroups/CU beat 2 rows/workgroup
    if ((dw_count + cand - 1) / cand < 1024) break;  // keep >= 4 workgroups per CU in the grid
    td = cand;
  }
  return td;
}

// what: 1 = diagonal + up (overwrite), 5 = the same + CSR Hnd, 4 = CSR Hnd only (accumulate), 101-103 = fused Lanczos
template <int NT, int TD, bool USE_LDS, bool PACKED, bool VEC, bool HDF>
static int launch_te(const NormalArgs& a, const double* vl, const double* vf, double* hv,
                     int what, hipStream_t st) {
  const int64_t nblk = (a.dw_count + TD - 1) / TD;
  const size_t lds = (USE_LDS ? (size_t)TD * ((a.dim_up + 2) & ~(int64_t)1) * sizeof(double) : 0) + 128 * sizeof(double);
  dim3 grid((unsigned)nblk), block(NT);
#define EDIGPU_LAUNCH_ROWS(LOC, NDF, LDSB, UL, PK, HF)                                           \
  do {                                                                                           \
    auto kern = normal_rows_kernel<NT, TD, UL, LOC, NDF, PK, VEC, HF, 0>;                            \
    if (ensure_dynamic_lds((const void*)kern, (LDSB))) return 1; \
    hipLaunchKernelGGL(kern, grid, block, (LDSB), st, a, vl, vf, hv);                            \
  } while (0)
  switch (what) {
    case 1: EDIGPU_LAUNCH_ROWS(true, false, lds, USE_LDS, PACKED, HDF); break;
    case 5: EDIGPU_LAUNCH_ROWS(true, true, lds, USE_LDS, PACKED, HDF); break;
    case 101: {  // fused Lanczos, first step
      auto kern = normal_rows_kernel<NT, TD, USE_LDS, true, false, PACKED, VEC, HDF, 1>;
      if (ensure_dynamic_lds((const void*)kern, lds)) return 1;
      hipLaunchKernelGGL(kern, grid, block, lds, st, a, vl, vf, hv);
      break;
    }
    case 102: {  // fused Lanczos, rotate + H*v
      auto kern = normal_rows_kernel<NT, TD, USE_LDS, true, false, PACKED, VEC, HDF, 2>;
      if (ensure_dynamic_lds((const void*)kern, lds)) return 1;
      hipLaunchKernelGGL(kern, grid, block, lds, st, a, vl, vf, hv);
      break;
    }
    case 103: {  // fused Lanczos, pending axpy + rotate + H*v
      auto kern = normal_rows_kernel<NT, TD, USE_LDS, true, false, PACKED, VEC, HDF, 3>;
      if (ensure_dynamic_lds((const void*)kern, lds)) return 1;
      hipLaunchKernelGGL(kern, grid, block, lds, st, a, vl, vf, hv);
      break;
    }
    case 4: EDIGPU_LAUNCH_ROWS(false, true, (size_t)0, false, false, false); break;
    default: set_error("launch_normal: bad term mask"); return 1;
  }
#undef EDIGPU_LAUNCH_ROWS
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

template <int TD, bool USE_LDS>
static int launch_td(const NormalArgs& a, bool packed, bool hdf, const double* vl, const double* vf,
                     double* hv, int what, hipStream_t st) {
  const bool vec = (a.dim_up % 2) == 0;
  // one workgroup per CU by LDS (a single row > 80 KiB): use all 1024 threads of it
  const bool big = TD == 1 && USE_LDS && (size_t)a.dim_up * sizeof(double) > 80 * 1024;
  constexpr int NTBIG = (TD == 1 && USE_LDS) ? 1024 : 512;  // only TD=1 instantiates the 1024 variant
#define EDIGPU_TD(PK, VC)                                                                  \
  (big ? (hdf ? launch_te<NTBIG, TD, USE_LDS, PK, VC, true>(a, vl, vf, hv, what, st)       \
              : launch_te<NTBIG, TD, USE_LDS, PK, VC, false>(a, vl, vf, hv, what, st))     \
       : (hdf ? launch_te<512, TD, USE_LDS, PK, VC, true>(a, vl, vf, hv, what, st)         \
              : launch_te<512, TD, USE_LDS, PK, VC, false>(a, vl, vf, hv, what, st)))
  if (vec) return packed ? EDIGPU_TD(true, true) : EDIGPU_TD(false, true);
  return packed ? EDIGPU_TD(true, false) : EDIGPU_TD(false, false);
#undef EDIGPU_TD
}

// rows longer than the LDS, staged in s->row_split column parts: one launch per part (see SPLIT)
template <bool VEC, bool HDF>
static int launch_split_t(const edigpu_sector* s, NormalArgs a, const double* vl, double* hv, hipStream_t st) {
  constexpr int NT = 1024;
  // equal even parts
  const int64_t part = (((a.dim_up + s->row_split - 1) / s->row_split) + 1) & ~(int64_t)1;
  auto kern = normal_rows_kernel<NT, 1, true, true, false, true, VEC, HDF, 0, true>;
  const size_t lds = (size_t)((part + 2) & ~(int64_t)1) * sizeof(double) + 128 * sizeof(double);
  if (ensure_dynamic_lds((const void*)kern, lds)) return 1;
  for (int64_t first = 0; first < a.dim_up; first += part) {
    a.split_first = first;
    a.split_count = std::min<int64_t>(part, a.dim_up - first);
    hipLaunchKernelGGL(kern, dim3((unsigned)a.dw_count), dim3(NT), lds, st, a, vl, nullptr, hv);
  }
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

static int launch_rows(const edigpu_sector* s, const NormalArgs& a, const double* vl,
                       const double* vf, double* hv, int what, hipStream_t st) {
  const bool packed = s->up_ell.pk != nullptr;
  const bool hdf = s->factored != 0;
  if (s->row_split > 1 && what != 4) {
    // what = 1: diagonal + up part; what = 5: the same, then the CSR Hnd pass (accumulate)
    if (what != 1 && what != 5) {
      set_error("launch_normal: fused Lanczos is not available for split rows");
      return 1;
    }
    const bool vec = (a.dim_up % 2) == 0;
    const int rc = vec ? (hdf ? launch_split_t<true, true>(s, a, vl, hv, st) : launch_split_t<true, false>(s, a, vl, hv, st))
                       : (hdf ? launch_split_t<false, true>(s, a, vl, hv, st) : launch_split_t<false, false>(s, a, vl, hv, st));
    if (rc || what == 1) return rc;
    return launch_td<1, false>(a, false, hdf, vl, vf, hv, 4, st);
  }
  switch (s->rows_per_block) {
    case 0: return launch_td<1, false>(a, packed, hdf, vl, vf, hv, what, st);
    case 1: return launch_td<1, true>(a, packed, hdf, vl, vf, hv, what, st);
    case 2: return launch_td<2, true>(a, packed, hdf, vl, vf, hv, what, st);
    case 4: return launch_td<4, true>(a, packed, hdf, vl, vf, hv, what, st);
    case 8: return launch_td<8, true>(a, packed, hdf, vl, vf, hv, what, st);
    default: set_error("launch_normal: bad rows_per_block"); return 1;
  }
}

static void fill_args(const edigpu_sector* s, NormalArgs& a) {
  a.dim_up = s->dim_up;
  a.dim_dw = s->dim_dw;
  a.dw_first = s->dw_first;
  a.dw_count = s->dw_count;
  a.hd = s->d_hd;
  a.eux = s->d_eux;
  a.ed = s->d_ed;
  a.impd = s->d_impd;
  a.ell_pk = s->up_ell.pk;
  a.ell_coef = s->up_ell.coef;
  a.ell_col = s->up_ell.col;
  a.ell_val = s->up_ell.val;
  a.ell_w = s->up_ell.width;
  a.ell_typed = s->up_ell.typed;
  a.ell_pitch = s->up_ell.pitch;
  a.dw_rowptr = s->dw.rowptr32;
  a.dw_col = s->dw.col;
  a.dw_val = s->dw.val;
  a.dw_maxrow = s->dw_maxrow;
  a.nd_rp32 = s->nd.rowptr32;
  a.nd_rp64 = s->nd.wide ? s->nd.rowptr64 : nullptr;
  a.nd_col = s->nd.col;
  a.nd_val = s->nd.val;
  a.has_nd = s->has_nd;
  a.nterms = s->factored ? s->fac_nterms : 0;
  a.nd_coef = s->d_ndcoef;
  a.jup = s->d_jup;
  a.jdw = s->d_jdw;
  a.panel_mode = s->panel_mode;
  a.tile_nchunks = s->tile_nchunks;
  a.tile_rows = s->tile_rows;
  a.tile_chunks = s->d_tile_chunks;
  a.tile_lbeg = s->d_tile_lbeg;
  a.tile_list_cap = s->tile_list_cap;
  a.tl_meta = s->d_tl_meta;
  a.tl_col = s->d_tl_col;
  a.tl_val = s->d_tl_val;
  a.tl_has_nd = s->tl_has_nd;
  a.blk_shift = 0;  // natural layout unless the caller runs the panel-major Lanczos loop
  a.blk_ps = 0;
  a.bl_meta = s->d_bl_meta;
  a.blk_rows = s->blk_rows;
  a.blk_list_cap = s->blk_list_cap;
  a.bl_lend = s->d_bl_lend;
  a.bl_ent = s->d_bl_ent;
  a.bl_wtab = s->d_bl_wtab;
  a.mx_rowptr = s->d_mx_rowptr;
  a.mx_col = s->d_mx_col;
  a.mx_val = s->d_mx_val;
  a.scal = nullptr;
  a.partial = nullptr;
  a.xout = nullptr;
  a.partial_cap = 0;
  a.lz_counter = nullptr;
  a.lz_nlanc = 0;
  a.lz_len = 0;
  a.split_first = 0;
  a.split_count = s->dim_up;
}

int launch_normal(const edigpu_sector* s, const double* v_local, const double* v_full, double* hv,
                  int phase, hipStream_t st) {
  NormalArgs a;
  fill_args(s, a);
  if (s->dw_count == 0) return 0;
  const bool fac = s->factored != 0;
  const bool sell_nd = !fac && s->has_nd && s->nd.sell;   // Hnd as a separate SELL pass (after the panels)
  const bool csr_nd = !fac && s->has_nd && !sell_nd;      // Hnd applied by the row kernel from CSR
  const bool fac_nd = fac && a.nterms > 0 && s->d_mx_rowptr != nullptr;     // Hnd applied by the panel kernel from the factored terms
  if (phase == 1) return launch_rows(s, a, v_local, v_full, hv, 1, st);
  if (phase == 3) {
    if (launch_rows(s, a, v_local, v_full, hv, csr_nd ? 5 : 1, st)) return 1;
    if (launch_dw_panels(a, true, fac_nd, v_full, hv, st)) return 1;
    return sell_nd ? launch_csr(s->nd, 0, v_full, hv, 1, st) : 0;
  }
  // phase 2: the terms that need the gathered vector, accumulated into hv
  if (csr_nd && launch_rows(s, a, v_local, v_full, hv, 4, st)) return 1;
  if (launch_dw_panels(a, true, fac_nd, v_full, hv, st)) return 1;
  return sell_nd ? launch_csr(s->nd, 0, v_full, hv, 1, st) : 0;
}

// The LDS row kernel on the padded panel layout of the impurity-block image, columns in POSITION order (IbDev::pr: Hup and
// the diagonal table permuted to positions, padding positions empty): the rows half of the product for sectors whose rows
// are short enough that this kernel beats the block rows kernels (config 2: 27 KB rows, four workgroups per CU), paired
// with the local-block columns kernel on the same vectors.  what: 1 plain; 101 / 102 / 103 the fused Lanczos forms (FUSE
// 1 / 2 / 3) with the new vector written to X.
int launch_normal_rows_pos(const edigpu_sector* s, const double* P, double* Q, double* X, int what, const double* scal, hipStream_t st) {
  const IbDev* d = s->ib;
  NormalArgs a;
  fill_args(s, a);
  a.dim_up = d->plen;
  a.split_count = d->plen;
  a.eux = d->pr.eux;
  a.ell_pk = d->pr.ell.pk;
  a.ell_coef = d->pr.ell.coef;
  a.ell_col = d->pr.ell.col;
  a.ell_val = d->pr.ell.val;
  a.ell_w = d->pr.ell.width;
  a.ell_typed = d->pr.ell.typed;
  a.ell_pitch = d->pr.ell.pitch;
  a.blk_shift = 4;
  a.blk_ps = d->ps;
  a.scal = scal;
  a.xout = X;
  const bool packed = d->pr.ell.pk != nullptr;
  switch (d->pr.td) {
    case 1: return launch_td<1, true>(a, packed, true, P, P, Q, what, st);
    case 2: return launch_td<2, true>(a, packed, true, P, P, Q, what, st);
    case 4: return launch_td<4, true>(a, packed, true, P, P, Q, what, st);
    case 8: return launch_td<8, true>(a, packed, true, P, P, Q, what, st);
  }
  set_error("launch_normal_rows_pos: rows per workgroup");
  return 1;
}

// Transposed exchange (reference spMatVec_mpi_normal_main :765-866): the two halves of the
// product on a whole-sector handle, each on the part of the vector a rank owns in that phase.
//   rows: hv_rows = (Hd + 1 (x) Hup) v for the down rows [dw_first, dw_first + dw_count), v_rows / hv_rows
//         hold those rows (row stride DimUp)
//   cols: hv_cols = (Hdw (x) 1 + Hnd) v for the up columns [col_first, col_first + ncol) of all rows, buffers
//         with halo columns on both sides (launch_dw_panel_cols)
// the electronic part of the handle can run as a row half + a column half (phonon blocks: one exchange per block)
bool normal_transposable_el(const edigpu_sector* s) {
  if (s->kind != 0 || s->nloc != s->dim || s->dw_count == 0) return false;
  if (!s->factored && s->has_nd) return false;  // explicit spH0nd couples arbitrary columns: all-gather form only
  if (s->factored && s->fac_nterms > 0 && s->d_mx_rowptr == nullptr) return false;
  return true;
}

bool normal_transposable(const edigpu_sector* s) { return s->nph == 0 && normal_transposable_el(s); }

int launch_normal_rows(const edigpu_sector* s, int64_t dw_first, int64_t dw_count, const double* v_rows,
                       double* hv_rows, hipStream_t st) {
  if (dw_count <= 0) return 0;
  NormalArgs a;
  fill_args(s, a);
  a.dw_first = dw_first;
  a.dw_count = dw_count;
  return launch_rows(s, a, v_rows, nullptr, hv_rows, 1, st);
}

int launch_normal_cols(const edigpu_sector* s, int64_t col_first, int64_t ncol, int64_t stride, int halo,
                       const double* w, double* hv, hipStream_t st) {
  NormalArgs a;
  fill_args(s, a);
  const bool fac_nd = s->factored && a.nterms > 0;
  return launch_dw_panel_cols(a, fac_nd, col_first, ncol, stride, halo, w, hv, st);
}

// One fused Lanczos step on a single-shard normal handle (see normal_rows_kernel FUSE):
//   row kernel : [rotate] + Q <- (Hd+Hup) v [- beta*P_old]
//   panel sweep: Q += (Hdw + Hnd) v ; per-workgroup partials of alpha = <v|Q>
// Returns the number of partials written.  The caller finalises alpha and runs the beta kernel.
bool normal_lanczos_fusable(const edigpu_sector* s) {
  if (s->kind != 0 || s->nloc != s->dim || s->dw_count == 0 || s->nph > 0) return false;
  if (s->rows_per_block == 0) return false;               // needs the LDS row kernel
  if (!s->factored && s->has_nd && !s->nd.sell) return false;  // CSR Hnd inside the row kernel needs the complete new vector
  if (getenv("EDIGPU_LANCZOS_UNFUSED")) return false;
  return true;
}

int launch_normal_lanczos(const edigpu_sector* s, double* P, double* Q, const double* scal,
                          double* partial, int64_t partial_cap, bool first, bool lazy_axpy, hipStream_t st, int* npartial,
                          int nlanc, bool* finalized, double* X, bool* in_x) {
  NormalArgs a;
  fill_args(s, a);
  a.scal = scal;
  a.partial = partial;
  a.partial_cap = partial_cap;
  // EDIGPU_LANCZOS_INKERNEL_FINALIZE=1: the sweep's last workgroup finalizes the step itself (lz_finalize.hpp) instead of
  // a separate 5 us kernel.  OPT-IN: it buys nothing.  With a release fence per workgroup (which writes back the XCD's
  // whole L2 on this part) it measured 306 against 162 us per step on config 2; in the fence-free form (device-scope
  // stores / loads for the partials) 167.8 against 162.3 (config 2), 20.9 against 20.1 (config 3), 10.8 against 11.4
  // (config 1): the reduction of a few thousand partials by one workgroup takes as long at the tail of the sweep as it
  // does in its own kernel.
  static const bool inkernel = getenv("EDIGPU_LANCZOS_INKERNEL_FINALIZE") && atoi(getenv("EDIGPU_LANCZOS_INKERNEL_FINALIZE")) != 0;
  const bool explicit_nd = !s->factored && s->has_nd && s->nd.sell;
  if (finalized) *finalized = false;
  if (inkernel && lazy_axpy && !explicit_nd && s->d_lzcnt && finalized) {
    a.lz_counter = s->d_lzcnt;
    a.lz_nlanc = nlanc;
    a.lz_len = s->lz_len;
    *finalized = true;
  }
  if (in_x) *in_x = false;
  if (s->lz_blocked && s->ib) {  // impurity-block image: its own two kernels on its own layout (kernels_ib.hip)
    if (finalized) *finalized = false;
    if (!X || !in_x) {
      set_error("launch_normal_lanczos: the impurity-block image needs a third buffer");
      return 1;
    }
    *in_x = !first && !s->ib->pr.on;  // (short rows: the position-order row kernel rotates in place)
    return launch_ib_lanczos(s, P, Q, X, scal, partial, partial_cap, first, lazy_axpy, st, npartial);
  }
  if (s->lz_blocked) {  // P, Q in the panel-major layout (lanczos_prepare)
    a.blk_shift = s->blk_shift;
    a.blk_ps = s->blk_ps;
  }
  if (launch_rows(s, a, P, P, Q, first ? 101 : (lazy_axpy ? 103 : 102), st)) return 1;
  if (s->lz_blocked) return launch_dw_blocked(a, a.nterms > 0, P, Q, st, true, npartial);
  const bool fac_nd = s->factored && a.nterms > 0 && s->d_mx_rowptr != nullptr;
  if (!s->factored && s->has_nd && s->nd.sell) {
    // explicit image (hand-over arrays): panels without the dot, then Q += Hnd v as a SELL pass whose
    // epilogue carries the <v|Q>, <Q|Q> partials
    if (launch_dw_panels(a, true, false, P, Q, st)) return 1;
    return launch_csr_lanczos(s->nd, 0, P, Q, partial, partial_cap, npartial, scal + SC_ALPHA, st);
  }
  return launch_dw_panels(a, true, fac_nd, P, Q, st, true, npartial);
}

// plain H*v on panel-major vectors (the product of the blocked Lanczos loop without the fused recurrence)
int launch_normal_blocked(const edigpu_sector* s, const double* v, double* hv, hipStream_t st) {
  if (s->ib) return launch_ib(s, v, hv, st);
  if (s->blk_shift == 0 || s->dw_count == 0) {
    set_error("launch_normal_blocked: the sector has no panel-major image");
    return 1;
  }
  NormalArgs a;
  fill_args(s, a);
  a.blk_shift = s->blk_shift;
  a.blk_ps = s->blk_ps;
  if (launch_rows(s, a, v, v, hv, 1, st)) return 1;
  return launch_dw_blocked(a, a.nterms > 0, v, hv, st, false, nullptr);
}

}  // namespace edigpu
