// Test-only C shim over the host builders of libedigpu (csrc/host_build.cpp): dense images of a normal-mode sector
// from (a) the explicit arrays (hd, Hup, Hdw, Hnd CSR) and (b) the factored tables the kernels consume (eux / ed /
// impd, partner tables jup / jdw / coef), so that the CPU suite can compare the builders with the oracle without a
// GPU.  Compiled with g++ by tests/test_host_builders.py; never part of the product.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "host_build.hpp"
using namespace edigpu;

static std::string g_err;

extern "C" const char* host_image_error() { return g_err.c_str(); }

// out: dim x dim row-major, zeroed here.  form 0 = explicit arrays, 1 = factored tables.
extern "C" int host_normal_dense(const edigpu_model* m, int nup, int ndw, int form, double* out, int64_t dim) {
  HostNormal hn;
  g_err = build_normal(*m, nup, ndw, 0, -1, hn, form == 0);
  if (!g_err.empty()) return 1;
  const int64_t du = hn.dim_up, dd = hn.dim_dw;
  if (du * dd != dim) { g_err = "host_normal_dense: dim mismatch"; return 2; }
  std::memset(out, 0, sizeof(double) * dim * dim);
  auto at = [&](int64_t i, int64_t j) -> double& { return out[i * dim + j]; };
  for (int64_t idw = 0; idw < dd; idw++)
    for (int64_t iup = 0; iup < du; iup++) {
      const int64_t i = iup + idw * du;
      if (form == 0) at(i, i) += hn.hd[i];
      else at(i, i) += hn.fac.eux[(size_t)hn.fac.impd[idw] * du + iup] + hn.fac.ed[idw];
      for (int64_t k = hn.up.rowptr[iup]; k < hn.up.rowptr[iup + 1]; k++) at(i, hn.up.col[k] + idw * du) += hn.up.val[k];
      for (int64_t k = hn.dw.rowptr[idw]; k < hn.dw.rowptr[idw + 1]; k++) at(i, iup + hn.dw.col[k] * du) += hn.dw.val[k];
      if (!hn.has_nd) continue;
      if (form == 0) {
        for (int64_t k = hn.nd.rowptr[i]; k < hn.nd.rowptr[i + 1]; k++) at(i, hn.nd.col[k]) += hn.nd.val[k];
      } else {
        for (int t = 0; t < hn.fac.nterms; t++) {
          const uint32_t pu = hn.fac.jup[(size_t)t * du + iup], pd = hn.fac.jdw[(size_t)t * dd + idw];
          if (pu == 0xFFFFFFFFu || pd == 0xFFFFFFFFu) continue;
          const double sg = ((pu ^ pd) & 0x80000000u) ? -1.0 : 1.0;
          at(i, (int64_t)(pu & 0x7FFFFFFFu) + (int64_t)(pd & 0x7FFFFFFFu) * du) += sg * hn.fac.coef[t];
        }
      }
    }
  if (form == 1 && hn.has_nd) {
    int64_t n = 0;
    for (int64_t i = 0; i < dim; i++)
      for (int t = 0; t < hn.fac.nterms; t++)
        n += hn.fac.jup[(size_t)t * du + i % du] != 0xFFFFFFFFu && hn.fac.jdw[(size_t)t * dd + i / du] != 0xFFFFFFFFu;
    if (n != hn.nd_nnz) { g_err = "host_normal_dense: nd_nnz disagrees with the factored tables"; return 3; }
  }
  return 0;
}

// dense image of a superc / nonsu2 sector from the stored CSR (re, im interleaved), or the refusal message
extern "C" int host_flat_dense(const edigpu_model* m, int sector, double* out, int64_t dim) {
  HostFlat hf;
  g_err = build_flat(*m, sector, 0, -1, hf);
  if (!g_err.empty()) return 1;
  if (hf.dim != dim) { g_err = "host_flat_dense: dim mismatch"; return 2; }
  std::memset(out, 0, sizeof(double) * 2 * dim * dim);
  for (int64_t i = 0; i < dim; i++)
    for (int64_t k = hf.h.rowptr[i]; k < hf.h.rowptr[i + 1]; k++) {
      out[2 * (i * dim + hf.h.col[k])] += hf.h.val[2 * k];
      out[2 * (i * dim + hf.h.col[k]) + 1] += hf.h.val[2 * k + 1];
    }
  return 0;
}

// the on-the-fly image (term list + diagonal tables of build_direct) evaluated the way direct_rows_kernel does
extern "C" int host_direct_dense(const edigpu_model* m, int sector, double* out, int64_t dim) {
  HostDirect hd;
  g_err = build_direct(*m, sector, 0, -1, hd);
  if (!g_err.empty()) return 1;
  if (hd.dim != dim) { g_err = "host_direct_dense: dim mismatch"; return 2; }
  std::memset(out, 0, sizeof(double) * 2 * dim * dim);
  const uint32_t impmask = (1u << hd.norb) - 1u;
  for (int64_t i = 0; i < dim; i++) {
    const uint32_t s = (uint32_t)hd.states[i];
    double dg = hd.xtab[(((s >> hd.ns) & impmask) << hd.norb) | (s & impmask)];
    for (int byte = 0; byte < 4; byte++) dg += hd.dtab[byte * 256 + ((s >> (8 * byte)) & 255u)];
    out[2 * (i * dim + i)] += dg;
    for (const DirectTerm& t : hd.terms) {
      if ((s & t.need_set) != t.need_set || (s & t.need_clear) != 0) continue;
      const uint32_t w = s ^ t.flip;
      const auto it = std::lower_bound(hd.states.begin(), hd.states.end(), (int32_t)w);
      if (it == hd.states.end() || (uint32_t)*it != w) { g_err = "host_direct_dense: term leaves the sector"; return 3; }
      const int64_t j = it - hd.states.begin();
      const double sg = ((__builtin_popcount(s & t.sign_mask) + t.csign) & 1) ? -1.0 : 1.0;
      out[2 * (i * dim + j)] += sg * t.cre;
      out[2 * (i * dim + j) + 1] += sg * t.cim;
    }
  }
  return 0;
}

extern "C" int host_direct_refuses(const edigpu_model* m, int sector) {
  HostDirect hd;
  g_err = build_direct(*m, sector, 0, -1, hd);
  return g_err.empty() ? 0 : 1;
}

// factor_handover on the arrays of one dw-shard: dense image (local rows x global columns, diagonal and Hnd only) from
// the recovered tables; returns -1 when the arrays are not of the factored form, else the number of terms
extern "C" int host_handover_dense(int64_t du, int64_t dd, int64_t dw_first, int64_t dw_count, const double* hd,
                                   const int64_t* nd_rowptr, const int32_t* nd_col, const double* nd_val, double* out,
                                   int* nclasses) {
  HostFactored fac;
  if (!factor_handover(du, dd, dw_first, dw_count, hd, nd_rowptr, nd_col, nd_val, 16, fac)) return -1;
  const int64_t nloc = du * dw_count, dim = du * dd;
  std::memset(out, 0, sizeof(double) * nloc * dim);
  for (int64_t r = 0; r < dw_count; r++)
    for (int64_t iup = 0; iup < du; iup++) {
      const int64_t i = iup + r * du, idw = dw_first + r;
      out[i * dim + (iup + idw * du)] += fac.eux[(size_t)fac.impd[idw] * du + iup] + fac.ed[idw];
      for (int t = 0; t < fac.nterms; t++) {
        const uint32_t pu = fac.jup[(size_t)t * du + iup], pd = fac.jdw[(size_t)t * dd + idw];
        if (pu == 0xFFFFFFFFu || pd == 0xFFFFFFFFu) continue;
        out[i * dim + (int64_t)(pu & 0x7FFFFFFFu) + (int64_t)(pd & 0x7FFFFFFFu) * du] +=
            (((pu ^ pd) & 0x80000000u) ? -1.0 : 1.0) * fac.coef[t];
      }
    }
  *nclasses = fac.nimp;
  return fac.nterms;
}

// the nonsu2 sector (Ntot, twoJz) of JZ_BASIS=T from the host builder: dense (re, im interleaved), or the refusal
extern "C" int host_flat_jz_dense(const edigpu_model* m, int ntot, int twojz, double* out, int64_t dim) {
  HostFlat hf;
  g_err = build_flat(*m, ntot, 0, -1, hf, true, twojz);
  if (!g_err.empty()) return 1;
  if (hf.dim != dim) { g_err = "host_flat_jz_dense: dim mismatch"; return 2; }
  std::memset(out, 0, sizeof(double) * 2 * dim * dim);
  for (int64_t i = 0; i < dim; i++)
    for (int64_t k = hf.h.rowptr[i]; k < hf.h.rowptr[i + 1]; k++) {
      out[2 * (i * dim + hf.h.col[k])] += hf.h.val[2 * k];
      out[2 * (i * dim + hf.h.col[k]) + 1] += hf.h.val[2 * k + 1];
    }
  return 0;
}

// the same sector from its ON-THE-FLY description (build_direct with jz_basis: what the device-built image and the
// direct kernel evaluate): dense (re, im interleaved).  Returns 3 if the two-table rank of a state of the sector is not
// its position in the map.
extern "C" int host_direct_jz_dense(const edigpu_model* m, int ntot, int twojz, double* out, int64_t dim) {
  HostDirect hd;
  g_err = build_direct(*m, ntot, 0, -1, hd, true, twojz);
  if (!g_err.empty()) return 1;
  if (hd.dim != dim) { g_err = "host_direct_jz_dense: dim mismatch"; return 2; }
  const int ns = hd.ns;
  const uint32_t lomask = (1u << ns) - 1u, impmask = (1u << hd.norb) - 1u;
  auto rank = [&](uint32_t w) { return (int64_t)hd.off_dw[w >> ns] + hd.rk_up[w & lomask]; };
  std::memset(out, 0, sizeof(double) * 2 * dim * dim);
  for (int64_t i = 0; i < dim; i++) {
    const uint32_t s = (uint32_t)hd.states[i];
    if (rank(s) != i) { g_err = "host_direct_jz_dense: two-table rank != position"; return 3; }
    out[2 * (i * dim + i)] += hd.dtab[s & 255u] + hd.dtab[256 + ((s >> 8) & 255u)] + hd.dtab[512 + ((s >> 16) & 255u)] +
                              hd.dtab[768 + (s >> 24)] + hd.xtab[(((s >> ns) & impmask) << hd.norb) | (s & impmask)];
    for (const DirectTerm& t : hd.terms) {
      if ((s & t.need_set) != t.need_set || (s & t.need_clear) != 0u) continue;
      const int64_t j = rank(s ^ t.flip);
      if (j < 0 || j >= dim) { g_err = "host_direct_jz_dense: partner outside the sector"; return 4; }
      const double sg = ((__builtin_popcount(s & t.sign_mask) + (t.csign & 1)) & 1) ? -1.0 : 1.0;
      out[2 * (i * dim + j)] += sg * t.cre;
      out[2 * (i * dim + j) + 1] += sg * t.cim;
    }
  }
  return 0;
}

// the _CMPLX_NORMAL sector as one real sector on the doubled up index (build_normal_doubled): its dense REAL image of
// size (2 dim) x (2 dim) from the factored tables, rows / columns ordered (idw, iup, re|im) -- i.e. the interleaved
// complex layout read as real
extern "C" int host_normal_doubled_dense(const edigpu_model* m, int nup, int ndw, double* out, int64_t dim2, int* nterms) {
  HostNormal hn;
  g_err = build_normal_doubled(*m, nup, ndw, hn, 16);
  if (!g_err.empty()) return 1;
  const int64_t du = hn.dim_up, dd = hn.dim_dw;
  if (du * dd != dim2) { g_err = "host_normal_doubled_dense: dim mismatch"; return 2; }
  std::memset(out, 0, sizeof(double) * dim2 * dim2);
  auto at = [&](int64_t i, int64_t j) -> double& { return out[i * dim2 + j]; };
  for (int64_t idw = 0; idw < dd; idw++)
    for (int64_t iup = 0; iup < du; iup++) {
      const int64_t i = iup + idw * du;
      at(i, i) += hn.fac.eux[(size_t)hn.fac.impd[idw] * du + iup] + hn.fac.ed[idw];
      for (int64_t k = hn.up.rowptr[iup]; k < hn.up.rowptr[iup + 1]; k++) at(i, hn.up.col[k] + idw * du) += hn.up.val[k];
      for (int64_t k = hn.dw.rowptr[idw]; k < hn.dw.rowptr[idw + 1]; k++) at(i, iup + hn.dw.col[k] * du) += hn.dw.val[k];
      for (int t = 0; t < hn.fac.nterms; t++) {
        const uint32_t pu = hn.fac.jup[(size_t)t * du + iup], pd = hn.fac.jdw[(size_t)t * dd + idw];
        if (pu == 0xFFFFFFFFu || pd == 0xFFFFFFFFu) continue;
        at(i, (int64_t)(pu & 0x7FFFFFFFu) + (int64_t)(pd & 0x7FFFFFFFu) * du) +=
            (((pu ^ pd) & 0x80000000u) ? -1.0 : 1.0) * hn.fac.coef[t];
      }
    }
  *nterms = hn.fac.nterms;
  return 0;
}
