// Test-only shim: the impurity-block image (csrc/host_ib.cpp) and the per-block routines the gfx950 kernels run
// (csrc/ib_core.hpp) evaluated on the host, against the explicit arrays of the same sector (hd, Hup, Hdw, Hnd CSR from
// csrc/host_build.cpp).  Compiled with g++ by tests/test_host_ib.py; never part of the product.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "host_build.hpp"
#include "host_ib.hpp"
#include "ib_core.hpp"
using namespace edigpu;

static std::string g_err;
extern "C" const char* host_ib_error() { return g_err.c_str(); }

namespace {

int64_t vec_len(const HostIb& ib) { return (int64_t)ib.npanels * kIbPanel * ib.dw.dim; }
int64_t vec_at(const HostIb& ib, int64_t row, int pos) {
  return (int64_t)(pos / kIbPanel) * ib.dw.dim * kIbPanel + row * kIbPanel + pos % kIbPanel;
}

template <int NORB>
void emulate_rows(const HostIb& ib, const std::vector<double>& v, std::vector<double>& hv) {
  const int plen = ib.npanels * kIbPanel, nimp = 1 << NORB;
  std::vector<double> img((size_t)ib.rimg_len, 0.0), res((size_t)ib.rimg_len, 0.0);
  ib::RowImage im;
  im.row = img.data();
  im.rank = ib.urank.data();
  for (int c = 0; c < 5; c++) {
    im.cb[c] = ib.rcb[c];
    im.cs[c] = ib.rcs[c];
  }
  for (int64_t r = 0; r < ib.dw.dim; r++) {
    for (int p = 0; p < plen; p++) img[ib.rmap[p]] = v[vec_at(ib, r, p)];   // (padding: zeros into the last word)
    res = img;   // words no block owns (slack, the zero word) leave as they came
    for (int n = 0; n <= NORB; n++)
      for (int q = ib.ucls[n]; q < ib.ucls[n + 1]; q++) {
        const uint16_t e = ib.ublist[q];
        const uint32_t b = e & 0x7FFFu;
        const uint32_t i = (uint32_t)(q - ib.ucls[n]);
        ib::for_class<NORB>(n, [&](auto N) {
          constexpr int nn = decltype(N)::value;
          double acc[ib::binom(NORB, nn)];
          ib::rows_block<NORB, nn>(im, b, i, ib.up.nb, ib.up.vtab.data(), ib.up.timp.data(), ib.ed[r],
                                   &ib.xu[(size_t)ib.impd[r] * nimp], acc, (int)ib.up.pmask.size(), ib.up.pmask.data(), ib.up.pt.data());
          if (!(e & kIbSkip))
            for (int j = 0; j < ib::binom(NORB, nn); j++) res[(size_t)ib.rcb[nn + 1] + (size_t)j * ib.rcs[nn + 1] + i] = acc[j];
        });
      }
    for (int p = 0; p < plen; p++) hv[vec_at(ib, r, p)] = res[ib.rmap[p]];
  }
}

// split rows: one half image at a time, the hop over the top bath level from the vector itself
template <int NORB>
void emulate_rows_split(const HostIb& ib, const std::vector<double>& v, std::vector<double>& hv) {
  const int nimp = 1 << NORB, top = ib.up.nb - 1;
  for (int h = 0; h < 2; h++) {
    const IbUpHalf& hf = ib.half[h];
    const int plen = hf.npanels * kIbPanel, p0 = hf.panel0 * kIbPanel;
    std::vector<double> img((size_t)hf.rimg_len, 0.0), res;
    ib::RowImage im;
    im.row = img.data();
    im.rank = ib.urank_low.data();
    for (int c = 0; c < 5; c++) {
      im.cb[c] = hf.rcb[c];
      im.cs[c] = hf.rcs[c];
    }
    const double eps_top = h ? ib.up.vtab[(size_t)top * 4 + 3] : 0.0;
    for (int64_t r = 0; r < ib.dw.dim; r++) {
      for (int p = 0; p < plen; p++) img[hf.rmap[p]] = v[vec_at(ib, r, p0 + p)];
      res = img;
      for (int n = 0; n <= NORB; n++)
        for (int q = hf.ucls[n]; q < hf.ucls[n + 1]; q++) {
          const uint16_t e = hf.ublist[q];
          const uint32_t bl = e & 0x7FFFu;
          const uint32_t i = (uint32_t)(q - hf.ucls[n]);
          ib::for_class<NORB>(n, [&](auto N) {
            constexpr int nn = decltype(N)::value;
            double acc[ib::binom(NORB, nn)];
            ib::rows_block<NORB, nn>(im, bl, i, top, ib.up.vtab.data(), ib.up.timp.data(), ib.ed[r] + eps_top,
                                     &ib.xu[(size_t)ib.impd[r] * nimp], acc);
            auto top_hop = [&](auto TS) {
              constexpr bool ts = decltype(TS)::value;
              constexpr int MP = ib::rows_top_words<NORB, nn, ts>();
              if constexpr (MP > 0) {
                double xp[MP];
                for (int j = 0; j < MP; j++) xp[j] = hf.utop[q] == kIbNone ? 0.0 : v[vec_at(ib, r, hf.utop[q] + j)];
                ib::rows_top<NORB, nn, ts>(bl, &ib.up.vtab[(size_t)top * 4], xp, acc);
              }
            };
            if (h) top_hop(std::true_type{}); else top_hop(std::false_type{});
            if (!(e & kIbSkip))
              for (int j = 0; j < ib::binom(NORB, nn); j++) res[(size_t)hf.rcb[nn + 1] + (size_t)j * hf.rcs[nn + 1] + i] = acc[j];
          });
        }
      for (int p = 0; p < plen; p++) hv[vec_at(ib, r, p0 + p)] = res[hf.rmap[p]];
    }
  }
}

template <int NORB>
void emulate_cols(const HostIb& ib, const std::vector<double>& v, std::vector<double>& hv) {
  const int64_t dd = ib.dw.dim, ps = dd * kIbPanel;
  const int nch = (int)ib.chunk_row.size() - 1;
  for (int pn = 0; pn < ib.npanels; pn++)
    for (int c = 0; c < nch; c++) {
      const int row0 = ib.chunk_row[c];
      const double* chunk = &v[(size_t)pn * ps + (size_t)row0 * kIbPanel];
      const int32_t* cls = &ib.dcls[(size_t)c * (kIbMaxNorb + 2)];
      for (int n = 0; n <= NORB; n++)
        for (int q = cls[n]; q < cls[n + 1]; q++) {
          const uint16_t e = ib.dblist[(size_t)ib.chunk_blk[c] + q];
          if (e & kIbSkip) continue;
          const uint32_t b = e & 0x7FFFu;
          const int own = ib.dw.first[b];
          for (int col = 0; col < kIbPanel; col += 2)
            ib::for_class<NORB>(n, [&](auto N) {
              constexpr int nn = decltype(N)::value;
              constexpr int M = ib::binom(NORB, nn);
              ib::Pair acc[M];
              for (int j = 0; j < M; j++) acc[j].x = acc[j].y = 0.0;
              auto gload = [&](int grow) -> ib::Pair {
                const double* g = &v[(size_t)pn * ps + (size_t)grow * kIbPanel + col];
                return ib::Pair{g[0], g[1]};
              };
              ib::cols_block<NORB, nn>(chunk, row0, b, own, &ib.dmeta[(size_t)b * 16], ib.dw.nb, ib.lowbits,
                                       ib.dw.vtab.data(), ib.dw.timp.data(), col, gload, acc, (int)ib.dw.pmask.size(),
                                       ib.dw.pmask.data(), ib.dw.pt.data(), ib.dmeta.data());
              if (ib.nterms > 0)
                ib::cols_block_nd<NORB, nn>(chunk, own - row0, col, ib.nterms, ib.ndcoef.data(), ib.nd_dw.data(),
                                            &ib.nd_up[(size_t)pn * kIbPanel], ib.npanels * kIbPanel, acc);
              for (int j = 0; j < M; j++) {
                double* h = &hv[(size_t)pn * ps + (size_t)(own + j) * kIbPanel + col];
                h[0] += acc[j].x;
                h[1] += acc[j].y;
              }
            });
        }
    }
}

}  // namespace

// H*v of the sector (nup, ndw) for a seeded vector, through the explicit arrays and through the impurity-block image.
// info: [0] valid, [1] lowbits, [2] chunks, [3] largest chunk, [4] panels, [5] Hnd terms, [6] padded columns.
// Returns 0 and *maxdiff = max |difference| / max |reference|; 1 when the image is refused (message in
// host_ib_error()); 2 on a builder error.
// lds_budget as build_ib takes it (bytes; < 0: always split; info[7] = halves).
extern "C" int host_ib_check2(const edigpu_model* m, int nup, int ndw, int max_chunk_rows, int lds_budget, int32_t* info,
                              double* maxdiff) {
  HostNormal hn;
  g_err = build_normal(*m, nup, ndw, 0, -1, hn, true);
  if (!g_err.empty()) return 2;
  HostIb ib;
  build_ib(hn, max_chunk_rows, ib, lds_budget);
  std::memset(info, 0, 8 * sizeof(int32_t));
  if (!ib.valid) {
    g_err = ib.why;
    return 1;
  }
  info[0] = 1;
  info[1] = ib.lowbits;
  info[2] = (int)ib.chunk_row.size() - 1;
  info[3] = ib.max_chunk_rows;
  info[4] = ib.npanels;
  info[5] = ib.nterms;
  info[6] = ib.npanels * kIbPanel - (int)hn.dim_up;
  info[7] = ib.nhalf;
  const int64_t du = hn.dim_up, dd = hn.dim_dw, dim = du * dd;
  std::vector<double> v((size_t)dim), ref((size_t)dim, 0.0);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (auto& x : v) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    x = (double)((int64_t)(s >> 11) - ((int64_t)1 << 52)) / (double)((int64_t)1 << 52);
  }
  for (int64_t idw = 0; idw < dd; idw++)
    for (int64_t iup = 0; iup < du; iup++) {
      const int64_t i = iup + idw * du;
      double t = hn.hd[i] * v[i];
      for (int64_t k = hn.up.rowptr[iup]; k < hn.up.rowptr[iup + 1]; k++) t += hn.up.val[k] * v[hn.up.col[k] + idw * du];
      for (int64_t k = hn.dw.rowptr[idw]; k < hn.dw.rowptr[idw + 1]; k++) t += hn.dw.val[k] * v[iup + hn.dw.col[k] * du];
      if (hn.has_nd)
        for (int64_t k = hn.nd.rowptr[i]; k < hn.nd.rowptr[i + 1]; k++) t += hn.nd.val[k] * v[hn.nd.col[k]];
      ref[i] = t;
    }
  std::vector<double> vi((size_t)vec_len(ib), 0.0), hi((size_t)vec_len(ib), 0.0);
  for (int64_t idw = 0; idw < dd; idw++)
    for (int64_t iup = 0; iup < du; iup++) vi[vec_at(ib, idw, ib.pos[iup])] = v[iup + idw * du];
  switch (ib.norb) {
    case 1: ib.nhalf == 2 ? emulate_rows_split<1>(ib, vi, hi) : emulate_rows<1>(ib, vi, hi); emulate_cols<1>(ib, vi, hi); break;
    case 2: ib.nhalf == 2 ? emulate_rows_split<2>(ib, vi, hi) : emulate_rows<2>(ib, vi, hi); emulate_cols<2>(ib, vi, hi); break;
    case 3: ib.nhalf == 2 ? emulate_rows_split<3>(ib, vi, hi) : emulate_rows<3>(ib, vi, hi); emulate_cols<3>(ib, vi, hi); break;
    default: g_err = "norb"; return 2;
  }
  double worst = 0.0, scale = 0.0;
  for (int64_t idw = 0; idw < dd; idw++)
    for (int64_t iup = 0; iup < du; iup++) {
      const double got = hi[vec_at(ib, idw, ib.pos[iup])], want = ref[iup + idw * du];
      worst = std::max(worst, std::fabs(got - want));
      scale = std::max(scale, std::fabs(want));
    }
  // the padding columns must stay zero
  {
    std::vector<char> real((size_t)ib.npanels * kIbPanel, 0);
    for (int64_t iup = 0; iup < du; iup++) real[ib.pos[iup]] = 1;
    for (int p = 0; p < ib.npanels * kIbPanel; p++)
      if (!real[p])
        for (int64_t idw = 0; idw < dd; idw++)
          if (hi[vec_at(ib, idw, p)] != 0.0) {
            g_err = "a padding column received a value";
            return 2;
          }
  }
  *maxdiff = scale > 0.0 ? worst / scale : worst;
  return 0;
}

extern "C" int host_ib_check(const edigpu_model* m, int nup, int ndw, int max_chunk_rows, int32_t* info, double* maxdiff) {
  int32_t tmp[8];
  const int rc = host_ib_check2(m, nup, ndw, max_chunk_rows, 0, tmp, maxdiff);
  std::memcpy(info, tmp, 7 * sizeof(int32_t));
  return rc;
}
