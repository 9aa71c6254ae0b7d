// Test-only shim: the local-block tables (csrc/host_sb.cpp) and the per-block routines the gfx950 kernels run
// (csrc/sb_core.hpp) evaluated on the host, against the explicit arrays of the same sector (hd, Hup, Hdw, Hnd CSR from
// csrc/host_build.cpp).  Compiled with g++ by tests/test_host_sb.py; never part of the product.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "host_build.hpp"
#include "host_ib.hpp"
#include "host_sb.hpp"
#include "sb_core.hpp"
using namespace edigpu;

static std::string g_err;
extern "C" const char* host_sb_error() { return g_err.c_str(); }

namespace {

int64_t vec_len(const HostIb& ib) { return (int64_t)ib.npanels * kIbPanel * ib.dw.dim; }
int64_t vec_at(const HostIb& ib, int64_t row, int pos) {
  return (int64_t)(pos / kIbPanel) * ib.dw.dim * kIbPanel + row * kIbPanel + pos % kIbPanel;
}

template <int NIMP, int NB0, int AMODE>
void emulate_rows(const HostIb& ib, const HostSb& sbt, const std::vector<double>& v, std::vector<double>& hv) {
  constexpr int NLOC = NIMP + NB0;
  const int plen = ib.npanels * kIbPanel, nimp = 1 << NIMP, nwv = sbt.rows_nt / 64;
  std::vector<double> img((size_t)sbt.rimg_len, 0.0), res((size_t)sbt.rimg_len, 0.0);
  sb::RowImage im;
  im.row = img.data();
  im.rank = sbt.urank.data();
  im.ebath = sbt.ebw.data();
  im.cs = sbt.rcs;
  for (int64_t r = 0; r < ib.dw.dim; r++) {
    for (int p = 0; p < plen; p++) img[sbt.rmap[p]] = v[vec_at(ib, r, p)];  // (padding: zeros into the last word)
    res = img;
    for (int s = 0; s < sbt.rows_nbt; s++)
      for (int wv = 0; wv < nwv; wv++) {
        const int32_t sd = sbt.uslot[(size_t)s * nwv + wv];
        if (sd < 0) continue;
        const int n = sd & 0xFF, i0 = sd >> 8;
        for (int l = 0; l < 64; l++) {
          const uint16_t e = sbt.ublist[((size_t)s * nwv + wv) * 64 + l];
          const uint32_t w = e & 0x7FFFu, i = (uint32_t)(i0 + l);
          sb::for_class<NLOC>(n, [&](auto N) {
            constexpr int nn = decltype(N)::value;
            double acc[sb::binom(NLOC, nn)];
            sb::rows_block<NIMP, NB0, AMODE, nn, 0>(im, w, i, sbt.up.nbw, sbt.up.vtab.data(), 4, sbt.up.korb.data(), sbt.up.tloc.data(),
                                                 ib.ed[r], &ib.xu[(size_t)ib.impd[r] * nimp], sbt.e0.data(), acc);
            if (!(e & kIbSkip))
              for (int j = 0; j < sb::binom(NLOC, nn); j++) res[(size_t)(sb::wbase(NLOC, nn) + j) * sbt.rcs + i] = acc[j];
          });
        }
      }
    for (int p = 0; p < plen; p++) hv[vec_at(ib, r, p)] = res[sbt.rmap[p]];
  }
}

// rows staged in halves: one image per value of the top walked bit, the hop over the top level from the vector itself
template <int NIMP, int NB0, int AMODE>
void emulate_rows_split(const HostIb& ib, const HostSb& sbt, const std::vector<double>& v, std::vector<double>& hv) {
  constexpr int NLOC = NIMP + NB0;
  const int nimp = 1 << NIMP, nwv = sbt.rows_nt / 64, top = sbt.up.nbw - 1;
  for (int h = 0; h < 2; h++) {
    const SbUpHalf& hf = sbt.half[h];
    const int plen = hf.npanels * kIbPanel, p00 = hf.panel0 * kIbPanel;
    std::vector<double> img((size_t)sbt.rimg_len, 0.0), res((size_t)sbt.rimg_len, 0.0);
    sb::RowImage im;
    im.row = img.data();
    im.rank = sbt.urank.data();
    im.ebath = hf.ebw.data();
    im.cs = sbt.rcs;
    for (int64_t r = 0; r < ib.dw.dim; r++) {
      for (int p = 0; p < plen; p++) img[hf.rmap[p]] = v[vec_at(ib, r, p00 + p)];
      res = img;
      for (int s = 0; s < sbt.rows_nbt; s++)
        for (int wv = 0; wv < nwv; wv++) {
          const int32_t sd = hf.uslot[(size_t)s * nwv + wv];
          if (sd < 0) continue;
          const int n = sd & 0xFF, i0 = sd >> 8;
          for (int l = 0; l < 64; l++) {
            const size_t at = ((size_t)s * nwv + wv) * 64 + l;
            const uint16_t e = hf.ublist[at];
            const uint32_t wl = e & 0x7FFFu, i = (uint32_t)(i0 + l);
            sb::for_class<NLOC>(n, [&](auto N) {
              constexpr int nn = decltype(N)::value;
              double acc[sb::binom(NLOC, nn)];
              sb::rows_block<NIMP, NB0, AMODE, nn, 0>(im, wl, i, top, sbt.up.vtab.data(), 4, sbt.up.korb.data(), sbt.up.tloc.data(),
                                                   ib.ed[r], &ib.xu[(size_t)ib.impd[r] * nimp], sbt.e0.data(), acc);
              auto top_hop = [&](auto TS) {
                constexpr bool ts = decltype(TS)::value;
                constexpr int MP = sb::rows_top_words<NLOC, nn, ts>();
                if constexpr (MP > 0) {
                  double xp[MP];
                  const int jg = hf.ugap[at] & 0x0F, g = hf.ugap[at] >> 4;
                  for (int j = 0; j < MP; j++)
                    xp[j] = hf.utop[at] == kIbNone ? 0.0 : v[vec_at(ib, r, hf.utop[at] + j + (j >= jg ? g : 0))];
                  sb::rows_top<NIMP, NB0, nn, ts>(wl, &sbt.up.vtab[(size_t)top * 4], xp, acc);
                }
              };
              if (h) top_hop(std::true_type{}); else top_hop(std::false_type{});
              if (!(e & kIbSkip))
                for (int j = 0; j < sb::binom(NLOC, nn); j++) res[(size_t)(sb::wbase(NLOC, nn) + j) * sbt.rcs + i] = acc[j];
            });
          }
        }
      for (int p = 0; p < plen; p++) hv[vec_at(ib, r, p00 + p)] = res[hf.rmap[p]];
    }
  }
}

template <int NIMP, int NB0, int AMODE, class T>
void emulate_cols(const HostIb& ib, const HostSb& sbt, const std::vector<double>& v, std::vector<double>& hv) {
  constexpr int NLOC = NIMP + NB0;
  constexpr int CW = (int)(sizeof(T) / sizeof(double));
  const int gs = sbt.cols_gs;
  const int64_t dd = ib.dw.dim, ps = dd * kIbPanel;
  const int nch = (int)sbt.chunk_row.size() - 1;
  for (int pn = 0; pn < ib.npanels; pn++)
    for (int c = 0; c < nch; c++) {
      const int row0 = sbt.chunk_row[c];
      const double* chunk = &v[(size_t)pn * ps + (size_t)row0 * kIbPanel];
      for (int q = sbt.chunk_slot[c]; q < sbt.chunk_slot[c + 1]; q++) {
        const int n = sbt.dslot[q];
        if (n < 0) continue;
        for (int g = 0; g < gs; g++) {
          const uint16_t e = sbt.dblist[(size_t)q * gs + g];
          if (e & kIbSkip) continue;
          const uint32_t w = e & 0x7FFFu;
          const int own = sbt.dw.first[w];
          for (int col = 0; col < kIbPanel; col += CW)
            sb::for_class<NLOC>(n, [&](auto N) {
              constexpr int nn = decltype(N)::value;
              constexpr int M = sb::binom(NLOC, nn);
              T acc[M];
              std::memset(acc, 0, sizeof(acc));
              auto gload = [&](int grow) -> const double* { return &v[(size_t)pn * ps + (size_t)grow * kIbPanel + col]; };
              sb::cols_block<NIMP, NB0, AMODE, nn, T>(chunk, row0, w, w >> sbt.lowbits, own, &sbt.dmeta[(size_t)w * 16], sbt.dw.nbw, sbt.lowbits,
                                                   sbt.dw.vtab.data(), 4, sbt.dw.korb.data(), sbt.dw.tloc.data(), col, gload, acc, [] {});
              if (ib.nterms > 0)
                sb::cols_block_nd<NIMP, NB0, nn, T>(chunk, own - row0, col, ib.nterms, ib.ndcoef.data(), sbt.nd_dw.data(),
                                                 &ib.nd_up[(size_t)pn * kIbPanel], ib.npanels * kIbPanel, acc);
              for (int j = 0; j < M; j++) {
                double* h = &hv[(size_t)pn * ps + (size_t)(own + j) * kIbPanel + col];
                for (int cc = 0; cc < CW; cc++) h[cc] += reinterpret_cast<const double*>(&acc[j])[cc];
              }
            });
        }
      }
    }
}

template <int NIMP, int NB0>
void emulate(const HostIb& ib, const HostSb& sbt, const std::vector<double>& v, std::vector<double>& hv) {
  if (sbt.amode == 1) {
    if constexpr (NIMP > 1) {
      if (sbt.nhalf == 2) emulate_rows_split<NIMP, NB0, 1>(ib, sbt, v, hv); else emulate_rows<NIMP, NB0, 1>(ib, sbt, v, hv);
      if (sbt.cols_gs == 8) emulate_cols<NIMP, NB0, 1, sb::Pair>(ib, sbt, v, hv); else emulate_cols<NIMP, NB0, 1, double>(ib, sbt, v, hv);
    }
  } else {
    if (sbt.nhalf == 2) emulate_rows_split<NIMP, NB0, 0>(ib, sbt, v, hv); else emulate_rows<NIMP, NB0, 0>(ib, sbt, v, hv);
    if (sbt.cols_gs == 8) emulate_cols<NIMP, NB0, 0, sb::Pair>(ib, sbt, v, hv); else emulate_cols<NIMP, NB0, 0, double>(ib, sbt, v, hv);
  }
}

}  // namespace

// H*v of the sector (nup, ndw) for a seeded vector, through the explicit arrays and through the local-block tables with
// nb0 low bath levels folded into the blocks.  info: [0] valid, [1] lowbits, [2] chunks, [3] largest chunk, [4] amode,
// [5] Hnd terms, [6] wave-slots of the rows kernel in use, [7] local levels.
// Returns 0 and *maxdiff = max |difference| / max |reference|; 1 when the tables are refused (message in
// host_sb_error()); 2 on a builder error.
static int sb_check(const edigpu_model* m, int nup, int ndw, int nb0, int max_chunk_rows, int rows_nt, int rows_nbt,
                    int cols_nw, int cols_gs, int lds_budget, int32_t* info, double* maxdiff);
extern "C" int host_sb_check(const edigpu_model* m, int nup, int ndw, int nb0, int max_chunk_rows, int rows_nt, int rows_nbt,
                             int cols_nw, int cols_gs, int32_t* info, double* maxdiff) {
  return sb_check(m, nup, ndw, nb0, max_chunk_rows, rows_nt, rows_nbt, cols_nw, cols_gs, 0, info, maxdiff);
}
// the same with the rows staged in halves (impurity-block image built with lds_budget < 0: always split)
extern "C" int host_sb_check_split(const edigpu_model* m, int nup, int ndw, int nb0, int max_chunk_rows, int rows_nt, int rows_nbt,
                                   int cols_nw, int cols_gs, int32_t* info, double* maxdiff) {
  return sb_check(m, nup, ndw, nb0, max_chunk_rows, rows_nt, rows_nbt, cols_nw, cols_gs, -1, info, maxdiff);
}
static int sb_check(const edigpu_model* m, int nup, int ndw, int nb0, int max_chunk_rows, int rows_nt, int rows_nbt,
                    int cols_nw, int cols_gs, int lds_budget, int32_t* info, double* maxdiff) {
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
  HostSb sbt;
  build_sb(hn, ib, nb0, max_chunk_rows, rows_nt, rows_nbt, cols_nw, sbt, cols_gs);
  if (!sbt.valid) {
    g_err = sbt.why;
    return 1;
  }
  info[0] = 1;
  info[1] = sbt.lowbits;
  info[2] = (int)sbt.chunk_row.size() - 1;
  info[3] = sbt.max_chunk_rows;
  info[4] = sbt.amode;
  info[5] = ib.nterms;
  for (int32_t sd : sbt.uslot) info[6] += sd >= 0;
  if (sbt.nhalf == 2) info[6] = 200 + (int)sbt.half[0].uslot.size();
  info[7] = sbt.nloc;
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
  const int key = sbt.norb * 10 + sbt.nb0;
  switch (key) {
    case 11: emulate<1, 1>(ib, sbt, vi, hi); break;
    case 12: emulate<1, 2>(ib, sbt, vi, hi); break;
    case 13: emulate<1, 3>(ib, sbt, vi, hi); break;
    case 14: emulate<1, 4>(ib, sbt, vi, hi); break;
    case 21: emulate<2, 1>(ib, sbt, vi, hi); break;
    case 22: emulate<2, 2>(ib, sbt, vi, hi); break;
    case 23: emulate<2, 3>(ib, sbt, vi, hi); break;
    case 31: emulate<3, 1>(ib, sbt, vi, hi); break;
    case 32: emulate<3, 2>(ib, sbt, vi, hi); break;
    case 33: emulate<3, 3>(ib, sbt, vi, hi); break;
    default: g_err = "no instantiation for this (norb, nb0)"; return 2;
  }
  double worst = 0.0, scale = 0.0;
  for (int64_t idw = 0; idw < dd; idw++)
    for (int64_t iup = 0; iup < du; iup++) {
      const double got = hi[vec_at(ib, idw, ib.pos[iup])], want = ref[iup + idw * du];
      worst = std::max(worst, std::fabs(got - want));
      scale = std::max(scale, std::fabs(want));
    }
  {  // the padding columns must stay zero
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
