// host_ib.cpp -- see host_ib.hpp.  Product code (host side of kernels_ib.hip); shares nothing with oracle/.
#include "host_ib.hpp"

#include <algorithm>
#include <cmath>

namespace edigpu {

namespace {

inline int popc(uint32_t x) { return __builtin_popcount(x); }

// the species tables; "" on success
std::string build_side(const HostNormal& hn, int sp, const CombBasis& bs, int npart, IbSide& s) {
  const int ns = hn.ns, norb = hn.norb, nb = ns - norb;
  s.ns = ns;
  s.npart = npart;
  s.norb = norb;
  s.nb = nb;
  s.dim = bs.size();
  const std::vector<double>& a = hn.ob_a[sp];
  const std::vector<double>& eps = hn.ob_eps[sp];
  if ((int)a.size() != ns * ns || (int)eps.size() != ns) return "no one-body data";
  if (s.dim >= 0xFFF0) return "more than 65519 states per species";
  for (int p = 0; p < ns; p++)
    for (int q = 0; q < ns; q++) {
      if (p == q) continue;
      if (a[p * ns + q] != a[q * ns + p]) return "hop matrix not symmetric";
    }
  s.pmask.clear();
  s.pt.clear();
  for (int p = norb; p < ns; p++)
    for (int q = p + 1; q < ns; q++)
      if (a[p * ns + q] != 0.0) {
        const int k1 = p - norb, k2 = q - norb;
        const uint32_t btw = ((1u << k2) - 1u) & ~((1u << (k1 + 1)) - 1u);
        s.pmask.push_back((1u << k1) | (1u << k2) | (btw << 16));
        s.pt.push_back(a[p * ns + q]);
      }
  if ((int)s.pmask.size() > kIbMaxPairs) return "more than 64 bath-bath hops";
  s.vtab.assign((size_t)nb * 4, 0.0);
  for (int k = 0; k < nb; k++)
    for (int ia = 0; ia < norb; ia++) s.vtab[(size_t)k * 4 + ia] = a[(norb + k) * ns + ia];
  for (int k = 0; k < nb; k++) s.vtab[(size_t)k * 4 + 3] = eps[norb + k];  // (norb <= 3: the fourth entry is free)
  s.timp.assign((size_t)norb * norb, 0.0);
  for (int p = 0; p < norb; p++)
    for (int q = 0; q < norb; q++)
      if (p != q) s.timp[(size_t)p * norb + q] = a[p * ns + q];
  const size_t nw = (size_t)1 << nb;
  s.first.assign(nw, kIbNone);
  for (int64_t i = 0; i < s.dim; i++) {
    const uint32_t st = (uint32_t)bs.states[i], b = st >> norb;
    if (s.first[b] == kIbNone) s.first[b] = (uint16_t)i;
    // the block is the ascending list of the impurity patterns with n bits: position inside = rank of the pattern
    const uint32_t p = st & ((1u << norb) - 1u);
    int r = 0;
    for (uint32_t q = 0; q < p; q++) r += popc(q) == popc(p);
    if ((int64_t)s.first[b] + r != i) return "basis not in block order";
  }
  s.ebath.assign(nw, 0.0);
  for (size_t b = 0; b < nw; b++) {
    double e = 0.0;
    for (int k = 0; k < nb; k++)
      if ((b >> k) & 1u) e += eps[norb + k];
    s.ebath[b] = e;
  }
  s.eimp.assign((size_t)1 << norb, 0.0);
  for (uint32_t p = 0; p < (1u << norb); p++) {
    double e = 0.0;
    for (int ia = 0; ia < norb; ia++)
      if ((p >> ia) & 1u) e += eps[ia];
    for (int ia = 0; ia < norb; ia++)
      for (int ib = ia + 1; ib < norb; ib++)
        if (((p >> ia) & 1u) && ((p >> ib) & 1u)) e += hn.samespin[(size_t)ia * norb + ib];
    s.eimp[p] = e;
  }
  return "";
}

}  // namespace

void build_ib(const HostNormal& hn, int max_chunk_rows, HostIb& out, int lds_budget) {
  out = HostIb();
  auto fail = [&](const std::string& w) {
    out.valid = false;
    out.why = w;
  };
  const int norb = hn.norb, ns = hn.ns, nb = ns - norb;
  if (norb < 1 || norb > kIbMaxNorb) return fail("impurity levels per species not in 1..3");
  if (nb < 1 || nb > kIbMaxBath) return fail("bath levels per species not in 1..14");
  if (hn.dw_first != 0 || hn.dw_count != hn.dim_dw) return fail("the handle is a shard");
  if (!hn.fac.valid || hn.fac.nterms > kIbMaxTerms) return fail("no factored tables / too many Hnd terms");
  out.norb = norb;
  std::string e = build_side(hn, 0, hn.bup, hn.nup, out.up);
  if (e.empty()) e = build_side(hn, 1, hn.bdw, hn.ndw, out.dw);
  if (!e.empty()) return fail(e);
  const int64_t du = hn.dim_up, dd = hn.dim_dw;
  const uint32_t impmask = (1u << norb) - 1u;
  const size_t nw = (size_t)1 << nb;
  auto cls_of = [&](const IbSide& s, uint32_t b) { return s.npart - popc(b); };
  auto rows_of = [&](const IbSide& s, uint32_t b) { return (int)binomial(norb, cls_of(s, b)); };

  // ---- up side: positions, block list ----
  const bool pad = hn.fac.nterms > 0;
  const int top = nb - 1;
  // lays the columns out (split: the first block with the top bath bit set starts a panel) and builds the row image
  auto build_up = [&](bool split) -> std::string {
    out.pos.assign((size_t)du, 0);
    out.upos.assign(nw, kIbNone);
    out.ublist.clear();
    int cur = 0, panel_b = -1;
    for (uint32_t b = 0; b < nw; b++) {
      if (out.up.first[b] == kIbNone) continue;
      const int m = rows_of(out.up, b);
      if (split && panel_b < 0 && (b >> top)) {
        cur = (cur + kIbPanel - 1) / kIbPanel * kIbPanel;
        panel_b = cur / kIbPanel;
      }
      if (pad && (cur % kIbPanel) + m > kIbPanel) cur = (cur / kIbPanel + 1) * kIbPanel;
      out.upos[b] = (uint16_t)cur;
      for (int j = 0; j < m; j++) out.pos[(size_t)out.up.first[b] + j] = cur + j;
      cur += m;
    }
    out.npanels = (cur + kIbPanel - 1) / kIbPanel;
    if (out.npanels * kIbPanel >= 0xFFF0) return "padded row longer than 65519 columns";
    for (int n = 0; n <= norb; n++) {
      out.ucls[n] = (int)out.ublist.size();
      uint16_t firstb = kIbNone;
      for (uint32_t b = 0; b < nw; b++)
        if (out.up.first[b] != kIbNone && cls_of(out.up, b) == n) {
          if (firstb == kIbNone) firstb = (uint16_t)b;
          out.ublist.push_back((uint16_t)b);
        }
      while (out.ublist.size() % 64) out.ublist.push_back((uint16_t)(firstb | kIbSkip));
    }
    out.ucls[norb + 1] = (int)out.ublist.size();
    {
      // LDS image of a row: classes one after the other, each as C(norb, n) arrays of (padded) class size
      out.urank.assign(nw, kIbNone);
      int at = 0, maxcs = 0;
      for (int n = 0; n <= norb; n++) {
        const int cs = out.ucls[n + 1] - out.ucls[n];
        out.rcb[n + 1] = at;
        out.rcs[n + 1] = cs;
        maxcs = std::max(maxcs, cs);
        at += (int)binomial(norb, n) * cs;
        for (int q = out.ucls[n]; q < out.ucls[n + 1]; q++)
          if (!(out.ublist[q] & kIbSkip)) out.urank[out.ublist[q]] = (uint16_t)(q - out.ucls[n]);
      }
      out.rimg_len = at + 2 * maxcs + 8;
      if (out.rimg_len >= 0xFFF0) return "row image longer than 65519 words";
      out.rmap.assign((size_t)out.npanels * kIbPanel, (uint16_t)(out.rimg_len - 1));
      for (uint32_t b = 0; b < nw; b++) {
        if (out.up.first[b] == kIbNone) continue;
        const int n = cls_of(out.up, b), m = rows_of(out.up, b);
        for (int j = 0; j < m; j++)
          out.rmap[(size_t)out.upos[b] + j] = (uint16_t)(out.rcb[n + 1] + j * out.rcs[n + 1] + out.urank[b]);
      }
    }
    out.nhalf = 1;
    if (!split) return "";
    if (panel_b <= 0 || panel_b >= out.npanels) return "a half of the split row is empty";
    const uint32_t lowmask = (1u << top) - 1u;
    out.urank_low.assign((size_t)1 << top, 0);
    {
      std::vector<int> cnt(nb + 1, 0);
      for (uint32_t w = 0; w <= lowmask; w++) out.urank_low[w] = (uint16_t)cnt[popc(w)]++;
    }
    for (int h = 0; h < 2; h++) {
      IbUpHalf& hf = out.half[h];
      hf = IbUpHalf();
      hf.panel0 = h ? panel_b : 0;
      hf.npanels = h ? out.npanels - panel_b : panel_b;
      int at = 0, maxcs = 0;
      for (int n = 0; n <= norb; n++) {
        hf.ucls[n] = (int)hf.ublist.size();
        uint16_t firstw = kIbNone;
        int count = 0;
        for (uint32_t b = 0; b < nw; b++)
          if ((int)(b >> top) == h && out.up.first[b] != kIbNone && cls_of(out.up, b) == n) {
            if (firstw == kIbNone) firstw = (uint16_t)(b & lowmask);
            // the class's part of the list must be in rank order: the position of a partner block is looked up in
            // urank_low
            if (out.urank_low[b & lowmask] != count) return "a half does not hold every low word of an occupation";
            hf.ublist.push_back((uint16_t)(b & lowmask));
            count++;
          }
        while (hf.ublist.size() % 64) hf.ublist.push_back((uint16_t)((firstw == kIbNone ? 0 : firstw) | kIbSkip));
        const int cs = (int)hf.ublist.size() - hf.ucls[n];
        hf.rcb[n + 1] = at;
        hf.rcs[n + 1] = cs;
        maxcs = std::max(maxcs, cs);
        at += (int)binomial(norb, n) * cs;
      }
      for (int n = norb + 1; n < kIbMaxNorb + 2; n++) hf.ucls[n] = (int)hf.ublist.size();
      hf.rimg_len = at + 2 * maxcs + 8;
      hf.rmap.assign((size_t)hf.npanels * kIbPanel, (uint16_t)(hf.rimg_len - 1));
      hf.utop.assign(hf.ublist.size(), kIbNone);
      for (int n = 0; n <= norb; n++)
        for (int q = hf.ucls[n]; q < hf.ucls[n + 1]; q++) {
          const uint32_t b = (uint32_t)(hf.ublist[q] & 0x7FFFu) | ((uint32_t)h << top);
          if (out.up.first[b] == kIbNone) continue;  // (an empty class's padding)
          hf.utop[q] = out.upos[b ^ (1u << top)];     // kIbNone when that block does not exist
          if (hf.ublist[q] & kIbSkip) continue;
          const int m = rows_of(out.up, b);
          for (int j = 0; j < m; j++)
            hf.rmap[(size_t)out.upos[b] - (size_t)hf.panel0 * kIbPanel + j] = (uint16_t)(hf.rcb[n + 1] + j * hf.rcs[n + 1] + (q - hf.ucls[n]));
        }
    }
    out.nhalf = 2;
    return "";
  };
  e = build_up(false);
  if (!e.empty()) return fail(e);
  // LDS of the rows kernel: the image, the amplitudes, the rank table of the bath words it looks partners up in
  auto rows_lds = [](int img_words, int nbits) { return (int64_t)img_words * 8 + (int64_t)(nbits + 2) * 32 + ((int64_t)2 << nbits); };
  if (lds_budget < 0 || (lds_budget > 0 && rows_lds(out.rimg_len, nb) > lds_budget)) {  // (< 0: always, tests)
    if (nb < 2) return fail("row image longer than the LDS");
    if (!out.up.pmask.empty()) return fail("bath-bath hops in a row staged in halves");
    e = build_up(true);
    if (!e.empty()) return fail(e);
    if (lds_budget > 0 && rows_lds(std::max(out.half[0].rimg_len, out.half[1].rimg_len), nb - 1) > lds_budget)
      return fail("half a row image is longer than the LDS");
  }

  // ---- diagonal ----
  out.xu.assign((size_t)(impmask + 1) * (impmask + 1), 0.0);
  for (uint32_t pd = 0; pd <= impmask; pd++)
    for (uint32_t pu = 0; pu <= impmask; pu++)
      out.xu[(size_t)pd * (impmask + 1) + pu] = out.up.eimp[pu] + hn.xt[((size_t)pu << norb) | pd];
  out.ed = hn.fac.ed;
  out.impd = hn.fac.impd;
  {  // the tables must reproduce the factored diagonal of the generic kernels
    double scale = 1.0, worst = 0.0;
    for (uint32_t c = 0; c <= impmask; c++)
      for (int64_t i = 0; i < du; i++) {
        const uint32_t st = (uint32_t)hn.bup.states[i];
        const double ref = hn.fac.eux[(size_t)c * du + i];
        const double got = out.up.ebath[st >> norb] + out.xu[(size_t)c * (impmask + 1) + (st & impmask)];
        scale = std::max(scale, std::fabs(ref));
        worst = std::max(worst, std::fabs(ref - got));
      }
    if (!(worst <= 1e-13 * scale)) return fail("diagonal tables disagree");
  }

  // ---- down side: chunks ----
  // rows that share the bath levels >= low form a contiguous run ("superblock"); the largest low whose superblocks
  // fit a chunk
  int low = -1;
  std::vector<int> sb_rows;
  for (int l = nb; l >= 0; l--) {
    std::vector<int> rows((size_t)1 << (nb - l), 0);
    for (uint32_t b = 0; b < nw; b++)
      if (out.dw.first[b] != kIbNone) rows[b >> l] += rows_of(out.dw, b);
    if (*std::max_element(rows.begin(), rows.end()) <= max_chunk_rows) {
      low = l;
      sb_rows = rows;
      break;
    }
  }
  if (low < 0) return fail("no chunk fits");
  if (nb - low > 2 * 3) return fail("more than 6 bath levels outside a chunk");
  out.lowbits = low;
  {
    // consecutive superblocks are merged while they fit
    int rows = 0, row0 = 0;
    std::vector<uint32_t> hs;
    auto flush = [&]() {
      if (hs.empty()) return;
      const int c = (int)out.chunk_row.size();
      out.chunk_row.push_back(row0);
      out.chunk_blk.push_back((int)out.dblist.size());
      out.dcls.resize((size_t)(c + 1) * (kIbMaxNorb + 2), 0);
      const int base = (int)out.dblist.size();
      for (int n = 0; n <= norb; n++) {
        out.dcls[(size_t)c * (kIbMaxNorb + 2) + n] = (int)out.dblist.size() - base;
        uint16_t firstb = kIbNone;
        for (uint32_t h : hs)
          for (uint32_t lo = 0; lo < (1u << low); lo++) {
            const uint32_t b = (h << low) | lo;
            if (out.dw.first[b] == kIbNone || cls_of(out.dw, b) != n) continue;
            if (firstb == kIbNone) firstb = (uint16_t)b;
            out.dblist.push_back((uint16_t)b);
          }
        while ((out.dblist.size() - base) % 8) out.dblist.push_back((uint16_t)(firstb | kIbSkip));
      }
      for (int n = norb + 1; n < kIbMaxNorb + 2; n++)
        out.dcls[(size_t)c * (kIbMaxNorb + 2) + n] = (int)out.dblist.size() - base;
      out.max_chunk_rows = std::max(out.max_chunk_rows, rows);
      row0 += rows;
      rows = 0;
      hs.clear();
    };
    for (uint32_t h = 0; h < (1u << (nb - low)); h++) {
      if (sb_rows[h] == 0) continue;
      if (rows + sb_rows[h] > max_chunk_rows) flush();
      hs.push_back(h);
      rows += sb_rows[h];
    }
    flush();
    out.chunk_row.push_back((int32_t)dd);
    out.chunk_blk.push_back((int32_t)out.dblist.size());
    if (row0 != dd) return fail("chunk plan does not cover the rows");
  }
  out.dmeta.assign(nw * 16, 0);
  for (uint32_t b = 0; b < nw; b++) {
    if (out.dw.first[b] == kIbNone) continue;
    uint16_t sg = 0;
    for (int k = 0; k < nb; k++) {
      out.dmeta[(size_t)b * 16 + k] = out.dw.first[b ^ (1u << k)];
      if (popc(b & ((1u << k) - 1u)) & 1) sg |= (uint16_t)(1u << k);
    }
    out.dmeta[(size_t)b * 16 + 14] = out.dw.first[b];
    out.dmeta[(size_t)b * 16 + 15] = sg;
  }

  // ---- factored Hnd, block-relative ----
  out.nterms = hn.fac.nterms;
  out.ndcoef = hn.fac.coef;
  out.nd_dw.assign((size_t)std::max(1, out.nterms) * (norb + 1) * 4, 0xFF);
  out.nd_up.assign((size_t)std::max(1, out.nterms) * out.npanels * kIbPanel, 0xFF);
  for (int t = 0; t < out.nterms; t++) {
    std::vector<int> seen((size_t)(norb + 1) * 4, -1);
    for (uint32_t b = 0; b < nw; b++) {
      if (out.dw.first[b] == kIbNone) continue;
      const int n = cls_of(out.dw, b), m = rows_of(out.dw, b);
      for (int j = 0; j < m; j++) {
        const uint32_t jd = hn.fac.jdw[(size_t)t * dd + out.dw.first[b] + j];
        int ent = 0xFF;
        if (jd != 0xFFFFFFFFu) {
          const int64_t rel = (int64_t)(jd & 0x7FFFFFFFu) - out.dw.first[b];
          if (rel < 0 || rel >= m) return fail("an Hnd term leaves its block of rows");
          ent = (int)rel | ((jd >> 31) ? 0x80 : 0);
        }
        int& sv = seen[(size_t)n * 4 + j];
        if (sv >= 0 && sv != ent) return fail("an Hnd term is not a function of the impurity pattern");
        sv = ent;
        out.nd_dw[((size_t)t * (norb + 1) + n) * 4 + j] = (uint8_t)ent;
      }
    }
    for (int64_t i = 0; i < du; i++) {
      const uint32_t ju = hn.fac.jup[(size_t)t * du + i];
      if (ju == 0xFFFFFFFFu) continue;
      const int64_t i2 = (int64_t)(ju & 0x7FFFFFFFu);
      const int p1 = out.pos[(size_t)i], p2 = out.pos[(size_t)i2];
      if (p1 / kIbPanel != p2 / kIbPanel || p2 - p1 < -8 || p2 - p1 > 7) return fail("an Hnd term leaves its panel");
      out.nd_up[(size_t)t * out.npanels * kIbPanel + p1] = (uint8_t)((p2 - p1 + 8) | ((ju >> 31) ? 0x80 : 0));
    }
  }
  out.valid = true;
}

}  // namespace edigpu
