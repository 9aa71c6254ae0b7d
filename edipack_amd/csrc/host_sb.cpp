// host_sb.cpp -- see host_sb.hpp.  Product code (host side of kernels_sb.hip); shares nothing with oracle/.
#include "host_sb.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace edigpu {

namespace {

inline int popc(uint32_t x) { return __builtin_popcount(x); }

// rank of the local pattern p among the patterns with the same number of bits
inline int pat_rank(uint32_t p) {
  int r = 0;
  for (uint32_t q = 0; q < p; q++) r += popc(q) == popc(p);
  return r;
}

std::string build_side(const HostNormal& hn, int sp, const CombBasis& bs, int npart, int nloc, SbSide& s) {
  const int ns = hn.ns, norb = hn.norb, nbw = ns - nloc;
  s.ns = ns;
  s.npart = npart;
  s.norb = norb;
  s.nloc = nloc;
  s.nbw = nbw;
  s.dim = bs.size();
  const std::vector<double>& a = hn.ob_a[sp];
  const std::vector<double>& eps = hn.ob_eps[sp];
  if ((int)a.size() != ns * ns || (int)eps.size() != ns) return "no one-body data";
  for (int p = norb; p < ns; p++)
    for (int q = norb; q < ns; q++)
      if (p != q && a[p * ns + q] != 0.0) return "bath-bath hops";
  s.vtab.assign((size_t)nbw * 4, 0.0);
  s.korb.assign((size_t)norb, 0u);
  s.single = true;
  for (int k = 0; k < nbw; k++) {
    int used = 0;
    for (int ia = 0; ia < norb; ia++) {
      const double v = a[(nloc + k) * ns + ia];
      s.vtab[(size_t)k * 4 + ia] = v;
      if (v != 0.0) {
        s.korb[ia] |= 1u << k;
        used++;
      }
    }
    if (used > 1) s.single = false;
    s.vtab[(size_t)k * 4 + 3] = eps[nloc + k];
  }
  s.tloc.assign((size_t)nloc * nloc, 0.0);
  for (int p = 0; p < nloc; p++)
    for (int q = 0; q < nloc; q++)
      if (p != q) s.tloc[(size_t)p * nloc + q] = a[p * ns + q];
  const size_t nw = (size_t)1 << nbw;
  const uint32_t lmask = (1u << nloc) - 1u;
  s.first.assign(nw, kIbNone);
  for (int64_t i = 0; i < s.dim; i++) {
    const uint32_t st = (uint32_t)bs.states[i], w = st >> nloc;
    if (s.first[w] == kIbNone) s.first[w] = (uint16_t)i;
    if ((int64_t)s.first[w] + pat_rank(st & lmask) != i) return "basis not in block order";
  }
  return "";
}

struct Slot {
  int cls, i0;
  long cost;
};

// longest-processing-time assignment of slots to nw waves with at most cap slots each; order[v] = the slots of wave v.
// false: they do not fit.
bool deal(std::vector<Slot> slots, int nw, int cap, std::vector<std::vector<Slot>>& order) {
  order.assign((size_t)nw, {});
  std::stable_sort(slots.begin(), slots.end(), [](const Slot& x, const Slot& y) { return x.cost > y.cost; });
  std::vector<long> load((size_t)nw, 0);
  for (const Slot& sl : slots) {
    int best = -1;
    for (int v = 0; v < nw; v++)
      if ((cap <= 0 || (int)order[v].size() < cap) && (best < 0 || load[v] < load[best])) best = v;
    if (best < 0) return false;
    order[best].push_back(sl);
    load[best] += sl.cost;
  }
  return true;
}

// LDS cycles of one block update (both kernels are bound by them: ~13 per walk step for the look-up and the amplitudes, ~7
// per partner word or row): a class-n block of a species with npart electrons walks nbw - (npart - n) empty levels (partner
// class n - 1) and npart - n occupied ones (partner class n + 1)
long slot_cost(int nloc, int nbw, int npart, int n) {
  const long m = (long)binomial(nloc, n);
  const long down = nbw - (npart - n), up = npart - n;
  long c = 20 * m;
  if (n >= 1) c += down * (13 + 7 * (long)binomial(nloc, n - 1));
  if (n < nloc) c += up * (13 + 7 * (long)binomial(nloc, n + 1));
  return c;
}

}  // namespace

int sb_rows_slots(const HostNormal& hn, int nb0, bool split) {
  const int nloc = hn.norb + nb0, nbw = hn.ns - nloc;
  if (nb0 < 1 || nloc > kSbMaxLoc || nbw < 1 || nbw > 14) return -1;
  if (split && nbw < 2) return -1;
  int most = 0;
  for (int h = 0; h < (split ? 2 : 1); h++) {
    std::vector<int> count((size_t)nloc + 1, 0);
    for (uint32_t w = 0; w < (1u << nbw); w++) {
      if (split && (int)(w >> (nbw - 1)) != h) continue;
      const int n = hn.nup - popc(w);
      if (n >= 0 && n <= nloc) count[n]++;
    }
    int slots = 0;
    for (int n = 0; n <= nloc; n++) slots += (count[n] + 63) / 64;
    most = std::max(most, slots);
  }
  return most;
}

void build_sb(const HostNormal& hn, const HostIb& ib, int nb0, int max_chunk_rows, int rows_nt, int rows_nbt, int cols_nw,
              HostSb& out, int cols_gs) {
  out = HostSb();
  const int gs = cols_gs;
  auto fail = [&](const std::string& w) {
    out.valid = false;
    out.why = w;
  };
  if (!ib.valid) return fail("no impurity-block image to derive the tables from");
  const bool split = ib.nhalf == 2;
  const int norb = hn.norb, ns = hn.ns, nloc = norb + nb0, nbw = ns - nloc;
  if (nb0 < 1 || nloc > kSbMaxLoc) return fail("local levels per block not in norb+1 .. 6");
  if (nbw < 1 || nbw > 14) return fail("walked bath levels per species not in 1..14");
  if (rows_nt % 64 || rows_nt < 64 || rows_nbt < 1 || cols_nw < 1 || (gs != 4 && gs != 8)) return fail("kernel geometry");
  out.cols_gs = gs;
  out.norb = norb;
  out.nb0 = nb0;
  out.nloc = nloc;
  std::string e = build_side(hn, 0, hn.bup, hn.nup, nloc, out.up);
  if (e.empty()) e = build_side(hn, 1, hn.bdw, hn.ndw, nloc, out.dw);
  if (!e.empty()) return fail(e);
  // The per-orbital walk (AMODE 1) issues only the multiply-adds of the orbital a level belongs to, but the lanes of a wave
  // then walk unequal numbers of levels per orbital and wait for the longest: measured on config 2 (two orbitals, six levels
  // each) 138 us per product against 126 with the all-orbital walk, whose extra multiply-adds (by amplitudes that are zero)
  // are cheaper than the extra LDS round trips.  On request only: EDIGPU_SB_AMODE=1.
  out.amode = 0;
  if (const char* e = getenv("EDIGPU_SB_AMODE")) out.amode = (atoi(e) != 0 && norb > 1 && out.up.single && out.dw.single) ? 1 : 0;
  const int64_t du = hn.dim_up, dd = hn.dim_dw;
  const uint32_t lmask = (1u << nloc) - 1u, impmask = (1u << norb) - 1u;
  const size_t nw = (size_t)1 << nbw;
  auto cls_of = [&](const SbSide& s, uint32_t w) { return s.npart - popc(w); };
  auto rows_of = [&](const SbSide& s, uint32_t w) { return (int)binomial(nloc, cls_of(s, w)); };

  // ---- up side: row image, wave-slots ----
  if (!split) {
    std::vector<std::vector<uint16_t>> bycls((size_t)nloc + 1);
    for (uint32_t w = 0; w < nw; w++)
      if (out.up.first[w] != kIbNone) bycls[(size_t)cls_of(out.up, w)].push_back((uint16_t)w);
    out.urank.assign(nw, kIbNone);
    int maxcnt = 1;
    for (int n = 0; n <= nloc; n++) maxcnt = std::max(maxcnt, (int)bycls[n].size());
    int cs = 64;
    while (cs < maxcnt) cs *= 2;
    cs += 1;  // odd: the words of one block, cs apart, fall into different LDS banks (the row moves in and out by position)
    out.rcs = cs;
    std::vector<int> wb((size_t)nloc + 2, 0);
    for (int n = 0; n <= nloc; n++) wb[n + 1] = wb[n] + (int)binomial(nloc, n);
    std::vector<Slot> slots;
    for (int n = 0; n <= nloc; n++) {
      const int cnt = (int)bycls[n].size();
      for (int q = 0; q < cnt; q++) out.urank[bycls[n][q]] = (uint16_t)q;
      for (int i0 = 0; i0 < cnt; i0 += 64) slots.push_back(Slot{n, i0, slot_cost(nloc, nbw, out.up.npart, n)});
    }
    out.rimg_len = (wb[nloc + 1] * cs + 8 + 1) & ~1;  // (even: the tables behind the image stay 16-byte aligned)
    if (out.rimg_len >= 0xFFF0) return fail("row image longer than 65519 words");
    const int plen = ib.npanels * kIbPanel;
    out.rmap.assign((size_t)plen, (uint16_t)(out.rimg_len - 1));
    for (int64_t i = 0; i < du; i++) {
      const uint32_t st = (uint32_t)hn.bup.states[i], w = st >> nloc;
      const int n = cls_of(out.up, w);
      out.rmap[(size_t)ib.pos[(size_t)i]] = (uint16_t)((wb[n] + pat_rank(st & lmask)) * cs + out.urank[w]);
    }
    const int nwv = rows_nt / 64;
    std::vector<std::vector<Slot>> order;
    if (!deal(slots, nwv, rows_nbt, order)) return fail("more wave-slots than the rows kernel holds");
    out.rows_nt = rows_nt;
    out.rows_nbt = rows_nbt;
    out.uslot.assign((size_t)rows_nbt * nwv, -1);
    out.ublist.assign((size_t)rows_nbt * nwv * 64, 0);
    for (int v = 0; v < nwv; v++)
      for (int s = 0; s < (int)order[v].size(); s++) {
        const Slot& sl = order[v][s];
        out.uslot[(size_t)s * nwv + v] = sl.cls | (sl.i0 << 8);
        const std::vector<uint16_t>& lst = bycls[(size_t)sl.cls];
        for (int l = 0; l < 64; l++) {
          const int i = sl.i0 + l;
          out.ublist[((size_t)s * nwv + v) * 64 + l] = i < (int)lst.size() ? lst[(size_t)i] : (uint16_t)(lst[0] | kIbSkip);
        }
      }
  }
  if (split) {
    // rows staged in halves (SbUpHalf): one image per value of the top walked bit
    if (nbw < 2) return fail("one walked level: nothing to split");
    const int top = nbw - 1;
    const uint32_t lowm = (1u << top) - 1u;
    out.nhalf = 2;
    out.urank.assign((size_t)1 << top, 0);
    {
      std::vector<int> cnt((size_t)nbw + 1, 0);
      for (uint32_t wl = 0; wl <= lowm; wl++) out.urank[wl] = (uint16_t)cnt[(size_t)popc(wl)]++;
    }
    std::vector<std::vector<uint16_t>> bycls[2];
    int maxcnt = 1;
    for (int h = 0; h < 2; h++) {
      bycls[h].assign((size_t)nloc + 1, {});
      for (uint32_t w = 0; w < nw; w++)
        if ((int)(w >> top) == h && out.up.first[w] != kIbNone) {
          std::vector<uint16_t>& lst = bycls[h][(size_t)cls_of(out.up, w)];
          // a class's list must be in rank order: a partner block is found through urank
          if (out.urank[w & lowm] != (int)lst.size()) return fail("a half does not hold every low word of an occupation");
          lst.push_back((uint16_t)(w & lowm));
        }
      for (int n = 0; n <= nloc; n++) maxcnt = std::max(maxcnt, (int)bycls[h][n].size());
    }
    int cs = 64;
    while (cs < maxcnt) cs *= 2;
    cs += 1;
    out.rcs = cs;
    std::vector<int> wb((size_t)nloc + 2, 0);
    for (int n = 0; n <= nloc; n++) wb[n + 1] = wb[n] + (int)binomial(nloc, n);
    out.rimg_len = (wb[nloc + 1] * cs + 8 + 1) & ~1;
    if (out.rimg_len >= 0xFFF0) return fail("row image longer than 65519 words");
    const int nwv = rows_nt / 64;
    out.rows_nt = rows_nt;
    out.rows_nbt = rows_nbt;
    const double eps_top = out.up.vtab[(size_t)top * 4 + 3];
    for (int h = 0; h < 2; h++) {
      SbUpHalf& hf = out.half[h];
      hf = SbUpHalf();
      hf.panel0 = ib.half[h].panel0;
      hf.npanels = ib.half[h].npanels;
      hf.rmap.assign((size_t)hf.npanels * kIbPanel, (uint16_t)(out.rimg_len - 1));
      for (int64_t i = 0; i < du; i++) {
        const uint32_t st = (uint32_t)hn.bup.states[i], w = st >> nloc;
        if ((int)(w >> top) != h) continue;
        const int64_t rel = (int64_t)ib.pos[(size_t)i] - (int64_t)hf.panel0 * kIbPanel;
        if (rel < 0 || rel >= (int64_t)hf.rmap.size()) return fail("a column of a half lies outside its panels");
        hf.rmap[(size_t)rel] = (uint16_t)((wb[cls_of(out.up, w)] + pat_rank(st & lmask)) * cs + out.urank[w & lowm]);
      }
      std::vector<Slot> slots;
      for (int n = 0; n <= nloc; n++)
        for (int i0 = 0; i0 < (int)bycls[h][n].size(); i0 += 64) slots.push_back(Slot{n, i0, slot_cost(nloc, top, out.up.npart - h, n)});
      std::vector<std::vector<Slot>> order;
      if (!deal(slots, nwv, rows_nbt, order)) return fail("more wave-slots than the rows kernel holds");
      hf.uslot.assign((size_t)rows_nbt * nwv, -1);
      hf.ublist.assign((size_t)rows_nbt * nwv * 64, 0);
      hf.utop.assign(hf.ublist.size(), kIbNone);
      hf.ugap.assign(hf.ublist.size(), 0x0F);
      for (int v = 0; v < nwv; v++)
        for (int sidx = 0; sidx < (int)order[v].size(); sidx++) {
          const Slot& sl = order[v][sidx];
          hf.uslot[(size_t)sidx * nwv + v] = sl.cls | (sl.i0 << 8);
          const std::vector<uint16_t>& lst = bycls[h][(size_t)sl.cls];
          for (int l = 0; l < 64; l++) {
            const int i = sl.i0 + l;
            const size_t at = ((size_t)sidx * nwv + v) * 64 + l;
            const uint16_t wl = i < (int)lst.size() ? lst[(size_t)i] : lst[0];
            hf.ublist[at] = i < (int)lst.size() ? wl : (uint16_t)(wl | kIbSkip);
            const uint32_t wp = (uint32_t)wl | ((uint32_t)(1 - h) << top);  // the block with the top bit toggled
            if (out.up.first[wp] == kIbNone) continue;
            const int64_t f = out.up.first[wp];
            const int m = rows_of(out.up, wp);
            const int p0 = ib.pos[(size_t)f];
            hf.utop[at] = (uint16_t)p0;
            int jg = -1, g = 0;
            for (int j = 1; j < m; j++) {
              const int d = ib.pos[(size_t)(f + j)] - (p0 + j);
              if (jg < 0 && d != 0) {
                jg = j;
                g = d;
              }
              if (d != (jg >= 0 && j >= jg ? g : 0)) return fail("a block's columns have more than one gap");
            }
            if (jg >= 0) {
              if (jg > 14 || g < 1 || g > 15) return fail("gap of a block's columns out of range");
              hf.ugap[at] = (uint8_t)(jg | (g << 4));
            }
          }
        }
      hf.ebw.assign((size_t)1 << top, 0.0);
      for (uint32_t wl = 0; wl <= lowm; wl++) {
        double e2 = h ? eps_top : 0.0;
        for (int k = 0; k < top; k++)
          if ((wl >> k) & 1u) e2 += out.up.vtab[(size_t)k * 4 + 3];
        hf.ebw[wl] = e2;
      }
    }
  }
  out.ebw.assign(nw, 0.0);
  for (uint32_t w = 0; w < nw; w++)
    for (int k = 0; k < nbw; k++)
      if ((w >> k) & 1u) out.ebw[w] += out.up.vtab[(size_t)k * 4 + 3];
  out.e0.assign((size_t)1 << nb0, 0.0);
  for (uint32_t lb = 0; lb < (1u << nb0); lb++)
    for (int k = 0; k < nb0; k++)
      if ((lb >> k) & 1u) out.e0[lb] += hn.ob_eps[0][(size_t)norb + k];
  {  // the tables must reproduce the factored diagonal of the generic kernels
    double scale = 1.0, worst = 0.0;
    for (uint32_t c = 0; c <= impmask; c++)
      for (int64_t i = 0; i < du; i++) {
        const uint32_t st = (uint32_t)hn.bup.states[i], w = st >> nloc, p = st & lmask;
        const double eb = out.ebw[w];
        const double ref = hn.fac.eux[(size_t)c * du + i];
        const double got = eb + out.e0[p >> norb] + ib.xu[(size_t)c * (impmask + 1) + (p & impmask)];
        scale = std::max(scale, std::fabs(ref));
        worst = std::max(worst, std::fabs(ref - got));
      }
    if (!(worst <= 1e-13 * scale)) return fail("diagonal tables disagree");
  }

  // ---- down side: chunks ----
  int low = -1;
  std::vector<int> sb_rows;
  for (int l = nbw; l >= 0; l--) {
    std::vector<int> rows((size_t)1 << (nbw - l), 0);
    for (uint32_t w = 0; w < nw; w++)
      if (out.dw.first[w] != kIbNone) rows[w >> l] += rows_of(out.dw, w);
    if (*std::max_element(rows.begin(), rows.end()) <= max_chunk_rows) {
      low = l;
      sb_rows = rows;
      break;
    }
  }
  if (low < 0) return fail("no chunk fits");
  if (nbw - low > 6) return fail("more than 6 bath levels outside a chunk");
  out.lowbits = low;
  out.cols_nw = cols_nw;
  {
    int rows = 0, row0 = 0;
    std::vector<uint32_t> hs;
    auto flush = [&]() {
      if (hs.empty()) return;
      out.chunk_row.push_back(row0);
      out.chunk_slot.push_back((int32_t)out.dslot.size());
      // groups of blocks with one class and one high word; Slot::cls indexes the group here
      std::vector<std::vector<uint16_t>> bycls;
      std::vector<int> gcls;
      for (uint32_t h : hs)
        for (int n = 0; n <= nloc; n++) {
          std::vector<uint16_t> g;
          for (uint32_t lo = 0; lo < (1u << low); lo++) {
            const uint32_t w = (h << low) | lo;
            if (out.dw.first[w] != kIbNone && cls_of(out.dw, w) == n) g.push_back((uint16_t)w);
          }
          if (!g.empty()) {
            bycls.push_back(g);
            gcls.push_back(n);
          }
        }
      std::vector<Slot> slots;
      for (int gi = 0; gi < (int)bycls.size(); gi++)
        for (int i0 = 0; i0 < (int)bycls[gi].size(); i0 += gs) slots.push_back(Slot{gi, i0, slot_cost(nloc, nbw, out.dw.npart, gcls[gi])});
      std::vector<std::vector<Slot>> order;
      deal(slots, cols_nw, 0, order);
      size_t rounds = 0;
      for (const auto& o : order) rounds = std::max(rounds, o.size());
      const size_t base = out.dslot.size();
      out.dslot.resize(base + rounds * cols_nw, -1);
      out.dblist.resize((base + rounds * cols_nw) * gs, 0);
      for (int v = 0; v < cols_nw; v++)
        for (size_t r = 0; r < order[v].size(); r++) {
          const Slot& sl = order[v][r];
          const size_t q = base + r * cols_nw + v;
          out.dslot[q] = gcls[(size_t)sl.cls];
          const std::vector<uint16_t>& lst = bycls[(size_t)sl.cls];
          for (int g = 0; g < gs; g++) {
            const int i = sl.i0 + g;
            out.dblist[q * gs + g] = i < (int)lst.size() ? lst[(size_t)i] : (uint16_t)(lst[(size_t)sl.i0] | kIbSkip);
          }
        }
      out.max_chunk_slots = std::max(out.max_chunk_slots, (int)(rounds * cols_nw));
      out.max_chunk_rows = std::max(out.max_chunk_rows, rows);
      row0 += rows;
      rows = 0;
      hs.clear();
    };
    for (uint32_t h = 0; h < (1u << (nbw - low)); h++) {
      if (sb_rows[h] == 0) continue;
      if (rows + sb_rows[h] > max_chunk_rows) flush();
      hs.push_back(h);
      rows += sb_rows[h];
    }
    flush();
    out.chunk_row.push_back((int32_t)dd);
    out.chunk_slot.push_back((int32_t)out.dslot.size());
    if (row0 != dd) return fail("chunk plan does not cover the rows");
  }
  out.dmeta.assign(nw * 16, 0);
  for (uint32_t w = 0; w < nw; w++) {
    if (out.dw.first[w] == kIbNone) continue;
    uint16_t sg = 0;
    for (int k = 0; k < nbw; k++) {
      out.dmeta[(size_t)w * 16 + k] = out.dw.first[w ^ (1u << k)];
      if (popc(w & ((1u << k) - 1u)) & 1) sg |= (uint16_t)(1u << k);
    }
    out.dmeta[(size_t)w * 16 + 14] = out.dw.first[w];
    out.dmeta[(size_t)w * 16 + 15] = sg;
  }

  {
    const int nch = (int)out.chunk_row.size() - 1;
    out.cdesc_off.assign((size_t)nch + 1, 0);
    for (int c = 0; c < nch; c++) {
      const int s0 = out.chunk_slot[c], nsl = out.chunk_slot[c + 1] - s0;
      const size_t base = out.cdesc.size();
      out.cdesc_off[c] = (int32_t)base;
      const size_t moff = (size_t)nsl * gs * 32, loff = moff + ((((size_t)nsl * gs * 2) + 15) & ~(size_t)15);
      out.cdesc.resize(base + sb_desc_bytes(nsl, gs), 0);
      uint16_t* meta = reinterpret_cast<uint16_t*>(out.cdesc.data() + base);
      uint16_t* lbl = reinterpret_cast<uint16_t*>(out.cdesc.data() + base + moff);
      int32_t* cls = reinterpret_cast<int32_t*>(out.cdesc.data() + base + loff);
      for (int q = 0; q < nsl; q++) {
        cls[q] = out.dslot[(size_t)s0 + q];
        for (int g = 0; g < gs; g++) {
          const uint16_t e = out.dblist[((size_t)s0 + q) * gs + g];
          lbl[q * gs + g] = e;
          for (int k = 0; k < 16; k++) meta[(size_t)(q * gs + g) * 16 + k] = out.dmeta[(size_t)(e & 0x7FFFu) * 16 + k];
        }
      }
    }
    out.cdesc_off[nch] = (int32_t)out.cdesc.size();
  }

  // ---- factored Hnd: the row table, block-relative (the column table is HostIb::nd_up) ----
  out.nd_dw.assign((size_t)std::max(1, ib.nterms) * (nloc + 1) * kSbNdStride, 0xFFu);
  for (int t = 0; t < ib.nterms; t++) {
    std::vector<int> seen((size_t)(nloc + 1) * kSbNdStride, -1);
    for (uint32_t w = 0; w < nw; w++) {
      if (out.dw.first[w] == kIbNone) continue;
      const int n = cls_of(out.dw, w), m = rows_of(out.dw, w);
      for (int j = 0; j < m; j++) {
        const uint32_t jd = hn.fac.jdw[(size_t)t * dd + out.dw.first[w] + j];
        int ent = 0xFF;
        if (jd != 0xFFFFFFFFu) {
          const int64_t rel = (int64_t)(jd & 0x7FFFFFFFu) - out.dw.first[w];
          if (rel < 0 || rel >= m) return fail("an Hnd term leaves its block of rows");
          ent = (int)rel | ((jd >> 31) ? 0x80 : 0);
        }
        int& sv = seen[(size_t)n * kSbNdStride + j];
        if (sv >= 0 && sv != ent) return fail("an Hnd term is not a function of the local pattern");
        sv = ent;
        out.nd_dw[((size_t)t * (nloc + 1) + n) * kSbNdStride + j] = (uint32_t)ent;
      }
    }
  }
  out.valid = true;
}

}  // namespace edigpu
