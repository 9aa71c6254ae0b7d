// host_build.cpp -- see host_build.hpp.  Product code; shares nothing with oracle/.
#include "host_build.hpp"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace edigpu {

using cplx = std::complex<double>;

int model_ns(const edigpu_model& m) {
  // Ns as in ed_setup_dimensions (ED_SETUP.f90:118-126)
  return m.bath_type == 1 ? m.nbath + m.norb : (m.nbath + 1) * m.norb;
}

int64_t binomial(int n, int k) {
  if (k < 0 || k > n) return 0;
  k = std::min(k, n - k);
  __int128 r = 1;
  for (int i = 1; i <= k; i++) r = r * (n - k + i) / i;
  return (int64_t)r;
}

static inline int popc(uint32_t x) { return __builtin_popcount(x); }

void CombBasis::init(int nbits_, int npart_) {
  nbits = nbits_;
  npart = npart_;
  hbits = nbits / 2;
  const int64_t n = binomial(nbits, npart);
  states.resize(n);
  off_hi.assign((size_t)1 << (nbits - hbits), 0);
  rank_lo.assign((size_t)1 << hbits, 0);
  if (n == 0) return;
  // ascending enumeration of fixed-popcount words (next-combination bit trick)
  uint32_t m = npart == 0 ? 0u : ((1u << npart) - 1u);
  const uint32_t lomask = (1u << hbits) - 1u;
  uint32_t last_hi = 0xffffffffu;
  for (int64_t k = 0; k < n; k++) {
    states[k] = (int32_t)m;
    uint32_t hi = m >> hbits, lo = m & lomask;
    if (hi != last_hi) {
      off_hi[hi] = (int32_t)k;
      last_hi = hi;
    }
    rank_lo[lo] = (int32_t)(k - off_hi[hi]);
    if (npart == 0) break;
    uint32_t c = m & (0u - m), r = m + c;
    m = (((r ^ m) >> 2) / c) | r;
  }
}

namespace {

struct Idx {
  const edigpu_model& m;
  explicit Idx(const edigpu_model& mm) : m(mm) {}
  double uloc(int a) const { return m.uloc[a]; }
  // the reference always reads the (a<b) entry (stored/H_local.f90:44-58)
  double ust(int a, int b) const { return m.ust[a * EDIGPU_MAXORB + b]; }
  double jh(int a, int b) const { return m.jh[a * EDIGPU_MAXORB + b]; }
  double jx(int a, int b) const { return m.jx[a * EDIGPU_MAXORB + b]; }
  double jp(int a, int b) const { return m.jp[a * EDIGPU_MAXORB + b]; }
  cplx hloc(int is, int js, int a, int b) const {
    const double* p = &m.hloc[((((is * 2) + js) * EDIGPU_MAXORB + a) * EDIGPU_MAXORB + b) * 2];
    return cplx(p[0], p[1]);
  }
  double bath(const double* arr, int is, int a, int k) const {
    return arr[(is * EDIGPU_MAXORB + a) * EDIGPU_MAXBATH + k];
  }
  // replica/general: hbath_tmp(is,js,a,b,k)
  cplx hb(int is, int js, int a, int b, int k) const {
    const double* p =
        &m.hb[(((((is * 2) + js) * EDIGPU_MAXORB + a) * EDIGPU_MAXORB + b) * EDIGPU_MAXBATH + k) * 2];
    return cplx(p[0], p[1]);
  }
  bool replica() const { return m.bath_type == 2 || m.bath_type == 3; }
  // 0-based level of bath site k of orbital a (getBathStride, ED_SETUP.f90:605-622)
  int bath_pos(int a, int k) const {
    switch (m.bath_type) {
      case 1: return m.norb + k;
      case 2:
      case 3: return a + (k + 1) * m.norb;
      default: return m.norb + a * m.nbath + k;
    }
  }
  int n_ebath_orb() const { return m.bath_type == 1 ? 1 : m.norb; }
};

std::string check_model(const edigpu_model& m) {
  if (m.norb < 1 || m.norb > EDIGPU_MAXORB) return "edigpu: norb out of range";
  if (m.nbath < 0 || m.nbath > EDIGPU_MAXBATH) return "edigpu: nbath out of range";
  if (m.nspin < 1 || m.nspin > 2) return "edigpu: nspin must be 1 or 2";
  if (m.bath_type < 0 || m.bath_type > 3) return "edigpu: unknown bath_type";
  int ns = model_ns(m);
  if (ns > 30) return "edigpu: Ns > 30 levels per spin not supported";
  return "";
}

// one-body data of one spin species: off-diagonal A(p,q) = coeff of c^+_p c_q, diagonal eps(p)
struct OneBody {
  int ns;
  std::vector<double> a;    // ns*ns
  std::vector<double> eps;  // ns
};

// spin: 0 = up, 1 = down.  The down species reads the (Nspin,Nspin) blocks.
OneBody one_body_normal(const edigpu_model& m, int spin) {
  Idx ix(m);
  const int ns = model_ns(m), norb = m.norb;
  const int s = spin == 0 ? 0 : m.nspin - 1;
  OneBody ob;
  ob.ns = ns;
  ob.a.assign((size_t)ns * ns, 0.0);
  ob.eps.assign(ns, 0.0);
  for (int a = 0; a < norb; a++) {
    for (int b = 0; b < norb; b++) {
      if (a == b) continue;
      ob.a[a * ns + b] += ix.hloc(s, s, a, b).real();
      // exciton field: (exc(1) +/- exc(4)) c+_a c_b, stored/H_up.f90:87-104, H_dw.f90
      ob.a[a * ns + b] += m.exc_field[0] + (spin == 0 ? 1.0 : -1.0) * m.exc_field[3];
    }
    double e = ix.hloc(s, s, a, a).real() - m.xmu;
    // spin_field(a,3) (n_up - n_dw): stored/H_local.f90:38-42
    e += (spin == 0 ? 1.0 : -1.0) * m.spin_field[a * 3 + 2];
    if (m.hfmode) {
      e -= 0.5 * ix.uloc(a);
      for (int b = 0; b < norb; b++) {
        if (b == a) continue;
        int lo = std::min(a, b), hi = std::max(a, b);
        e -= 0.5 * ix.ust(lo, hi) + 0.5 * (ix.ust(lo, hi) - ix.jh(lo, hi));
      }
    }
    ob.eps[a] = e;
    for (int k = 0; k < m.nbath; k++) {
      const int p = ix.bath_pos(a, k);
      const double v = ix.bath(m.bv, s, a, k);
      ob.a[p * ns + a] += v;
      ob.a[a * ns + p] += v;
    }
  }
  if (ix.replica()) {
    // bath levels and inter-orbital bath hops of replica k: hbath_tmp(s,s,a,b,k)
    // (stored/H_local.f90 bath_diag, stored/H_up.f90:26-50; real part in the real-valued normal mode)
    for (int k = 0; k < m.nbath; k++)
      for (int a = 0; a < norb; a++)
        for (int b = 0; b < norb; b++) {
          const double h = ix.hb(s, s, a, b, k).real();
          if (a == b)
            ob.eps[ix.bath_pos(a, k)] += h;
          else
            ob.a[ix.bath_pos(a, k) * ns + ix.bath_pos(b, k)] += h;  // c^+_{a,k} c_{b,k}
        }
    return ob;
  }
  for (int a = 0; a < ix.n_ebath_orb(); a++)
    for (int k = 0; k < m.nbath; k++) ob.eps[ix.bath_pos(a, k)] += ix.bath(m.be, s, a, k);
  return ob;
}

inline uint32_t between_mask(int p, int q) {
  const int lo = std::min(p, q), hi = std::max(p, q);
  return ((1u << hi) - 1u) & ~((1u << (lo + 1)) - 1u);
}

// rows of the hopping matrix of one species in a fixed-N basis
void hop_csr(const OneBody& ob, const CombBasis& b, HostCsr& out) {
  const int ns = ob.ns;
  const int64_t n = b.size();
  out.nrow = out.ncol = n;
  out.is_complex = false;
  out.rowptr.assign(n + 1, 0);
  out.col.clear();
  out.val.clear();
  // the hops that exist at all
  struct Hop { int q, p; double t; uint32_t btw; };
  std::vector<Hop> hops;
  for (int q = 0; q < ns; q++)
    for (int p = 0; p < ns; p++)
      if (p != q && ob.a[q * ns + p] != 0.0) hops.push_back({q, p, ob.a[q * ns + p], between_mask(p, q)});
  out.col.reserve((size_t)n * 8);
  out.val.reserve((size_t)n * 8);
  for (int64_t i = 0; i < n; i++) {
    const uint32_t mi = (uint32_t)b.states[i];
    for (const Hop& h : hops) {
      if (!((mi >> h.q) & 1u) || ((mi >> h.p) & 1u)) continue;
      const uint32_t mj = mi ^ (1u << h.q) ^ (1u << h.p);
      const double sg = (popc(mi & h.btw) & 1) ? -1.0 : 1.0;
      out.col.push_back(b.rank(mj));
      out.val.push_back(h.t * sg);
    }
    out.rowptr[i + 1] = (int64_t)out.col.size();
  }
}

}  // namespace

// _CMPLX_NORMAL: H = S + iA with S built from the real parts (the ordinary real-valued build) and A, real and
// antisymmetric, from the imaginary parts of the one-body hops (impHloc off-diagonal, replica / general bath
// matrices).  Returns the model whose ordinary build gives the hop matrices of A (everything else zero);
// any = false when all imaginary parts vanish.
edigpu_model imag_part_model(const edigpu_model& m, bool& any) {
  edigpu_model r = m;
  any = false;
  r.hfmode = 0;
  r.xmu = 0.0;
  r.nph = 0;
  std::fill(std::begin(r.uloc), std::end(r.uloc), 0.0);
  std::fill(std::begin(r.ust), std::end(r.ust), 0.0);
  std::fill(std::begin(r.jh), std::end(r.jh), 0.0);
  std::fill(std::begin(r.jx), std::end(r.jx), 0.0);
  std::fill(std::begin(r.jp), std::end(r.jp), 0.0);
  std::fill(std::begin(r.be), std::end(r.be), 0.0);
  std::fill(std::begin(r.bv), std::end(r.bv), 0.0);
  std::fill(std::begin(r.g_ph), std::end(r.g_ph), 0.0);
  std::fill(std::begin(r.spin_field), std::end(r.spin_field), 0.0);
  std::fill(std::begin(r.exc_field), std::end(r.exc_field), 0.0);
  r.nsundry = 0;
  // (re, im) -> (im, 0) off the orbital diagonal, 0 on it (a Hermitian matrix has a real diagonal)
  const size_t nh = sizeof(r.hloc) / sizeof(double) / 2;
  for (size_t i = 0; i < nh; i++) {
    const size_t ab = i % ((size_t)EDIGPU_MAXORB * EDIGPU_MAXORB);
    const bool diag = ab / EDIGPU_MAXORB == ab % EDIGPU_MAXORB;
    const double im = diag ? 0.0 : m.hloc[2 * i + 1];
    r.hloc[2 * i] = im;
    r.hloc[2 * i + 1] = 0.0;
    any = any || im != 0.0;
  }
  const size_t nb = sizeof(r.hb) / sizeof(double) / 2;
  for (size_t i = 0; i < nb; i++) {
    const size_t ab = (i / EDIGPU_MAXBATH) % ((size_t)EDIGPU_MAXORB * EDIGPU_MAXORB);
    const bool diag = ab / EDIGPU_MAXORB == ab % EDIGPU_MAXORB;
    const double im = (diag || !(m.bath_type == 2 || m.bath_type == 3)) ? 0.0 : m.hb[2 * i + 1];
    r.hb[2 * i] = im;
    r.hb[2 * i + 1] = 0.0;
    any = any || im != 0.0;
  }
  return r;
}

// Electron-phonon operator O = sum_{ab} g_ph(a,b) sum_sigma c^+_{a sigma} c_{b sigma} (stored/H_e_ph.f90: the
// electronic factor of g (b + b^+)) as the model whose ordinary sector build gives exactly O: impHloc = g, all else 0.
bool eph_offdiagonal(const edigpu_model& m) {
  for (int a = 0; a < m.norb; a++)
    for (int b = 0; b < m.norb; b++)
      if (a != b && m.g_ph[a * EDIGPU_MAXORB + b] != 0.0) return true;
  return false;
}

edigpu_model eph_operator_model(const edigpu_model& m) {
  edigpu_model r;
  std::memset(&r, 0, sizeof(r));
  r.ed_mode = m.ed_mode;
  r.bath_type = m.bath_type;
  r.norb = m.norb;
  r.nbath = m.nbath;
  r.nspin = m.nspin;
  for (int s = 0; s < 2; s++)
    for (int a = 0; a < m.norb; a++)
      for (int b = 0; b < m.norb; b++)
        r.hloc[((((s * 2) + s) * EDIGPU_MAXORB + a) * EDIGPU_MAXORB + b) * 2] = m.g_ph[a * EDIGPU_MAXORB + b];
  return r;
}

std::string sector_dim(const edigpu_model& m, int q1, int q2, int64_t& dim) {
  std::string e = check_model(m);
  if (!e.empty()) return e;
  const int ns = model_ns(m);
  switch (m.ed_mode) {
    case 0: dim = binomial(ns, q1) * binomial(ns, q2); return "";
    case 1: {  // superc, Sz = q1: sum over nup-ndw = Sz
      int64_t d = 0;
      for (int nd = 0; nd <= ns; nd++) d += binomial(ns, nd) * binomial(ns, nd + q1);
      dim = d;
      return "";
    }
    case 2: dim = binomial(2 * ns, q1); return "";
    default: return "edigpu: unknown ed_mode";
  }
}


// Regroup the factored Hnd terms: sum_t c_t (Pdw_t (x) Pup_t) with equal coefficients and pairwise disjoint supports
// collapses, wherever the terms form a full product set A x B of up-maps and down-maps, into one term
// (U_{a in A} Pup_a) (x) (U_{b in B} Pdw_b).  With Jx = Jp the four terms of an orbital pair {a, b} (exchange and pair
// hopping, both directions) are such a set: 12 terms of a 3-orbital model become 3.  Every matrix element is produced
// by exactly one (term, row) as before, so the product is unchanged; the kernels do a quarter of the partner look-ups.
static void merge_factored_terms(HostFactored& fac, int64_t du, int64_t dd) {
  const int nt = fac.nterms;
  if (nt < 2) return;
  auto dedupe = [&](const std::vector<uint32_t>& tab, int64_t n, std::vector<int>& id, std::vector<int>& rep) {
    id.assign(nt, -1);
    for (int t = 0; t < nt; t++) {
      for (size_t r = 0; r < rep.size() && id[t] < 0; r++)
        if (std::equal(tab.begin() + (size_t)t * n, tab.begin() + (size_t)(t + 1) * n, tab.begin() + (size_t)rep[r] * n))
          id[t] = (int)r;
      if (id[t] < 0) {
        id[t] = (int)rep.size();
        rep.push_back(t);
      }
    }
  };
  std::vector<int> uid, did, urep, drep;
  dedupe(fac.jup, du, uid, urep);
  dedupe(fac.jdw, dd, did, drep);
  const int nu = (int)urep.size(), nd = (int)drep.size();
  std::vector<double> c((size_t)nu * nd, 0.0);
  std::vector<char> has((size_t)nu * nd, 0);
  for (int t = 0; t < nt; t++) {
    c[(size_t)uid[t] * nd + did[t]] += fac.coef[t];
    has[(size_t)uid[t] * nd + did[t]] = 1;
  }
  auto disjoint = [](const std::vector<uint32_t>& tab, int64_t n, int a, int b) {
    for (int64_t i = 0; i < n; i++)
      if (tab[(size_t)a * n + i] != 0xFFFFFFFFu && tab[(size_t)b * n + i] != 0xFFFFFFFFu) return false;
    return true;
  };
  std::vector<double> coef;
  std::vector<uint32_t> jup, jdw;
  int nout = 0;
  for (int u0 = 0; u0 < nu; u0++)
    for (int d0 = 0; d0 < nd; d0++) {
      if (!has[(size_t)u0 * nd + d0]) continue;
      const double cc = c[(size_t)u0 * nd + d0];
      std::vector<int> A{u0}, B{d0};
      for (bool grew = true; grew;) {
        grew = false;
        for (int u = 0; u < nu; u++) {
          if (std::find(A.begin(), A.end(), u) != A.end()) continue;
          bool ok = true;
          for (int d : B) ok = ok && has[(size_t)u * nd + d] && c[(size_t)u * nd + d] == cc;
          for (int a : A) ok = ok && disjoint(fac.jup, du, urep[u], urep[a]);
          if (ok) A.push_back(u), grew = true;
        }
        for (int d = 0; d < nd; d++) {
          if (std::find(B.begin(), B.end(), d) != B.end()) continue;
          bool ok = true;
          for (int u : A) ok = ok && has[(size_t)u * nd + d] && c[(size_t)u * nd + d] == cc;
          for (int b : B) ok = ok && disjoint(fac.jdw, dd, drep[d], drep[b]);
          if (ok) B.push_back(d), grew = true;
        }
      }
      for (int u : A)
        for (int d : B) has[(size_t)u * nd + d] = 0;
      coef.push_back(cc);
      jup.resize((size_t)(nout + 1) * du, 0xFFFFFFFFu);
      jdw.resize((size_t)(nout + 1) * dd, 0xFFFFFFFFu);
      for (int u : A)
        for (int64_t i = 0; i < du; i++)
          if (fac.jup[(size_t)urep[u] * du + i] != 0xFFFFFFFFu) jup[(size_t)nout * du + i] = fac.jup[(size_t)urep[u] * du + i];
      for (int d : B)
        for (int64_t i = 0; i < dd; i++)
          if (fac.jdw[(size_t)drep[d] * dd + i] != 0xFFFFFFFFu) jdw[(size_t)nout * dd + i] = fac.jdw[(size_t)drep[d] * dd + i];
      nout++;
    }
  fac.nterms = nout;
  fac.coef.swap(coef);
  fac.jup.swap(jup);
  fac.jdw.swap(jdw);
}

std::string build_normal(const edigpu_model& m, int nup, int ndw, int64_t dw_first,
                         int64_t dw_count, HostNormal& out, bool explicit_arrays) {
  std::string e = check_model(m);
  if (!e.empty()) return e;
  if (m.ed_mode != 0) return "edigpu_normal_build: model.ed_mode is not normal";
  Idx ix(m);
  const int ns = model_ns(m), norb = m.norb;
  if (nup < 0 || nup > ns || ndw < 0 || ndw > ns) return "edigpu_normal_build: bad sector";
  out.ns = ns;
  out.nup = nup;
  out.ndw = ndw;
  out.bup.init(ns, nup);
  out.bdw.init(ns, ndw);
  out.dim_up = out.bup.size();
  out.dim_dw = out.bdw.size();
  if (out.dim_up * out.dim_dw >= ((int64_t)1 << 31))
    return "edigpu_normal_build: sector dimension >= 2^31 (the reference's integer range)";
  if (dw_count < 0) {
    dw_first = 0;
    dw_count = out.dim_dw;
  }
  if (dw_first < 0 || dw_first + dw_count > out.dim_dw) return "edigpu_normal_build: bad shard";
  out.dw_first = dw_first;
  out.dw_count = dw_count;
  const int64_t DimUp = out.dim_up;

  OneBody oup = one_body_normal(m, 0), odw = one_body_normal(m, 1);
  hop_csr(oup, out.bup, out.up);
  hop_csr(odw, out.bdw, out.dw);

  // ---- diagonal: Hd(iup,idw) = Eu(iup) + Ed(idw) + X(imp bits up, imp bits dw) ----
  const uint32_t impmask = (1u << norb) - 1u;
  auto species_energy = [&](const OneBody& ob, const CombBasis& b, std::vector<double>& en) {
    en.resize(b.size());
    for (int64_t i = 0; i < b.size(); i++) {
      const uint32_t s = (uint32_t)b.states[i];
      double x = 0.0;
      for (int p = 0; p < ns; p++)
        if ((s >> p) & 1u) x += ob.eps[p];
      for (int a = 0; a < norb; a++)
        for (int bb = a + 1; bb < norb; bb++)
          if (((s >> a) & 1u) && ((s >> bb) & 1u)) x += ix.ust(a, bb) - ix.jh(a, bb);
      en[i] = x;
    }
  };
  std::vector<double> eu, ed;
  species_energy(oup, out.bup, eu);
  species_energy(odw, out.bdw, ed);
  std::vector<double> xt((size_t)1 << (2 * norb));
  double cst = 0.0;
  if (m.hfmode) {
    for (int a = 0; a < norb; a++) cst += 0.25 * ix.uloc(a);
    for (int a = 0; a < norb; a++)
      for (int b = a + 1; b < norb; b++) cst += 0.5 * ix.ust(a, b) + 0.5 * (ix.ust(a, b) - ix.jh(a, b));
  }
  for (uint32_t iu = 0; iu <= impmask; iu++)
    for (uint32_t id = 0; id <= impmask; id++) {
      double x = cst;
      for (int a = 0; a < norb; a++) {
        const int nu = (iu >> a) & 1, nd = (id >> a) & 1;
        x += ix.uloc(a) * nu * nd;
        for (int b = a + 1; b < norb; b++) {
          const int nub = (iu >> b) & 1, ndb = (id >> b) & 1;
          x += ix.ust(a, b) * (nu * ndb + nub * nd);
        }
      }
      xt[(iu << norb) | id] = x;
    }
  out.norb = norb;
  out.ob_a[0] = oup.a;
  out.ob_a[1] = odw.a;
  out.ob_eps[0] = oup.eps;
  out.ob_eps[1] = odw.eps;
  out.xt = xt;
  out.samespin.assign((size_t)norb * norb, 0.0);
  for (int a = 0; a < norb; a++)
    for (int bb = a + 1; bb < norb; bb++) out.samespin[(size_t)a * norb + bb] = ix.ust(a, bb) - ix.jh(a, bb);
  // factored diagonal tables
  HostFactored& fac = out.fac;
  fac = HostFactored();
  fac.valid = true;
  fac.nimp = 1 << norb;
  fac.eux.resize((size_t)fac.nimp * DimUp);
  for (uint32_t id = 0; id <= impmask; id++)
    for (int64_t iup = 0; iup < DimUp; iup++) {
      const uint32_t mu = (uint32_t)out.bup.states[iup];
      fac.eux[(size_t)id * DimUp + iup] = eu[iup] + xt[((mu & impmask) << norb) | id];
    }
  fac.ed = ed;
  fac.impd.resize(out.dim_dw);
  for (int64_t idw = 0; idw < out.dim_dw; idw++) fac.impd[idw] = (uint8_t)((uint32_t)out.bdw.states[idw] & impmask);
  out.hd.clear();
  if (explicit_arrays) out.hd.resize((size_t)(dw_count * DimUp));
  for (int64_t r = 0; explicit_arrays && r < dw_count; r++) {
    const int64_t idw = dw_first + r;
    const uint32_t md = (uint32_t)out.bdw.states[idw];
    const double edr = ed[idw];
    const double* xrow_base = xt.data();
    double* dst = &out.hd[(size_t)(r * DimUp)];
    for (int64_t iup = 0; iup < DimUp; iup++) {
      const uint32_t mu = (uint32_t)out.bup.states[iup];
      dst[iup] = eu[iup] + edr + xrow_base[((mu & impmask) << norb) | (md & impmask)];
    }
  }

  // ---- non-local block: spin exchange + pair hopping (only touches impurity bits) ----
  bool any_jx = false, any_jp = false;
  for (int a = 0; a < norb; a++)
    for (int b = 0; b < norb; b++) {
      if (ix.jx(a, b) != 0.0) any_jx = true;
      if (ix.jp(a, b) != 0.0) any_jp = true;
    }
  // coulomb_sundry lines as per-species operator strings in application order (c_l, cd_j, c_k, cd_i), each operator
  // acting on the word of its own spin with that word's sign only (stored/H_sundry.f90:36-98), so every line is a
  // product (O_dw (x) O_up) like the exchange and pair-hopping terms
  struct SpeciesOp { int pos; bool create; };
  struct Sundry { std::vector<SpeciesOp> up, dw; double u; };
  std::vector<Sundry> sundry;
  if (m.nsundry < 0 || m.nsundry > EDIGPU_MAXSUNDRY) return "edigpu_normal_build: nsundry out of range";
  for (int il = 0; il < m.nsundry; il++) {
    const int32_t* op = &m.sundry_op[il * 8];
    Sundry sl;
    sl.u = m.sundry_u[il];
    const int order[4] = {3, 1, 2, 0};  // l, j, k, i
    int balance[2] = {0, 0};
    for (int k = 0; k < 4; k++) {
      const int w = order[k], orb = op[2 * w], sp = op[2 * w + 1];
      if (orb < 1 || orb > norb || sp < 1 || sp > 2) return "edigpu_normal_build: coulomb_sundry orbital / spin out of range";
      const bool create = w < 2;
      balance[sp - 1] += create ? 1 : -1;
      (sp == 1 ? sl.up : sl.dw).push_back({orb - 1, create});
    }
    if (balance[0] != 0 || balance[1] != 0)
      return "edigpu_normal_build: in normal mode coulomb_sundry operators that change the total spin are forbidden";
    if (sl.u != 0.0) sundry.push_back(sl);
  }
  // a string on one word: false when it annihilates the state, else the new word and the sign
  auto apply_ops = [](const std::vector<SpeciesOp>& ops, uint32_t st, uint32_t& res, int& neg) {
    neg = 0;
    for (const SpeciesOp& o : ops) {
      const uint32_t b = 1u << o.pos;
      if (o.create ? (st & b) != 0 : (st & b) == 0) return false;
      neg ^= popc(st & (b - 1u)) & 1;
      st ^= b;
    }
    res = st;
    return true;
  };
  out.has_nd = (norb > 1 && (any_jx || any_jp)) || !sundry.empty();
  out.nd = HostCsr();
  if (out.has_nd) {
    // factored form: one (Pdw (x) Pup) pair per (kind, a, b)
    auto partner = [&](const CombBasis& bs, int from, int to, std::vector<uint32_t>& dst) {
      // c^+_to c_from on every basis state: partner index | sign
      const uint32_t bf = 1u << from, bt = 1u << to, btw = between_mask(from, to);
      for (int64_t i = 0; i < bs.size(); i++) {
        const uint32_t st = (uint32_t)bs.states[i];
        if ((st & bf) && !(st & bt)) {
          const uint32_t j = (uint32_t)bs.rank(st ^ bf ^ bt);
          dst.push_back(j | ((popc(st & btw) & 1) ? 0x80000000u : 0u));
        } else {
          dst.push_back(0xFFFFFFFFu);
        }
      }
    };
    for (int a = 0; a < norb; a++)
      for (int b = 0; b < norb; b++) {
        if (a == b) continue;
        if (ix.jx(a, b) != 0.0) {  // up b->a, down a->b
          fac.coef.push_back(ix.jx(a, b));
          partner(out.bup, b, a, fac.jup);
          partner(out.bdw, a, b, fac.jdw);
          fac.nterms++;
        }
        if (ix.jp(a, b) != 0.0) {  // up b->a, down b->a
          fac.coef.push_back(ix.jp(a, b));
          partner(out.bup, b, a, fac.jup);
          partner(out.bdw, b, a, fac.jdw);
          fac.nterms++;
        }
      }
    auto partner_ops = [&](const CombBasis& bs, const std::vector<SpeciesOp>& ops, std::vector<uint32_t>& dst) {
      for (int64_t i = 0; i < bs.size(); i++) {
        uint32_t res;
        int neg;
        if (apply_ops(ops, (uint32_t)bs.states[i], res, neg))
          dst.push_back((uint32_t)bs.rank(res) | (neg ? 0x80000000u : 0u));
        else
          dst.push_back(0xFFFFFFFFu);
      }
    };
    for (const Sundry& sl : sundry) {
      fac.coef.push_back(sl.u);
      partner_ops(out.bup, sl.up, fac.jup);
      partner_ops(out.bdw, sl.dw, fac.jdw);
      fac.nterms++;
    }
    if (!getenv("EDIGPU_ND_NO_MERGE")) merge_factored_terms(fac, DimUp, out.dim_dw);
    struct Term { uint32_t xu, xd; double val; };
    std::vector<std::vector<Term>> terms((size_t)1 << (2 * norb));
    for (uint32_t iu = 0; iu <= impmask; iu++)
      for (uint32_t id = 0; id <= impmask; id++) {
        auto& tl = terms[(iu << norb) | id];
        for (int a = 0; a < norb; a++)
          for (int b = 0; b < norb; b++) {
            if (a == b) continue;
            const uint32_t ba = 1u << a, bb = 1u << b, btw = between_mask(a, b);
            const double sg = ((popc(iu & btw) + popc(id & btw)) & 1) ? -1.0 : 1.0;
            // spin exchange: up b->a, down a->b
            if (any_jx && (iu & bb) && !(iu & ba) && (id & ba) && !(id & bb) && ix.jx(a, b) != 0.0)
              tl.push_back({ba | bb, ba | bb, ix.jx(a, b) * sg});
            // pair hopping: up b->a, down b->a
            if (any_jp && (iu & bb) && !(iu & ba) && (id & bb) && !(id & ba) && ix.jp(a, b) != 0.0)
              tl.push_back({ba | bb, ba | bb, ix.jp(a, b) * sg});
          }
        for (const Sundry& sl : sundry) {
          uint32_t ru, rd;
          int nu, nd;
          if (apply_ops(sl.up, iu, ru, nu) && apply_ops(sl.dw, id, rd, nd))
            tl.push_back({iu ^ ru, id ^ rd, (nu ^ nd) ? -sl.u : sl.u});
        }
      }
    // nnz(Hnd) of the local rows from the impurity-pattern histograms (O(DimUp + DimDw))
    {
      std::vector<int64_t> cu((size_t)impmask + 1, 0), cd((size_t)impmask + 1, 0);
      for (int64_t iup = 0; iup < DimUp; iup++) cu[(uint32_t)out.bup.states[iup] & impmask]++;
      for (int64_t r = 0; r < dw_count; r++) cd[(uint32_t)out.bdw.states[dw_first + r] & impmask]++;
      out.nd_nnz = 0;
      for (uint32_t iu = 0; iu <= impmask; iu++)
        for (uint32_t id = 0; id <= impmask; id++)
          out.nd_nnz += cu[iu] * cd[id] * (int64_t)terms[(iu << norb) | id].size();
    }
    HostCsr& nd = out.nd;
    nd.nrow = dw_count * DimUp;
    nd.ncol = out.dim_up * out.dim_dw;
    if (!explicit_arrays) return "";
    nd.rowptr.assign(nd.nrow + 1, 0);
    int64_t nnz = 0;
    for (int64_t r = 0; r < dw_count; r++) {
      const uint32_t md = (uint32_t)out.bdw.states[dw_first + r] & impmask;
      for (int64_t iup = 0; iup < DimUp; iup++) {
        const uint32_t mu = (uint32_t)out.bup.states[iup] & impmask;
        nnz += (int64_t)terms[(mu << norb) | md].size();
        nd.rowptr[r * DimUp + iup + 1] = nnz;
      }
    }
    nd.col.resize(nnz);
    nd.val.resize(nnz);
    for (int64_t r = 0; r < dw_count; r++) {
      const uint32_t sd = (uint32_t)out.bdw.states[dw_first + r];
      for (int64_t iup = 0; iup < DimUp; iup++) {
        const uint32_t su = (uint32_t)out.bup.states[iup];
        const auto& tl = terms[((su & impmask) << norb) | (sd & impmask)];
        int64_t k = nd.rowptr[r * DimUp + iup];
        for (const Term& t : tl) {
          const int64_t jup = out.bup.rank(su ^ t.xu), jdw = out.bdw.rank(sd ^ t.xd);
          nd.col[k] = (int32_t)(jup + jdw * DimUp);
          nd.val[k] = t.val;
          k++;
        }
      }
    }
  }
  return "";
}

// --------------------------------------------------------------------------------------
// flat row-CSR sectors (superc: fixed Sz = Nup-Ndw; nonsu2: fixed Ntot)
// --------------------------------------------------------------------------------------
namespace {

// Basis ordered as build_sector does for these modes (ED_SECTOR.f90:263-281, :351-368):
// state = iup + idw*2^Ns, idw outer / iup inner, i.e. ascending integer.  Ranking is
// off_dw[idw] + rk_up[iup] (rk_up = rank among words of equal popcount).
struct SpinBasis {
  int ns = 0;
  std::vector<int32_t> states, off_dw, rk_up;
  int mode = 1, q = 0;
  int nup_for(int ndw) const { return mode == 1 ? ndw + q : q - ndw; }
  void init(int ns_, int mode_, int q_) {
    ns = ns_;
    mode = mode_;
    q = q_;
    const uint32_t nw = 1u << ns;
    rk_up.assign(nw, 0);
    std::vector<int32_t> cnt(ns + 1, 0);
    for (uint32_t w = 0; w < nw; w++) rk_up[w] = cnt[popc(w)]++;
    off_dw.assign(nw, 0);
    int64_t tot = 0;
    for (uint32_t d = 0; d < nw; d++) {
      off_dw[d] = (int32_t)tot;
      tot += binomial(ns, nup_for(popc(d)));
    }
    states.resize(tot);
    for (uint32_t d = 0; d < nw; d++) {
      const int nu = nup_for(popc(d));
      if (nu < 0 || nu > ns) continue;
      int64_t k = off_dw[d];
      uint32_t u = nu == 0 ? 0u : ((1u << nu) - 1u);
      const int64_t n = binomial(ns, nu);
      for (int64_t t = 0; t < n; t++) {
        states[k++] = (int32_t)(u | (d << ns));
        if (nu == 0) break;
        uint32_t c = u & (0u - u), r = u + c;
        u = (((r ^ u) >> 2) / c) | r;
      }
    }
  }
  // Jz_basis=T sectors of nonsu2 (ED_SECTOR.f90:289-350): Ntot = q and twoJz = (Nup - Ndw) + twoLz with twoLz = sum over
  // the levels iorb + Norb * ibath of 2 Lzdiag(iorb) (n_up + n_dw), Lzdiag = [-1, +1, 0] (ED_VARS_GLOBAL.f90:283).  The
  // states stay in ascending order.  The up words that go with a down word d are those of ONE (occupation, Lz) class,
  // (Ntot - n(d), the Lz that completes twoJz), so the two-table rank still holds with rk_up = rank of a word inside
  // its (occupation, Lz) class -- for states OF the sector.  rank() is asked about states that may lie outside (a model
  // that does not conserve Jz), so the host builder searches; the device kernels use the tables after
  // build_direct has checked every term against the conservation law.
  bool jz = false;
  void init_jz(int ns_, int norb, int ntot, int twojz) {
    static const int lzdiag[3] = {-1, +1, 0};
    ns = ns_;
    mode = 2;
    q = ntot;
    jz = true;
    const uint32_t nw = 1u << ns;
    std::vector<int8_t> lz(nw, 0);  // twoLz / 2 of a word
    for (uint32_t w = 0; w < nw; w++) {
      int x = 0;
      for (int p = 0; p < ns; p++)
        if ((w >> p) & 1u) x += lzdiag[p % norb];
      lz[w] = (int8_t)x;
    }
    states.clear();
    // rank of a word inside its (occupation, Lz) class; Lz of ns levels lies in [-ns, ns]
    rk_up.assign(nw, 0);
    {
      std::vector<int32_t> cnt((size_t)(ns + 1) * (2 * ns + 1), 0);
      for (uint32_t w = 0; w < nw; w++) rk_up[w] = cnt[(size_t)popc(w) * (2 * ns + 1) + (lz[w] + ns)]++;
    }
    off_dw.assign(nw, 0);
    for (uint32_t d = 0; d < nw; d++) {
      off_dw[d] = (int32_t)states.size();
      for (uint32_t u = 0; u < nw; u++) {
        const int nu = popc(u), nd = popc(d);
        if (nu + nd == ntot && (nu - nd) + 2 * (lz[u] + lz[d]) == twojz) states.push_back((int32_t)(u | (d << ns)));
      }
    }
  }

  inline int64_t rank(uint32_t s) const {
    if (jz) {
      auto it = std::lower_bound(states.begin(), states.end(), (int32_t)s);
      return (it != states.end() && *it == (int32_t)s) ? (int64_t)(it - states.begin()) : -1;
    }
    return (int64_t)off_dw[s >> ns] + rk_up[s & ((1u << ns) - 1u)];
  }
};

// apply c (create=false) or c^+ (create=true) at level pos; returns false if it annihilates
inline bool apply_op(uint32_t& s, int pos, bool create, double& sg) {
  const uint32_t b = 1u << pos;
  if (create ? (s & b) : !(s & b)) return false;
  if (popc(s & (b - 1u)) & 1) sg = -sg;
  s ^= b;
  return true;
}

struct OpTerm {           // coef * op[n-1] ... op[1] op[0]  (op[0] acts first)
  int n;
  int pos[4];
  bool create[4];
  cplx coef;              // value inserted at (row = source state, col = result state)
};

}  // namespace

// Sector map as build_sector leaves it (ED_SECTOR.f90:165-373): which = 0 / 1: H(1)%map (up) / H(2)%map (down) of a
// normal-mode sector (q1, q2) = (Nup, Ndw); superc / nonsu2: the single map of sector q1 (Sz / Ntot), which ignored.
std::string sector_map_jz(const edigpu_model& m, int ntot, int twojz, std::vector<int32_t>& out) {
  std::string e = check_model(m);
  if (!e.empty()) return e;
  const int ns = model_ns(m);
  if (m.ed_mode != 2 || m.norb != 3 || 2 * ns > 31 || ntot < 0 || ntot > 2 * ns) return "edigpu_sector_map_jz: needs nonsu2, Norb = 3";
  SpinBasis sb;
  sb.init_jz(ns, m.norb, ntot, twojz);
  out = sb.states;
  return "";
}

std::string sector_map(const edigpu_model& m, int q1, int q2, int which, std::vector<int32_t>& out) {
  std::string e = check_model(m);
  if (!e.empty()) return e;
  const int ns = model_ns(m);
  if (m.ed_mode == 0) {
    if (q1 < 0 || q1 > ns || q2 < 0 || q2 > ns || which < 0 || which > 1) return "edigpu_sector_map: bad sector";
    CombBasis b;
    b.init(ns, which == 0 ? q1 : q2);
    out = b.states;
    return "";
  }
  if (m.ed_mode == 1 ? (q1 < -ns || q1 > ns) : (q1 < 0 || q1 > 2 * ns)) return "edigpu_sector_map: bad sector";
  if (2 * ns > 31) return "edigpu_sector_map: 2 Ns > 31 bits";
  SpinBasis sb;
  sb.init(ns, m.ed_mode, q1);
  out = sb.states;
  return "";
}

// Operator strings and diagonal data of the superc / nonsu2 Hamiltonians over the 2*Ns spin-orbital
// levels (up: 0..Ns-1, down: Ns..2Ns-1); shared by the stored (CSR) and the direct (on-the-fly) builders.
static void flat_physics(const edigpu_model& m, std::vector<OpTerm>& terms, std::vector<double>& eps_out,
                         double& cst_out) {
  Idx ix(m);
  const int ns = model_ns(m), norb = m.norb, nbath = m.nbath;
  const int sd = m.nspin - 1;  // spin index used for the down species
  // ---- operator strings; coefficient = what lands at (row i, col j=O|i>) ----
  auto hop = [&](int p, int q, cplx h) {  // c^+_p c_q with amplitude h: entry conj(h)
    if (h == cplx(0.0)) return;
    terms.push_back({2, {q, p, 0, 0}, {false, true, false, false}, std::conj(h)});
  };
  for (int a = 0; a < norb; a++)
    for (int b = 0; b < norb; b++) {
      if (a != b) {
        hop(a, b, ix.hloc(0, 0, a, b));
        hop(a + ns, b + ns, ix.hloc(sd, sd, a, b));
      }
      if (m.ed_mode == 2) {  // spin-flip local terms (ED_NONSU2/stored/Himp.f90:80-108)
        hop(a, b + ns, ix.hloc(0, 1, a, b));
        hop(a + ns, b, ix.hloc(1, 0, a, b));
      }
    }
  for (int a = 0; a < norb; a++)
    for (int k = 0; k < nbath; k++) {
      const int p = ix.bath_pos(a, k);
      const double vu = ix.bath(m.bv, 0, a, k), vd = ix.bath(m.bv, sd, a, k);
      hop(p, a, vu);
      hop(a, p, vu);
      hop(p + ns, a + ns, vd);
      hop(a + ns, p + ns, vd);
      if (m.ed_mode == 2 && !ix.replica()) {  // spin-flip hybridisation (ED_NONSU2/stored/Himp_bath.f90:77-136)
        const double uu = ix.bath(m.bu, 0, a, k), ud = ix.bath(m.bu, sd, a, k);
        hop(p + ns, a, uu);
        hop(a, p + ns, uu);
        hop(p, a + ns, ud);
        hop(a + ns, p, ud);
      }
    }
  if (ix.replica()) {
    // replica / general bath: matrices hbath_tmp(is,js,a,b,k) (ED_SUPERC/stored/Hbath.f90:40-92,130-186,
    // ED_NONSU2/stored/Hbath.f90:38-116)
    const int n1 = m.ed_mode == 1 ? 1 : sd;  // index of the second diagonal block (Nambu hole / spin down)
    for (int k = 0; k < nbath; k++)
      for (int a = 0; a < norb; a++)
        for (int b = 0; b < norb; b++) {
          const int pa = ix.bath_pos(a, k), pb = ix.bath_pos(b, k);
          if (a != b) {
            hop(pa, pb, ix.hb(0, 0, a, b, k));  // up: c^+_{a} c_{b}
            if (m.ed_mode == 1) {
              // Nambu hole block: c_{a,dw} c^+_{b,dw} with entry conj(h)
              const cplx h = ix.hb(n1, n1, a, b, k);
              if (h != cplx(0.0))
                terms.push_back({2, {pb + ns, pa + ns, 0, 0}, {true, false, false, false}, std::conj(h)});
            } else {
              hop(pa + ns, pb + ns, ix.hb(n1, n1, a, b, k));
            }
          }
          if (m.ed_mode == 1) {
            // anomalous blocks: c^+_{a,up} c^+_{b,dw} (0,1) and c_{a,dw} c_{b,up} (1,0)
            const cplx h01 = ix.hb(0, 1, a, b, k), h10 = ix.hb(1, 0, a, b, k);
            if (h01 != cplx(0.0))
              terms.push_back({2, {pb + ns, pa, 0, 0}, {true, true, false, false}, std::conj(h01)});
            if (h10 != cplx(0.0))
              terms.push_back({2, {pb, pa + ns, 0, 0}, {false, false, false, false}, std::conj(h10)});
          } else if (m.ed_mode == 2 && m.nspin == 2) {
            hop(pa, pb + ns, ix.hb(0, 1, a, b, k));       // c^+_{a,up} c_{b,dw}
            hop(pa + ns, pb, ix.hb(1, 0, a, b, k));       // c^+_{a,dw} c_{b,up}
          }
        }
  }
  if (m.ed_mode == 1 && !ix.replica()) {
    // bath pairing (ED_SUPERC/stored/Hbath.f90:94-128): d c_{dw} c_{up} and d c^+_{up} c^+_{dw}
    for (int a = 0; a < ix.n_ebath_orb(); a++)
      for (int k = 0; k < nbath; k++) {
        const int p = ix.bath_pos(a, k);
        const double d = ix.bath(m.bd, 0, a, k);
        if (d == 0.0) continue;
        terms.push_back({2, {p, p + ns, 0, 0}, {false, false, false, false}, cplx(d)});
        terms.push_back({2, {p + ns, p, 0, 0}, {true, true, false, false}, cplx(d)});
      }
  }
  if (m.ed_mode == 1) {
    // local pair field (ED_SUPERC/stored/Himp.f90:84-124)
    for (int a = 0; a < norb; a++) {
      const double f = m.pair_field[a];
      if (f == 0.0) continue;
      terms.push_back({2, {a, a + ns, 0, 0}, {false, false, false, false}, cplx(f)});
      terms.push_back({2, {a + ns, a, 0, 0}, {true, true, false, false}, cplx(f)});
    }
  }
  // spin exchange and pair hopping (ED_SUPERC/stored/Hint.f90:60-121 == ED_NONSU2/stored/Hint.f90)
  if (norb > 1)
    for (int a = 0; a < norb; a++)
      for (int b = 0; b < norb; b++) {
        if (a == b) continue;
        if (ix.jx(a, b) != 0.0)
          terms.push_back({4, {b, a + ns, b + ns, a}, {false, false, true, true}, cplx(ix.jx(a, b))});
        if (ix.jp(a, b) != 0.0)
          terms.push_back({4, {b, b + ns, a + ns, a}, {false, false, true, true}, cplx(ix.jp(a, b))});
      }

  // coulomb_sundry (ED_SUPERC/stored/Hint.f90:127-178 == ED_NONSU2/stored/Hint.f90:127-181): U cd_i cd_j c_k c_l applied
  // right to left as c_l, cd_j, c_k, cd_i on the 2 Ns-level word (ranges checked by check_flat_fields)
  for (int il = 0; il < m.nsundry; il++) {
    const int32_t* op = &m.sundry_op[il * 8];
    if (m.sundry_u[il] == 0.0) continue;
    auto lev = [&](int k) { return (op[2 * k] - 1) + ns * (op[2 * k + 1] - 1); };
    terms.push_back({4, {lev(3), lev(1), lev(2), lev(0)}, {false, true, false, true}, cplx(m.sundry_u[il])});
  }
  if (m.ed_mode == 2) {
    // exc_field = (F_0, F_x, F_y, F_z), F.T (ED_NONSU2/stored/Himp.f90:113-228): entries as inserted, no conjugation
    const double f0 = m.exc_field[0], fx = m.exc_field[1], fy = m.exc_field[2], fz = m.exc_field[3];
    auto entry = [&](int q, int p, cplx v) {  // c^+_p c_q, value v at (i, j = O|i>)
      if (v != cplx(0.0)) terms.push_back({2, {q, p, 0, 0}, {false, true, false, false}, v});
    };
    if (f0 != 0.0 || fx != 0.0 || fy != 0.0 || fz != 0.0)
      for (int a = 0; a < norb; a++)
        for (int b = 0; b < norb; b++) {
          if (a != b) {
            entry(b + ns, a + ns, cplx(f0 - fz));
            entry(b, a, cplx(f0 + fz));
          }
          entry(b, a + ns, cplx(fx, -fy));
          entry(b + ns, a, cplx(fx, fy));
        }
    // spin_field(a, x|y|z), F.S (Himp.f90:235-296); the z part goes to the level energies below
    for (int a = 0; a < norb; a++) {
      const double sx = m.spin_field[a * 3 + 0], sy = m.spin_field[a * 3 + 1];
      entry(a, a + ns, cplx(sx, -sy));
      entry(a + ns, a, cplx(sx, sy));
    }
  }

  // ---- diagonal pieces ----
  std::vector<double>& eps = eps_out;
  eps.assign(2 * ns, 0.0);
  for (int a = 0; a < norb; a++) {
    double shift = -m.xmu;
    if (m.hfmode) {
      shift -= 0.5 * ix.uloc(a);
      for (int b = 0; b < norb; b++) {
        if (b == a) continue;
        const int lo = std::min(a, b), hi = std::max(a, b);
        shift -= 0.5 * ix.ust(lo, hi) + 0.5 * (ix.ust(lo, hi) - ix.jh(lo, hi));
      }
    }
    eps[a] = ix.hloc(0, 0, a, a).real() + shift;
    eps[a + ns] = ix.hloc(sd, sd, a, a).real() + shift;
    if (m.ed_mode == 2) {
      // F_z (n_up - n_dw).  Himp.f90:237-240 adds this to a variable the blocks before it leave set (`htmp = htmp +`
      // with no reset), i.e. the reference inserts F_z S^z plus a stale value; built here is what its comment states.
      eps[a] += m.spin_field[a * 3 + 2];
      eps[a + ns] -= m.spin_field[a * 3 + 2];
    }
  }
  if (ix.replica()) {
    for (int a = 0; a < norb; a++)
      for (int k = 0; k < nbath; k++) {
        eps[ix.bath_pos(a, k)] += ix.hb(0, 0, a, a, k).real();
        // superc: minus the Nambu hole block (Hbath.f90:44-52); nonsu2: the spin-down block
        eps[ix.bath_pos(a, k) + ns] +=
            m.ed_mode == 1 ? -ix.hb(1, 1, a, a, k).real() : ix.hb(sd, sd, a, a, k).real();
      }
  } else {
    for (int a = 0; a < ix.n_ebath_orb(); a++)
      for (int k = 0; k < nbath; k++) {
        eps[ix.bath_pos(a, k)] += ix.bath(m.be, 0, a, k);
        eps[ix.bath_pos(a, k) + ns] += ix.bath(m.be, sd, a, k);
      }
  }
  double& cst = cst_out;
  cst = 0.0;
  if (m.hfmode) {
    for (int a = 0; a < norb; a++) cst += 0.25 * ix.uloc(a);
    for (int a = 0; a < norb; a++)
      for (int b = a + 1; b < norb; b++) cst += 0.5 * ix.ust(a, b) + 0.5 * (ix.ust(a, b) - ix.jh(a, b));
  }

}

// coulomb_sundry lines in range and inside the sector family; the superc files of the reference hold no spin_field /
// exc_field terms (they would be ignored there): refused, so that nobody relies on them
static std::string check_flat_fields(const edigpu_model& m, const char* who) {
  if (m.nsundry < 0 || m.nsundry > EDIGPU_MAXSUNDRY) return std::string(who) + ": nsundry out of range";
  for (int il = 0; il < m.nsundry; il++) {
    const int32_t* op = &m.sundry_op[il * 8];
    int dsz = 0;
    for (int k = 0; k < 4; k++) {
      const int orb = op[2 * k], sp = op[2 * k + 1];
      if (orb < 1 || orb > m.norb || sp < 1 || sp > 2) return std::string(who) + ": coulomb_sundry orbital / spin out of range";
      dsz += (k < 2 ? 1 : -1) * (sp == 1 ? 1 : -1);  // cd_i, cd_j create; c_k, c_l annihilate
    }
    if (m.ed_mode == 1 && dsz != 0 && m.sundry_u[il] != 0.0)
      return std::string(who) + ": coulomb_sundry line changes Sz (the reference stops with 'impossible operator')";
  }
  if (m.ed_mode == 1) {
    bool any = false;
    for (double x : m.spin_field) any = any || x != 0.0;
    for (double x : m.exc_field) any = any || x != 0.0;
    if (any) return std::string(who) + ": spin_field / exc_field have no terms in the superc Hamiltonian";
  }
  return "";
}

std::string build_flat(const edigpu_model& m, int sector, int64_t row_first, int64_t row_count,
                       HostFlat& out, bool jz_basis, int twojz) {
  std::string e = check_model(m);
  if (!e.empty()) return e;
  if (m.ed_mode != 1 && m.ed_mode != 2) return "edigpu_flat_build: model.ed_mode must be superc or nonsu2";
  if (m.ed_mode == 2 && m.nspin != 2) return "edigpu_flat_build: nonsu2 needs nspin=2";
  e = check_flat_fields(m, "edigpu_flat_build");
  if (!e.empty()) return e;
  Idx ix(m);
  const int ns = model_ns(m), norb = m.norb;
  if (2 * ns > 30) return "edigpu_flat_build: 2*Ns > 30 bits (the reference's integer range)";
  SpinBasis sb;
  if (jz_basis) {
    if (m.ed_mode != 2 || norb != 3) return "edigpu_flat_build_jz: Jz_basis needs ed_mode = nonsu2 and Norb = 3 (Lzdiag = [-1, +1, 0])";
    if (!(m.bath_type == 2 || m.bath_type == 3 || m.nbath == 1))
      return "edigpu_flat_build_jz: Jz_basis labels the levels as iorb + Norb * ibath: replica / general bath (or Nbath = 1)";
    if (sector < 0 || sector > 2 * ns) return "edigpu_flat_build_jz: bad sector";
    sb.init_jz(ns, norb, sector, twojz);
  } else {
    sb.init(ns, m.ed_mode, sector);
  }
  out.ns = ns;
  out.dim = (int64_t)sb.states.size();
  if (row_count < 0) {
    row_first = 0;
    row_count = out.dim;
  }
  if (row_first < 0 || row_first + row_count > out.dim) return "edigpu_flat_build: bad shard";
  out.row_first = row_first;
  out.row_count = row_count;

  std::vector<OpTerm> terms;
  std::vector<double> eps;
  double cst = 0.0;
  flat_physics(m, terms, eps, cst);

  HostCsr& H = out.h;
  H.nrow = row_count;
  H.ncol = out.dim;
  H.is_complex = true;
  H.rowptr.assign(row_count + 1, 0);
  H.col.clear();
  H.val.clear();
  H.col.reserve((size_t)row_count * 24);
  H.val.reserve((size_t)row_count * 48);
  std::vector<int32_t> rc;
  std::vector<cplx> rv;
  for (int64_t r = 0; r < row_count; r++) {
    const int64_t i = row_first + r;
    const uint32_t s = (uint32_t)sb.states[i];
    rc.clear();
    rv.clear();
    double dg = cst;
    for (int p = 0; p < 2 * ns; p++)
      if ((s >> p) & 1u) dg += eps[p];
    for (int a = 0; a < norb; a++) {
      const int nu = (s >> a) & 1, nd = (s >> (a + ns)) & 1;
      dg += ix.uloc(a) * nu * nd;
      for (int b = a + 1; b < norb; b++) {
        const int nub = (s >> b) & 1, ndb = (s >> (b + ns)) & 1;
        dg += ix.ust(a, b) * (nu * ndb + nub * nd);
        dg += (ix.ust(a, b) - ix.jh(a, b)) * (nu * nub + nd * ndb);
      }
    }
    rc.push_back((int32_t)i);
    rv.push_back(cplx(dg, 0.0));
    for (const OpTerm& t : terms) {
      uint32_t w = s;
      double sg = 1.0;
      bool ok = true;
      for (int k = 0; k < t.n && ok; k++) ok = apply_op(w, t.pos[k], t.create[k], sg);
      if (!ok) continue;
      const int32_t j = (int32_t)sb.rank(w);
      if (j < 0) return "edigpu_flat_build: a matrix element leaves the sector (Jz_basis: the model does not conserve Jz)";
      const cplx v = t.coef * sg;
      bool merged = false;
      for (size_t k = 0; k < rc.size(); k++)
        if (rc[k] == j) {
          rv[k] += v;
          merged = true;
          break;
        }
      if (!merged) {
        rc.push_back(j);
        rv.push_back(v);
      }
    }
    for (size_t k = 0; k < rc.size(); k++) {
      H.col.push_back(rc[k]);
      H.val.push_back(rv[k].real());
      H.val.push_back(rv[k].imag());
    }
    H.rowptr[r + 1] = (int64_t)H.col.size();
  }
  out.states = std::move(sb.states);
  return "";
}

std::string build_orbs(const edigpu_model& m, const int* nups, const int* ndws, HostOrbs& out, bool explicit_diag) {
  std::string e = check_model(m);
  if (!e.empty()) return e;
  if (m.ed_mode != 0) return "edigpu_orbs_build: model.ed_mode is not normal";
  if (m.bath_type != 0) return "edigpu_orbs_build: ed_total_ud=F needs bath_type=normal";
  Idx ix(m);
  const int norb = m.norb, nbath = m.nbath, nso = 1 + nbath, sd = m.nspin - 1;
  out = HostOrbs();
  out.naxes = 2 * norb;
  out.dims.resize(out.naxes);
  out.fac.resize(out.naxes);
  out.eax.resize(out.naxes);
  out.impbit.resize(out.naxes);
  out.factored = true;
  out.dim = 1;
  for (int k = 0; k < out.naxes; k++) {
    const int a = k < norb ? k : k - norb, s = k < norb ? 0 : sd;
    const int n = k < norb ? nups[a] : ndws[a];
    if (n < 0 || n > nso) return "edigpu_orbs_build: bad sector";
    CombBasis b;  // chain of orbital a: level 0 = impurity, level kp = bath site kp
    b.init(nso, n);
    const int64_t d = b.size();
    out.dims[k] = d;
    if (out.dim * d >= ((int64_t)1 << 31)) return "edigpu_orbs_build: sector dimension >= 2^31";
    out.dim *= d;
    // one-body chain: hybridisation hops imp <-> bath, bath energies (stored/Orbs/H_up.f90, H_local.f90)
    OneBody ob;
    ob.ns = nso;
    ob.a.assign((size_t)nso * nso, 0.0);
    ob.eps.assign(nso, 0.0);
    for (int kp = 0; kp < nbath; kp++) {
      const double v = ix.bath(m.bv, s, a, kp);
      ob.a[0 * nso + (1 + kp)] += v;
      ob.a[(1 + kp) * nso + 0] += v;
      ob.eps[1 + kp] = ix.bath(m.be, s, a, kp);
    }
    hop_csr(ob, b, out.fac[k]);
    out.eax[k].resize(d);
    out.impbit[k].resize(d);
    for (int64_t i = 0; i < d; i++) {
      const uint32_t st = (uint32_t)b.states[i];
      double x = 0.0;
      for (int p = 1; p < nso; p++)
        if ((st >> p) & 1u) x += ob.eps[p];
      out.eax[k][i] = x;
      out.impbit[k][i] = (uint8_t)(st & 1u);
    }
  }
  // impurity part of the diagonal as a table over the 2*Norb impurity occupations (bit k = axis k)
  out.xtab.assign((size_t)1 << out.naxes, 0.0);
  for (uint32_t bits = 0; bits < (1u << out.naxes); bits++) {
    double x = 0.0;
    auto nu = [&](int a) { return (double)((bits >> a) & 1u); };
    auto nd = [&](int a) { return (double)((bits >> (norb + a)) & 1u); };
    for (int a = 0; a < norb; a++) {
      x += ix.hloc(0, 0, a, a).real() * nu(a) + ix.hloc(sd, sd, a, a).real() * nd(a) - m.xmu * (nu(a) + nd(a));
      x += m.spin_field[a * 3 + 2] * (nu(a) - nd(a));  // stored/Orbs/H_local.f90:19-24
      x += ix.uloc(a) * nu(a) * nd(a);
      for (int b = a + 1; b < norb; b++) {
        x += ix.ust(a, b) * (nu(a) * nd(b) + nu(b) * nd(a));
        x += (ix.ust(a, b) - ix.jh(a, b)) * (nu(a) * nu(b) + nd(a) * nd(b));
      }
    }
    if (m.hfmode) {
      for (int a = 0; a < norb; a++) x += -0.5 * ix.uloc(a) * (nu(a) + nd(a)) + 0.25 * ix.uloc(a);
      for (int a = 0; a < norb; a++)
        for (int b = a + 1; b < norb; b++) {
          // NB: 0.25 per pair here, 0.5 in the ed_total_ud=T builder -- as in the reference
          // (stored/Orbs/H_local.f90:61-62 vs stored/H_local.f90:63-64)
          const double ust = ix.ust(a, b), ujj = ust - ix.jh(a, b), nn = nu(a) + nd(a) + nu(b) + nd(b);
          x += -0.5 * ust * nn + 0.25 * ust - 0.5 * ujj * nn + 0.25 * ujj;
        }
    }
    out.xtab[bits] = x;
  }
  if (explicit_diag) {
    out.hd.resize((size_t)out.dim);
    std::vector<int64_t> idx(out.naxes, 0);
    for (int64_t i = 0; i < out.dim; i++) {
      double x = 0.0;
      uint32_t bits = 0;
      for (int k = 0; k < out.naxes; k++) {
        x += out.eax[k][idx[k]];
        bits |= (uint32_t)out.impbit[k][idx[k]] << k;
      }
      out.hd[i] = x + out.xtab[bits];
      for (int k = 0; k < out.naxes; k++) {
        if (++idx[k] < out.dims[k]) break;
        idx[k] = 0;
      }
    }
  }
  return "";
}

int twojz_of_level(int p, int ns, int norb) {
  static const int lzdiag[3] = {-1, +1, 0};
  return (p < ns ? 1 : -1) + 2 * lzdiag[(p % ns) % norb];
}

std::string build_direct(const edigpu_model& m, int sector, int64_t row_first, int64_t row_count,
                         HostDirect& out, bool jz_basis, int twojz) {
  std::string e = check_model(m);
  if (!e.empty()) return e;
  if (m.ed_mode != 1 && m.ed_mode != 2) return "edigpu_direct_build: model.ed_mode must be superc or nonsu2";
  if (m.ed_mode == 2 && m.nspin != 2) return "edigpu_direct_build: nonsu2 needs nspin=2";
  e = check_flat_fields(m, "edigpu_direct_build");
  if (!e.empty()) return e;
  Idx ix(m);
  const int ns = model_ns(m), norb = m.norb;
  if (2 * ns > 30) return "edigpu_direct_build: 2*Ns > 30 bits (the reference's integer range)";
  SpinBasis sb;
  if (jz_basis) {
    if (m.ed_mode != 2 || norb != 3) return "Jz_basis needs ed_mode = nonsu2 and Norb = 3 (Lzdiag = [-1, +1, 0])";
    if (m.nbath > 1 && m.bath_type != 2 && m.bath_type != 3)
      return "Jz_basis labels the levels as iorb + Norb * ibath: replica / general bath (or Nbath = 1)";
    if (m.nph > 0) return "phonon sectors are not built in the Jz basis";
    if (sector < 0 || sector > 2 * ns) return "Jz_basis: bad sector";
    sb.init_jz(ns, norb, sector, twojz);
  } else {
    sb.init(ns, m.ed_mode, sector);
  }
  out.ns = ns;
  out.norb = norb;
  out.dim = (int64_t)sb.states.size();
  if (out.dim == 0) return "edigpu_direct_build: empty sector";
  if (row_count < 0) {
    row_first = 0;
    row_count = out.dim;
  }
  if (row_first < 0 || row_first + row_count > out.dim) return "edigpu_direct_build: bad shard";
  out.row_first = row_first;
  out.row_count = row_count;
  out.states.assign(sb.states.begin() + row_first, sb.states.begin() + row_first + row_count);
  out.off_dw = sb.off_dw;
  out.rk_up = sb.rk_up;

  std::vector<OpTerm> terms;
  std::vector<double> eps;
  double cst = 0.0;
  flat_physics(m, terms, eps, cst);
  out.terms.clear();
  for (const OpTerm& t : terms) {
    DirectTerm d{};
    uint32_t flipped = 0, touched = 0;
    int cs = 0;
    bool zero = false;
    for (int k = 0; k < t.n && !zero; k++) {
      const uint32_t b = 1u << t.pos[k];
      if (touched & b) {
        // a level met again (coulomb_sundry lines such as n_i c^+_j c_k): its occupation is known from the operators
        // before -- set iff (it had to be set) xor (it has been flipped)
        const bool occ = ((d.need_set & b) != 0) != ((flipped & b) != 0);
        if (occ == t.create[k]) zero = true;  // c^+ on a filled / c on an empty level: the line is identically zero
      } else if (t.create[k]) {
        d.need_clear |= b;
      } else {
        d.need_set |= b;
      }
      touched |= b;
      // popc((s ^ flipped) & below) = popc(s & below) + popc(flipped & below)  (mod 2)
      d.sign_mask ^= (b - 1u);
      cs ^= popc(flipped & (b - 1u)) & 1;
      flipped ^= b;
    }
    if (zero) continue;
    d.flip = flipped;
    d.csign = cs;
    d.cre = t.coef.real();
    d.cim = t.coef.imag();
    if (jz_basis) {
      // the two-table rank is only defined inside the sector: a term that changes twoJz would be ranked to a wrong
      // row silently (the reference's binary_search fails on it)
      int dj = 0;
      for (int b = 0; b < 2 * ns; b++) {
        if (((flipped & d.need_clear) >> b) & 1u) dj += twojz_of_level(b, ns, norb);
        if (((flipped & d.need_set) >> b) & 1u) dj -= twojz_of_level(b, ns, norb);
      }
      if (dj != 0 && (d.cre != 0.0 || d.cim != 0.0))
        return "a matrix element leaves the sector (Jz_basis: the model does not conserve Jz)";
    }
    out.terms.push_back(d);
  }
  // Order the terms by where they lead: a term takes state s to s + delta with delta = sum(+2^b over the levels it
  // fills) - sum(2^b over the levels it empties), and ranks grow with the state.  A lane pops its applicable terms in
  // list order, so with ascending delta the lanes of a wave (consecutive rows) walk their partners in ascending
  // column order together -- what the column-sorted rows of the stored SELL image give that kernel.
  // EDIGPU_DIRECT_NOSORT=1 keeps the order of the operator list.
  if (!getenv("EDIGPU_DIRECT_NOSORT")) {
    auto delta = [](const DirectTerm& t) {
      int64_t d = 0;
      for (int b = 0; b < 32; b++) {
        if ((t.flip & t.need_clear) >> b & 1u) d += (int64_t)1 << b;
        if ((t.flip & t.need_set) >> b & 1u) d -= (int64_t)1 << b;
      }
      return d;
    };
    std::stable_sort(out.terms.begin(), out.terms.end(),
                     [&](const DirectTerm& x, const DirectTerm& y) { return delta(x) < delta(y); });
  }
  // (an earlier version merged every hop with its reverse into one "pair" term; with the two-instruction
  // applicability test of the kernels a plain term list is faster -- see kernels_direct.hip)
  // diagonal: byte-wise sums of the one-body energies + impurity interaction table
  out.dtab.assign(4 * 256, 0.0);
  for (int byte = 0; byte < 4; byte++)
    for (int v = 0; v < 256; v++) {
      double x = 0.0;
      for (int bit = 0; bit < 8; bit++) {
        const int p = byte * 8 + bit;
        if (p < 2 * ns && ((v >> bit) & 1)) x += eps[p];
      }
      out.dtab[byte * 256 + v] = x;
    }
  const uint32_t impmask = (1u << norb) - 1u;
  out.xtab.assign((size_t)1 << (2 * norb), 0.0);
  for (uint32_t iu = 0; iu <= impmask; iu++)
    for (uint32_t id = 0; id <= impmask; id++) {
      double x = cst;
      for (int a = 0; a < norb; a++) {
        const int nu = (iu >> a) & 1, nd = (id >> a) & 1;
        x += ix.uloc(a) * nu * nd;
        for (int b = a + 1; b < norb; b++) {
          const int nub = (iu >> b) & 1, ndb = (id >> b) & 1;
          x += ix.ust(a, b) * (nu * ndb + nub * nd);
          x += (ix.ust(a, b) - ix.jh(a, b)) * (nu * nub + nd * ndb);
        }
      }
      out.xtab[(id << norb) | iu] = x;
    }
  return "";
}

}  // namespace edigpu

// --------------------------------------------------------------------------------------
// Hand-over images (edigpu_normal_create): recover the factored tables from the arrays the reference built
// --------------------------------------------------------------------------------------
namespace edigpu {

// spH0d as Hd(iup, idw) = P_c(idw)(iup) + Hd(0, idw): the rows of one class c share the profile P_c(iup) = Hd(iup, r) -
// Hd(0, r) (in an impurity model: rows with equal impurity occupations of the down word).  Accepted when every local
// element is reproduced within `tol`.  Tables of rows outside the shard stay zero.
static bool factor_diagonal(int64_t du, int64_t dd, int64_t dw_first, int64_t dw_count, const double* hd, double tol,
                            HostFactored& fac) {
  const int kMaxClasses = 64;
  std::vector<int64_t> rep;  // class -> local row of its representative
  fac.ed.assign((size_t)dd, 0.0);
  fac.impd.assign((size_t)dd, 0);
  int64_t samples[4] = {du / 5, (2 * du) / 5 + (du > 1), (3 * du) / 4, du - 1};
  for (int64_t r = 0; r < dw_count; r++) {
    const double* row = hd + r * du;
    int cls = -1;
    for (size_t c = 0; c < rep.size() && cls < 0; c++) {
      const double* pr = hd + rep[c] * du;
      bool ok = true;
      for (int k = 0; k < 4 && ok; k++) {
        const int64_t i = std::min(samples[k], du - 1);
        ok = std::fabs((row[i] - row[0]) - (pr[i] - pr[0])) <= tol;
      }
      for (int64_t i = 0; i < du && ok; i++) ok = std::fabs((row[i] - row[0]) - (pr[i] - pr[0])) <= tol;
      if (ok) cls = (int)c;
    }
    if (cls < 0) {
      if ((int)rep.size() == kMaxClasses) return false;
      cls = (int)rep.size();
      rep.push_back(r);
    }
    fac.ed[(size_t)(dw_first + r)] = row[0];
    fac.impd[(size_t)(dw_first + r)] = (uint8_t)cls;
  }
  fac.nimp = std::max<int>(1, (int)rep.size());
  fac.eux.assign((size_t)fac.nimp * du, 0.0);
  for (size_t c = 0; c < rep.size(); c++) {
    const double* pr = hd + rep[c] * du;
    for (int64_t i = 0; i < du; i++) fac.eux[c * du + i] = pr[i] - pr[0];
  }
  return true;
}

// spH0nd (local rows, global columns) as sum_t coef_t (Pdw_t (x) Pup_t).  For every (idw -> jdw) pair the entries form
// an operator on the up index; the distinct ones (up to a sign) are few, each is cut into partial maps of one magnitude
// and the terms are (operator, map, occurrence of that operator among the pairs of one idw).  The result is checked
// against the CSR entry by entry (exact: a term contributes +/- coef_t), so a matrix that is not of this form only
// costs the attempt.
static bool factor_nonlocal(int64_t du, int64_t dd, int64_t dw_first, int64_t dw_count, const int64_t* rp,
                            const int32_t* col, const double* val, int max_terms, HostFactored& fac) {
  struct Ent { int32_t iup, jup; double v; };
  struct UpOp { std::vector<Ent> e; };
  std::vector<UpOp> ops;
  std::unordered_multimap<uint64_t, int> op_index;
  struct Pair { int32_t jdw; int op; bool neg; };
  std::vector<std::vector<Pair>> pairs((size_t)dw_count);
  std::vector<std::pair<int32_t, std::vector<Ent>>> groups;  // of the current dw row
  for (int64_t r = 0; r < dw_count; r++) {
    groups.clear();
    for (int64_t iup = 0; iup < du; iup++) {
      const int64_t i = iup + r * du;
      for (int64_t k = rp[i]; k < rp[i + 1]; k++) {
        const int32_t jdw = (int32_t)(col[k] / du), jup = (int32_t)(col[k] % du);
        size_t g = 0;
        while (g < groups.size() && groups[g].first != jdw) g++;
        if (g == groups.size()) {
          if (groups.size() >= 64) return false;
          groups.push_back({jdw, {}});
        }
        groups[g].second.push_back({(int32_t)iup, jup, val[k]});
      }
    }
    for (auto& g : groups) {
      std::vector<Ent>& e = g.second;
      std::sort(e.begin(), e.end(), [](const Ent& a, const Ent& b) { return a.iup != b.iup ? a.iup < b.iup : a.jup < b.jup; });
      for (size_t k = 1; k < e.size(); k++)
        if (e[k].iup == e[k - 1].iup && e[k].jup == e[k - 1].jup) return false;  // duplicate entry
      if (e.empty() || e[0].v == 0.0) return false;
      const bool neg = e[0].v < 0.0;
      uint64_t hsh = 1469598103934665603ull;
      for (Ent& x : e) {
        if (neg) x.v = -x.v;
        uint64_t bits;
        std::memcpy(&bits, &x.v, 8);
        for (uint64_t w : {(uint64_t)(uint32_t)x.iup, (uint64_t)(uint32_t)x.jup, bits}) hsh = (hsh ^ w) * 1099511628211ull;
      }
      int found = -1;
      auto range = op_index.equal_range(hsh);
      for (auto it = range.first; it != range.second && found < 0; ++it) {
        const std::vector<Ent>& o = ops[it->second].e;
        if (o.size() == e.size() &&
            std::equal(o.begin(), o.end(), e.begin(),
                       [](const Ent& a, const Ent& b) { return a.iup == b.iup && a.jup == b.jup && a.v == b.v; }))
          found = it->second;
      }
      if (found < 0) {
        if ((int)ops.size() >= max_terms) return false;
        found = (int)ops.size();
        ops.push_back({e});
        op_index.insert({hsh, found});
      }
      pairs[r].push_back({g.first, found, neg});
    }
  }
  // partial maps of one magnitude per operator
  struct Sub { double mag; std::vector<uint32_t> jup; };
  std::vector<std::vector<Sub>> subs(ops.size());
  for (size_t o = 0; o < ops.size(); o++) {
    const std::vector<Ent>& e = ops[o].e;
    for (size_t k = 0; k < e.size();) {
      size_t k1 = k;
      while (k1 < e.size() && e[k1].iup == e[k].iup) k1++;
      for (size_t q = k; q < k1; q++) {  // the entries of row iup: each goes to a map of its magnitude that is still free here
        const double mag = std::fabs(e[q].v);
        size_t sidx = 0;
        for (; sidx < subs[o].size(); sidx++)
          if (subs[o][sidx].mag == mag && subs[o][sidx].jup[e[q].iup] == 0xFFFFFFFFu) break;
        if (sidx == subs[o].size()) {
          if ((int)subs[o].size() >= max_terms) return false;
          subs[o].push_back({mag, std::vector<uint32_t>((size_t)du, 0xFFFFFFFFu)});
        }
        subs[o][sidx].jup[e[q].iup] = (uint32_t)e[q].jup | (e[q].v < 0.0 ? 0x80000000u : 0u);
      }
      k = k1;
    }
  }
  // occurrences of one operator among the pairs of a dw row
  std::vector<int> nslots(ops.size(), 0);
  for (int64_t r = 0; r < dw_count; r++) {
    std::vector<int> cnt(ops.size(), 0);
    for (const Pair& p : pairs[r]) nslots[p.op] = std::max(nslots[p.op], ++cnt[p.op]);
  }
  int nterms = 0;
  std::vector<std::vector<int>> first_term(ops.size());  // [op][slot] -> first term (one per sub map)
  for (size_t o = 0; o < ops.size(); o++)
    for (int sl = 0; sl < nslots[o]; sl++) {
      first_term[o].push_back(nterms);
      nterms += (int)subs[o].size();
    }
  if (nterms > max_terms) return false;
  fac.nterms = nterms;
  fac.coef.assign((size_t)nterms, 0.0);
  fac.jup.assign((size_t)nterms * du, 0xFFFFFFFFu);
  fac.jdw.assign((size_t)nterms * dd, 0xFFFFFFFFu);
  for (size_t o = 0; o < ops.size(); o++)
    for (int sl = 0; sl < nslots[o]; sl++)
      for (size_t sb = 0; sb < subs[o].size(); sb++) {
        const int t = first_term[o][sl] + (int)sb;
        fac.coef[t] = subs[o][sb].mag;
        std::copy(subs[o][sb].jup.begin(), subs[o][sb].jup.end(), fac.jup.begin() + (size_t)t * du);
      }
  for (int64_t r = 0; r < dw_count; r++) {
    std::vector<int> cnt(ops.size(), 0);
    for (const Pair& p : pairs[r]) {
      const int sl = cnt[p.op]++;
      for (size_t sb = 0; sb < subs[p.op].size(); sb++)
        fac.jdw[(size_t)(first_term[p.op][sl] + (int)sb) * dd + dw_first + r] = (uint32_t)p.jdw | (p.neg ? 0x80000000u : 0u);
    }
  }
  // entry-by-entry check
  std::vector<std::pair<int64_t, double>> want, got;
  for (int64_t r = 0; r < dw_count; r++)
    for (int64_t iup = 0; iup < du; iup++) {
      const int64_t i = iup + r * du;
      want.clear();
      got.clear();
      for (int64_t k = rp[i]; k < rp[i + 1]; k++) want.push_back({col[k], val[k]});
      for (int t = 0; t < nterms; t++) {
        const uint32_t pu = fac.jup[(size_t)t * du + iup], pd = fac.jdw[(size_t)t * dd + dw_first + r];
        if (pu == 0xFFFFFFFFu || pd == 0xFFFFFFFFu) continue;
        got.push_back({(int64_t)(pu & 0x7FFFFFFFu) + (int64_t)(pd & 0x7FFFFFFFu) * du,
                       ((pu ^ pd) & 0x80000000u) ? -fac.coef[t] : fac.coef[t]});
      }
      if (want.size() != got.size()) return false;
      std::sort(want.begin(), want.end());
      std::sort(got.begin(), got.end());
      if (want != got) return false;
    }
  return true;
}

bool factor_handover(int64_t dim_up, int64_t dim_dw, int64_t dw_first, int64_t dw_count, const double* hd,
                     const int64_t* nd_rowptr, const int32_t* nd_col, const double* nd_val, int max_terms,
                     HostFactored& fac) {
  fac = HostFactored();
  if (dw_count <= 0 || dim_dw >= ((int64_t)1 << 31) || dim_up >= ((int64_t)1 << 31)) return false;
  double amax = 0.0;
  const int64_t nloc = dim_up * dw_count;
  for (int64_t i = 0; i < nloc; i++) amax = std::max(amax, std::fabs(hd[i]));
  const double tol = 8.0 * 2.220446049250313e-16 * amax;
  if (!factor_diagonal(dim_up, dim_dw, dw_first, dw_count, hd, tol, fac)) return false;
  if (nd_rowptr && nd_rowptr[nloc] > 0 &&
      !factor_nonlocal(dim_up, dim_dw, dw_first, dw_count, nd_rowptr, nd_col, nd_val, max_terms, fac))
    return false;
  fac.valid = true;
  return true;
}

}  // namespace edigpu

// --------------------------------------------------------------------------------------
// _CMPLX_NORMAL as ONE real sector on a doubled up index
// --------------------------------------------------------------------------------------
namespace edigpu {

// H = S + iA (S real symmetric: the ordinary build of the real parts; A real antisymmetric: the hop matrices of the
// imaginary parts of impHloc / the replica matrices).  On interleaved complex vectors, read as real arrays with the up
// index doubled, u' = 2 iup + c (c = 0 real part, 1 imaginary part), H is the REAL symmetric operator
//   H'[(i,c),(j,c')] = S[i,j] delta(c,c') + A[i,j] eps(c,c'),   eps = [[0,-1],[1,0]]
// (real part of H z: S xr - A xi, imaginary part: S xi + A xr).  Its up factor is S_up (x) 1 + A_up (x) eps, an ordinary
// sparse matrix on 2 DimUp columns; its down factor is S_dw; the diagonal tables and the factored Hnd terms are those of
// S with every up index doubled; and (A_dw (x) eps), the down hops with imaginary amplitudes, is a handful of further
// factored terms (a partial map of the down index per magnitude and hop (x) the signed swap of the two components).  The
// kernels of the real sector then compute the complex product in one pass over 2 Dim elements -- about twice a real
// product instead of four real products and two layout passes.
std::string build_normal_doubled(const edigpu_model& m, int nup, int ndw, HostNormal& out, int max_terms) {
  HostNormal hs, ha;
  std::string e = build_normal(m, nup, ndw, 0, -1, hs, false);
  if (!e.empty()) return e;
  bool any = false;
  const edigpu_model mi = imag_part_model(m, any);
  if (any) {
    e = build_normal(mi, nup, ndw, 0, -1, ha, false);
    if (!e.empty()) return e;
  }
  const int64_t du = hs.dim_up, dd = hs.dim_dw, du2 = 2 * du;
  if (du2 * dd >= ((int64_t)1 << 31)) return "complex sector: 2 Dim >= 2^31";
  out = HostNormal();
  out.ns = hs.ns;
  out.nup = nup;
  out.ndw = ndw;
  out.dim_up = du2;
  out.dim_dw = dd;
  out.dw_first = 0;
  out.dw_count = dd;
  // ---- up factor ----
  HostCsr& up = out.up;
  up.nrow = up.ncol = du2;
  up.rowptr.assign(du2 + 1, 0);
  for (int64_t i = 0; i < du; i++)
    for (int c = 0; c < 2; c++) {
      for (int64_t k = hs.up.rowptr[i]; k < hs.up.rowptr[i + 1]; k++) {
        up.col.push_back((int32_t)(2 * hs.up.col[k] + c));
        up.val.push_back(hs.up.val[k]);
      }
      if (any)
        for (int64_t k = ha.up.rowptr[i]; k < ha.up.rowptr[i + 1]; k++) {
          up.col.push_back((int32_t)(2 * ha.up.col[k] + (1 - c)));
          up.val.push_back(c == 0 ? -ha.up.val[k] : ha.up.val[k]);
        }
      up.rowptr[2 * i + c + 1] = (int64_t)up.col.size();
    }
  out.dw = hs.dw;
  // ---- diagonal and the Hnd terms of S, up index doubled ----
  HostFactored& f = out.fac;
  const HostFactored& fs = hs.fac;
  f = HostFactored();
  f.valid = true;
  f.nimp = fs.nimp;
  f.ed = fs.ed;
  f.impd = fs.impd;
  f.eux.resize((size_t)fs.nimp * du2);
  for (int c0 = 0; c0 < fs.nimp; c0++)
    for (int64_t i = 0; i < du; i++) {
      f.eux[(size_t)c0 * du2 + 2 * i] = fs.eux[(size_t)c0 * du + i];
      f.eux[(size_t)c0 * du2 + 2 * i + 1] = fs.eux[(size_t)c0 * du + i];
    }
  f.nterms = fs.nterms;
  f.coef = fs.coef;
  f.jdw = fs.jdw;
  f.jup.assign((size_t)fs.nterms * du2, 0xFFFFFFFFu);
  for (int t = 0; t < fs.nterms; t++)
    for (int64_t i = 0; i < du; i++) {
      const uint32_t j = fs.jup[(size_t)t * du + i];
      if (j == 0xFFFFFFFFu) continue;
      for (int c = 0; c < 2; c++)
        f.jup[(size_t)t * du2 + 2 * i + c] = (2u * (j & 0x7FFFFFFFu) + (uint32_t)c) | (j & 0x80000000u);
    }
  // ---- A_dw (x) eps: partial maps of the down index, one magnitude each ----
  if (any) {
    struct Sub { double mag; std::vector<uint32_t> jdw; };
    std::vector<Sub> subs;
    for (int64_t r = 0; r < dd; r++)
      for (int64_t k = ha.dw.rowptr[r]; k < ha.dw.rowptr[r + 1]; k++) {
        const double v = ha.dw.val[k], mag = std::fabs(v);
        if (mag == 0.0) continue;
        size_t sidx = 0;
        for (; sidx < subs.size(); sidx++)
          if (subs[sidx].mag == mag && subs[sidx].jdw[(size_t)r] == 0xFFFFFFFFu) break;
        if (sidx == subs.size()) subs.push_back({mag, std::vector<uint32_t>((size_t)dd, 0xFFFFFFFFu)});
        subs[sidx].jdw[(size_t)r] = (uint32_t)ha.dw.col[k] | (v < 0.0 ? 0x80000000u : 0u);
      }
    for (const Sub& sb : subs) {
      f.coef.push_back(sb.mag);
      f.jdw.insert(f.jdw.end(), sb.jdw.begin(), sb.jdw.end());
      const size_t base = f.jup.size();
      f.jup.resize(base + (size_t)du2);
      for (int64_t i = 0; i < du; i++) {
        f.jup[base + 2 * i] = (uint32_t)(2 * i + 1) | 0x80000000u;  // real part <- - a * imaginary part
        f.jup[base + 2 * i + 1] = (uint32_t)(2 * i);                 // imaginary part <- + a * real part
      }
      f.nterms++;
    }
  }
  if (f.nterms > max_terms) return "complex sector: too many factored terms for the doubled real image";
  out.has_nd = f.nterms > 0;
  out.nd_nnz = 0;
  for (int t = 0; t < f.nterms; t++) {
    int64_t nu = 0, nd = 0;
    for (int64_t i = 0; i < du2; i++) nu += f.jup[(size_t)t * du2 + i] != 0xFFFFFFFFu;
    for (int64_t i = 0; i < dd; i++) nd += f.jdw[(size_t)t * dd + i] != 0xFFFFFFFFu;
    out.nd_nnz += nu * nd;
  }
  out.has_nd = out.nd_nnz > 0;
  return "";
}

}  // namespace edigpu
