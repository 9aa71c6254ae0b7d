// CPU-side sanitizer harness (ASan + UBSan) for the host builders of libedigpu (csrc/host_build.cpp): every bath type,
// mode and sector of a few small models, shards included.  Built and run by tests/test_host_sanitizers.py.
#include <cstdio>
#include <cstring>
#include <random>
#include "host_build.hpp"
#include "host_ib.hpp"
using namespace edigpu;
static void fill(edigpu_model& m, int mode, int bath, int norb, int nbath) {
  memset(&m, 0, sizeof(m));
  m.ed_mode = mode; m.bath_type = bath; m.norb = norb; m.nbath = nbath; m.nspin = mode == 2 ? 2 : 1; m.hfmode = 1;
  std::mt19937 g(7); std::uniform_real_distribution<double> u(0.1, 0.6), e(-2, 2);
  for (int a = 0; a < norb; a++) { m.uloc[a] = 2.0; for (int b = 0; b < norb; b++) if (a != b) { m.ust[a*EDIGPU_MAXORB+b] = 1.5; m.jh[a*EDIGPU_MAXORB+b] = 0.25; m.jx[a*EDIGPU_MAXORB+b] = bath==0&&mode==0? 0.25:0.25; m.jp[a*EDIGPU_MAXORB+b] = 0.25; } }
  for (int s = 0; s < 2; s++) for (int a = 0; a < norb; a++) for (int k = 0; k < nbath; k++) {
    m.bv[(s*EDIGPU_MAXORB+a)*EDIGPU_MAXBATH+k] = u(g); m.be[(s*EDIGPU_MAXORB+a)*EDIGPU_MAXBATH+k] = e(g);
    m.bd[(s*EDIGPU_MAXORB+a)*EDIGPU_MAXBATH+k] = 0.05; m.bu[(s*EDIGPU_MAXORB+a)*EDIGPU_MAXBATH+k] = 0.1; }
  for (int is = 0; is < 2; is++) for (int js = 0; js < 2; js++) for (int a = 0; a < norb; a++) for (int b = 0; b < norb; b++) for (int k = 0; k < nbath; k++) {
    double* p = &m.hb[(((((is*2)+js)*EDIGPU_MAXORB+a)*EDIGPU_MAXORB+b)*EDIGPU_MAXBATH+k)*2];
    p[0] = (a==b && is==js) ? e(g) : 0.1; p[1] = 0.0; }
}
static bool mode_jz(int bath, int norb, int nbath) { return norb == 3 && (bath >= 2 || nbath == 1) && 2 * (bath == 1 ? nbath + norb : (nbath + 1) * norb) <= 14; }
int main() {
  int nfail = 0, nib = 0, njz = 0;
  for (int bath = 0; bath < 4; bath++) for (int norb = 1; norb <= 3; norb++) for (int nbath = 1; nbath <= 3; nbath++) {
    edigpu_model m; fill(m, 0, bath, norb, nbath);
    int ns = model_ns(m);
    for (int nup = 0; nup <= ns; nup += 1) for (int ndw = 0; ndw <= ns; ndw += 2) {
      HostNormal hn; std::string e = build_normal(m, nup, ndw, 0, -1, hn, (nup + ndw) % 3 == 0);
      if (!e.empty()) { printf("normal %d %d %d (%d,%d): %s\n", bath, norb, nbath, nup, ndw, e.c_str()); nfail++; }
      if (hn.dim_dw > 2) { HostNormal h2; e = build_normal(m, nup, ndw, 1, hn.dim_dw - 2, h2, true); if (!e.empty()) nfail++; }
      // the impurity-block image (csrc/host_ib.cpp): whole rows, rows in two halves, small chunks; a refusal is fine
      if (e.empty() && hn.dim_up > 0 && hn.dim_dw > 0) {
        HostNormal hf2; std::string e3 = build_normal(m, nup, ndw, 0, -1, hf2, true);
        if (e3.empty()) { HostIb a, b, c; build_ib(hf2, 480, a); build_ib(hf2, 6, b, -1); build_ib(hf2, 24, c, 64); nib += a.valid + b.valid + c.valid; }
      }
    }
    for (int mode = 1; mode <= 2; mode++) {
      fill(m, mode, bath, norb, nbath);
      if (2 * ns > 14) continue;
      for (int sec = (mode == 1 ? -ns : 0); sec <= (mode == 1 ? ns : 2 * ns); sec++) {
        HostFlat hf; std::string e = build_flat(m, sec, 0, -1, hf);
        HostDirect hd; std::string e2 = build_direct(m, sec, 0, -1, hd);
        if (!e.empty() || !e2.empty()) { printf("flat %d %d %d %d sec %d: %s %s\n", mode, bath, norb, nbath, sec, e.c_str(), e2.c_str()); nfail++; }
        if (hf.dim != hd.dim) nfail++;
        if (hf.dim > 3) { HostFlat h2; build_flat(m, sec, 1, hf.dim - 2, h2); HostDirect d2; build_direct(m, sec, 2, hf.dim - 3, d2); }
      }
    }
    if (mode_jz(bath, norb, nbath)) {
      // Jz_basis sectors: the map and the on-the-fly description (two-table rank inside (occupation, Lz) classes); this
      // model does not conserve Jz, so build_direct must refuse it after building the tables
      fill(m, 2, bath, norb, nbath);
      for (int ntot = 0; ntot <= 2 * ns; ntot += 3) for (int tj = -3; tj <= 3; tj += 2) {
        std::vector<int32_t> st; std::string e = sector_map_jz(m, ntot, tj, st);
        if (!e.empty()) { printf("jz map: %s\n", e.c_str()); nfail++; }
        HostDirect hd; e = build_direct(m, ntot, 0, -1, hd, true, tj);
        njz += (int)st.size() > 0;
        if (e.empty() && hd.dim != (int64_t)st.size()) nfail++;
      }
    }
    if (bath == 0) {
      fill(m, 0, 0, norb, nbath);
      int nups[3] = {1, 0, nbath + 1}, ndws[3] = {0, 1, 1};
      HostOrbs ho; std::string e = build_orbs(m, nups, ndws, ho, true);
      if (!e.empty()) { printf("orbs: %s\n", e.c_str()); nfail++; }
    }
  }
  printf("host builders under sanitizers: %d failures (%d impurity-block images, %d Jz sectors)\n", nfail, nib, njz);
  if (nib == 0 || njz == 0) nfail++;
  return nfail != 0;
}
