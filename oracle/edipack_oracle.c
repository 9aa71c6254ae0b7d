/*
 * edipack_oracle.c -- TEST INFRASTRUCTURE ONLY (see edipack_oracle.h).
 *
 * Plain-C restatement of the EDIpack H*v hot path, written to follow the
 * reference loop by loop so that it can serve as the parity oracle for the HIP
 * product path.  It is deliberately simple and slow (linear duplicate scans,
 * O(pos) fermionic sign loops, recursive-style binary search): it restates the
 * reference algorithm, it is not the product.
 *
 * Paths cited are relative to /root/reference/src/singlesite.
 */
#include "edipack_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small helpers                                                       */
/* ------------------------------------------------------------------ */

static void *xcalloc(size_t n, size_t s) {
  void *p = calloc(n ? n : 1, s);
  if (!p) {
    fprintf(stderr, "edipack_oracle: out of memory\n");
    abort();
  }
  return p;
}

/* ED_SETUP.f90:118-126 */
int orc_ns(const orc_model *m) {
  switch (m->bath_type) {
    case 1: return m->nbath + m->norb;
    default: return (m->nbath + 1) * m->norb;
  }
}

/* ED_SETUP.f90:1074-1092 (floating point product, rounded) */
int64_t orc_binomial(int n1, int n2) {
  double xh = 1.0;
  if (n2 < 0) return 0;
  if (n2 == 0) return 1;
  for (int i = 1; i <= n2; i++) xh = xh * (double)(n1 + 1 - i) / (double)i;
  return (int64_t)(xh + 0.5);
}

/* ED_SETUP.f90:605-622 getBathStride(iorb,kp), 1-based site position */
int orc_bath_stride(const orc_model *m, int iorb, int kp) {
  switch (m->bath_type) {
    case 1: return m->norb + kp;
    case 2:
    case 3: return iorb + kp * m->norb;
    default: return m->norb + (iorb - 1) * m->nbath + kp;
  }
}

/* ED_AUX_FUNX.f90:334-358: destruction operator at 1-based pos, sign = parity of
 * occupied levels below pos, computed with the reference's O(pos) loop. */
int orc_c(int pos, int32_t in, int32_t *out, double *sgn) {
  if (!((in >> (pos - 1)) & 1)) return 1;
  double f = 1.0;
  for (int l = 1; l <= pos - 1; l++)
    if ((in >> (l - 1)) & 1) f = -f;
  *sgn = f;
  *out = in & ~((int32_t)1 << (pos - 1));
  return 0;
}

/* ED_AUX_FUNX.f90:360-384 */
int orc_cdg(int pos, int32_t in, int32_t *out, double *sgn) {
  if ((in >> (pos - 1)) & 1) return 1;
  double f = 1.0;
  for (int l = 1; l <= pos - 1; l++)
    if ((in >> (l - 1)) & 1) f = -f;
  *sgn = f;
  *out = in | ((int32_t)1 << (pos - 1));
  return 0;
}

/* ED_AUX_FUNX.f90:463-480: recursive bisection on a sorted array, 1-based
 * result, 0 when absent. */
int64_t orc_binary_search(const int32_t *a, int64_t n, int32_t value) {
  if (n == 0) return 0;
  int64_t mid = n / 2 + 1; /* 1-based */
  if (a[mid - 1] > value) return orc_binary_search(a, mid - 1, value);
  if (a[mid - 1] < value) {
    int64_t r = orc_binary_search(a + mid, n - mid, value);
    return r ? mid + r : 0;
  }
  return mid;
}

static int popcnt32(uint32_t x) { return __builtin_popcount(x); }

/* ------------------------------------------------------------------ */
/* sectors: ED_SECTOR.f90:165-373                                      */
/* ------------------------------------------------------------------ */

/* normal mode, ed_total_ud=T: ED_SECTOR.f90:217-242 */
int orc_build_sector_normal(int ns, int nup, int ndw, int32_t *mapup, int32_t *mapdw) {
  int64_t dim = 0;
  for (int32_t iup = 0; iup < ((int32_t)1 << ns); iup++) {
    if (popcnt32((uint32_t)iup) != nup) continue;
    mapup[dim++] = iup;
  }
  dim = 0;
  for (int32_t idw = 0; idw < ((int32_t)1 << ns); idw++) {
    if (popcnt32((uint32_t)idw) != ndw) continue;
    mapdw[dim++] = idw;
  }
  return 0;
}

/* superc: ED_SECTOR.f90:263-281 (idw outer, iup inner, state = iup + idw*2**Ns) */
int64_t orc_build_sector_superc(int ns, int sz, int32_t *map) {
  int64_t dim = 0;
  for (int32_t idw = 0; idw < ((int32_t)1 << ns); idw++) {
    int ndw_ = popcnt32((uint32_t)idw);
    for (int32_t iup = 0; iup < ((int32_t)1 << ns); iup++) {
      int nup_ = popcnt32((uint32_t)iup);
      if (nup_ - ndw_ == sz) {
        if (map) map[dim] = iup + idw * ((int32_t)1 << ns);
        dim++;
      }
    }
  }
  return dim;
}

/* nonsu2, Jz_basis=F: ED_SECTOR.f90:351-368 */
int64_t orc_build_sector_nonsu2(int ns, int ntot, int32_t *map) {
  int64_t dim = 0;
  for (int32_t idw = 0; idw < ((int32_t)1 << ns); idw++) {
    int ndw_ = popcnt32((uint32_t)idw);
    for (int32_t iup = 0; iup < ((int32_t)1 << ns); iup++) {
      int nup_ = popcnt32((uint32_t)iup);
      if (nup_ + ndw_ != ntot) continue;
      if (map) map[dim] = iup + idw * ((int32_t)1 << ns);
      dim++;
    }
  }
  return dim;
}

/* nonsu2, Jz_basis=T: ED_SECTOR.f90:289-350.  The sector holds the states with Ntot = ntot and
 * twoJz = (Nup - Ndw) + twoLz, twoLz = sum over the levels iorb + Norb*ibath (ibath = 0: impurity) of
 * 2 Lzdiag(iorb) (n_up + n_dw), Lzdiag = [-1, +1, 0] (ED_VARS_GLOBAL.f90:283): three orbitals. */
int64_t orc_build_sector_nonsu2_jz(int norb, int nbath, int ntot, int twojz, int32_t *map) {
  static const int lzdiag[3] = {-1, +1, 0};
  const int ns = norb * (nbath + 1);
  if (norb != 3) return -1;
  int64_t dim = 0;
  for (int32_t idw = 0; idw < ((int32_t)1 << ns); idw++) {
    int ndw_ = popcnt32((uint32_t)idw);
    for (int32_t iup = 0; iup < ((int32_t)1 << ns); iup++) {
      int nup_ = popcnt32((uint32_t)iup);
      int nt_ = nup_ + ndw_, twosz_ = nup_ - ndw_, twolz_ = 0;
      for (int ibath = 0; ibath <= nbath; ibath++)
        for (int iorb = 1; iorb <= norb; iorb++) {
          int lev = iorb + norb * ibath; /* 1-based level of bdecomp */
          twolz_ += 2 * lzdiag[iorb - 1] * ((iup >> (lev - 1)) & 1) + 2 * lzdiag[iorb - 1] * ((idw >> (lev - 1)) & 1);
        }
      if (nt_ != ntot || twojz != twosz_ + twolz_) continue;
      if (map) map[dim] = iup + idw * ((int32_t)1 << ns);
      dim++;
    }
  }
  return dim;
}

/* ------------------------------------------------------------------ */
/* sparse container: ED_SPARSE_MATRIX.f90:16-41, :328-360              */
/* ------------------------------------------------------------------ */

/* Triplets are recorded in call order and turned into rows afterwards; a row
 * keeps first-occurrence order and later hits on the same column accumulate
 * into it, which is what sp_insert_element does one call at a time. */
typedef struct {
  int64_t n, cap;
  int is_complex;
  int64_t *row;
  int32_t *col;
  double *val; /* 1 or 2 doubles per entry */
} coo_t;

static void coo_init(coo_t *c, int is_complex) {
  memset(c, 0, sizeof(*c));
  c->is_complex = is_complex;
}

static void coo_push(coo_t *c, int64_t i, int64_t j, double re, double im) {
  if (c->n == c->cap) {
    c->cap = c->cap ? 2 * c->cap : 1024;
    c->row = realloc(c->row, c->cap * sizeof(int64_t));
    c->col = realloc(c->col, c->cap * sizeof(int32_t));
    c->val = realloc(c->val, c->cap * sizeof(double) * (c->is_complex ? 2 : 1));
    if (!c->row || !c->col || !c->val) abort();
  }
  c->row[c->n] = i;
  c->col[c->n] = (int32_t)j;
  if (c->is_complex) {
    c->val[2 * c->n] = re;
    c->val[2 * c->n + 1] = im;
  } else {
    c->val[c->n] = re;
  }
  c->n++;
}

static void coo_free(coo_t *c) {
  free(c->row);
  free(c->col);
  free(c->val);
  memset(c, 0, sizeof(*c));
}

static void csr_free(orc_csr *a) {
  free(a->rowptr);
  free(a->col);
  free(a->val);
  memset(a, 0, sizeof(*a));
}

/* sp_insert_element semantics applied to the recorded call sequence. */
static void coo_to_csr(const coo_t *c, int64_t nrow, int64_t ncol, orc_csr *a) {
  int w = c->is_complex ? 2 : 1;
  a->nrow = nrow;
  a->ncol = ncol;
  a->is_complex = c->is_complex;
  int64_t *cnt = xcalloc(nrow + 1, sizeof(int64_t));
  for (int64_t k = 0; k < c->n; k++) cnt[c->row[k] + 1]++;
  for (int64_t i = 0; i < nrow; i++) cnt[i + 1] += cnt[i];
  /* upper bound layout, then compact */
  int32_t *col = xcalloc(c->n, sizeof(int32_t));
  double *val = xcalloc(c->n * w, sizeof(double));
  int64_t *fill = xcalloc(nrow, sizeof(int64_t));
  for (int64_t k = 0; k < c->n; k++) {
    int64_t i = c->row[k], base = cnt[i];
    int64_t hit = -1;
    for (int64_t p = 0; p < fill[i]; p++)
      if (col[base + p] == c->col[k]) {
        hit = p;
        break;
      }
    if (hit >= 0) {
      for (int q = 0; q < w; q++) val[(base + hit) * w + q] += c->val[k * w + q];
    } else {
      col[base + fill[i]] = c->col[k];
      for (int q = 0; q < w; q++) val[(base + fill[i]) * w + q] = c->val[k * w + q];
      fill[i]++;
    }
  }
  a->rowptr = xcalloc(nrow + 1, sizeof(int64_t));
  for (int64_t i = 0; i < nrow; i++) a->rowptr[i + 1] = a->rowptr[i] + fill[i];
  a->nnz = a->rowptr[nrow];
  a->col = xcalloc(a->nnz, sizeof(int32_t));
  a->val = xcalloc(a->nnz * w, sizeof(double));
  for (int64_t i = 0; i < nrow; i++) {
    memcpy(a->col + a->rowptr[i], col + cnt[i], fill[i] * sizeof(int32_t));
    memcpy(a->val + a->rowptr[i] * w, val + cnt[i] * w, fill[i] * w * sizeof(double));
  }
  free(cnt);
  free(col);
  free(val);
  free(fill);
}

/* ED_SPARSE_MATRIX.f90:778-793 sp_matvec_matrix_csr_d */
void orc_csr_matvec_d(const orc_csr *a, const double *x, double *y) {
  for (int64_t i = 0; i < a->nrow; i++) {
    double s = 0.0;
    for (int64_t k = a->rowptr[i]; k < a->rowptr[i + 1]; k++) s += a->val[k] * x[a->col[k]];
    y[i] = s;
  }
}

void orc_csr_matvec_z(const orc_csr *a, const double *x, double *y) {
  for (int64_t i = 0; i < a->nrow; i++) {
    double sr = 0.0, si = 0.0;
    for (int64_t k = a->rowptr[i]; k < a->rowptr[i + 1]; k++) {
      double ar = a->val[2 * k], ai = a->val[2 * k + 1];
      double xr = x[2 * (int64_t)a->col[k]], xi = x[2 * (int64_t)a->col[k] + 1];
      sr += ar * xr - ai * xi;
      si += ar * xi + ai * xr;
    }
    y[2 * i] = sr;
    y[2 * i + 1] = si;
  }
}

/* ------------------------------------------------------------------ */
/* normal mode builder                                                 */
/* ------------------------------------------------------------------ */

/* diag_hybr / bath_diag / hbath_tmp as set at
 * ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:76-117 (real build: the real
 * part is taken by the real(8) assignment). */
static double diag_hybr(const orc_model *m, int ispin, int iorb, int kp) { /* 1-based */
  switch (m->bath_type) {
    case 2: return m->vr[kp - 1];
    case 3: return m->vg[(iorb - 1) + m->norb * (ispin - 1)][kp - 1];
    default: return m->bv[ispin - 1][iorb - 1][kp - 1];
  }
}

static int bath_diag_norb(const orc_model *m) { /* size(bath_diag,2) */
  return m->bath_type == 1 ? 1 : m->norb;
}

static double bath_diag(const orc_model *m, int ispin, int iorb, int kp) {
  switch (m->bath_type) {
    case 2:
    case 3: return m->hb_re[ispin - 1][ispin - 1][iorb - 1][iorb - 1][kp - 1];
    default: return m->be[ispin - 1][iorb - 1][kp - 1];
  }
}

static int nonloc_condition(const orc_model *m) {
  if (m->norb <= 1) return 0;
  for (int i = 0; i < m->norb; i++)
    for (int j = 0; j < m->norb; j++)
      if (m->jx[i][j] != 0.0 || m->jp[i][j] != 0.0) return 1;
  return 0;
}

/* stored/H_up.f90:1-108 and stored/H_dw.f90:1-108: one spin species.
 * spin = 1 (up) or 2 (dw); matrix element goes to (row = result state, col = source). */
static void build_h_spin(const orc_model *m, int ns, int spin, const int32_t *map, int64_t dim,
                         coo_t *coo) {
  int norb = m->norb, nbath = m->nbath;
  int isp = (spin == 1) ? 1 : m->nspin; /* impHloc(Nspin,Nspin), diag_hybr(Nspin) for dw */
  int mfs = (spin == 1) ? 0 : 1;        /* mfHloc(1,1) / mfHloc(2,2) */
  int n[32];
  for (int64_t j = 0; j < dim; j++) {
    int32_t ms = map[j];
    for (int l = 0; l < ns; l++) n[l + 1] = (ms >> l) & 1; /* bdecomp, 1-based */
    int32_t k1, k2;
    double sg1, sg2;
    /* H_imp off-diagonal: stored/H_up.f90:9-24 */
    for (int iorb = 1; iorb <= norb; iorb++)
      for (int jorb = 1; jorb <= norb; jorb++) {
        double t = m->hloc_re[isp - 1][isp - 1][iorb - 1][jorb - 1] +
                   m->mfh_re[mfs][mfs][iorb - 1][jorb - 1];
        if (t != 0.0 && n[jorb] == 1 && n[iorb] == 0) {
          orc_c(jorb, ms, &k1, &sg1);
          orc_cdg(iorb, k1, &k2, &sg2);
          int64_t i = orc_binary_search(map, dim, k2);
          coo_push(coo, i - 1, j, t * sg1 * sg2, 0.0);
        }
      }
    /* replica/general inter-orbital bath hopping: stored/H_up.f90:28-52 */
    if (m->bath_type == 2 || m->bath_type == 3) {
      for (int kp = 1; kp <= nbath; kp++)
        for (int iorb = 1; iorb <= norb; iorb++)
          for (int jorb = 1; jorb <= norb; jorb++) {
            int ialfa = orc_bath_stride(m, iorb, kp), ibeta = orc_bath_stride(m, jorb, kp);
            double t = m->hb_re[isp - 1][isp - 1][iorb - 1][jorb - 1][kp - 1];
            if (t != 0.0 && n[ibeta] == 1 && n[ialfa] == 0) {
              orc_c(ibeta, ms, &k1, &sg1);
              orc_cdg(ialfa, k1, &k2, &sg2);
              int64_t i = orc_binary_search(map, dim, k2);
              coo_push(coo, i - 1, j, t * sg1 * sg2, 0.0);
            }
          }
    }
    /* hybridisation imp <-> bath: stored/H_up.f90:56-82 */
    for (int iorb = 1; iorb <= norb; iorb++)
      for (int kp = 1; kp <= nbath; kp++) {
        int ialfa = orc_bath_stride(m, iorb, kp);
        double vv = diag_hybr(m, isp, iorb, kp);
        if (vv != 0.0 && n[iorb] == 1 && n[ialfa] == 0) {
          orc_c(iorb, ms, &k1, &sg1);
          orc_cdg(ialfa, k1, &k2, &sg2);
          int64_t i = orc_binary_search(map, dim, k2);
          coo_push(coo, i - 1, j, vv * sg1 * sg2, 0.0);
        }
        if (vv != 0.0 && n[iorb] == 0 && n[ialfa] == 1) {
          orc_c(ialfa, ms, &k1, &sg1);
          orc_cdg(iorb, k1, &k2, &sg2);
          int64_t i = orc_binary_search(map, dim, k2);
          coo_push(coo, i - 1, j, vv * sg1 * sg2, 0.0);
        }
      }
    /* exciton field: stored/H_up.f90:87-104 / H_dw.f90 (sign flip on exc_field(4) for dw) */
    if (m->exc_field[0] != 0.0 || m->exc_field[1] != 0.0 || m->exc_field[2] != 0.0 ||
        m->exc_field[3] != 0.0) {
      for (int iorb = 1; iorb <= norb; iorb++)
        for (int jorb = 1; jorb <= norb; jorb++)
          if (n[jorb] == 1 && n[iorb] == 0) {
            orc_c(jorb, ms, &k1, &sg1);
            orc_cdg(iorb, k1, &k2, &sg2);
            int64_t i = orc_binary_search(map, dim, k2);
            coo_push(coo, i - 1, j, m->exc_field[0] * sg1 * sg2, 0.0);
            coo_push(coo, i - 1, j, (spin == 1 ? 1.0 : -1.0) * m->exc_field[3] * sg1 * sg2, 0.0);
          }
    }
  }
}

/* ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:26-267, serial branch, DimPh=1. */
orc_hnormal *orc_buildh_normal_main(const orc_model *m, int nup_tot, int ndw_tot) {
  orc_hnormal *h = xcalloc(1, sizeof(*h));
  int ns = orc_ns(m), norb = m->norb, nbath = m->nbath, nspin = m->nspin;
  h->ns = ns;
  h->nup = nup_tot;
  h->ndw = ndw_tot;
  h->dimup = orc_binomial(ns, nup_tot);
  h->dimdw = orc_binomial(ns, ndw_tot);
  h->dim = h->dimup * h->dimdw;
  h->mapup = xcalloc(h->dimup, sizeof(int32_t));
  h->mapdw = xcalloc(h->dimdw, sizeof(int32_t));
  orc_build_sector_normal(ns, nup_tot, ndw_tot, h->mapup, h->mapdw);
  int64_t DimUp = h->dimup;
  h->hd = xcalloc(h->dim, sizeof(double));
  /* either_condition = nonloc_condition .OR. sundry_condition (ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:56-58) */
  h->has_nd = nonloc_condition(m) || m->nsundry > 0;

  int any_sfz = 0;
  for (int io = 0; io < norb; io++)
    if (m->spin_field[io][2] != 0.0) any_sfz = 1;

  /* ---- stored/H_local.f90:1-88 ---- */
  for (int64_t i = 1; i <= h->dim; i++) {
    int64_t iup = i % DimUp;
    if (iup == 0) iup = DimUp;               /* iup_index, ED_SECTOR.f90:1705 */
    int64_t idw = (i - 1) / DimUp + 1;        /* idw_index, ED_SECTOR.f90:1712 */
    int32_t mup = h->mapup[iup - 1], mdw = h->mapdw[idw - 1];
    double nup[32], ndw[32];
    for (int l = 0; l < ns; l++) {
      nup[l + 1] = (double)((mup >> l) & 1);
      ndw[l + 1] = (double)((mdw >> l) & 1);
    }
    double htmp = 0.0;
    for (int io = 1; io <= norb; io++) {
      htmp += (m->hloc_re[0][0][io - 1][io - 1] + m->mfh_re[0][0][io - 1][io - 1]) * nup[io];
      htmp += m->hloc_re[nspin - 1][nspin - 1][io - 1][io - 1] * ndw[io];
      htmp += m->mfh_re[1][1][io - 1][io - 1] * ndw[io];
      htmp -= m->xmu * (nup[io] + ndw[io]);
    }
    if (any_sfz)
      for (int io = 1; io <= norb; io++) htmp += m->spin_field[io - 1][2] * (nup[io] - ndw[io]);
    for (int io = 1; io <= norb; io++) htmp += m->uloc[io - 1] * nup[io] * ndw[io];
    if (norb > 1) {
      for (int io = 1; io <= norb; io++)
        for (int jo = io + 1; jo <= norb; jo++)
          htmp += m->ust[io - 1][jo - 1] * (nup[io] * ndw[jo] + nup[jo] * ndw[io]);
      for (int io = 1; io <= norb; io++)
        for (int jo = io + 1; jo <= norb; jo++)
          htmp += (m->ust[io - 1][jo - 1] - m->jh[io - 1][jo - 1]) *
                  (nup[io] * nup[jo] + ndw[io] * ndw[jo]);
    }
    if (m->hfmode) {
      for (int io = 1; io <= norb; io++)
        htmp = htmp - 0.5 * m->uloc[io - 1] * (nup[io] + ndw[io]) + 0.25 * m->uloc[io - 1];
      if (norb > 1)
        for (int io = 1; io <= norb; io++)
          for (int jo = io + 1; jo <= norb; jo++) {
            double ust = m->ust[io - 1][jo - 1], ujj = ust - m->jh[io - 1][jo - 1];
            htmp = htmp - 0.5 * ust * (nup[io] + ndw[io] + nup[jo] + ndw[jo]) + 0.5 * ust;
            htmp = htmp - 0.5 * ujj * (nup[io] + ndw[io] + nup[jo] + ndw[jo]) + 0.5 * ujj;
          }
    }
    for (int io = 1; io <= bath_diag_norb(m); io++)
      for (int kp = 1; kp <= nbath; kp++) {
        int ialfa = orc_bath_stride(m, io, kp);
        htmp += bath_diag(m, 1, io, kp) * nup[ialfa];
        htmp += bath_diag(m, nspin, io, kp) * ndw[ialfa];
      }
    h->hd[i - 1] = htmp;
  }

  /* ---- stored/H_non_local.f90:4-84 ---- */
  if (h->has_nd) {
    coo_t coo;
    coo_init(&coo, 0);
    int any_jx = 0, any_jp = 0;
    for (int a = 0; a < norb; a++)
      for (int b = 0; b < norb; b++) {
        if (m->jx[a][b] != 0.0) any_jx = 1;
        if (m->jp[a][b] != 0.0) any_jp = 1;
      }
    for (int64_t i = 1; i <= h->dim; i++) {
      int64_t iup = i % DimUp;
      if (iup == 0) iup = DimUp;
      int64_t idw = (i - 1) / DimUp + 1;
      int32_t mup = h->mapup[iup - 1], mdw = h->mapdw[idw - 1];
      int nup[32], ndw[32];
      for (int l = 0; l < ns; l++) {
        nup[l + 1] = (mup >> l) & 1;
        ndw[l + 1] = (mdw >> l) & 1;
      }
      int32_t k1, k2, k3, k4;
      double sg1, sg2, sg3, sg4;
      if (norb > 1 && any_jx)
        for (int io = 1; io <= norb; io++)
          for (int jo = 1; jo <= norb; jo++)
            if (io != jo && nup[jo] == 1 && ndw[io] == 1 && ndw[jo] == 0 && nup[io] == 0) {
              orc_c(io, mdw, &k1, &sg1);
              orc_cdg(jo, k1, &k2, &sg2);
              int64_t jdw = orc_binary_search(h->mapdw, h->dimdw, k2);
              orc_c(jo, mup, &k3, &sg3);
              orc_cdg(io, k3, &k4, &sg4);
              int64_t jup = orc_binary_search(h->mapup, h->dimup, k4);
              double htmp = m->jx[io - 1][jo - 1] * sg1 * sg2 * sg3 * sg4;
              int64_t j = jup + (jdw - 1) * DimUp;
              coo_push(&coo, i - 1, j - 1, htmp, 0.0);
            }
      if (norb > 1 && any_jp)
        for (int io = 1; io <= norb; io++)
          for (int jo = 1; jo <= norb; jo++)
            if (nup[jo] == 1 && ndw[jo] == 1 && ndw[io] == 0 && nup[io] == 0) {
              orc_c(jo, mdw, &k1, &sg1);
              orc_cdg(io, k1, &k2, &sg2);
              int64_t jdw = orc_binary_search(h->mapdw, h->dimdw, k2);
              orc_c(jo, mup, &k3, &sg3);
              orc_cdg(io, k3, &k4, &sg4);
              int64_t jup = orc_binary_search(h->mapup, h->dimup, k4);
              double htmp = m->jp[io - 1][jo - 1] * sg1 * sg2 * sg3 * sg4;
              int64_t j = jup + (jdw - 1) * DimUp;
              coo_push(&coo, i - 1, j - 1, htmp, 0.0);
            }
    }
    /* ---- stored/H_sundry.f90:1-111: generic two-body terms U cd_i cd_j c_k c_l, applied right to left as
     * c_l, cd_j, c_k, cd_i on the up / down word of their spin (no cross-spin sign) ---- */
    for (int64_t i = 1; m->nsundry > 0 && i <= h->dim; i++) {
      int64_t iup = i % DimUp;
      if (iup == 0) iup = DimUp;
      int64_t idw = (i - 1) / DimUp + 1;
      int32_t mup = h->mapup[iup - 1], mdw = h->mapdw[idw - 1];
      for (int il = 0; il < m->nsundry; il++) {
        const int *op = m->sundry_op[il];
        const int orbvec_dag[2] = {op[0], op[2]}, spinvec_dag[2] = {op[1], op[3]};
        const int orbvec[2] = {op[4], op[6]}, spinvec[2] = {op[5], op[7]};
        int32_t pu = mup, pd = mdw, t;
        double sg[4];
        int err; /* orc_c / orc_cdg: non-zero = the operator annihilates the state (the reference's ierr / Jcondition) */
        /* last annihilation, last creation, first annihilation, first creation */
        if (spinvec[1] == 1) { err = orc_c(orbvec[1], pu, &t, &sg[0]); pu = t; } else { err = orc_c(orbvec[1], pd, &t, &sg[0]); pd = t; }
        if (err) continue;
        if (spinvec_dag[1] == 1) { err = orc_cdg(orbvec_dag[1], pu, &t, &sg[1]); pu = t; } else { err = orc_cdg(orbvec_dag[1], pd, &t, &sg[1]); pd = t; }
        if (err) continue;
        if (spinvec[0] == 1) { err = orc_c(orbvec[0], pu, &t, &sg[2]); pu = t; } else { err = orc_c(orbvec[0], pd, &t, &sg[2]); pd = t; }
        if (err) continue;
        if (spinvec_dag[0] == 1) { err = orc_cdg(orbvec_dag[0], pu, &t, &sg[3]); pu = t; } else { err = orc_cdg(orbvec_dag[0], pd, &t, &sg[3]); pd = t; }
        if (err) continue;
        int64_t jdw = orc_binary_search(h->mapdw, h->dimdw, pd), jup = orc_binary_search(h->mapup, h->dimup, pu);
        if (jup == 0 || jdw == 0) continue; /* the reference STOPs ("impossible operator"): a spin-changing line */
        coo_push(&coo, i - 1, jup + (jdw - 1) * DimUp - 1, m->sundry_u[il] * sg[0] * sg[1] * sg[2] * sg[3], 0.0);
      }
    }
    coo_to_csr(&coo, h->dim, h->dim, &h->nd);
    coo_free(&coo);
  }

  /* ---- stored/H_up.f90, stored/H_dw.f90 ---- */
  {
    coo_t coo;
    coo_init(&coo, 0);
    build_h_spin(m, ns, 1, h->mapup, h->dimup, &coo);
    coo_to_csr(&coo, h->dimup, h->dimup, &h->up);
    coo_free(&coo);
    coo_init(&coo, 0);
    build_h_spin(m, ns, 2, h->mapdw, h->dimdw, &coo);
    coo_to_csr(&coo, h->dimdw, h->dimdw, &h->dw);
    coo_free(&coo);
  }
  return h;
}

void orc_hnormal_free(orc_hnormal *h) {
  if (!h) return;
  free(h->mapup);
  free(h->mapdw);
  free(h->hd);
  csr_free(&h->up);
  csr_free(&h->dw);
  csr_free(&h->nd);
  free(h);
}

void orc_hnormal_sizes(const orc_hnormal *h, int64_t out[8]) {
  out[0] = h->dimup;
  out[1] = h->dimdw;
  out[2] = h->dim;
  out[3] = h->up.nnz;
  out[4] = h->dw.nnz;
  out[5] = h->has_nd ? h->nd.nnz : 0;
  out[6] = h->has_nd;
  out[7] = h->ns;
}

/* ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650 with DimPh=1: the
 * same four sweeps in the same order (diagonal; DW with iup outer / idw inner;
 * UP with idw outer / iup inner; non-local). */
void orc_spmatvec_normal_main(const orc_hnormal *h, const double *v, double *hv) {
  int64_t DimUp = h->dimup, DimDw = h->dimdw, N = h->dim;
  for (int64_t i = 0; i < N; i++) hv[i] = 0.0;
  /* :542-554 */
  for (int64_t i = 0; i < N; i++) hv[i] = hv[i] + h->hd[i] * v[i];
  /* :558-575 DW */
  for (int64_t iup = 0; iup < DimUp; iup++)
    for (int64_t idw = 0; idw < DimDw; idw++) {
      int64_t i = iup + idw * DimUp;
      for (int64_t jj = h->dw.rowptr[idw]; jj < h->dw.rowptr[idw + 1]; jj++) {
        int64_t j = iup + (int64_t)h->dw.col[jj] * DimUp;
        hv[i] = hv[i] + h->dw.val[jj] * v[j];
      }
    }
  /* :578-595 UP */
  for (int64_t idw = 0; idw < DimDw; idw++)
    for (int64_t iup = 0; iup < DimUp; iup++) {
      int64_t i = iup + idw * DimUp;
      for (int64_t jj = h->up.rowptr[iup]; jj < h->up.rowptr[iup + 1]; jj++) {
        int64_t j = (int64_t)h->up.col[jj] + idw * DimUp;
        hv[i] = hv[i] + h->up.val[jj] * v[j];
      }
    }
  /* :634-648 non-local */
  if (h->has_nd)
    for (int64_t i = 0; i < N; i++)
      for (int64_t jj = h->nd.rowptr[i]; jj < h->nd.rowptr[i + 1]; jj++)
        hv[i] = hv[i] + h->nd.val[jj] * v[h->nd.col[jj]];
}

/* spMatVec_normal_main with DimPh > 1 (:517-650 incl. :597-629) */
void orc_spmatvec_normal_ph(const orc_hnormal *h, const orc_model *m, const double *v, double *hv) {
  const int64_t N = h->dim, DimUp = h->dimup;
  const int dimph = m->nph + 1, norb = m->norb;
  /* electronic part, the same for every phonon number */
  for (int iph = 0; iph < dimph; iph++) orc_spmatvec_normal_main(h, v + iph * N, hv + iph * N);
  for (int iph = 1; iph <= dimph; iph++)
    for (int64_t i_el = 1; i_el <= N; i_el++) {
      int64_t i = (i_el - 1) + (int64_t)(iph - 1) * N;
      /* PHONON: stored/H_ph.f90 -- diagonal w0*(iph-1); A*sqrt(iph) at (iph+1,iph), A*sqrt(iph-1) at (iph-1,iph) */
      hv[i] += m->w0_ph * (double)(iph - 1) * v[i];
      if (m->a_ph != 0.0) {
        if (iph > 1) hv[i] += m->a_ph * sqrt((double)(iph - 1)) * v[i - N]; /* row iph, col iph-1 */
        if (iph < dimph) hv[i] += m->a_ph * sqrt((double)iph) * v[i + N];   /* row iph, col iph+1 */
      }
      /* ELECTRON-PHONON: (rows of spH0e_eph) x (b + b^+): stored/H_e_ph.f90 */
      int64_t iup = (i_el - 1) % DimUp, idw = (i_el - 1) / DimUp;
      int32_t mup = h->mapup[iup], mdw = h->mapdw[idw];
      for (int side = 0; side < 2; side++) { /* phonon neighbour: iph-1, iph+1 */
        int jph = side == 0 ? iph - 1 : iph + 1;
        if (jph < 1 || jph > dimph) continue;
        double bval = side == 0 ? sqrt((double)(iph - 1)) : sqrt((double)iph);
        int64_t joff = (int64_t)(jph - 1) * N;
        double gdiag = 0.0;
        for (int io = 0; io < norb; io++) gdiag += m->g_ph[io][io] * (double)(((mup >> io) & 1) + ((mdw >> io) & 1));
        hv[i] += gdiag * bval * v[(i_el - 1) + joff];
        /* off-diagonal g: the stored entry (row j, col i) = g(io,jo) c^+_io c_jo |i>; the matrix is symmetric
         * (g Hermitian, real here), so row i holds the same values at the columns its hops reach */
        for (int io = 1; io <= norb; io++)
          for (int jo = 1; jo <= norb; jo++) {
            if (io == jo || m->g_ph[io - 1][jo - 1] == 0.0) continue;
            int32_t k1, k2;
            double sg1, sg2;
            if (((mup >> (jo - 1)) & 1) == 1 && ((mup >> (io - 1)) & 1) == 0) {
              orc_c(jo, mup, &k1, &sg1);
              orc_cdg(io, k1, &k2, &sg2);
              int64_t jup = orc_binary_search(h->mapup, h->dimup, k2) - 1;
              hv[i] += m->g_ph[io - 1][jo - 1] * sg1 * sg2 * bval * v[jup + idw * DimUp + joff];
            }
            if (((mdw >> (jo - 1)) & 1) == 1 && ((mdw >> (io - 1)) & 1) == 0) {
              orc_c(jo, mdw, &k1, &sg1);
              orc_cdg(io, k1, &k2, &sg2);
              int64_t jdw = orc_binary_search(h->mapdw, h->dimdw, k2) - 1;
              hv[i] += m->g_ph[io - 1][jo - 1] * sg1 * sg2 * bval * v[iup + jdw * DimUp + joff];
            }
          }
      }
    }
}

void orc_spmatvec_normal_arrays(int64_t dimup, int64_t dimdw, const double *hd,
                                const int64_t *up_rowptr, const int32_t *up_col, const double *up_val,
                                const int64_t *dw_rowptr, const int32_t *dw_col, const double *dw_val,
                                const int64_t *nd_rowptr, const int32_t *nd_col, const double *nd_val,
                                const double *v, double *hv) {
  orc_hnormal h;
  memset(&h, 0, sizeof(h));
  h.dimup = dimup;
  h.dimdw = dimdw;
  h.dim = dimup * dimdw;
  h.hd = (double *)hd;
  h.up.nrow = dimup;
  h.up.rowptr = (int64_t *)up_rowptr;
  h.up.col = (int32_t *)up_col;
  h.up.val = (double *)up_val;
  h.dw.nrow = dimdw;
  h.dw.rowptr = (int64_t *)dw_rowptr;
  h.dw.col = (int32_t *)dw_col;
  h.dw.val = (double *)dw_val;
  h.has_nd = nd_rowptr != NULL;
  h.nd.nrow = h.dim;
  h.nd.rowptr = (int64_t *)nd_rowptr;
  h.nd.col = (int32_t *)nd_col;
  h.nd.val = (double *)nd_val;
  orc_spmatvec_normal_main(&h, v, hv);
}

/* The same product with the reference's MPI decomposition (spMatVec_mpi_normal_main, :765-929: every rank
 * owns a block of down indices) mapped on OpenMP threads of one host: thread t = rank t.  Shared memory
 * replaces the two MPI_Alltoallv transposes and the MPI_Allgatherv (all threads read the same v).
 * Used only as the multi-core CPU baseline of bench.py. */
void orc_spmatvec_normal_arrays_mt(int64_t dimup, int64_t dimdw, const double *hd,
                                   const int64_t *up_rowptr, const int32_t *up_col, const double *up_val,
                                   const int64_t *dw_rowptr, const int32_t *dw_col, const double *dw_val,
                                   const int64_t *nd_rowptr, const int32_t *nd_col, const double *nd_val,
                                   const double *v, double *hv, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t idw = 0; idw < dimdw; idw++) {
    for (int64_t iup = 0; iup < dimup; iup++) {
      int64_t i = iup + idw * dimup;
      double acc = hd[i] * v[i];
      for (int64_t jj = up_rowptr[iup]; jj < up_rowptr[iup + 1]; jj++)
        acc += up_val[jj] * v[(int64_t)up_col[jj] + idw * dimup];
      hv[i] = acc;
    }
    for (int64_t jj = dw_rowptr[idw]; jj < dw_rowptr[idw + 1]; jj++) {
      const double w = dw_val[jj];
      const double *src = v + (int64_t)dw_col[jj] * dimup;
      double *dst = hv + idw * dimup;
      for (int64_t iup = 0; iup < dimup; iup++) dst[iup] += w * src[iup];
    }
    if (nd_rowptr)
      for (int64_t i = idw * dimup; i < (idw + 1) * dimup; i++)
        for (int64_t jj = nd_rowptr[i]; jj < nd_rowptr[i + 1]; jj++) hv[i] += nd_val[jj] * v[nd_col[jj]];
  }
}

/* row-partitioned CSR product (spMatVec_mpi_superc_main / _nonsu2_main data flow) on OpenMP threads */
void orc_csr_matvec_z_mt(const orc_csr *a, const double *x, double *y, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t i = 0; i < a->nrow; i++) {
    double sr = 0.0, si = 0.0;
    for (int64_t k = a->rowptr[i]; k < a->rowptr[i + 1]; k++) {
      const double ar = a->val[2 * k], ai = a->val[2 * k + 1];
      const double xr = x[2 * (int64_t)a->col[k]], xi = x[2 * (int64_t)a->col[k] + 1];
      sr += ar * xr - ai * xi;
      si += ar * xi + ai * xr;
    }
    y[2 * i] = sr;
    y[2 * i + 1] = si;
  }
}

/* ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:209-262 (Hmat dump), row-major. */
void orc_hnormal_dense(const orc_hnormal *h, double *hmat) {
  int64_t N = h->dim, DimUp = h->dimup, DimDw = h->dimdw;
  memset(hmat, 0, sizeof(double) * N * N);
  for (int64_t i = 0; i < N; i++) hmat[i * N + i] += h->hd[i];
  if (h->has_nd)
    for (int64_t i = 0; i < N; i++)
      for (int64_t jj = h->nd.rowptr[i]; jj < h->nd.rowptr[i + 1]; jj++)
        hmat[i * N + h->nd.col[jj]] += h->nd.val[jj];
  for (int64_t idw = 0; idw < DimDw; idw++)
    for (int64_t jj = h->dw.rowptr[idw]; jj < h->dw.rowptr[idw + 1]; jj++)
      for (int64_t iup = 0; iup < DimUp; iup++)
        hmat[(iup + idw * DimUp) * N + (iup + (int64_t)h->dw.col[jj] * DimUp)] += h->dw.val[jj];
  for (int64_t iup = 0; iup < DimUp; iup++)
    for (int64_t jj = h->up.rowptr[iup]; jj < h->up.rowptr[iup + 1]; jj++)
      for (int64_t idw = 0; idw < DimDw; idw++)
        hmat[(iup + idw * DimUp) * N + ((int64_t)h->up.col[jj] + idw * DimUp)] += h->up.val[jj];
}

/* ------------------------------------------------------------------ */
/* Lanczos tridiagonalisation (SciFortran sp_lanc_tridiag, restated)   */
/* ------------------------------------------------------------------ */

/* One step of the three-term recurrence as SciFortran's lanczos_iteration
 * does it (third-party; source not under /root/reference, algorithm restated):
 *   iter==1: vin <- vin/|vin| ; else (vin, vout) <- (vout/beta, -beta*vin)
 *   vout <- vout + H vin ; alfa = <vin|vout> ; vout <- vout - alfa vin ;
 *   beta = |vout|
 * The consumer reads alanc(1:N), blanc(2:N) (ED_NORMAL/ED_GF_NORMAL.f90:410-411). */
typedef void (*matvec_fn)(const void *ctx, const double *v, double *hv);

static int lanc_tridiag(matvec_fn mv, const void *ctx, int64_t n, int w, double *vin, int nitermax,
                        double *alanc, double *blanc, double threshold) {
  double *vout = xcalloc(n * w, sizeof(double));
  double *tmp = xcalloc(n * w, sizeof(double));
  double a_ = 0.0, b_ = 0.0;
  int done = 0;
  for (int k = 0; k < nitermax; k++) alanc[k] = blanc[k] = 0.0;
  for (int iter = 1; iter <= nitermax; iter++) {
    if (iter == 1) {
      double nrm = 0.0;
      for (int64_t i = 0; i < n * w; i++) nrm += vin[i] * vin[i];
      nrm = sqrt(nrm);
      if (nrm == 0.0) break;
      for (int64_t i = 0; i < n * w; i++) vin[i] /= nrm;
    } else {
      for (int64_t i = 0; i < n * w; i++) {
        double t = vin[i];
        vin[i] = vout[i] / b_;
        vout[i] = -b_ * t;
      }
    }
    mv(ctx, vin, tmp);
    for (int64_t i = 0; i < n * w; i++) vout[i] += tmp[i];
    /* alfa = <vin|vout>: real part (H Hermitian) */
    double s = 0.0;
    for (int64_t i = 0; i < n * w; i++) s += vin[i] * vout[i];
    a_ = s;
    for (int64_t i = 0; i < n * w; i++) vout[i] -= a_ * vin[i];
    s = 0.0;
    for (int64_t i = 0; i < n * w; i++) s += vout[i] * vout[i];
    b_ = sqrt(s);
    alanc[iter - 1] = a_;
    done = iter;
    if (fabs(b_) < threshold) break;
    if (iter < nitermax) blanc[iter] = b_;
  }
  free(vout);
  free(tmp);
  return done;
}

static void mv_normal(const void *ctx, const double *v, double *hv) {
  orc_spmatvec_normal_main((const orc_hnormal *)ctx, v, hv);
}

int orc_lanc_tridiag_normal(const orc_hnormal *h, double *vin, int nitermax, double *alanc,
                            double *blanc, double threshold) {
  return lanc_tridiag(mv_normal, h, h->dim, 1, vin, nitermax, alanc, blanc, threshold);
}

typedef struct {
  const orc_hnormal *h;
  const orc_model *m;
} ph_ctx;

static void mv_normal_ph(const void *ctx, const double *v, double *hv) {
  const ph_ctx *c = (const ph_ctx *)ctx;
  orc_spmatvec_normal_ph(c->h, c->m, v, hv);
}

int orc_lanc_tridiag_normal_ph(const orc_hnormal *h, const orc_model *m, double *vin, int nitermax, double *alanc,
                               double *blanc, double threshold) {
  ph_ctx c = {h, m};
  return lanc_tridiag(mv_normal_ph, &c, h->dim * (m->nph + 1), 1, vin, nitermax, alanc, blanc, threshold);
}

/* ------------------------------------------------------------------ */
/* flat CSR modes (superc / nonsu2)                                     */
/* ------------------------------------------------------------------ */

void orc_hflat_free(orc_hflat *h) {
  if (!h) return;
  free(h->map);
  csr_free(&h->h);
  free(h);
}

void orc_hflat_sizes(const orc_hflat *h, int64_t out[4]) {
  out[0] = h->dim;
  out[1] = h->h.nnz;
  out[2] = h->ns;
  out[3] = h->h.is_complex;
}

/* ED_SUPERC/ED_HAMILTONIAN_SUPERC_STORED_HxV.f90:312-362,
 * ED_NONSU2/ED_HAMILTONIAN_NONSU2_STORED_HxV.f90:194-209 (DimPh=1) */
void orc_spmatvec_flat_z(const orc_hflat *h, const double *v, double *hv) {
  int64_t N = h->dim;
  for (int64_t i = 0; i < 2 * N; i++) hv[i] = 0.0;
  for (int64_t i = 0; i < N; i++)
    for (int64_t k = h->h.rowptr[i]; k < h->h.rowptr[i + 1]; k++) {
      double ar = h->h.val[2 * k], ai = h->h.val[2 * k + 1];
      int64_t j = h->h.col[k];
      hv[2 * i] += ar * v[2 * j] - ai * v[2 * j + 1];
      hv[2 * i + 1] += ar * v[2 * j + 1] + ai * v[2 * j];
    }
}

void orc_hflat_dense(const orc_hflat *h, double *hmat) {
  int64_t N = h->dim;
  memset(hmat, 0, sizeof(double) * 2 * N * N);
  for (int64_t i = 0; i < N; i++)
    for (int64_t k = h->h.rowptr[i]; k < h->h.rowptr[i + 1]; k++) {
      hmat[2 * (i * N + h->h.col[k])] += h->h.val[2 * k];
      hmat[2 * (i * N + h->h.col[k]) + 1] += h->h.val[2 * k + 1];
    }
}

static void mv_flat(const void *ctx, const double *v, double *hv) {
  orc_spmatvec_flat_z((const orc_hflat *)ctx, v, hv);
}

/* complex vectors: <vin|vout> = sum conj(vin)*vout, real part kept (H Hermitian) */
int orc_lanc_tridiag_flat(const orc_hflat *h, double *vin, int nitermax, double *alanc,
                          double *blanc, double threshold) {
  return lanc_tridiag(mv_flat, h, h->dim, 2, vin, nitermax, alanc, blanc, threshold);
}

/* spMatVec_superc_main / spMatVec_nonsu2_main with DimPh > 1 */
void orc_spmatvec_flat_ph(const orc_hflat *h, const orc_model *m, const double *v, double *hv) {
  const int64_t N = h->dim;
  const int dimph = m->nph + 1, norb = m->norb, ns = h->ns;
  for (int iph = 0; iph < dimph; iph++) orc_spmatvec_flat_z(h, v + 2 * iph * N, hv + 2 * iph * N);
  for (int iph = 1; iph <= dimph; iph++)
    for (int64_t i_el = 0; i_el < N; i_el++) {
      const int64_t i = i_el + (int64_t)(iph - 1) * N;
      const int32_t st = h->map[i_el];
      double ar = m->w0_ph * (double)(iph - 1) * v[2 * i], ai = m->w0_ph * (double)(iph - 1) * v[2 * i + 1];
      for (int side = 0; side < 2; side++) {
        const int jph = side == 0 ? iph - 1 : iph + 1;
        if (jph < 1 || jph > dimph) continue;
        const double bval = side == 0 ? sqrt((double)(iph - 1)) : sqrt((double)iph);
        const int64_t joff = (int64_t)(jph - 1) * N;
        double gd = m->a_ph; /* A (b + b^+) has the same phonon structure */
        for (int io = 0; io < norb; io++) gd += m->g_ph[io][io] * (double)(((st >> io) & 1) + ((st >> (io + ns)) & 1));
        ar += gd * bval * v[2 * (i_el + joff)];
        ai += gd * bval * v[2 * (i_el + joff) + 1];
        for (int io = 1; io <= norb; io++)
          for (int jo = 1; jo <= norb; jo++) {
            if (io == jo || m->g_ph[io - 1][jo - 1] == 0.0) continue;
            for (int sp = 0; sp < 2; sp++) {
              const int pi = io + sp * ns, pj = jo + sp * ns;
              if (((st >> (pj - 1)) & 1) == 1 && ((st >> (pi - 1)) & 1) == 0) {
                int32_t k1, k2;
                double sg1, sg2;
                orc_c(pj, st, &k1, &sg1);
                orc_cdg(pi, k1, &k2, &sg2);
                const int64_t j = orc_binary_search(h->map, N, k2) - 1;
                const double w = m->g_ph[io - 1][jo - 1] * sg1 * sg2 * bval;
                ar += w * v[2 * (j + joff)];
                ai += w * v[2 * (j + joff) + 1];
              }
            }
          }
      }
      hv[2 * i] += ar;
      hv[2 * i + 1] += ai;
    }
}

typedef struct {
  const orc_hflat *h;
  const orc_model *m;
} phf_ctx;

static void mv_flat_ph(const void *ctx, const double *v, double *hv) {
  const phf_ctx *c = (const phf_ctx *)ctx;
  orc_spmatvec_flat_ph(c->h, c->m, v, hv);
}

int orc_lanc_tridiag_flat_ph(const orc_hflat *h, const orc_model *m, double *vin, int nitermax, double *alanc,
                             double *blanc, double threshold) {
  phf_ctx c = {h, m};
  return lanc_tridiag(mv_flat_ph, &c, h->dim * (m->nph + 1), 2, vin, nitermax, alanc, blanc, threshold);
}

#include "edipack_oracle_flat.inc"
#include "edipack_oracle_orbs.inc"
