/*
 * edipack_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the EDIpack Hamiltonian-times-vector hot path.
 * It exists to CHECK the HIP product path (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).  Nothing under edipack_amd/ may include, link
 * or call it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/src/singlesite unless noted).
 *
 * Parity status: PINNED against the reference's own golden fixtures
 * (test/src/{NORMAL,HYBRID,REPLICA,GENERAL}_{NORMAL,SUPERC,NONSU2}/{evals,dens,docc}.check), see
 * tests/test_oracle_golden.py.  The Lanczos recurrence (SciFortran
 * sp_lanc_tridiag, third party, un-vendored, un-pinned "master") is restated
 * from its published algorithm and pinned through Sigma_momenta.check of
 * NORMAL_NORMAL / HYBRID_NORMAL (functions of the alpha/beta it returns; tests/gf_normal.py).
 */
#ifndef EDIPACK_ORACLE_H
#define EDIPACK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXORB 5
#define ORC_MAXBATH 16

/* Model parameters: the module-global state of the reference that the
 * Hamiltonian builders read (ED_INPUT_VARS / ED_VARS_GLOBAL / dmft_bath). */
#define ORC_MAXSUNDRY 64
typedef struct {
  int ed_mode;   /* 0 normal, 1 superc, 2 nonsu2                (ED_INPUT_VARS ED_MODE)   */
  int bath_type; /* 0 normal, 1 hybrid, 2 replica, 3 general    (ED_INPUT_VARS BATH_TYPE) */
  int norb, nbath, nspin;
  int hfmode;
  double xmu;
  /* interaction tables after set_umatrix (ED_PARSE_UMATRIX.f90:92-142) */
  double uloc[ORC_MAXORB];
  double ust[ORC_MAXORB][ORC_MAXORB];
  double jh[ORC_MAXORB][ORC_MAXORB];
  double jx[ORC_MAXORB][ORC_MAXORB];
  double jp[ORC_MAXORB][ORC_MAXORB];
  /* impHloc(ispin,jspin,iorb,jorb) and mfHloc, complex */
  double hloc_re[2][2][ORC_MAXORB][ORC_MAXORB];
  double hloc_im[2][2][ORC_MAXORB][ORC_MAXORB];
  double mfh_re[2][2][ORC_MAXORB][ORC_MAXORB];
  double mfh_im[2][2][ORC_MAXORB][ORC_MAXORB];
  double anom_re[ORC_MAXORB][ORC_MAXORB]; /* impHloc_anomalous(1,1,:,:) */
  double anom_im[ORC_MAXORB][ORC_MAXORB];
  double pair_field[ORC_MAXORB];
  double spin_field[ORC_MAXORB][3];
  double exc_field[4];
  /* dmft_bath%e,v,d,u(ispin,iorb,k); hybrid uses iorb=0 only for e,d */
  double be[2][ORC_MAXORB][ORC_MAXBATH];
  double bv[2][ORC_MAXORB][ORC_MAXBATH];
  double bd[2][ORC_MAXORB][ORC_MAXBATH];
  double bu[2][ORC_MAXORB][ORC_MAXBATH];
  /* replica/general: build_Hreplica(lambda) result hbath_tmp(is,js,io,jo,k) */
  double hb_re[2][2][ORC_MAXORB][ORC_MAXORB][ORC_MAXBATH];
  double hb_im[2][2][ORC_MAXORB][ORC_MAXORB][ORC_MAXBATH];
  double vr[ORC_MAXBATH];                 /* replica: item(k)%v            */
  double vg[2 * ORC_MAXORB][ORC_MAXBATH]; /* general: item(k)%vg(io+Norb*(is-1)) */
  /* phonons (ED_INPUT_VARS.f90:184-198): Nph = cut-off (DimPh = Nph+1), W0_PH, A_PH, g_ph(iorb,jorb) */
  int nph;
  double w0_ph, a_ph;
  double g_ph[ORC_MAXORB][ORC_MAXORB];
  /* coulomb_sundry(:) (ED_VARS_GLOBAL.f90:299, filled by ED_PARSE_UMATRIX): U cd_i cd_j c_k c_l, per line
   * [cd_i orb, cd_i spin, cd_j orb, cd_j spin, c_k orb, c_k spin, c_l orb, c_l spin], orbitals 1-based, spin 1 up / 2 down */
  int nsundry;
  int sundry_op[ORC_MAXSUNDRY][8];
  double sundry_u[ORC_MAXSUNDRY];
} orc_model;

/* CSR matrix in the reference's row order (insertion order inside a row,
 * duplicates accumulated into the first occurrence: ED_SPARSE_MATRIX.f90:328-360).
 * Indices are 0-based here.  val holds nnz doubles (real) or 2*nnz (complex). */
typedef struct {
  int64_t nrow, ncol, nnz;
  int is_complex;
  int64_t *rowptr;
  int32_t *col;
  double *val;
} orc_csr;

/* normal-mode Kronecker pieces (ED_VARS_GLOBAL.f90:190-195: spH0d, spH0ups(1), spH0dws(1), spH0nd) */
typedef struct {
  int ns, nup, ndw;
  int64_t dimup, dimdw, dim;
  int32_t *mapup, *mapdw;
  double *hd; /* spH0d: one entry per row */
  orc_csr up, dw, nd;
  int has_nd;
} orc_hnormal;

int orc_ns(const orc_model *m);
int64_t orc_binomial(int n1, int n2);
int orc_bath_stride(const orc_model *m, int iorb, int kp); /* 1-based in, 1-based out */

/* ED_SECTOR.f90:165-373 */
int orc_build_sector_normal(int ns, int nup, int ndw, int32_t *mapup, int32_t *mapdw);
int64_t orc_build_sector_superc(int ns, int sz, int32_t *map);  /* map may be NULL: count only */
int64_t orc_build_sector_nonsu2(int ns, int ntot, int32_t *map);
/* Jz_basis=T (ED_SECTOR.f90:289-350): Norb = 3, levels iorb + Norb*ibath; -1 for other Norb */
int64_t orc_build_sector_nonsu2_jz(int norb, int nbath, int ntot, int twojz, int32_t *map);

/* ED_AUX_FUNX.f90:334-384, :463-480 */
int orc_c(int pos, int32_t in, int32_t *out, double *sgn);
int orc_cdg(int pos, int32_t in, int32_t *out, double *sgn);
int64_t orc_binary_search(const int32_t *a, int64_t n, int32_t value); /* 1-based result, 0 = not found */

/* ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:26-267 (+ stored/H_local, H_up, H_dw, H_non_local) */
orc_hnormal *orc_buildh_normal_main(const orc_model *m, int nup, int ndw);
void orc_hnormal_free(orc_hnormal *h);
/* ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650 */
void orc_spmatvec_normal_main(const orc_hnormal *h, const double *v, double *hv);
/* the same with the phonon branches (DimPh = Nph+1 > 1, :597-629): H = 1 (x) H_el + H_ph (x) 1 + (b + b^+) (x) G_el with
 * H_ph = w0 b^+ b + A (b + b^+) (stored/H_ph.f90) and G_el = sum_ab g_ab c^+_a c_b over both spins
 * (stored/H_e_ph.f90); vectors of length dim * (Nph+1), index i_el + iph * dim.  PARITY UNPINNED against
 * fixtures (the reference ships none with phonons): checked through the g = A = 0 limit and the Lang-Firsov
 * atomic limit in tests/test_oracle_golden.py. */
void orc_spmatvec_normal_ph(const orc_hnormal *h, const orc_model *m, const double *v, double *hv);
/* dense dump: ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:209-262 (column-major nothing: symmetric, row-major out) */
void orc_hnormal_dense(const orc_hnormal *h, double *hmat);

/* same product on caller-owned arrays (bench.py cpu_baseline leg: the matrices of the timed
 * workload are handed over so that CPU and GPU multiply the identical Hamiltonian) */
void orc_spmatvec_normal_arrays(int64_t dimup, int64_t dimdw, const double *hd,
                                const int64_t *up_rowptr, const int32_t *up_col, const double *up_val,
                                const int64_t *dw_rowptr, const int32_t *dw_col, const double *dw_val,
                                const int64_t *nd_rowptr, const int32_t *nd_col, const double *nd_val,
                                const double *v, double *hv);

/* multi-core CPU baseline of bench.py: the reference's MPI row decompositions (spMatVec_mpi_normal_main
 * :765-929, spMatVec_mpi_superc_main :366-432) with one OpenMP thread per "rank" */
void orc_spmatvec_normal_arrays_mt(int64_t dimup, int64_t dimdw, const double *hd,
                                   const int64_t *up_rowptr, const int32_t *up_col, const double *up_val,
                                   const int64_t *dw_rowptr, const int32_t *dw_col, const double *dw_val,
                                   const int64_t *nd_rowptr, const int32_t *nd_col, const double *nd_val,
                                   const double *v, double *hv, int nthreads);

/* flat CSR modes: ED_SUPERC/ED_HAMILTONIAN_SUPERC_STORED_HxV.f90:29-293, ED_NONSU2/..._STORED_HxV.f90:29-175 */
typedef struct {
  int ns;
  int64_t dim;
  int32_t *map;
  orc_csr h; /* complex */
} orc_hflat;
orc_hflat *orc_buildh_superc_main(const orc_model *m, int sz);
orc_hflat *orc_buildh_nonsu2_main(const orc_model *m, int ntot);
/* the same builder on a Jz_basis=T sector (Ntot, twoJz): build_sector's map is the only difference */
orc_hflat *orc_buildh_nonsu2_jz(const orc_model *m, int ntot, int twojz);
/* on-the-fly product of the nonsu2 mode (directMatVec_nonsu2_main, ED_NONSU2/ED_HAMILTONIAN_NONSU2_DIRECT_HxV.f90:22-126);
 * v, hv: complex, re/im interleaved.  _rows: rows [row0, row1) with the sector map handed in (one call per thread) */
int orc_directmatvec_nonsu2_main(const orc_model *m, int ntot, const double *v, double *hv);
void orc_directmatvec_nonsu2_rows(const orc_model *m, const int32_t *map, int64_t dim, int64_t row0, int64_t row1,
                                  const double *v, double *hv);
void orc_hflat_free(orc_hflat *h);
/* ED_SUPERC/..._STORED_HxV.f90:312-362, ED_NONSU2/..._STORED_HxV.f90:194-209 */
void orc_spmatvec_flat_z(const orc_hflat *h, const double *v, double *hv);
void orc_hflat_dense(const orc_hflat *h, double *hmat_re_im);
/* phonon branches of spMatVec_superc_main / _nonsu2_main (ED_SUPERC/..._STORED_HxV.f90:336-360, stored/H_ph.f90,
 * H_e_ph.f90): vectors of dim * (Nph+1) complex elements.  PARITY UNPINNED against fixtures (none exist). */
void orc_spmatvec_flat_ph(const orc_hflat *h, const orc_model *m, const double *v, double *hv);
int orc_lanc_tridiag_flat_ph(const orc_hflat *h, const orc_model *m, double *vin, int nitermax, double *alanc,
                             double *blanc, double threshold);

/* generic CSR y = A x in the reference's loop order (ED_SPARSE_MATRIX.f90:778-793) */
void orc_csr_matvec_d(const orc_csr *a, const double *x, double *y);
void orc_csr_matvec_z(const orc_csr *a, const double *x, double *y);
void orc_csr_matvec_z_mt(const orc_csr *a, const double *x, double *y, int nthreads);

/* SciFortran SF_SP_LINALG sp_lanc_tridiag restated (three-term recurrence);
 * call sites ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:360-365 etc.  vin is
 * overwritten (as in the reference).  Returns the number of iterations done. */
int orc_lanc_tridiag_normal(const orc_hnormal *h, double *vin, int nitermax, double *alanc,
                            double *blanc, double threshold);
int orc_lanc_tridiag_normal_ph(const orc_hnormal *h, const orc_model *m, double *vin, int nitermax, double *alanc,
                               double *blanc, double threshold);
int orc_lanc_tridiag_flat(const orc_hflat *h, double *vin, int nitermax, double *alanc,
                          double *blanc, double threshold);

/* ed_total_ud = F ("orbs") variant: per-orbital quantum numbers, H = Hd + sum_a (.. (x) Hup_a (x) ..) +
 * (.. (x) Hdw_a (x) ..) with (1+Nbath)-level factors.  ed_buildh_normal_orbs / spMatVec_normal_orbs
 * (ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:273-496, :652-761).  fac[k], maps[k], dims[k]:
 * k < Norb up-factor of orbital k, k >= Norb down-factor of orbital k - Norb. */
typedef struct {
  int norb, nsorb;
  int nups[ORC_MAXORB], ndws[ORC_MAXORB];
  int64_t dims[2 * ORC_MAXORB], dim;
  int32_t *maps[2 * ORC_MAXORB];
  double *hd;
  orc_csr fac[2 * ORC_MAXORB];
} orc_horbs;
orc_horbs *orc_buildh_normal_orbs(const orc_model *m, const int *nups, const int *ndws);
void orc_horbs_free(orc_horbs *h);
void orc_horbs_sizes(const orc_horbs *h, int64_t out[16]);
void orc_spmatvec_normal_orbs(const orc_horbs *h, const double *v, double *hv);
void orc_horbs_dense(const orc_horbs *h, double *hmat);
int orc_lanc_tridiag_orbs(const orc_horbs *h, double *vin, int nitermax, double *alanc, double *blanc,
                          double threshold);

/* accessors for ctypes */
void orc_hnormal_sizes(const orc_hnormal *h, int64_t out[8]);
void orc_hflat_sizes(const orc_hflat *h, int64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
