/*
 * edigpu.h -- C ABI of the MI355X (gfx950) Lanczos H*v engine for EDIpack.
 *
 * This is the drop-in boundary for EDIpack's Hamiltonian-times-vector hot path.
 * Reference paths below are relative to /root/reference/src/singlesite.
 *
 * In the reference the path sits behind two Fortran procedure pointers,
 *   procedure(dd_sparse_HxV),pointer :: spHtimesV_p     (ED_VARS_GLOBAL.f90:111-122,196)
 *   procedure(cc_sparse_HxV),pointer :: spHtimesV_cc    (ED_VARS_GLOBAL.f90:125-132,197)
 * which build_Hv_sector_<mode> points at spMatVec_* / directMatVec_* and which
 * SciFortran's sp_eigh / sp_lanc_eigh / sp_lanc_tridiag call once per iteration.
 * A thin ISO_C_BINDING module (fortran/edigpu_shim.f90) binds
 * the entry points below and is assigned to those pointers.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; edigpu_last_error()
 *    then holds a message (the Fortran shim does `if(ierr/=0) stop msg`, matching
 *    the reference's `stop "..."` convention, e.g. ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:797).
 *  - all indices are 0-based; row pointers are int64, column ids int32 (the
 *    reference is limited to default 32-bit integers as well: Dim < 2^31).
 *  - complex numbers are interleaved (re,im) doubles == Fortran complex(8).
 *  - a handle describes ONE sector Hamiltonian (the reference keeps exactly one
 *    live at a time in module globals, ED_VARS_GLOBAL.f90:190-195); several
 *    handles may coexist here.
 *  - "host" pointers are ordinary CPU memory; "dev" pointers are HIP device
 *    memory on the handle's device; `stream` is a hipStream_t passed as void*
 *    (NULL = the HIP default stream, as in every HIP API; the host-pointer entry
 *    points use a private stream of the handle and synchronise it before returning).
 *  - There is NO CPU fallback: every entry point fails with an error if no
 *    HIP device is usable.
 */
#ifndef EDIGPU_H
#define EDIGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDIGPU_MAXORB 5
#define EDIGPU_MAXSUNDRY 64
#define EDIGPU_MAXBATH 16

typedef struct edigpu_sector *edigpu_handle;

/* ----------------------------------------------------------------------- */
/* library / device                                                         */
/* ----------------------------------------------------------------------- */
const char *edigpu_last_error(void);
int edigpu_version(void);
/* sizeof(edigpu_model) as this library was compiled: a binding checks its own mirror of the struct against it
 * (ctypes: sizeof(EdigpuModel); Fortran: c_sizeof(edigpu_model_t)) before the first build call. */
int64_t edigpu_model_sizeof(void);
/* number of visible HIP devices (0 and an error if the runtime is unusable) */
int edigpu_device_count(int *count);
/* select the device used by handles created afterwards from this thread
 * (replaces nothing in the reference: it has no device notion; ED_MAIN.f90:164 is
 * where a GPU-enabled ed_solve would call it once per rank). */
int edigpu_init(int device);

/* ----------------------------------------------------------------------- */
/* normal mode: Kronecker-stored sector Hamiltonian                          */
/*   H = diag(Hd) + 1 (x) Hup + Hdw (x) 1 + Hnd                              */
/* replaces spH0d, spH0ups(1), spH0dws(1), spH0nd (ED_VARS_GLOBAL.f90:190-192) as */
/* filled by ed_buildh_normal_main (ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:26-267) */
/* ----------------------------------------------------------------------- */
/*
 * Vector layout (ED_SECTOR.f90:1705-1717): i = iup + idw*DimUp.
 * A shard owns the down-index range [dw_first, dw_first+dw_count), i.e. vector
 * rows [dw_first*DimUp, (dw_first+dw_count)*DimUp) -- the reference's MPI split
 * of the down index (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:129-142).  hd and the rows
 * of nd are given for the owned rows only (as the reference builds them,
 * stored/H_local.f90:1, stored/H_non_local.f90:4); nd column ids are global.
 * up/dw are the full DimUp x DimUp / DimDw x DimDw factors (replicated on every
 * rank in the reference too).  nd_rowptr may be NULL (no spin-exchange /
 * pair-hopping block).  Columns inside a row may come in any order.
 */
int edigpu_normal_create(edigpu_handle *h, int64_t dim_up, int64_t dim_dw, int64_t dw_first,
                         int64_t dw_count, const double *hd, const int64_t *up_rowptr,
                         const int32_t *up_col, const double *up_val, const int64_t *dw_rowptr,
                         const int32_t *dw_col, const double *dw_val, const int64_t *nd_rowptr,
                         const int32_t *nd_col, const double *nd_val);

/* ----------------------------------------------------------------------- */
/* superc / nonsu2 modes: one flat row-CSR matrix                            */
/* replaces spH0 with its loc / non-loc row blocks (ED_SPARSE_MATRIX.f90:26-41,   */
/* :462-470) as filled by ed_buildH_superc_main / ed_buildH_nonsu2_main       */
/* ----------------------------------------------------------------------- */
/*
 * The shard owns global rows [row_first, row_first+nrow_local) of an
 * ncol_global-column matrix.  One CSR with GLOBAL column ids is passed; the
 * library splits it into the local diagonal block (columns inside the shard,
 * multiplied before the exchanged vector arrives) and the non-local block,
 * which is what sp_insert_element does on insertion in the reference.
 * `_d` = real(8) values, `_z` = complex(8) values.
 */
int edigpu_csr_create_d(edigpu_handle *h, int64_t nrow_local, int64_t ncol_global,
                        int64_t row_first, const int64_t *rowptr, const int32_t *col,
                        const double *val);
int edigpu_csr_create_z(edigpu_handle *h, int64_t nrow_local, int64_t ncol_global,
                        int64_t row_first, const int64_t *rowptr, const int32_t *col,
                        const double *val_re_im);

/* ----------------------------------------------------------------------- */
/* sector construction from model parameters                                 */
/* replaces build_sector (ED_SECTOR.f90:165-373) + ed_buildh_*_main: the       */
/* matrices are generated by the library instead of handed over.              */
/* ----------------------------------------------------------------------- */
typedef struct edigpu_model {
  int32_t ed_mode;   /* 0 normal, 1 superc, 2 nonsu2          (ED_INPUT_VARS.f90 ED_MODE)   */
  int32_t bath_type; /* 0 normal, 1 hybrid, 2 replica, 3 general (ED_INPUT_VARS.f90 BATH_TYPE) */
  int32_t norb, nbath, nspin;
  int32_t hfmode;
  double xmu;
  /* Uloc_internal(a), Ust/Jh/Jx/Jp_internal(a,b) (ED_VARS_GLOBAL.f90:216-220), row-major [a][b] */
  double uloc[EDIGPU_MAXORB];
  double ust[EDIGPU_MAXORB * EDIGPU_MAXORB];
  double jh[EDIGPU_MAXORB * EDIGPU_MAXORB];
  double jx[EDIGPU_MAXORB * EDIGPU_MAXORB];
  double jp[EDIGPU_MAXORB * EDIGPU_MAXORB];
  /* impHloc + mfHloc, [ispin][jspin][iorb][jorb][re,im] */
  double hloc[2 * 2 * EDIGPU_MAXORB * EDIGPU_MAXORB * 2];
  /* pair_field(a) (superc) */
  double pair_field[EDIGPU_MAXORB];
  /* dmft_bath%e,v,d,u [ispin][iorb][k]; hybrid: e,d use iorb=0 only */
  double be[2 * EDIGPU_MAXORB * EDIGPU_MAXBATH];
  double bv[2 * EDIGPU_MAXORB * EDIGPU_MAXBATH];
  double bd[2 * EDIGPU_MAXORB * EDIGPU_MAXBATH];
  double bu[2 * EDIGPU_MAXORB * EDIGPU_MAXBATH];
  /* replica / general baths (bath_type 2, 3): the per-replica matrices
   * hbath_tmp(is,js,iorb,jorb,k) = build_Hreplica / build_Hgeneral(dmft_bath%item(k)%lambda)
   * (ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:99-122), [is][js][iorb][jorb][k][re,im]; is,js run over
   * Nspin (normal, nonsu2) or over the Nambu index (superc).  The hybridisations go in bv:
   * replica bv[is][iorb][k] = item(k)%v for every is,iorb; general bv[is][iorb][k] = item(k)%vg(iorb+Norb*(is-1)).
   * be, bd, bu are not read for these bath types. */
  double hb[2 * 2 * EDIGPU_MAXORB * EDIGPU_MAXORB * EDIGPU_MAXBATH * 2];
  /* phonons, all modes (ED_INPUT_VARS.f90:184-198; ED_NORMAL/stored/H_ph.f90, H_e_ph.f90): nph = phonon cut-off
   * Nph (DimPh = Nph + 1; 0 = no phonons), w0_ph, a_ph, g_ph[iorb][jorb] (real symmetric; density couplings g_aa
   * run inside the phonon pass, a general matrix -- GPHFILE in the reference -- as one extra product per phonon block
   * with the operator sum_ab g_ab sum_s c+_as c_bs held as its own sector).  With nph > 0 edigpu_normal_build /
   * edigpu_flat_build / edigpu_direct_build make a handle whose vectors have dim_el * (Nph + 1) elements, index
   * i_el + iph * dim_el.  Whole sectors serve every entry point; with density couplings edigpu_flat_build /
   * edigpu_direct_build also make row shards (the same rows of every phonon block, index i_loc + iph * row_count) for
   * the sharded calls (edigpu_apply_sharded_*, edigpu_lanczos_tridiag_sharded), normal mode shards through the whole-
   * sector handle there. */
  int32_t nph;
  int32_t pad_;
  double w0_ph, a_ph;
  double g_ph[EDIGPU_MAXORB * EDIGPU_MAXORB];
  /* spin_field[iorb][x,y,z] (ED_INPUT_VARS.f90 SPIN_FIELD_X/Y/Z).  Normal mode: the z component enters H_local as
   *   sum_a spin_field(a,3) (n_a,up - n_a,dw)  (ED_NORMAL/stored/H_local.f90:38-42); x, y are not read in this mode.
   *   nonsu2: F.S with all three components (ED_NONSU2/stored/Himp.f90:235-296; the z term as that file's comment
   *   states it -- the code adds it to a variable the blocks before it leave set, see oracle/edipack_oracle_flat.inc).
   *   superc: the reference's files have no such term; the builders refuse a model that sets it.
   * exc_field[4] (EXC_FIELD) = (F_0, F_x, F_y, F_z).  Normal mode: (1) and (4) add (exc(1) +/- exc(4)) c+_a,s c_b,s,
   *   a != b, + for up / - for down (ED_NORMAL/stored/H_up.f90:87-104, H_dw.f90); (2), (3) are not read.  nonsu2: F.T
   *   with all four (Himp.f90:113-228).  superc: refused as above.
   * coulomb_sundry (ED_VARS_GLOBAL.f90 coulomb_sundry(:), read from the umatrix file): nsundry lines
   *   sundry_u[l] cd_i cd_j c_k c_l with sundry_op[l][8] = (orb_i, spin_i, orb_j, spin_j, orb_k, spin_k, orb_l,
   *   spin_l), orbitals 1-based, spin 1 = up / 2 = down, applied right to left as c_l, cd_j, c_k, cd_i.  Normal mode: on
   *   the word of their spin (ED_NORMAL/stored/H_sundry.f90:1-111); lines that change N_up or N_dw are refused.
   *   superc / nonsu2: on the 2 Ns-level word (stored/Hint.f90:127-181), stored and on the fly; superc refuses lines that
   *   change Sz (the reference stops with "impossible operator"). */
  double spin_field[EDIGPU_MAXORB * 3];
  double exc_field[4];
  int32_t nsundry;
  int32_t pad2_;
  int32_t sundry_op[EDIGPU_MAXSUNDRY * 8];
  double sundry_u[EDIGPU_MAXSUNDRY];
} edigpu_model;

/* normal mode sector (N_up, N_dw); the shard owns down-indices [dw_first, dw_first+dw_count)
 * (dw_count < 0: all).  Equivalent to build_Hv_sector_normal(isector) with ed_sparse_H=T
 * (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:31-206). */
int edigpu_normal_build(edigpu_handle *h, const edigpu_model *model, int nup, int ndw,
                        int64_t dw_first, int64_t dw_count);
/* The same sector with complex algebra: the reference's -D_CMPLX_NORMAL build (CMakeLists.txt:43-48), where
 * spH0ups / spH0dws and the vectors of ed_mode=normal are complex(8) and the imaginary parts of impHloc(1,1,a,b)
 * and of the replica / general bath matrices enter the hops (ED_NORMAL/stored/H_up.f90:8-50, H_dw.f90).  The handle
 * is complex (edigpu_apply_z, interleaved re/im vectors, real alpha / beta) and holds the whole sector.  Inside,
 * H = S + iA (S: the real parts = the ordinary build; A: the antisymmetric hop matrices of the imaginary parts) is held
 * as ONE real sector on the doubled up index 2 iup + (re | im) -- the interleaved complex vectors are its real
 * vectors, H' = S (x) 1 + A (x) [[0,-1],[1,0]] -- so the kernels, the panel-major layout and the fused recurrence of
 * the real sectors do the complex product in one pass over 2 Dim elements (measured 1.9-2.6x a real product, complex
 * Lanczos step 2.0-2.4x).  More than 16 factored terms (complex replica matrices with many imaginary inter-orbital
 * hops) or EDIGPU_CMPLX_FOURPRODUCTS=1: four products of the two real handles on planar work vectors (3.4-4.9x).
 * No phonons. */
int edigpu_normal_build_z(edigpu_handle *h, const edigpu_model *model, int nup, int ndw);
/* superc sector Sz / nonsu2 sector Ntot; the shard owns rows [row_first, row_first+row_count)
 * (row_count < 0: all).  Equivalent to build_Hv_sector_superc / _nonsu2
 * (ED_SUPERC/ED_HAMILTONIAN_SUPERC.f90:33-135, ED_NONSU2/ED_HAMILTONIAN_NONSU2.f90:31-128). */
int edigpu_flat_build(edigpu_handle *h, const edigpu_model *model, int sector, int64_t row_first,
                      int64_t row_count);

/*
 * nonsu2 sector of JZ_BASIS=T (ED_INPUT_VARS.f90:757; build_sector, ED_SECTOR.f90:289-350): the states with Ntot = ntot
 * and twoJz = (Nup - Ndw) + twoLz, twoLz = sum over the levels iorb + Norb*ibath of 2 Lzdiag(iorb) (n_up + n_dw),
 * Lzdiag = [-1, +1, 0] (ED_VARS_GLOBAL.f90:283) -- three orbitals, replica / general bath (or Nbath = 1), the level
 * order that labelling assumes.  The Hamiltonian is the nonsu2 one (ed_buildH_nonsu2_main) on that map; a model whose
 * terms leave the sector (Jz not conserved: the reference's binary_search would fail) is refused.  The up words that go
 * with a down word all have one (occupation, Lz), so a state's row is still the sum of two table entries and the
 * sector has the same three forms as an Ntot sector: the stored image generated on the device (edigpu_flat_build_jz;
 * EDIGPU_FLAT_HOSTBUILD=1: host CSR), the on-the-fly product (edigpu_direct_build_jz), and edigpu_apply_op_flat /
 * edigpu_apply_cops_flat between two Jz sectors (the destination must be (Ntot +- 1, twoJz +- (spin + 2 Lz of the level))).
 * apply / Lanczos / eigensolver entry points as for edigpu_flat_build, shards by rows.  edigpu_sector_map_jz returns the
 * map (as edigpu_sector_map).
 */
int edigpu_flat_build_jz(edigpu_handle *h, const edigpu_model *model, int ntot, int twojz, int64_t row_first,
                         int64_t row_count);
int edigpu_direct_build_jz(edigpu_handle *h, const edigpu_model *model, int ntot, int twojz, int64_t row_first,
                           int64_t row_count);
int edigpu_sector_map_jz(const edigpu_model *model, int ntot, int twojz, int32_t *map, int64_t *n);

/*
 * On-the-fly ("direct", ED_SPARSE_H=F) sector: nothing of H is stored; every H*v regenerates the
 * matrix elements from the sector map, as directMatVec_nonsu2_main / directMatVec_MPI_nonsu2_main
 * (ED_NONSU2/ED_HAMILTONIAN_NONSU2_DIRECT_HxV.f90:22-252) and directMatVec_[MPI_]superc_main
 * (ED_SUPERC/ED_HAMILTONIAN_SUPERC_DIRECT_HxV.f90:22-311) do.  superc: sector = Sz, nonsu2: sector = Ntot.
 * Sharded handles need the gathered vector for the whole product (the reference gathers first, :220-223):
 * edigpu_apply_local_dev only zeroes hv, edigpu_apply_remote_dev does the work.
 */
int edigpu_direct_build(edigpu_handle *h, const edigpu_model *model, int sector, int64_t row_first,
                        int64_t row_count);

/*
 * ed_total_ud = F ("orbs") normal-mode sector: quantum numbers (Nup_a, Ndw_a) per orbital, bath_type = normal,
 * no Jx/Jp.  The vector is the tensor [iup_1..iup_Norb, idw_1..idw_Norb] (first index fastest, state2indices,
 * ED_AUX_FUNX.f90) and H = Hd + sum over the 2*Norb axes of one (1+Nbath)-level factor each.
 * Takes the place of build_Hv_sector_normal with ed_total_ud=F -> ed_buildh_normal_orbs
 * (ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:273-496) and spMatVec_normal_orbs (:652-761).
 * edigpu_orbs_build holds the whole sector; edigpu_orbs_build_rows below is the row shard of the MPI variant
 * (:932-1082).  All apply / Lanczos entry points work on a whole-sector handle.
 */
int edigpu_orbs_build(edigpu_handle *h, const edigpu_model *model, const int32_t *nups, const int32_t *ndws);
/* The same sector as a row shard (rows [row_first, row_first + row_count) of the tensor-ordered vector; row_count < 0:
 * to the end), for spMatVec_mpi_normal_orbs (ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:932-1082) in the
 * all-gather form: edigpu_apply_local_dev zeroes hv, edigpu_apply_remote_dev computes the shard's rows from the
 * gathered vector (the factored tables are O(sum of the axis sizes), every rank holds them whole). */
int edigpu_orbs_build_rows(edigpu_handle *h, const edigpu_model *model, const int32_t *nups, const int32_t *ndws,
                           int64_t row_first, int64_t row_count);
/* hand-over of the reference's own arrays: spH0d (dim values) and spH0ups(1:Norb), spH0dws(1:Norb) as ONE
 * CSR with the rows of the 2*Norb factors stacked (rowptr of sum(dims)+1 entries, columns local to their
 * factor, 0-based).  dims[k]: k < Norb = DimUps(k+1), k >= Norb = DimDws(k-Norb+1). */
int edigpu_orbs_create(edigpu_handle *h, int naxes, const int64_t *dims, const double *hd,
                       const int64_t *fac_rowptr, const int32_t *fac_col, const double *fac_val);

/* sector dimensions (get_normal/superc/nonsu2_sector_dimension, ED_SETUP.f90:998-1033) */
int edigpu_sector_dim(const edigpu_model *model, int q1, int q2, int64_t *dim);

/* The sector map build_sector leaves in Hsector%H(k)%map (ED_SECTOR.f90:165-373), bit-exact: normal mode (q1, q2) =
 * (Nup, Ndw), which = 0 the up map (DimUp ascending integers with Nup bits set), 1 the down map; superc (q1 = Sz) /
 * nonsu2 (q1 = Ntot): the one map of states iup + idw 2^Ns (idw outer, iup inner), q2 and which ignored.  Host-only
 * (no device needed).  *n: in = capacity of map (ignored when map == NULL), out = number of states. */
int edigpu_sector_map(const edigpu_model *model, int q1, int q2, int which, int32_t *map, int64_t *n);

/* ----------------------------------------------------------------------- */
/* queries                                                                    */
/* ----------------------------------------------------------------------- */
/* info[0]=global dim, [1]=local rows (vecDim_Hv_sector_*), [2]=first local row,
 * [3]=is_complex, [4]=kind (0 normal Kronecker, 1 flat CSR, 2 direct), [5]=DimUp, [6]=DimDw,
 * [7]=nnz(up)+nnz(dw) or nnz(loc), [8]=nnz(nd) or nnz(nonloc), [9]=device id */
int edigpu_info(edigpu_handle h, int64_t info[10]);
/* Device image of a normal-mode handle (diagnostics, tests): image[0] = 1 when the kernels run on the factored tables
 * (separable diagonal + Hnd as a sum of signed partial permutations; library-built sectors, and hand-over sectors
 * whose arrays edigpu_normal_create could factor), 0 for the explicit image (spH0d + spH0nd as given);
 * [1] = Hnd terms, [2] = diagonal classes, [3] = panel sweep variant (0 one column per lane, 1 two, 2 LDS-tiled),
 * [4] = columns per panel of the panel-major vector layout the device-resident Lanczos loops of this sector run on
 * (0: natural layout; DESIGN.md "panel-major vectors"), [5] = 1 when those loops run the impurity-block kernels
 * (16-column padded panels), 2 when their rows are staged in two halves (rows longer than the LDS), else 0. */
int edigpu_image_info(edigpu_handle h, int32_t image[6]);
/* algorithmic bytes of one H*v in the reference's storage format */
int edigpu_algorithmic_bytes(edigpu_handle h, double *bytes_hv, double *bytes_lanczos_step);
/* copy the built matrices back to the host (tests: compare with the oracle).
 * Any pointer may be NULL.  Normal kind: hd[local rows], up/dw/nd CSR (nd over local rows).
 * Sizes come from edigpu_info. */
int edigpu_normal_export(edigpu_handle h, double *hd, int64_t *up_rowptr, int32_t *up_col,
                         double *up_val, int64_t *dw_rowptr, int32_t *dw_col, double *dw_val,
                         int64_t *nd_rowptr, int32_t *nd_col, double *nd_val);
/* flat kind: one CSR over the local rows with global columns (loc and non-loc merged) */
int edigpu_csr_export(edigpu_handle h, int64_t *rowptr, int32_t *col, double *val);

/* ----------------------------------------------------------------------- */
/* H*v                                                                        */
/* ----------------------------------------------------------------------- */
/*
 * Callback-compatible product with host vectors: same contract as
 * dd_sparse_HxV / cc_sparse_HxV (ED_VARS_GLOBAL.f90:111-132): Hv is overwritten,
 * v is untouched, nloc must equal the local row count.  Single-shard handles
 * only (a sharded handle needs the gathered vector: use the _dev entry points).
 * Replaces spMatVec_normal_main (ED_NORMAL/..._STORED_HxV.f90:517-650),
 * spMatVec_superc_main (ED_SUPERC/..._STORED_HxV.f90:312-362),
 * spMatVec_nonsu2_main (ED_NONSU2/..._STORED_HxV.f90:194-209).
 */
int edigpu_apply_d(edigpu_handle h, int64_t nloc, const double *v_host, double *hv_host);
int edigpu_apply_z(edigpu_handle h, int64_t nloc, const double *v_host, double *hv_host);

/*
 * Device-resident product.  v_full_dev holds the WHOLE vector (global dim
 * elements; for a single shard that is just the vector), hv_dev the local rows.
 * Enqueued on `stream`, no host synchronisation.
 */
int edigpu_apply_dev(edigpu_handle h, const void *v_full_dev, void *hv_dev, void *stream);
/*
 * Two-phase form for sharded handles, mirroring spMatVec_mpi_* :
 *  _local  : the terms that need only the shard's own slice of v (normal: diagonal
 *            + up part, ..._STORED_HxV.f90:800-832; flat: the `loc` block, ED_SUPERC/
 *            ..._STORED_HxV.f90:395-403).  Overwrites hv.  Runs while the exchange is in flight.
 *  _remote : the terms that need the gathered vector (normal: down part + Hnd,
 *            :847-927; flat: non-local block, :423-430).  Accumulates into hv.
 */
int edigpu_apply_local_dev(edigpu_handle h, const void *v_local_dev, void *hv_dev, void *stream);
int edigpu_apply_remote_dev(edigpu_handle h, const void *v_full_dev, void *hv_dev, void *stream);

/*
 * Transposed exchange for normal mode on N > 1 GPUs.  Replaces the two vector_transpose_MPI calls per product
 * of spMatVec_mpi_normal_main (ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:834-866; the transpose itself
 * ED_NORMAL/ED_HAMILTONIAN_NORMAL_COMMON.f90:66-167) AND its allgather_vector_MPI for spH0nd (:906-927).
 * Every rank builds the WHOLE sector (edigpu_normal_build with dw_count < 0: only O(DimUp + DimDw) tables) and
 * owns, of the vector, q = ceil(DimDw / N) down rows in the row phase and pcol = ceil(DimUp / N) up columns of
 * ALL rows in the column phase:
 *   edigpu_transpose_pack          row shard v_rows[nrows][DimUp] -> send[N][q][pcol + 2 halo]
 *   (equal-split all-to-all, e.g. torch.distributed.all_to_all_single over RCCL)
 *                                  -> recv[N*q][pcol + 2 halo] = the column shard, already in row order
 *   edigpu_normal_apply_rows_dev   hv_rows  = (Hd + 1 (x) Hup) v on the row shard (runs during the exchange)
 *   edigpu_normal_apply_cols_dev   hv_cols  = (Hdw (x) 1 + Hnd) v on the column shard (same layout as recv;
 *                                  only the col_count owned columns of the DimDw real rows are written;
 *                                  row_stride = pcol + 2 halo, also on the last rank, which owns fewer columns)
 *   (equal-split all-to-all back: block r of hv_cols goes to rank r, no packing)
 *   edigpu_transpose_unpack_add    hv_rows += the received blocks
 * halo (edigpu_normal_transpose_info): the spin-exchange / pair-hopping terms of Hnd move one electron between
 * impurity levels of the up configuration, i.e. reach a column at most `halo` away -- a few columns travel
 * twice instead of the whole vector being gathered.  All calls are enqueued on `stream` without host
 * synchronisation.  Handles holding explicit spH0nd arrays or phonons are refused (use the all-gather form).
 */
int edigpu_normal_transpose_info(edigpu_handle h, int32_t *halo);
int edigpu_normal_apply_rows_dev(edigpu_handle h, int64_t dw_first, int64_t dw_count, const void *v_rows_dev,
                                 void *hv_rows_dev, void *stream);
int edigpu_normal_apply_cols_dev(edigpu_handle h, int64_t col_first, int64_t col_count, int64_t row_stride,
                                 int32_t halo, const void *w_cols_dev, void *hv_cols_dev, void *stream);
int edigpu_transpose_pack(int64_t dim_up, int64_t nrows, int64_t q, int32_t world, int64_t pcol, int32_t halo,
                          const void *v_rows_dev, void *send_dev, void *stream);
int edigpu_transpose_unpack_add(int64_t dim_up, int64_t nrows, int64_t q, int32_t world, int64_t pcol,
                                int32_t halo, const void *recv_dev, void *hv_rows_dev, void *stream);
/*
 * The same two steps fused with the vector updates of lanczos_iteration (one all-reduce per step instead of
 * two, three passes over the shard less):
 *   _rotate_pack      first = 0: alpha = ab[0], beta^2 = ab[1] - alpha^2 (the all-reduced <v|w>, <w|w> of the
 *                     previous step); w -= alpha v, (v, w) <- (w / beta, -beta v), new v -> send buffer.
 *                     first = 1: only packs v.
 *   _unpack_add_dot2  w += hv_rows + received blocks; out2 = this rank's (<v|w>, <w|w>);
 *                     work: edigpu_vec_work_doubles() doubles.
 * beta^2 = <w|w> - alpha^2 loses digits when beta << |alpha|; the caller checks the history and repeats the
 * run with the exact two-reduction recurrence in that case (edipack_amd/sharding.py).
 */
int edigpu_transpose_rotate_pack(int32_t first, int64_t dim_up, int64_t nrows, int64_t q, int32_t world,
                                 int64_t pcol, int32_t halo, void *vin_dev, void *vout_dev, const void *ab_dev,
                                 void *send_dev, void *stream);
int edigpu_transpose_unpack_add_dot2(int64_t dim_up, int64_t nrows, int64_t q, int32_t world, int64_t pcol,
                                     int32_t halo, const void *vin_dev, void *vout_dev, const void *tmp_dev,
                                     const void *back_dev, void *out2_dev, void *work_dev, void *stream);

/* ----------------------------------------------------------------------- */
/* device-resident Lanczos                                                    */
/* ----------------------------------------------------------------------- */
/*
 * Partial tridiagonalisation with the vector kept in HBM: replaces the
 * sp_lanc_tridiag(spHtimesV_p, vvinit, alanc, blanc) call inside
 * tridiag_Hv_sector_* (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:360-365,
 * ED_SUPERC/ED_HAMILTONIAN_SUPERC.f90:263-268, ED_NONSU2/ED_HAMILTONIAN_NONSU2.f90:235-240).
 * vin_host (local rows; real or complex per the handle) is normalised internally
 * exactly as lanczos_iteration does on its first step; alanc/blanc have nlanc
 * entries, blanc[0] is left 0 (the consumer reads blanc(2:N), ED_GF_NORMAL.f90:410-411).
 * The recurrence stops early when |beta| < threshold; *niter_done is the number
 * of alpha values produced.  Single-shard handles.
 */
int edigpu_lanczos_tridiag(edigpu_handle h, const double *vin_host, int nlanc, double *alanc,
                           double *blanc, double threshold, int *niter_done);
/*
 * Lowest `neigen` eigenpairs by thick-restart Lanczos with full re-orthogonalisation on an ncv-dimensional,
 * device-resident Krylov basis: the role ARPACK plays behind sp_eigh in the reference's
 * default LANC_METHOD=arpack path (ED_NORMAL/ED_DIAG_NORMAL.f90:179-196, ncv = lanc_ncv_factor*Neigen +
 * lanc_ncv_add; ED_SUPERC / ED_NONSU2 likewise).  ncv <= 0 picks max(2*neigen+10, 20); ncv is capped at 128 and
 * at the sector dimension.  v0 (host or device, may be NULL = seeded random), evecs (host or device, may be
 * NULL): neigen vectors of the sector's length, consecutive.  nconv = number of leading eigenpairs whose
 * residual |H x - theta x| <= tol * max(|theta|, 1).  As with ARPACK, a degenerate eigenvalue is found with
 * the multiplicity the start vector (and rounding) exposes.
 */
int edigpu_lanczos_eigh_multi(edigpu_handle h, int neigen, int ncv, double tol, int maxrestart, const double *v0,
                              double *evals, double *evecs, int *nconv, int *nmatvec);

/*
 * Device-resident neighbours of the tridiagonalisation: the seed of a Green's-function
 * tridiagonalisation is c / c^+ applied to an eigenvector (apply_op_C / apply_op_CDG, ED_SECTOR.f90:465-536,
 * called from ED_NORMAL/ED_GF_NORMAL.f90:141-175 on the master rank, then scattered).  Here it maps a
 * device vector of one normal-mode sector to a device vector of the neighbouring sector; both handles
 * must come from edigpu_normal_build (whole sectors).  iorb 0-based, ispin 0 = up / 1 = down, create != 0 = c^+.
 * The down-spin sign counts down electrons only, as the reference does.  Returns after the stream finished.
 */
int edigpu_apply_op_normal(edigpu_handle src, edigpu_handle dst, const double *v_src_dev, double *v_dst_dev,
                           int iorb, int ispin, int create, void *stream);
/*
 * apply_Cops (ED_SECTOR.f90:839-960; the seeds of the off-diagonal Green's functions, ED_NORMAL/ED_GF_NORMAL.f90:
 * 216-261: (c^+_a + c^+_b)|gs>, (c_a + c_b)|gs>): v_dst = sum_s coef[s] * O_s v_src with O_s = c^+ (create[s] > 0)
 * or c (create[s] <= 0; the reference's Os = +1 / -1) of orbital iorb[s] (0-based), spin ispin[s] (0 up, 1 down).
 * Every term must lead from the source sector to the same destination sector.  Real coefficients (the complex
 * combinations c_a + i c_b belong to the _CMPLX_NORMAL build).
 */
int edigpu_apply_cops_normal(edigpu_handle src, edigpu_handle dst, const double *v_src_dev, double *v_dst_dev,
                             int nops, const double *coef, const int32_t *create, const int32_t *iorb,
                             const int32_t *ispin, void *stream);
/* the same for superc / nonsu2 sectors (handles from edigpu_flat_build / edigpu_direct_build, complex vectors):
 * superc sectors are labelled by Sz, so c^+_up / c_dw lead to Sz+1 and c^+_dw / c_up to Sz-1; nonsu2 by Ntot.
 * The sign counts every occupied level below the operator's level in the 2*Ns-bit state (up levels first). */
int edigpu_apply_op_flat(edigpu_handle src, edigpu_handle dst, const double *v_src_dev, double *v_dst_dev,
                         int iorb, int ispin, int create, void *stream);
/* apply_Cops for superc / nonsu2 sectors (ED_SECTOR.f90:654-839: the six two-operator combinations behind the NONSU2
 * exciton order parameters, ED_NONSU2/ED_OBSERVABLES_NONSU2.f90:325-425, and the off-diagonal Green's-function seeds):
 * v_dst = sum_s coef[s] * O_s v_src with COMPLEX coefficients coef_re_im[2 s], [2 s + 1] (e.g. c_a,up - i c_b,dw). */
int edigpu_apply_cops_flat(edigpu_handle src, edigpu_handle dst, const double *v_src_dev, double *v_dst_dev, int nops,
                           const double *coef_re_im, const int32_t *create, const int32_t *iorb, const int32_t *ispin,
                           void *stream);
/* edigpu_lanczos_tridiag with the seed in device memory (e.g. the output of edigpu_apply_op_normal; it must be
 * complete when the call is made) and norm2 = <vin|vin> returned as tridiag_Hv_sector_* does.
 * edigpu_lanczos_eigh likewise accepts device pointers for v0 and for the eigenvector. */
int edigpu_lanczos_tridiag_dev(edigpu_handle h, const double *vin_dev, int nlanc, double *alanc, double *blanc,
                               double threshold, int *niter_done, double *norm2);

/*
 * Lowest eigenpair by plain Lanczos (lanc_method="lanczos": sp_lanc_eigh call
 * sites ED_NORMAL/ED_DIAG_NORMAL.f90:206-214): iterate until the lowest Ritz value
 * moves by less than tol (checked every `check_every` steps) or nitermax, then
 * rebuild the Ritz vector with a second pass.  evec_host may be NULL.
 */
int edigpu_lanczos_eigh(edigpu_handle h, int nitermax, double tol, int check_every,
                        const double *v0_host, double *eval, double *evec_host,
                        int *niter_done);

/*
 * Vector kernels of the three-term recurrence on caller-owned device buffers (the sharded N>1 loop:
 * the scalars cross ranks with a 1-element all-reduce, everything else stays on the device).
 * n counts doubles (2 per complex element).  Scalars are device pointers; no host synchronisation.
 * They restate the elementwise part of SciFortran's lanczos_iteration (see edigpu_lanczos_tridiag).
 *   edigpu_vec_rotate   : (vin, vout) <- (vout/beta, -beta*vin),  beta = sqrt(*beta2_dev)
 *   edigpu_vec_add_dot  : vout += tmp ; *out_dev = sum(vin*vout)          (local partial of alpha)
 *   edigpu_vec_axpy_nrm2: vout -= (*alpha_dev)*vin ; *out_dev = sum(vout^2) (local partial of beta^2)
 *   edigpu_vec_scale    : v *= 1/sqrt(*nrm2_dev)
 * `work_dev` must hold at least edigpu_vec_work_doubles() doubles.
 */
int edigpu_vec_work_doubles(void);
int edigpu_vec_rotate(int64_t n, double *vin_dev, double *vout_dev, const double *beta2_dev, void *stream);
int edigpu_vec_add_dot(int64_t n, const double *vin_dev, double *vout_dev, const double *tmp_dev,
                       double *out_dev, double *work_dev, void *stream);
int edigpu_vec_axpy_nrm2(int64_t n, const double *vin_dev, double *vout_dev, const double *alpha_dev,
                         double *out_dev, double *work_dev, void *stream);
int edigpu_vec_scale(int64_t n, double *v_dev, const double *nrm2_dev, void *stream);
/* One-reduction form of the same step (one all-reduce of two doubles per step instead of two of one):
 *   edigpu_vec_add_dot2     w += tmp; out2 = this rank's (<v|w>, <w|w>)
 *   edigpu_vec_rotate_lazy  ab = the summed (<v|w>, <w|w>) of the previous step: alpha = ab[0],
 *                           beta^2 = ab[1] - alpha^2; (v, w) <- ((w - alpha v) / beta, -beta v)
 * The caller checks the history for cancellation in beta^2 and falls back to the two-reduction calls above. */
int edigpu_vec_rotate_lazy(int64_t n, double *vin_dev, double *vout_dev, const double *ab_dev, void *stream);
int edigpu_vec_add_dot2(int64_t n, const double *vin_dev, double *vout_dev, const double *tmp_dev, double *out2_dev,
                        double *work_dev, void *stream);

/* ----------------------------------------------------------------------- */
/* N > 1 inside the library: communicator, sharded product, sharded Lanczos    */
/* ----------------------------------------------------------------------- */
/*
 * One process (MPI rank) per GPU.  The communicator takes the place of MpiComm in the reference's distributed
 * products and in SciFortran's MPI Lanczos driver: it is RCCL over xGMI (edigpu_comm_create; the unique id is made
 * on rank 0 with edigpu_comm_unique_id and broadcast by the host, e.g. MPI_Bcast of 128 bytes), or a host-staged
 * transport through POSIX shared memory for ranks of one node that share a GPU (edigpu_comm_create_shm: tests of the
 * N > 1 data flow on a one-GPU box, hosts without RCCL).  RCCL is loaded at run time (dlopen: the copy the process
 * already mapped, else /opt/rocm's).
 *
 * Shards (edigpu_shard_plan): rank r owns units [r q, min((r+1) q, units)), q = ceil(units / world); units = DimDw
 * down rows (normal mode) or Dim rows (superc / nonsu2).  The reference puts the remainder on the first / last
 * ranks instead (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:129-142, ED_SUPERC/ED_HAMILTONIAN_SUPERC.f90:82-88); the choice is
 * not observable in the results.
 *
 * Handles: normal mode with the transposed exchange -- every rank builds the WHOLE sector (edigpu_normal_build with
 * dw_count < 0: only O(DimUp + DimDw) tables) and the calls below run two all-to-alls per product, the first beside
 * the row half of H*v (spMatVec_mpi_normal_main, ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:765-929); every other
 * case -- hand-over arrays, superc / nonsu2 stored or on the fly -- each rank builds ITS shard (first / count of
 * edigpu_shard_plan) and the calls run one all-gather of the vector beside the shard-local block
 * (spMatVec_mpi_superc_main, ED_SUPERC/ED_HAMILTONIAN_SUPERC_STORED_HxV.f90:366-432; spMatVec_mpi_nonsu2_main,
 * ED_NONSU2/ED_HAMILTONIAN_NONSU2_STORED_HxV.f90:213-267; directMatVec_MPI_*).  All ranks call collectively.
 */
typedef struct edigpu_comm_s *edigpu_comm;
#define EDIGPU_UNIQUE_ID_BYTES 128
int edigpu_shard_plan(int64_t units, int32_t world, int32_t rank, int64_t *first, int64_t *count, int64_t *q);
/* Which exchange a sharded call on (h, c) takes: info[0] = 0 all-gather (the handle is the rank's shard), 1 transposed
 * exchange on column blocks with halo columns (spMatVec_mpi_normal_main's vector_transpose_MPI,
 * ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:834-866), 2 the same exchange on the padded 16-column panels of the
 * local-block kernels (no packing, no halo; whole normal-mode sectors with that image, EDIGPU_SHARD_GENERIC=1 switches
 * it off); info[1] = q (units per rank), info[2] = columns (1) / panels (2) per rank, info[3] = halo columns. */
int edigpu_shard_info(edigpu_handle h, edigpu_comm c, int32_t info[4]);
/* Host-only (no GPU needed): the index maps of the transposed exchange of normal mode exactly as the library's kernels
 * compute them (reference: vector_transpose_MPI, ED_NORMAL/ED_HAMILTONIAN_NORMAL_COMMON.f90:66-167).
 *   send_map: src[e], e < world*q*(pcol+2*halo) = element i*dim_up+col of the rank's row shard that send slot e
 *             carries, or -1 for a zero (rows past nrows, columns outside the sector)
 *   back_map: slot[i*dim_up+col] = where the column half of H*v for that element sits in the buffer the second
 *             all-to-all delivers
 * For hosts that stage the exchange themselves; the CPU test suite exchanges through them between gloo ranks. */
int edigpu_exchange_send_map(int64_t dim_up, int64_t nrows, int64_t q, int32_t world, int64_t pcol, int32_t halo,
                             int64_t *src);
int edigpu_exchange_back_map(int64_t dim_up, int64_t nrows, int64_t q, int32_t world, int64_t pcol, int32_t halo,
                             int64_t *slot);
int edigpu_comm_unique_id(void *id128);
/* id128 may be NULL when world == 1 (no RCCL communicator is made then) */
int edigpu_comm_create(edigpu_comm *c, int32_t rank, int32_t world, const void *id128);
/* name: POSIX shared-memory name common to the ranks; slot_bytes >= the largest message a rank sends in one collective
 * (8 * the padded shard length * world for the transposed exchange) */
int edigpu_comm_create_shm(edigpu_comm *c, int32_t rank, int32_t world, const char *name, int64_t slot_bytes);
int edigpu_comm_info(edigpu_comm c, int32_t *rank, int32_t *world, int32_t *kind /* 0 RCCL, 1 shared memory */);
int edigpu_comm_destroy(edigpu_comm c);
/*
 * The (Nloc, v, Hv) contract of spMatVec_mpi_* / directMatVec_MPI_* on shards (ED_VARS_GLOBAL.f90:111-132 with
 * Nloc = vecDim_Hv_sector_*): v_shard_host / hv_shard_host hold this rank's nloc elements; the exchange happens
 * inside.  Assigned to spHtimesV_p / spHtimesV_cc in a -D_MPI build (fortran/edigpu_shim.f90).
 * A complex normal-mode handle (edigpu_normal_build_z, whole sector on every rank) is served through its doubled real
 * sector with the transposed exchange: nloc = this rank's down rows * DimUp complex elements.
 * A normal-mode phonon handle (nph > 0, whole sector on every rank, density couplings g_ph(a,a)): the shard is
 * (Nph + 1) blocks of this rank's down rows, i = iup + idw_local * DimUp + iph * DimUp * count, the layout of
 * spMatVec_mpi_normal_main (ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:820-904); every block goes through the
 * exchange on its own, the phonon and electron-phonon terms are local.  superc / nonsu2 phonon handles: row shards
 * (edigpu_flat_build / edigpu_direct_build with the first / count of edigpu_shard_plan over the ELECTRONIC dimension),
 * nloc = count * (Nph + 1), one all-gather per phonon block.  A general g_ph(a,b) is refused here (one GPU).
 */
int edigpu_apply_sharded_d(edigpu_handle h, edigpu_comm c, int64_t nloc, const double *v_shard_host,
                           double *hv_shard_host);
int edigpu_apply_sharded_z(edigpu_handle h, edigpu_comm c, int64_t nloc, const double *v_shard_host,
                           double *hv_shard_host);
/*
 * sp_lanc_tridiag(MpiComm, spHtimesV_p, vvloc, alanc, blanc) with the vectors resident in HBM (call sites
 * ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:357-365 and the superc / nonsu2 twins): vin_shard (host or device memory) is
 * this rank's slice of the seed as scatter_vector_MPI leaves it (ED_AUX_FUNX.f90:598-692); it is normalised inside and
 * *norm2 = <vin|vin> over all ranks (NULL allowed).  Per step the ranks exchange the vector (all-to-all pair or
 * all-gather, on a side stream) and all-reduce three doubles; nothing else crosses a link and -- with RCCL -- the
 * host is not synchronised until the coefficients are read back.  Every rank receives the same alanc / blanc.
 */
int edigpu_lanczos_tridiag_sharded(edigpu_handle h, edigpu_comm c, const double *vin_shard, int nlanc, double *alanc,
                                   double *blanc, double threshold, int *niter_done, double *norm2);
/*
 * sp_eigh(MpiComm, spHtimesV_p, eval, evec, Nblock, Nitermax, tol) -- the default LANC_METHOD=arpack path under MPI -- and
 * sp_lanc_eigh(MpiComm, spHtimesV_p, eval, evec, Nitermax) with every vector a device-resident shard (call sites
 * ED_NORMAL/ED_DIAG_NORMAL.f90:179-214, ED_SUPERC/ED_DIAG_SUPERC.f90:161-196, ED_NONSU2/ED_DIAG_NONSU2.f90:179-214): the
 * thick-restart solver of edigpu_lanczos_eigh_multi on this rank's shard, the product through the sharded exchange, the
 * Gram-Schmidt coefficients (j numbers per step, the reduction SciFortran's MPI drivers do with MPI_AllReduce) and the
 * norms through all-reduces.  v0_shard (host or device, NULL = seeded random) and evecs_shard (host or device, NULL
 * allowed; neigen consecutive shards) hold this rank's nloc elements as scatter_vector_MPI / es_return_dvector leave
 * them (ED_EIGENSPACE.f90:723-793).  Every rank receives the same eigenvalues.  Handles as for edigpu_apply_sharded_*.
 */
int edigpu_lanczos_eigh_multi_sharded(edigpu_handle h, edigpu_comm c, int neigen, int ncv, double tol, int maxrestart,
                                      const double *v0_shard, double *evals, double *evecs_shard, int *nconv,
                                      int *nmatvec);
int edigpu_lanczos_eigh_sharded(edigpu_handle h, edigpu_comm c, int nitermax, double tol, const double *v0_shard,
                                double *eval, double *evec_shard, int *nmatvec);
/*
 * apply_op_C / apply_op_CDG / apply_Cops on shards: the Green's-function seeds the reference builds on the master rank
 * from the gathered eigenvector and scatters again (ED_NORMAL/ED_GF_NORMAL.f90:141-175, ED_AUX_FUNX.f90:598-692).  Here
 * v_src_shard (host or device: this rank's shard of a vector of sector `src`) is all-gathered on the device and every
 * rank computes ITS shard of the destination vector; v_dst_shard (host or device) receives the nloc elements of this
 * rank's shard of sector `dst`.  Handles as for the other sharded calls (normal mode: whole sectors from
 * edigpu_normal_build; superc / nonsu2: this rank's row shards); coefficients as in edigpu_apply_cops_flat (normal mode:
 * imaginary parts must be zero).
 */
int edigpu_apply_cops_sharded(edigpu_handle src, edigpu_handle dst, edigpu_comm c, const double *v_src_shard,
                              double *v_dst_shard, int nops, const double *coef_re_im, const int32_t *create,
                              const int32_t *iorb, const int32_t *ispin);
/* bench.py --gpus N: `warmup` + `steps` sharded Lanczos steps on a seeded random vector; wall time per step between
 * two collectives that act as barriers, and the bytes this rank sends per product */
int edigpu_lanczos_bench_sharded(edigpu_handle h, edigpu_comm c, int warmup, int steps, double *ms_per_step,
                                 int64_t *exchange_bytes);

/* bench.py --gpus N: *rccl_ranks = ncclCommCount of the communicator (0: shared-memory transport, -1: not available),
 * *ms_exchange = average duration of the collectives of one product + step alone (the two all-to-alls or the all-gather,
 * and the 3-double all-reduce), HIP events on the communicator's stream; either pointer may be NULL. */
int edigpu_exchange_bench(edigpu_handle h, edigpu_comm c, int steps, int32_t *rccl_ranks, double *ms_exchange);

/* ----------------------------------------------------------------------- */
/* Per-solve cache of sector handles                     */
/* ----------------------------------------------------------------------- */
/*
 * The reference rebuilds the sector Hamiltonian in every tridiag_Hv_sector_* call (build_Hv_sector_* ...
 * delete_Hv_sector_*, ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:351-368): one build per Green's-function channel and state.
 * edigpu_cache_get returns the handle of (model, kind, sector) -- building it on a miss with edigpu_normal_build
 * (kind 0: q1, q2 = Nup, Ndw), edigpu_flat_build (1: q1 = Sz / Ntot), edigpu_direct_build (2) or edigpu_normal_build_z
 * (3), whole sectors -- and keeps it on the device for the next request.  The handle belongs to the cache: do not
 * edigpu_destroy it.  Entries are evicted least recently used once their device memory (measured at build time + the
 * Lanczos workspace) exceeds max_device_bytes; the two most recently returned handles are never evicted (the
 * Green's-function loop holds the eigenstate's sector and the target sector at once).  The key includes every byte of
 * the model: a new bath is a new key; edigpu_cache_clear drops everything (e.g. at the start of a DMFT iteration).
 * stats = hits, misses, evictions, device bytes in use, entries.  Thread-safe.
 */
typedef struct edigpu_cache_s *edigpu_cache;
int edigpu_cache_create(edigpu_cache *c, int64_t max_device_bytes);
int edigpu_cache_get(edigpu_cache c, const edigpu_model *model, int kind, int q1, int q2, edigpu_handle *h);
int edigpu_cache_stats(edigpu_cache c, int64_t stats[5]);
int edigpu_cache_clear(edigpu_cache c);
int edigpu_cache_destroy(edigpu_cache c);

/*
 * Timing helper for bench.py: runs `warmup` untimed and `steps` timed H*v
 * products (device-resident, random unit vector) on the handle's stream and
 * returns the average duration of one step measured with HIP events on that stream.  lanczos = 0: the boundary
 * product (edigpu_apply_dev: vectors in the reference's layout); 1: full Lanczos steps; 2: the plain product as the
 * device-resident Lanczos loops compute it (on panel-major vectors for the sectors that use them).
 */
int edigpu_time_apply(edigpu_handle h, int warmup, int steps, int lanczos, double *ms_per_step);

/*
 * Measurement helper for bench.py: `warmup` + `steps` full Lanczos steps (H*v + the vector
 * recurrence) on a random unit start vector.  *ms_wall_per_step = host wall clock of the timed
 * steps (stream synchronised on both sides) / steps; *ms_hv_per_launch = average duration of the
 * H*v launches alone inside the timed steps, from HIP events recorded around every launch on the
 * stream they are launched on.
 */
int edigpu_lanczos_bench(edigpu_handle h, int warmup, int steps, double *ms_wall_per_step,
                         double *ms_hv_per_launch);

/*
 * Device buffers for hosts without a HIP binding of their own (the Fortran host keeps eigenvectors and Green's-function
 * seeds resident between edigpu_lanczos_eigh*, edigpu_apply_op_* and edigpu_lanczos_tridiag_dev this way; PyTorch
 * hosts use their own tensors).  Synchronous copies on the device selected by edigpu_init.
 */
int edigpu_dev_alloc(int64_t bytes, void **dev_ptr);
int edigpu_dev_free(void *dev_ptr);
int edigpu_dev_upload(void *dst_dev, const void *src_host, int64_t bytes);
int edigpu_dev_download(void *dst_host, const void *src_dev, int64_t bytes);

/* Measurement helper for bench.py: streaming ceilings of the device the calling thread selected -- gbs3[0] read,
 * [1] copy (bytes read + written), [2] triad (two reads + one write), GB/s on buffers of `bytes` each (use >= 1 GB
 * to leave the 256 MiB Infinity Cache; SURVEY.md 8d asks for the measured ceiling next to the 8 TB/s spec). */
int edigpu_membw(int64_t bytes, double *gbs3);

int edigpu_destroy(edigpu_handle h);

#ifdef __cplusplus
}
#endif
#endif /* EDIGPU_H */
