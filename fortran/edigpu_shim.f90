!> EDIGPU_SHIM -- thin ISO_C_BINDING layer between EDIpack's Fortran host and libedigpu.so
!!
!! The reference keeps its H*v behind two procedure pointers,
!!   procedure(dd_sparse_HxV),pointer :: spHtimesV_p    (ED_VARS_GLOBAL.f90:111-122,196)
!!   procedure(cc_sparse_HxV),pointer :: spHtimesV_cc   (ED_VARS_GLOBAL.f90:125-132,197)
!! assigned in build_Hv_sector_<mode> (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:177-203,
!! ED_SUPERC/ED_HAMILTONIAN_SUPERC.f90:118-132, ED_NONSU2/ED_HAMILTONIAN_NONSU2.f90:111-125).
!! This module provides procedures with exactly those interfaces,
!!   spMatVec_gpu_d(Nloc,v,Hv)   and   spMatVec_gpu_c(Nloc,v,Hv),
!! which forward to the C ABI (include/edigpu.h), plus the calls that hand the sector
!! Hamiltonian over (flattened sparse_matrix_csr rows) and the device-resident replacement of
!! sp_lanc_tridiag.  Error convention of the reference: `stop "message"`.
!!
!! One sector is live at a time (module variable `gpu_sector`), like Hsector/spH0* in the
!! reference (ED_VARS_GLOBAL.f90:190-197).  See INTEGRATION.md for the three-line patch of
!! build_Hv_sector_* / delete_Hv_sector_* / tridiag_Hv_sector_* that uses it.
module EDIGPU_SHIM
  use, intrinsic :: iso_c_binding
  implicit none
  private

  type(c_ptr), save :: gpu_sector = c_null_ptr   !< the live edigpu_handle
  type(c_ptr), save :: gpu_comm = c_null_ptr     !< the live edigpu_comm (MpiComm's counterpart; N > 1 only)
  type(c_ptr), save :: gpu_cache = c_null_ptr    !< the per-solve cache of sector handles (gpu_cache_create)
  logical, save :: gpu_sector_borrowed = .false. !< the live handle belongs to gpu_cache: gpu_delete_sector keeps it

  integer, parameter, public :: EDIGPU_MAXORB = 5, EDIGPU_MAXBATH = 16, EDIGPU_MAXSUNDRY = 64
  !> breakdown threshold of the tridiagonalisation (sp_lanc_tridiag is called without one: SciFortran's default)
  real(c_double), public :: lanc_threshold = 1.0e-12_c_double

  !> struct edigpu_model (include/edigpu.h): the module globals the reference's builders read.  C arrays
  !! are row-major, so the Fortran index order is reversed: uloc(iorb), ust(jorb,iorb) ... ,
  !! hloc(re/im, jorb, iorb, jspin, ispin), be(k, iorb, ispin), hb(re/im, k, jorb, iorb, js, is).
  type, bind(C), public :: edigpu_model_t
     integer(c_int32_t) :: ed_mode = 0      !< 0 normal, 1 superc, 2 nonsu2
     integer(c_int32_t) :: bath_type = 0    !< 0 normal, 1 hybrid, 2 replica, 3 general
     integer(c_int32_t) :: norb = 1, nbath = 1, nspin = 1
     integer(c_int32_t) :: hfmode = 1
     real(c_double) :: xmu = 0d0
     real(c_double) :: uloc(EDIGPU_MAXORB) = 0d0
     real(c_double) :: ust(EDIGPU_MAXORB, EDIGPU_MAXORB) = 0d0
     real(c_double) :: jh(EDIGPU_MAXORB, EDIGPU_MAXORB) = 0d0
     real(c_double) :: jx(EDIGPU_MAXORB, EDIGPU_MAXORB) = 0d0
     real(c_double) :: jp(EDIGPU_MAXORB, EDIGPU_MAXORB) = 0d0
     real(c_double) :: hloc(2, EDIGPU_MAXORB, EDIGPU_MAXORB, 2, 2) = 0d0
     real(c_double) :: pair_field(EDIGPU_MAXORB) = 0d0
     real(c_double) :: be(EDIGPU_MAXBATH, EDIGPU_MAXORB, 2) = 0d0
     real(c_double) :: bv(EDIGPU_MAXBATH, EDIGPU_MAXORB, 2) = 0d0
     real(c_double) :: bd(EDIGPU_MAXBATH, EDIGPU_MAXORB, 2) = 0d0
     real(c_double) :: bu(EDIGPU_MAXBATH, EDIGPU_MAXORB, 2) = 0d0
     real(c_double) :: hb(2, EDIGPU_MAXBATH, EDIGPU_MAXORB, EDIGPU_MAXORB, 2, 2) = 0d0
     integer(c_int32_t) :: nph = 0          !< phonon cut-off Nph (0: none); DimPh = Nph+1
     integer(c_int32_t) :: pad_ = 0
     real(c_double) :: w0_ph = 0d0, a_ph = 0d0
     real(c_double) :: g_ph(EDIGPU_MAXORB, EDIGPU_MAXORB) = 0d0   !< g_ph(jorb,iorb) = C [iorb][jorb]
     real(c_double) :: spin_field(3, EDIGPU_MAXORB) = 0d0         !< spin_field(xyz,iorb) = transpose(spin_field)
     real(c_double) :: exc_field(4) = 0d0
     integer(c_int32_t) :: nsundry = 0      !< size(coulomb_sundry)
     integer(c_int32_t) :: pad2_ = 0
     !> sundry_op(:,l) = [cd_i(1:2), cd_j(1:2), c_k(1:2), c_l(1:2)] of coulomb_sundry(l) (orbital, spin; 1-based)
     integer(c_int32_t) :: sundry_op(8, EDIGPU_MAXSUNDRY) = 0
     real(c_double) :: sundry_u(EDIGPU_MAXSUNDRY) = 0d0           !< coulomb_sundry(l)%U
  end type edigpu_model_t

  interface
     function edigpu_last_error() bind(C, name="edigpu_last_error") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function edigpu_last_error
     function edigpu_model_sizeof() bind(C, name="edigpu_model_sizeof") result(n)
       import :: c_int64_t
       integer(c_int64_t) :: n
     end function edigpu_model_sizeof
     function edigpu_init(device) bind(C, name="edigpu_init") result(ierr)
       import :: c_int
       integer(c_int), value :: device
       integer(c_int) :: ierr
     end function edigpu_init
     function edigpu_normal_create(h, dim_up, dim_dw, dw_first, dw_count, hd, &
          up_rowptr, up_col, up_val, dw_rowptr, dw_col, dw_val, nd_rowptr, nd_col, nd_val) &
          bind(C, name="edigpu_normal_create") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr) :: h                                   ! edigpu_handle* (by reference)
       integer(c_int64_t), value :: dim_up, dim_dw, dw_first, dw_count
       type(c_ptr), value :: hd, up_rowptr, up_col, up_val, dw_rowptr, dw_col, dw_val
       type(c_ptr), value :: nd_rowptr, nd_col, nd_val
       integer(c_int) :: ierr
     end function edigpu_normal_create
     function edigpu_normal_build(h, model, nup, ndw, dw_first, dw_count) &
          bind(C, name="edigpu_normal_build") result(ierr)
       import :: c_ptr, c_int, c_int64_t, edigpu_model_t
       type(c_ptr) :: h
       type(edigpu_model_t), intent(in) :: model
       integer(c_int), value :: nup, ndw
       integer(c_int64_t), value :: dw_first, dw_count
       integer(c_int) :: ierr
     end function edigpu_normal_build
     function edigpu_normal_build_z(h, model, nup, ndw) bind(C, name="edigpu_normal_build_z") result(ierr)
       import :: c_ptr, c_int, edigpu_model_t
       type(c_ptr) :: h
       type(edigpu_model_t), intent(in) :: model
       integer(c_int), value :: nup, ndw
       integer(c_int) :: ierr
     end function edigpu_normal_build_z
     function edigpu_flat_build(h, model, sector, row_first, row_count) &
          bind(C, name="edigpu_flat_build") result(ierr)
       import :: c_ptr, c_int, c_int64_t, edigpu_model_t
       type(c_ptr) :: h
       type(edigpu_model_t), intent(in) :: model
       integer(c_int), value :: sector
       integer(c_int64_t), value :: row_first, row_count
       integer(c_int) :: ierr
     end function edigpu_flat_build
     function edigpu_flat_build_jz(h, model, ntot, twojz, row_first, row_count) &
          bind(C, name="edigpu_flat_build_jz") result(ierr)
       import :: c_ptr, c_int, c_int64_t, edigpu_model_t
       type(c_ptr) :: h
       type(edigpu_model_t), intent(in) :: model
       integer(c_int), value :: ntot, twojz
       integer(c_int64_t), value :: row_first, row_count
       integer(c_int) :: ierr
     end function edigpu_flat_build_jz
     function edigpu_direct_build_jz(h, model, ntot, twojz, row_first, row_count) &
          bind(C, name="edigpu_direct_build_jz") result(ierr)
       import :: c_ptr, c_int, c_int64_t, edigpu_model_t
       type(c_ptr) :: h
       type(edigpu_model_t), intent(in) :: model
       integer(c_int), value :: ntot, twojz
       integer(c_int64_t), value :: row_first, row_count
       integer(c_int) :: ierr
     end function edigpu_direct_build_jz
     function edigpu_direct_build(h, model, sector, row_first, row_count) &
          bind(C, name="edigpu_direct_build") result(ierr)
       import :: c_ptr, c_int, c_int64_t, edigpu_model_t
       type(c_ptr) :: h
       type(edigpu_model_t), intent(in) :: model
       integer(c_int), value :: sector
       integer(c_int64_t), value :: row_first, row_count
       integer(c_int) :: ierr
     end function edigpu_direct_build
     function edigpu_orbs_build(h, model, nups, ndws) bind(C, name="edigpu_orbs_build") result(ierr)
       import :: c_ptr, c_int, c_int32_t, edigpu_model_t
       type(c_ptr) :: h
       type(edigpu_model_t), intent(in) :: model
       integer(c_int32_t), intent(in) :: nups(*), ndws(*)
       integer(c_int) :: ierr
     end function edigpu_orbs_build
     function edigpu_csr_create_d(h, nrow_local, ncol_global, row_first, rowptr, col, val) &
          bind(C, name="edigpu_csr_create_d") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr) :: h
       integer(c_int64_t), value :: nrow_local, ncol_global, row_first
       type(c_ptr), value :: rowptr, col, val
       integer(c_int) :: ierr
     end function edigpu_csr_create_d
     function edigpu_csr_create_z(h, nrow_local, ncol_global, row_first, rowptr, col, val) &
          bind(C, name="edigpu_csr_create_z") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr) :: h
       integer(c_int64_t), value :: nrow_local, ncol_global, row_first
       type(c_ptr), value :: rowptr, col, val
       integer(c_int) :: ierr
     end function edigpu_csr_create_z
     function edigpu_apply_d(h, nloc, v, hv) bind(C, name="edigpu_apply_d") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int64_t), value :: nloc
       real(c_double), intent(in) :: v(*)
       real(c_double), intent(inout) :: hv(*)
       integer(c_int) :: ierr
     end function edigpu_apply_d
     function edigpu_apply_z(h, nloc, v, hv) bind(C, name="edigpu_apply_z") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double_complex
       type(c_ptr), value :: h
       integer(c_int64_t), value :: nloc
       complex(c_double_complex), intent(in) :: v(*)
       complex(c_double_complex), intent(inout) :: hv(*)
       integer(c_int) :: ierr
     end function edigpu_apply_z
     function edigpu_lanczos_tridiag(h, vin, nlanc, alanc, blanc, threshold, niter) &
          bind(C, name="edigpu_lanczos_tridiag") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h, vin
       integer(c_int), value :: nlanc
       real(c_double), intent(inout) :: alanc(*), blanc(*)
       real(c_double), value :: threshold
       integer(c_int), intent(out) :: niter
       integer(c_int) :: ierr
     end function edigpu_lanczos_tridiag
     function edigpu_lanczos_tridiag_dev(h, vin_dev, nlanc, alanc, blanc, threshold, niter, norm2) &
          bind(C, name="edigpu_lanczos_tridiag_dev") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h, vin_dev
       integer(c_int), value :: nlanc
       real(c_double), intent(inout) :: alanc(*), blanc(*)
       real(c_double), value :: threshold
       integer(c_int), intent(out) :: niter
       real(c_double), intent(out) :: norm2
       integer(c_int) :: ierr
     end function edigpu_lanczos_tridiag_dev
     function edigpu_lanczos_eigh(h, nitermax, tol, check_every, v0, eval, evec, niter) &
          bind(C, name="edigpu_lanczos_eigh") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h, v0, evec          ! host or device memory, may be null
       integer(c_int), value :: nitermax, check_every
       real(c_double), value :: tol
       real(c_double), intent(out) :: eval
       integer(c_int), intent(out) :: niter
       integer(c_int) :: ierr
     end function edigpu_lanczos_eigh
     function edigpu_lanczos_eigh_multi(h, neigen, ncv, tol, maxrestart, v0, evals, evecs, nconv, nmatvec) &
          bind(C, name="edigpu_lanczos_eigh_multi") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h, v0, evecs          ! host or device memory, may be null
       integer(c_int), value :: neigen, ncv, maxrestart
       real(c_double), value :: tol
       real(c_double), intent(inout) :: evals(*)
       integer(c_int), intent(out) :: nconv, nmatvec
       integer(c_int) :: ierr
     end function edigpu_lanczos_eigh_multi
     function edigpu_apply_op_normal(src, dst, v_src_dev, v_dst_dev, iorb, ispin, create, stream) &
          bind(C, name="edigpu_apply_op_normal") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: src, dst, v_src_dev, v_dst_dev, stream
       integer(c_int), value :: iorb, ispin, create
       integer(c_int) :: ierr
     end function edigpu_apply_op_normal
     function edigpu_apply_op_flat(src, dst, v_src_dev, v_dst_dev, iorb, ispin, create, stream) &
          bind(C, name="edigpu_apply_op_flat") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: src, dst, v_src_dev, v_dst_dev, stream
       integer(c_int), value :: iorb, ispin, create
       integer(c_int) :: ierr
     end function edigpu_apply_op_flat
     function edigpu_apply_cops_normal(src, dst, v_src_dev, v_dst_dev, nops, coef, create, iorb, ispin, stream) &
          bind(C, name="edigpu_apply_cops_normal") result(ierr)
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: src, dst, v_src_dev, v_dst_dev, stream
       integer(c_int), value :: nops
       real(c_double), intent(in) :: coef(*)
       integer(c_int32_t), intent(in) :: create(*), iorb(*), ispin(*)
       integer(c_int) :: ierr
     end function edigpu_apply_cops_normal
     function edigpu_info(h, info) bind(C, name="edigpu_info") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: h
       integer(c_int64_t), intent(out) :: info(10)
       integer(c_int) :: ierr
     end function edigpu_info
     function edigpu_dev_alloc(bytes, p) bind(C, name="edigpu_dev_alloc") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       integer(c_int64_t), value :: bytes
       type(c_ptr), intent(out) :: p
       integer(c_int) :: ierr
     end function edigpu_dev_alloc
     function edigpu_dev_free(p) bind(C, name="edigpu_dev_free") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: p
       integer(c_int) :: ierr
     end function edigpu_dev_free
     function edigpu_dev_upload(dst, src, bytes) bind(C, name="edigpu_dev_upload") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: dst, src
       integer(c_int64_t), value :: bytes
       integer(c_int) :: ierr
     end function edigpu_dev_upload
     function edigpu_dev_download(dst, src, bytes) bind(C, name="edigpu_dev_download") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: dst, src
       integer(c_int64_t), value :: bytes
       integer(c_int) :: ierr
     end function edigpu_dev_download
     ! ---- N > 1 inside the library (include/edigpu.h, "communicator, sharded product, sharded Lanczos") ----
     function edigpu_shard_plan(units, world, rank, first, count, q) bind(C, name="edigpu_shard_plan") result(ierr)
       import :: c_int, c_int32_t, c_int64_t
       integer(c_int64_t), value :: units
       integer(c_int32_t), value :: world, rank
       integer(c_int64_t), intent(out) :: first, count, q
       integer(c_int) :: ierr
     end function edigpu_shard_plan
     function edigpu_cache_create(c, max_device_bytes) bind(C, name="edigpu_cache_create") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr) :: c
       integer(c_int64_t), value :: max_device_bytes
       integer(c_int) :: ierr
     end function edigpu_cache_create
     function edigpu_cache_get(c, model, kind, q1, q2, h) bind(C, name="edigpu_cache_get") result(ierr)
       import :: c_ptr, c_int, edigpu_model_t
       type(c_ptr), value :: c
       type(edigpu_model_t), intent(in) :: model
       integer(c_int), value :: kind, q1, q2
       type(c_ptr) :: h
       integer(c_int) :: ierr
     end function edigpu_cache_get
     function edigpu_cache_stats(c, stats) bind(C, name="edigpu_cache_stats") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: c
       integer(c_int64_t) :: stats(5)
       integer(c_int) :: ierr
     end function edigpu_cache_stats
     function edigpu_cache_clear(c) bind(C, name="edigpu_cache_clear") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: c
       integer(c_int) :: ierr
     end function edigpu_cache_clear
     function edigpu_cache_destroy(c) bind(C, name="edigpu_cache_destroy") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: c
       integer(c_int) :: ierr
     end function edigpu_cache_destroy
     function edigpu_comm_unique_id(id) bind(C, name="edigpu_comm_unique_id") result(ierr)
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id(128)
       integer(c_int) :: ierr
     end function edigpu_comm_unique_id
     function edigpu_comm_create(c, rank, world, id) bind(C, name="edigpu_comm_create") result(ierr)
       import :: c_ptr, c_int, c_int32_t, c_char
       type(c_ptr) :: c
       integer(c_int32_t), value :: rank, world
       character(kind=c_char), intent(in) :: id(128)
       integer(c_int) :: ierr
     end function edigpu_comm_create
     function edigpu_comm_create_shm(c, rank, world, name, slot_bytes) bind(C, name="edigpu_comm_create_shm") result(ierr)
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_char
       type(c_ptr) :: c
       integer(c_int32_t), value :: rank, world
       character(kind=c_char), intent(in) :: name(*)
       integer(c_int64_t), value :: slot_bytes
       integer(c_int) :: ierr
     end function edigpu_comm_create_shm
     function edigpu_comm_destroy(c) bind(C, name="edigpu_comm_destroy") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: c
       integer(c_int) :: ierr
     end function edigpu_comm_destroy
     function edigpu_apply_sharded_d(h, c, nloc, v, hv) bind(C, name="edigpu_apply_sharded_d") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h, c
       integer(c_int64_t), value :: nloc
       real(c_double), intent(in) :: v(*)
       real(c_double), intent(inout) :: hv(*)
       integer(c_int) :: ierr
     end function edigpu_apply_sharded_d
     function edigpu_apply_sharded_z(h, c, nloc, v, hv) bind(C, name="edigpu_apply_sharded_z") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double_complex
       type(c_ptr), value :: h, c
       integer(c_int64_t), value :: nloc
       complex(c_double_complex), intent(in) :: v(*)
       complex(c_double_complex), intent(inout) :: hv(*)
       integer(c_int) :: ierr
     end function edigpu_apply_sharded_z
     function edigpu_lanczos_tridiag_sharded(h, c, vin, nlanc, alanc, blanc, threshold, niter, norm2) &
          bind(C, name="edigpu_lanczos_tridiag_sharded") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h, c, vin
       integer(c_int), value :: nlanc
       real(c_double), intent(inout) :: alanc(*), blanc(*)
       real(c_double), value :: threshold
       integer(c_int), intent(out) :: niter
       real(c_double), intent(out) :: norm2
       integer(c_int) :: ierr
     end function edigpu_lanczos_tridiag_sharded
     function edigpu_lanczos_eigh_multi_sharded(h, c, neigen, ncv, tol, maxrestart, v0, evals, evecs, nconv, nmatvec) &
          bind(C, name="edigpu_lanczos_eigh_multi_sharded") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h, c, v0, evecs
       integer(c_int), value :: neigen, ncv, maxrestart
       real(c_double), value :: tol
       real(c_double), intent(inout) :: evals(*)
       integer(c_int), intent(out) :: nconv, nmatvec
       integer(c_int) :: ierr
     end function edigpu_lanczos_eigh_multi_sharded
     function edigpu_apply_cops_sharded(src, dst, c, v_src, v_dst, nops, coef2, create, iorb, ispin) &
          bind(C, name="edigpu_apply_cops_sharded") result(ierr)
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: src, dst, c, v_src, v_dst
       integer(c_int), value :: nops
       real(c_double), intent(in) :: coef2(*)
       integer(c_int32_t), intent(in) :: create(*), iorb(*), ispin(*)
       integer(c_int) :: ierr
     end function edigpu_apply_cops_sharded
     function edigpu_destroy(h) bind(C, name="edigpu_destroy") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int) :: ierr
     end function edigpu_destroy
     function c_strlen(s) bind(C, name="strlen") result(n)
       import :: c_ptr, c_size_t
       type(c_ptr), value :: s
       integer(c_size_t) :: n
     end function c_strlen
  end interface

  public :: gpu_init, gpu_delete_sector
  public :: gpu_set_normal, gpu_set_csr_d, gpu_set_csr_c
  public :: gpu_model_set_kanamori, gpu_model_set_hloc, gpu_model_set_bath
  public :: gpu_build_normal, gpu_build_normal_cmplx, gpu_build_flat, gpu_build_flat_jz, gpu_build_orbs
  public :: spMatVec_gpu_d, spMatVec_gpu_c
  public :: gpu_lanc_tridiag_d, gpu_lanc_tridiag_c
  public :: flatten_rows_count
  ! "next" rows of SURVEY.md 8(f): the eigensolvers and the device-resident neighbours of the tridiagonalisation
  public :: gpu_sector_handle, gpu_sector_swap, gpu_sector_dim, gpu_sector_destroy
  public :: gpu_cache_create, gpu_cache_clear, gpu_cache_destroy, gpu_cache_stats, gpu_build_cached
  public :: gpu_sp_lanc_eigh_d, gpu_sp_lanc_eigh_c, gpu_sp_eigh_d, gpu_sp_eigh_c
  public :: gpu_vec_alloc, gpu_vec_free, gpu_vec_upload_d, gpu_vec_download_d, gpu_vec_upload_c, gpu_vec_download_c
  public :: gpu_sp_lanc_eigh_dev, gpu_apply_op, gpu_apply_cops, gpu_lanc_tridiag_dev
  ! N > 1: communicator + the MPI twins of the product and of the tridiagonalisation
  public :: gpu_comm_unique_id, gpu_comm_create, gpu_comm_create_shm, gpu_comm_destroy, gpu_shard_plan
  public :: spMatVec_mpi_gpu_d, spMatVec_mpi_gpu_c, gpu_lanc_tridiag_mpi_d, gpu_lanc_tridiag_mpi_c
  public :: gpu_sp_eigh_mpi_d, gpu_sp_eigh_mpi_c, gpu_apply_op_mpi_d

contains

  !> `stop` with the library's message: the reference's error convention
  !! (e.g. ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:797).
  subroutine gpu_check(ierr, where)
    integer(c_int), intent(in) :: ierr
    character(len=*), intent(in) :: where
    type(c_ptr) :: cmsg
    character(kind=c_char), pointer :: fmsg(:)
    character(len=512) :: msg
    integer :: i, n
    if (ierr == 0) return
    msg = ""
    cmsg = edigpu_last_error()
    if (c_associated(cmsg)) then
       n = min(int(c_strlen(cmsg)), len(msg))
       call c_f_pointer(cmsg, fmsg, [n])
       do i = 1, n
          msg(i:i) = fmsg(i)
       end do
    end if
    write(*,"(A)") "EDIGPU ERROR in "//trim(where)//": "//trim(msg)
    error stop "EDIGPU error"
  end subroutine gpu_check

  !> once per rank, e.g. from ed_solve (ED_MAIN.f90:164): device = local MPI rank
  subroutine gpu_init(device)
    integer, intent(in) :: device
    type(edigpu_model_t) :: probe
    ! the Fortran mirror of struct edigpu_model must be the library's (include/edigpu.h)
    if (int(c_sizeof(probe), c_int64_t) /= edigpu_model_sizeof()) then
       write(0, *) "edigpu_shim: edigpu_model_t has ", c_sizeof(probe), " bytes, libedigpu expects ", edigpu_model_sizeof()
       stop 2
    end if
    call gpu_check(edigpu_init(int(device, c_int)), "gpu_init")
  end subroutine gpu_init

  !> helper for the flattening of sparse_matrix_csr rows (ED_SPARSE_MATRIX.f90:16-41):
  !! rowptr(0:n) from the per-row sizes, 0-based as the C side wants it
  subroutine flatten_rows_count(sizes, rowptr)
    integer, intent(in) :: sizes(:)
    integer(c_int64_t), intent(out) :: rowptr(0:)
    integer :: i
    rowptr(0) = 0_c_int64_t
    do i = 1, size(sizes)
       rowptr(i) = rowptr(i-1) + int(sizes(i), c_int64_t)
    end do
  end subroutine flatten_rows_count

  !> hand over spH0d / spH0ups(1) / spH0dws(1) / spH0nd (already flattened, 0-based columns).
  !! dw_first/dw_count = the rank's share of the down index (mpiIshift/DimUp, mpiQdw;
  !! ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:129-142).  Pass has_nd=.false. when spH0nd is not built.
  subroutine gpu_set_normal(DimUp, DimDw, dw_first, dw_count, hd, up_rowptr, up_col, up_val, &
       dw_rowptr, dw_col, dw_val, has_nd, nd_rowptr, nd_col, nd_val)
    integer, intent(in) :: DimUp, DimDw, dw_first, dw_count
    real(c_double), intent(in), target :: hd(:), up_val(:), dw_val(:), nd_val(:)
    integer(c_int64_t), intent(in), target :: up_rowptr(0:), dw_rowptr(0:), nd_rowptr(0:)
    integer(c_int32_t), intent(in), target :: up_col(:), dw_col(:), nd_col(:)
    logical, intent(in) :: has_nd
    type(c_ptr) :: pr, pc, pv
    if (c_associated(gpu_sector)) stop "gpu_set_normal: a sector is already allocated"
    pr = c_null_ptr; pc = c_null_ptr; pv = c_null_ptr
    if (has_nd) then
       pr = c_loc(nd_rowptr); pc = c_loc(nd_col); pv = c_loc(nd_val)
    end if
    call gpu_check(edigpu_normal_create(gpu_sector, int(DimUp, c_int64_t), int(DimDw, c_int64_t), &
         int(dw_first, c_int64_t), int(dw_count, c_int64_t), c_loc(hd), &
         c_loc(up_rowptr), c_loc(up_col), c_loc(up_val), &
         c_loc(dw_rowptr), c_loc(dw_col), c_loc(dw_val), pr, pc, pv), "gpu_set_normal")
  end subroutine gpu_set_normal

  !> ---- library-built sectors: skip ed_buildh_* altogether (factored normal image, device-built
  !! superc / nonsu2 image, on-the-fly kernel).  Fill edigpu_model_t from the module globals: ----

  !> Uloc_internal, Ust/Jh/Jx/Jp_internal (ED_VARS_GLOBAL.f90:216-220)
  subroutine gpu_model_set_kanamori(m, uloc, ust, jh, jx, jp)
    type(edigpu_model_t), intent(inout) :: m
    real(8), intent(in) :: uloc(:), ust(:,:), jh(:,:), jx(:,:), jp(:,:)
    integer :: a, b, no
    no = size(uloc)
    m%uloc(1:no) = uloc
    do a = 1, no
       do b = 1, no     ! C [a][b] = Fortran (b,a)
          m%ust(b,a) = ust(a,b); m%jh(b,a) = jh(a,b); m%jx(b,a) = jx(a,b); m%jp(b,a) = jp(a,b)
       end do
    end do
  end subroutine gpu_model_set_kanamori

  !> impHloc + mfHloc, Fortran shape (Nspin,Nspin,Norb,Norb) as in ED_VARS_GLOBAL
  subroutine gpu_model_set_hloc(m, hloc)
    type(edigpu_model_t), intent(inout) :: m
    complex(8), intent(in) :: hloc(:,:,:,:)
    integer :: is, js, a, b
    do is = 1, size(hloc,1)
       do js = 1, size(hloc,2)
          do a = 1, size(hloc,3)
             do b = 1, size(hloc,4)
                m%hloc(1, b, a, js, is) = dble(hloc(is,js,a,b))
                m%hloc(2, b, a, js, is) = aimag(hloc(is,js,a,b))
             end do
          end do
       end do
    end do
  end subroutine gpu_model_set_hloc

  !> dmft_bath%e, %v (and %d for superc, %u for nonsu2), Fortran shape (Nspin, Norb|1, Nbath)
  subroutine gpu_model_set_bath(m, e, v, d, u)
    type(edigpu_model_t), intent(inout) :: m
    real(8), intent(in) :: e(:,:,:), v(:,:,:)
    real(8), intent(in), optional :: d(:,:,:), u(:,:,:)
    integer :: is, a, k
    do is = 1, size(e,1)
       do a = 1, size(e,2)
          do k = 1, size(e,3)
             m%be(k, a, is) = e(is,a,k)
             if (present(d)) m%bd(k, a, is) = d(is,a,k)
          end do
       end do
    end do
    do is = 1, size(v,1)
       do a = 1, size(v,2)
          do k = 1, size(v,3)
             m%bv(k, a, is) = v(is,a,k)
             if (present(u)) m%bu(k, a, is) = u(is,a,k)
          end do
       end do
    end do
  end subroutine gpu_model_set_bath

  !> build_Hv_sector_normal(isector) with ed_total_ud=T: the (Nup,Ndw) sector; dw_first/dw_count as in
  !! gpu_set_normal (dw_count < 0: the whole sector)
  subroutine gpu_build_normal(m, nup, ndw, dw_first, dw_count)
    type(edigpu_model_t), intent(in) :: m
    integer, intent(in) :: nup, ndw, dw_first, dw_count
    if (c_associated(gpu_sector)) stop "gpu_build_normal: a sector is already allocated"
    call gpu_check(edigpu_normal_build(gpu_sector, m, int(nup, c_int), int(ndw, c_int), &
         int(dw_first, c_int64_t), int(dw_count, c_int64_t)), "gpu_build_normal")
  end subroutine gpu_build_normal

  !> build_Hv_sector_normal in a -D_CMPLX_NORMAL build: complex impHloc / bath matrices, complex vectors
  !> (use the complex spMatVec / tridiag wrappers afterwards)
  subroutine gpu_build_normal_cmplx(m, nup, ndw)
    type(edigpu_model_t), intent(in) :: m
    integer, intent(in) :: nup, ndw
    if (c_associated(gpu_sector)) stop "gpu_build_normal_cmplx: a sector is already allocated"
    call gpu_check(edigpu_normal_build_z(gpu_sector, m, int(nup, c_int), int(ndw, c_int)), "gpu_build_normal_cmplx")
  end subroutine gpu_build_normal_cmplx

  !> build_Hv_sector_superc / _nonsu2: sector = Sz / Ntot; direct=.true. selects ed_sparse_H=F
  subroutine gpu_build_flat(m, sector, row_first, row_count, direct)
    type(edigpu_model_t), intent(in) :: m
    integer, intent(in) :: sector, row_first, row_count
    logical, intent(in) :: direct
    if (c_associated(gpu_sector)) stop "gpu_build_flat: a sector is already allocated"
    if (direct) then
       call gpu_check(edigpu_direct_build(gpu_sector, m, int(sector, c_int), int(row_first, c_int64_t), &
            int(row_count, c_int64_t)), "gpu_build_flat(direct)")
    else
       call gpu_check(edigpu_flat_build(gpu_sector, m, int(sector, c_int), int(row_first, c_int64_t), &
            int(row_count, c_int64_t)), "gpu_build_flat")
    end if
  end subroutine gpu_build_flat

  !> build_Hv_sector_nonsu2 with Jz_basis=T: sector (getN(isector), gettwoJz(isector)) (ED_SECTOR.f90:289-350);
  !> direct=.true. selects ed_sparse_H=F (nothing stored)
  subroutine gpu_build_flat_jz(m, ntot, twojz, row_first, row_count, direct)
    type(edigpu_model_t), intent(in) :: m
    integer, intent(in) :: ntot, twojz, row_first, row_count
    logical, intent(in), optional :: direct
    logical :: direct_
    if (c_associated(gpu_sector)) stop "gpu_build_flat_jz: a sector is already allocated"
    direct_ = .false.
    if (present(direct)) direct_ = direct
    if (direct_) then
       call gpu_check(edigpu_direct_build_jz(gpu_sector, m, int(ntot, c_int), int(twojz, c_int), int(row_first, c_int64_t), &
            int(row_count, c_int64_t)), "gpu_build_flat_jz(direct)")
    else
       call gpu_check(edigpu_flat_build_jz(gpu_sector, m, int(ntot, c_int), int(twojz, c_int), int(row_first, c_int64_t), &
            int(row_count, c_int64_t)), "gpu_build_flat_jz")
    end if
  end subroutine gpu_build_flat_jz

  !> build_Hv_sector_normal with ed_total_ud=F: per-orbital (Nups, Ndws)
  subroutine gpu_build_orbs(m, nups, ndws)
    type(edigpu_model_t), intent(in) :: m
    integer, intent(in) :: nups(:), ndws(:)
    integer(c_int32_t) :: a(size(nups)), b(size(ndws))
    if (c_associated(gpu_sector)) stop "gpu_build_orbs: a sector is already allocated"
    a = int(nups, c_int32_t); b = int(ndws, c_int32_t)
    call gpu_check(edigpu_orbs_build(gpu_sector, m, a, b), "gpu_build_orbs")
  end subroutine gpu_build_orbs

  !> hand over a flat real CSR (sp_matvec-type use, e.g. a real spH0)
  subroutine gpu_set_csr_d(nrow_local, ncol_global, row_first, rowptr, col, val)
    integer, intent(in) :: nrow_local, ncol_global, row_first
    integer(c_int64_t), intent(in), target :: rowptr(0:)
    integer(c_int32_t), intent(in), target :: col(:)
    real(c_double), intent(in), target :: val(:)
    if (c_associated(gpu_sector)) stop "gpu_set_csr_d: a sector is already allocated"
    call gpu_check(edigpu_csr_create_d(gpu_sector, int(nrow_local, c_int64_t), &
         int(ncol_global, c_int64_t), int(row_first, c_int64_t), &
         c_loc(rowptr), c_loc(col), c_loc(val)), "gpu_set_csr_d")
  end subroutine gpu_set_csr_d

  !> hand over spH0 of the superc / nonsu2 modes: the rank's rows (loc and non-loc entries merged
  !! back into one row, global 0-based columns); row_first = mpiIshift
  !! (ED_SUPERC/ED_HAMILTONIAN_SUPERC.f90:82-88, ED_NONSU2/ED_HAMILTONIAN_NONSU2.f90:73-79)
  subroutine gpu_set_csr_c(nrow_local, ncol_global, row_first, rowptr, col, val)
    integer, intent(in) :: nrow_local, ncol_global, row_first
    integer(c_int64_t), intent(in), target :: rowptr(0:)
    integer(c_int32_t), intent(in), target :: col(:)
    complex(c_double_complex), intent(in), target :: val(:)
    if (c_associated(gpu_sector)) stop "gpu_set_csr_c: a sector is already allocated"
    call gpu_check(edigpu_csr_create_z(gpu_sector, int(nrow_local, c_int64_t), &
         int(ncol_global, c_int64_t), int(row_first, c_int64_t), &
         c_loc(rowptr), c_loc(col), c_loc(val)), "gpu_set_csr_c")
  end subroutine gpu_set_csr_c

  !> dd_sparse_HxV-compatible (ED_VARS_GLOBAL.f90:111-122): assign with  spHtimesV_p => spMatVec_gpu_d
  subroutine spMatVec_gpu_d(Nloc, v, Hv)
    integer :: Nloc
    real(8), dimension(Nloc) :: v, Hv
    if (.not. c_associated(gpu_sector)) stop "spMatVec_gpu_d: Hsector NOT allocated"
    call gpu_check(edigpu_apply_d(gpu_sector, int(Nloc, c_int64_t), v, Hv), "spMatVec_gpu_d")
  end subroutine spMatVec_gpu_d

  !> cc_sparse_HxV-compatible (ED_VARS_GLOBAL.f90:125-132): spHtimesV_cc => spMatVec_gpu_c
  subroutine spMatVec_gpu_c(Nloc, v, Hv)
    integer :: Nloc
    complex(8), dimension(Nloc) :: v, Hv
    if (.not. c_associated(gpu_sector)) stop "spMatVec_gpu_c: Hsector NOT allocated"
    call gpu_check(edigpu_apply_z(gpu_sector, int(Nloc, c_int64_t), v, Hv), "spMatVec_gpu_c")
  end subroutine spMatVec_gpu_c

  !> device-resident replacement of  call sp_lanc_tridiag(spHtimesV_p, vvinit, alanc, blanc)
  !! The reference passes no threshold there, i.e. SciFortran's default breakdown threshold applies (believed
  !! 1d-12; the SciFortran source is not part of the reference tree): the recurrence exits when |beta| falls below
  !! it, and the library additionally stops on an exact zero whatever the threshold is.
  !! (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:360-365): the vector never leaves HBM between steps
  subroutine gpu_lanc_tridiag_d(vin, alanc, blanc)
    real(8), intent(in), target :: vin(:)
    real(8), intent(inout) :: alanc(:), blanc(:)
    integer(c_int) :: niter
    if (.not. c_associated(gpu_sector)) stop "gpu_lanc_tridiag_d: Hsector NOT allocated"
    call gpu_check(edigpu_lanczos_tridiag(gpu_sector, c_loc(vin), int(size(alanc), c_int), &
         alanc, blanc, lanc_threshold, niter), "gpu_lanc_tridiag_d")
  end subroutine gpu_lanc_tridiag_d

  subroutine gpu_lanc_tridiag_c(vin, alanc, blanc)
    complex(8), intent(in), target :: vin(:)
    real(8), intent(inout) :: alanc(:), blanc(:)
    integer(c_int) :: niter
    if (.not. c_associated(gpu_sector)) stop "gpu_lanc_tridiag_c: Hsector NOT allocated"
    call gpu_check(edigpu_lanczos_tridiag(gpu_sector, c_loc(vin), int(size(alanc), c_int), &
         alanc, blanc, lanc_threshold, niter), "gpu_lanc_tridiag_c")
  end subroutine gpu_lanc_tridiag_c

  ! ==============================================================================================
  ! SURVEY.md 8(f) rows f1 / f3 and the N > 1 path, as the Fortran host sees them
  ! ==============================================================================================

  !> the live sector as an opaque handle / make another handle the live one: the Green's-function loop keeps TWO
  !! sectors alive (the eigenstate's and the one c / c^+ leads to, ED_NORMAL/ED_GF_NORMAL.f90:141-175) while the
  !! reference's module globals hold one.  gpu_sector_swap(h) installs h and returns the previous live handle in h.
  function gpu_sector_handle() result(h)
    type(c_ptr) :: h
    h = gpu_sector
  end function gpu_sector_handle

  subroutine gpu_sector_swap(h)
    type(c_ptr), intent(inout) :: h
    type(c_ptr) :: t
    t = gpu_sector; gpu_sector = h; h = t
  end subroutine gpu_sector_swap

  !> vecDim_Hv_sector_* of a handle (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:286-313): local rows
  function gpu_sector_dim(h) result(n)
    type(c_ptr), intent(in) :: h
    integer :: n
    integer(c_int64_t) :: info(10)
    call gpu_check(edigpu_info(h, info), "gpu_sector_dim")
    n = int(info(2))
  end function gpu_sector_dim

  subroutine gpu_sector_destroy(h)
    type(c_ptr), intent(inout) :: h
    if (c_associated(h)) call gpu_check(edigpu_destroy(h), "gpu_sector_destroy")
    h = c_null_ptr
  end subroutine gpu_sector_destroy

  !> call sp_lanc_eigh(spHtimesV_p, eig_values(1), eig_basis(:,1), Nitermax, iverbose=..., threshold=lanc_tolerance)
  !! (lanc_method = "lanczos", ED_NORMAL/ED_DIAG_NORMAL.f90:206-214): lowest eigenpair, vectors resident in HBM
  subroutine gpu_sp_lanc_eigh_d(eval, evec, Nitermax, threshold)
    real(8), intent(out) :: eval
    real(8), intent(inout), target :: evec(:)
    integer, intent(in) :: Nitermax
    real(8), intent(in), optional :: threshold
    real(c_double) :: tol
    integer(c_int) :: niter
    if (.not. c_associated(gpu_sector)) stop "gpu_sp_lanc_eigh_d: Hsector NOT allocated"
    tol = 1d-12; if (present(threshold)) tol = max(threshold, 1d-15)
    call gpu_check(edigpu_lanczos_eigh(gpu_sector, int(Nitermax, c_int), tol, 10_c_int, c_null_ptr, eval, &
         c_loc(evec), niter), "gpu_sp_lanc_eigh_d")
  end subroutine gpu_sp_lanc_eigh_d

  subroutine gpu_sp_lanc_eigh_c(eval, evec, Nitermax, threshold)
    real(8), intent(out) :: eval
    complex(8), intent(inout), target :: evec(:)
    integer, intent(in) :: Nitermax
    real(8), intent(in), optional :: threshold
    real(c_double) :: tol
    integer(c_int) :: niter
    if (.not. c_associated(gpu_sector)) stop "gpu_sp_lanc_eigh_c: Hsector NOT allocated"
    tol = 1d-12; if (present(threshold)) tol = max(threshold, 1d-15)
    call gpu_check(edigpu_lanczos_eigh(gpu_sector, int(Nitermax, c_int), tol, 10_c_int, c_null_ptr, eval, &
         c_loc(evec), niter), "gpu_sp_lanc_eigh_c")
  end subroutine gpu_sp_lanc_eigh_c

  !> the same with the eigenvector LEFT ON THE DEVICE (evec_dev from gpu_vec_alloc): the state never visits the host
  !! between the diagonalisation and the Green's-function seeds (row f3)
  subroutine gpu_sp_lanc_eigh_dev(eval, evec_dev, Nitermax, threshold)
    real(8), intent(out) :: eval
    type(c_ptr), intent(in) :: evec_dev
    integer, intent(in) :: Nitermax
    real(8), intent(in), optional :: threshold
    real(c_double) :: tol
    integer(c_int) :: niter
    if (.not. c_associated(gpu_sector)) stop "gpu_sp_lanc_eigh_dev: Hsector NOT allocated"
    tol = 1d-12; if (present(threshold)) tol = max(threshold, 1d-15)
    call gpu_check(edigpu_lanczos_eigh(gpu_sector, int(Nitermax, c_int), tol, 10_c_int, c_null_ptr, eval, &
         evec_dev, niter), "gpu_sp_lanc_eigh_dev")
  end subroutine gpu_sp_lanc_eigh_dev

  !> call sp_eigh(spHtimesV_p, eig_values, eig_basis, Nblock, Nitermax, tol=lanc_tolerance)
  !! (the default lanc_method = "arpack", ED_NORMAL/ED_DIAG_NORMAL.f90:179-196): the lowest size(eig_values)
  !! eigenpairs by thick-restart Lanczos on an Nblock-dimensional device-resident basis; Nitermax bounds the
  !! number of restarts as ARPACK's maxiter does
  subroutine gpu_sp_eigh_d(eig_values, eig_basis, Nblock, Nitermax, tol)
    real(8), intent(inout) :: eig_values(:)
    real(8), intent(inout), target :: eig_basis(:,:)
    integer, intent(in) :: Nblock, Nitermax
    real(8), intent(in), optional :: tol
    real(c_double) :: tol_
    integer(c_int) :: nconv, nmv
    if (.not. c_associated(gpu_sector)) stop "gpu_sp_eigh_d: Hsector NOT allocated"
    ! lanc_tolerance defaults to 1d-18 in the reference, below what a residual reaches in double precision: ask for
    ! 1d-12 (relative residual) at most.  Should a sector not even allow that, the solver has stopped at rounding
    ! level (include/edigpu.h) and the pairs are taken if they meet 1d-9.
    tol_ = 1d-12; if (present(tol)) tol_ = max(tol, 1d-12)
    call gpu_check(edigpu_lanczos_eigh_multi(gpu_sector, int(size(eig_values), c_int), int(Nblock, c_int), tol_, &
         int(Nitermax, c_int), c_null_ptr, eig_values, c_loc(eig_basis), nconv, nmv), "gpu_sp_eigh_d")
    if (nconv < size(eig_values) .and. tol_ < 1d-9) then   ! (a caller's looser tolerance is not tightened)
       call gpu_check(edigpu_lanczos_eigh_multi(gpu_sector, int(size(eig_values), c_int), int(Nblock, c_int), 1d-9, &
            int(Nitermax, c_int), c_null_ptr, eig_values, c_loc(eig_basis), nconv, nmv), "gpu_sp_eigh_d")
    end if
    if (nconv < size(eig_values)) stop "gpu_sp_eigh_d: not all eigenpairs converged"
  end subroutine gpu_sp_eigh_d

  subroutine gpu_sp_eigh_c(eig_values, eig_basis, Nblock, Nitermax, tol)
    real(8), intent(inout) :: eig_values(:)
    complex(8), intent(inout), target :: eig_basis(:,:)
    integer, intent(in) :: Nblock, Nitermax
    real(8), intent(in), optional :: tol
    real(c_double) :: tol_
    integer(c_int) :: nconv, nmv
    if (.not. c_associated(gpu_sector)) stop "gpu_sp_eigh_c: Hsector NOT allocated"
    ! lanc_tolerance defaults to 1d-18 in the reference, below what a residual reaches in double precision: ask for
    ! 1d-12 (relative residual) at most.  Should a sector not even allow that, the solver has stopped at rounding
    ! level (include/edigpu.h) and the pairs are taken if they meet 1d-9.
    tol_ = 1d-12; if (present(tol)) tol_ = max(tol, 1d-12)
    call gpu_check(edigpu_lanczos_eigh_multi(gpu_sector, int(size(eig_values), c_int), int(Nblock, c_int), tol_, &
         int(Nitermax, c_int), c_null_ptr, eig_values, c_loc(eig_basis), nconv, nmv), "gpu_sp_eigh_c")
    if (nconv < size(eig_values) .and. tol_ < 1d-9) then   ! (a caller's looser tolerance is not tightened)
       call gpu_check(edigpu_lanczos_eigh_multi(gpu_sector, int(size(eig_values), c_int), int(Nblock, c_int), 1d-9, &
            int(Nitermax, c_int), c_null_ptr, eig_values, c_loc(eig_basis), nconv, nmv), "gpu_sp_eigh_c")
    end if
    if (nconv < size(eig_values)) stop "gpu_sp_eigh_c: not all eigenpairs converged"
  end subroutine gpu_sp_eigh_c

  !> device vectors of n real(8) (ncomplex = 2 n for complex(8)) elements
  function gpu_vec_alloc(n) result(p)
    integer, intent(in) :: n
    type(c_ptr) :: p
    call gpu_check(edigpu_dev_alloc(8_c_int64_t * int(n, c_int64_t), p), "gpu_vec_alloc")
  end function gpu_vec_alloc

  subroutine gpu_vec_free(p)
    type(c_ptr), intent(inout) :: p
    call gpu_check(edigpu_dev_free(p), "gpu_vec_free")
    p = c_null_ptr
  end subroutine gpu_vec_free

  subroutine gpu_vec_upload_d(p, v)
    type(c_ptr), intent(in) :: p
    real(8), intent(in), target :: v(:)
    call gpu_check(edigpu_dev_upload(p, c_loc(v), 8_c_int64_t * size(v, kind=c_int64_t)), "gpu_vec_upload_d")
  end subroutine gpu_vec_upload_d

  subroutine gpu_vec_download_d(v, p)
    real(8), intent(inout), target :: v(:)
    type(c_ptr), intent(in) :: p
    call gpu_check(edigpu_dev_download(c_loc(v), p, 8_c_int64_t * size(v, kind=c_int64_t)), "gpu_vec_download_d")
  end subroutine gpu_vec_download_d

  subroutine gpu_vec_upload_c(p, v)
    type(c_ptr), intent(in) :: p
    complex(8), intent(in), target :: v(:)
    call gpu_check(edigpu_dev_upload(p, c_loc(v), 16_c_int64_t * size(v, kind=c_int64_t)), "gpu_vec_upload_c")
  end subroutine gpu_vec_upload_c

  subroutine gpu_vec_download_c(v, p)
    complex(8), intent(inout), target :: v(:)
    type(c_ptr), intent(in) :: p
    call gpu_check(edigpu_dev_download(c_loc(v), p, 16_c_int64_t * size(v, kind=c_int64_t)), "gpu_vec_download_c")
  end subroutine gpu_vec_download_c

  !> vvinit = apply_op_CDG(v_state, iorb, ispin, isector, jsector) / apply_op_C (ED_SECTOR.f90:465-536, called from
  !! ED_NORMAL/ED_GF_NORMAL.f90:155,167 on the master rank): device vector of sector hsrc -> device vector of sector
  !! hdst.  iorb, ispin 1-based as in the reference; cdg = .true. for c^+.  Normal-mode handles from gpu_build_normal,
  !! superc / nonsu2 handles from gpu_build_flat.
  subroutine gpu_apply_op(hsrc, hdst, v_src_dev, v_dst_dev, iorb, ispin, cdg, flat)
    type(c_ptr), intent(in) :: hsrc, hdst, v_src_dev, v_dst_dev
    integer, intent(in) :: iorb, ispin
    logical, intent(in) :: cdg
    logical, intent(in), optional :: flat
    integer(c_int) :: cr
    logical :: flat_
    cr = 0; if (cdg) cr = 1
    flat_ = .false.; if (present(flat)) flat_ = flat
    if (flat_) then
       call gpu_check(edigpu_apply_op_flat(hsrc, hdst, v_src_dev, v_dst_dev, int(iorb-1, c_int), int(ispin-1, c_int), &
            cr, c_null_ptr), "gpu_apply_op(flat)")
    else
       call gpu_check(edigpu_apply_op_normal(hsrc, hdst, v_src_dev, v_dst_dev, int(iorb-1, c_int), int(ispin-1, c_int), &
            cr, c_null_ptr), "gpu_apply_op")
    end if
  end subroutine gpu_apply_op

  !> vvinit = apply_Cops(v_state, coefs, Os, orbs, spins, isector, jsector) (ED_SECTOR.f90:839-960; the mixed seeds
  !! of the off-diagonal Green's functions, ED_NORMAL/ED_GF_NORMAL.f90:216-261): Os(i) = +1 for c^+, -1 for c
  subroutine gpu_apply_cops(hsrc, hdst, v_src_dev, v_dst_dev, coefs, Os, orbs, spins)
    type(c_ptr), intent(in) :: hsrc, hdst, v_src_dev, v_dst_dev
    real(8), intent(in) :: coefs(:)
    integer, intent(in) :: Os(:), orbs(:), spins(:)
    integer(c_int32_t) :: o(size(Os)), io(size(Os)), is(size(Os))
    o = int(Os, c_int32_t); io = int(orbs - 1, c_int32_t); is = int(spins - 1, c_int32_t)
    call gpu_check(edigpu_apply_cops_normal(hsrc, hdst, v_src_dev, v_dst_dev, int(size(Os), c_int), coefs, o, io, is, &
         c_null_ptr), "gpu_apply_cops")
  end subroutine gpu_apply_cops

  !> tridiag_Hv_sector_*(jsector, vvinit, alfa_, beta_, norm2) with the seed already on the device: the live sector
  !! is the one the seed belongs to; norm2 = <vvinit|vvinit> as the reference returns it
  subroutine gpu_lanc_tridiag_dev(vin_dev, alanc, blanc, norm2)
    type(c_ptr), intent(in) :: vin_dev
    real(8), intent(inout) :: alanc(:), blanc(:)
    real(8), intent(out) :: norm2
    integer(c_int) :: niter
    if (.not. c_associated(gpu_sector)) stop "gpu_lanc_tridiag_dev: Hsector NOT allocated"
    call gpu_check(edigpu_lanczos_tridiag_dev(gpu_sector, vin_dev, int(size(alanc), c_int), alanc, blanc, &
         lanc_threshold, niter, norm2), "gpu_lanc_tridiag_dev")
  end subroutine gpu_lanc_tridiag_dev

  ! ---- N > 1: one MPI rank per GPU -------------------------------------------------------------

  !> rank 0 makes the 128-byte RCCL id; the host broadcasts it (call MPI_Bcast(id, 128, MPI_BYTE, 0, MpiComm, ierr))
  subroutine gpu_comm_unique_id(id)
    character(kind=c_char), intent(out) :: id(128)
    call gpu_check(edigpu_comm_unique_id(id), "gpu_comm_unique_id")
  end subroutine gpu_comm_unique_id

  !> MpiComm's counterpart: RCCL over xGMI, rank = get_Rank_MPI(MpiComm), world = get_Size_MPI(MpiComm)
  !! (where the reference builds its communicator: ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:99-125)
  subroutine gpu_comm_create(rank, world, id)
    integer, intent(in) :: rank, world
    character(kind=c_char), intent(in) :: id(128)
    if (c_associated(gpu_comm)) stop "gpu_comm_create: a communicator is already live"
    call gpu_check(edigpu_comm_create(gpu_comm, int(rank, c_int32_t), int(world, c_int32_t), id), "gpu_comm_create")
  end subroutine gpu_comm_create

  !> host-staged transport through POSIX shared memory: ranks of one node that share a GPU (tests, hosts without RCCL)
  subroutine gpu_comm_create_shm(rank, world, name, slot_bytes)
    integer, intent(in) :: rank, world
    character(len=*), intent(in) :: name
    integer(c_int64_t), intent(in) :: slot_bytes
    character(kind=c_char) :: cname(len_trim(name)+1)
    integer :: i
    if (c_associated(gpu_comm)) stop "gpu_comm_create_shm: a communicator is already live"
    do i = 1, len_trim(name)
       cname(i) = name(i:i)
    end do
    cname(len_trim(name)+1) = c_null_char
    call gpu_check(edigpu_comm_create_shm(gpu_comm, int(rank, c_int32_t), int(world, c_int32_t), cname, slot_bytes), &
         "gpu_comm_create_shm")
  end subroutine gpu_comm_create_shm

  subroutine gpu_comm_destroy()
    if (c_associated(gpu_comm)) call gpu_check(edigpu_comm_destroy(gpu_comm), "gpu_comm_destroy")
    gpu_comm = c_null_ptr
  end subroutine gpu_comm_destroy

  !> this rank's share of `units` (DimDw in normal mode, Dim in superc / nonsu2): first is 0-based
  subroutine gpu_shard_plan(units, rank, world, first, count)
    integer, intent(in) :: units, rank, world
    integer, intent(out) :: first, count
    integer(c_int64_t) :: f, n, q
    call gpu_check(edigpu_shard_plan(int(units, c_int64_t), int(world, c_int32_t), int(rank, c_int32_t), f, n, q), &
         "gpu_shard_plan")
    first = int(f); count = int(n)
  end subroutine gpu_shard_plan

  !> spMatVec_mpi_normal_main / spMatVec_mpi_superc_main / ... (dd_sparse_HxV interface on SHARDS, Nloc =
  !! vecDim_Hv_sector_*): the exchange happens inside the library.  spHtimesV_p => spMatVec_mpi_gpu_d
  subroutine spMatVec_mpi_gpu_d(Nloc, v, Hv)
    integer :: Nloc
    real(8), dimension(Nloc) :: v, Hv
    if (.not. c_associated(gpu_sector)) stop "spMatVec_mpi_gpu_d: Hsector NOT allocated"
    if (.not. c_associated(gpu_comm)) stop "spMatVec_mpi_gpu_d: no communicator"
    call gpu_check(edigpu_apply_sharded_d(gpu_sector, gpu_comm, int(Nloc, c_int64_t), v, Hv), "spMatVec_mpi_gpu_d")
  end subroutine spMatVec_mpi_gpu_d

  subroutine spMatVec_mpi_gpu_c(Nloc, v, Hv)
    integer :: Nloc
    complex(8), dimension(Nloc) :: v, Hv
    if (.not. c_associated(gpu_sector)) stop "spMatVec_mpi_gpu_c: Hsector NOT allocated"
    if (.not. c_associated(gpu_comm)) stop "spMatVec_mpi_gpu_c: no communicator"
    call gpu_check(edigpu_apply_sharded_z(gpu_sector, gpu_comm, int(Nloc, c_int64_t), v, Hv), "spMatVec_mpi_gpu_c")
  end subroutine spMatVec_mpi_gpu_c

  !> call sp_lanc_tridiag(MpiComm, spHtimesV_p, vvloc, alanc, blanc) (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:357-365):
  !! vvloc = this rank's slice as scatter_vector_MPI leaves it; norm2 over all ranks (the reference computes it on
  !! the master before scattering, :344-349)
  subroutine gpu_lanc_tridiag_mpi_d(vvloc, alanc, blanc, norm2)
    real(8), intent(in), target :: vvloc(:)
    real(8), intent(inout) :: alanc(:), blanc(:)
    real(8), intent(out), optional :: norm2
    integer(c_int) :: niter
    real(c_double) :: n2
    if (.not. c_associated(gpu_sector)) stop "gpu_lanc_tridiag_mpi_d: Hsector NOT allocated"
    if (.not. c_associated(gpu_comm)) stop "gpu_lanc_tridiag_mpi_d: no communicator"
    call gpu_check(edigpu_lanczos_tridiag_sharded(gpu_sector, gpu_comm, c_loc(vvloc), int(size(alanc), c_int), &
         alanc, blanc, lanc_threshold, niter, n2), "gpu_lanc_tridiag_mpi_d")
    if (present(norm2)) norm2 = n2
  end subroutine gpu_lanc_tridiag_mpi_d

  subroutine gpu_lanc_tridiag_mpi_c(vvloc, alanc, blanc, norm2)
    complex(8), intent(in), target :: vvloc(:)
    real(8), intent(inout) :: alanc(:), blanc(:)
    real(8), intent(out), optional :: norm2
    integer(c_int) :: niter
    real(c_double) :: n2
    if (.not. c_associated(gpu_sector)) stop "gpu_lanc_tridiag_mpi_c: Hsector NOT allocated"
    if (.not. c_associated(gpu_comm)) stop "gpu_lanc_tridiag_mpi_c: no communicator"
    call gpu_check(edigpu_lanczos_tridiag_sharded(gpu_sector, gpu_comm, c_loc(vvloc), int(size(alanc), c_int), &
         alanc, blanc, lanc_threshold, niter, n2), "gpu_lanc_tridiag_mpi_c")
    if (present(norm2)) norm2 = n2
  end subroutine gpu_lanc_tridiag_mpi_c

  !> call sp_eigh(MpiComm, spHtimesV_p, eig_values, eig_basis, Nblock, Nitermax, tol=lanc_tolerance) -- the spectrum
  !! solve of a -D_MPI build (ED_NORMAL/ED_DIAG_NORMAL.f90:221-242): eig_basis(:, k) = this rank's shard of eigenvector
  !! k, as PARPACK leaves it (and as es_return_dvector_mpi hands it on, ED_EIGENSPACE.f90:723-793).  Every vector of
  !! the solve is a device-resident shard; only the Gram-Schmidt coefficients and norms cross ranks.
  subroutine gpu_sp_eigh_mpi_d(eig_values, eig_basis, Nblock, Nitermax, tol)
    real(8), intent(inout) :: eig_values(:)
    real(8), intent(inout), target :: eig_basis(:,:)
    integer, intent(in) :: Nblock, Nitermax
    real(8), intent(in), optional :: tol
    real(c_double) :: tol_
    integer(c_int) :: nconv, nmv
    type(c_ptr) :: pv
    if (.not. c_associated(gpu_sector)) stop "gpu_sp_eigh_mpi_d: Hsector NOT allocated"
    if (.not. c_associated(gpu_comm)) stop "gpu_sp_eigh_mpi_d: no communicator"
    tol_ = 1d-12; if (present(tol)) tol_ = max(tol, 1d-12)
    pv = c_null_ptr; if (size(eig_basis) > 0) pv = c_loc(eig_basis)
    call gpu_check(edigpu_lanczos_eigh_multi_sharded(gpu_sector, gpu_comm, int(size(eig_values), c_int), int(Nblock, c_int), &
         tol_, int(Nitermax, c_int), c_null_ptr, eig_values, pv, nconv, nmv), "gpu_sp_eigh_mpi_d")
    if (nconv < size(eig_values) .and. tol_ < 1d-9) then
       call gpu_check(edigpu_lanczos_eigh_multi_sharded(gpu_sector, gpu_comm, int(size(eig_values), c_int), &
            int(Nblock, c_int), 1d-9, int(Nitermax, c_int), c_null_ptr, eig_values, pv, nconv, nmv), "gpu_sp_eigh_mpi_d")
    end if
    if (nconv < size(eig_values)) stop "gpu_sp_eigh_mpi_d: not all eigenpairs converged"
  end subroutine gpu_sp_eigh_mpi_d

  subroutine gpu_sp_eigh_mpi_c(eig_values, eig_basis, Nblock, Nitermax, tol)
    real(8), intent(inout) :: eig_values(:)
    complex(8), intent(inout), target :: eig_basis(:,:)
    integer, intent(in) :: Nblock, Nitermax
    real(8), intent(in), optional :: tol
    real(c_double) :: tol_
    integer(c_int) :: nconv, nmv
    type(c_ptr) :: pv
    if (.not. c_associated(gpu_sector)) stop "gpu_sp_eigh_mpi_c: Hsector NOT allocated"
    if (.not. c_associated(gpu_comm)) stop "gpu_sp_eigh_mpi_c: no communicator"
    tol_ = 1d-12; if (present(tol)) tol_ = max(tol, 1d-12)
    pv = c_null_ptr; if (size(eig_basis) > 0) pv = c_loc(eig_basis)
    call gpu_check(edigpu_lanczos_eigh_multi_sharded(gpu_sector, gpu_comm, int(size(eig_values), c_int), int(Nblock, c_int), &
         tol_, int(Nitermax, c_int), c_null_ptr, eig_values, pv, nconv, nmv), "gpu_sp_eigh_mpi_c")
    if (nconv < size(eig_values) .and. tol_ < 1d-9) then
       call gpu_check(edigpu_lanczos_eigh_multi_sharded(gpu_sector, gpu_comm, int(size(eig_values), c_int), &
            int(Nblock, c_int), 1d-9, int(Nitermax, c_int), c_null_ptr, eig_values, pv, nconv, nmv), "gpu_sp_eigh_mpi_c")
    end if
    if (nconv < size(eig_values)) stop "gpu_sp_eigh_mpi_c: not all eigenpairs converged"
  end subroutine gpu_sp_eigh_mpi_c

  !> vvloc = apply_op_C / apply_op_CDG(v_state, ...) followed by scatter_vector_MPI (ED_NORMAL/ED_GF_NORMAL.f90:141-175),
  !! without the master rank: v_shard = this rank's shard of the eigenvector (sector hsrc), vv_shard = this rank's shard
  !! of the seed (sector hdst).  iorb 1-based, ispin 1 = up / 2 = down as in the reference.
  subroutine gpu_apply_op_mpi_d(hsrc, hdst, v_shard, vv_shard, iorb, ispin, create)
    type(c_ptr), intent(in) :: hsrc, hdst
    real(8), intent(in), target :: v_shard(:)
    real(8), intent(inout), target :: vv_shard(:)
    integer, intent(in) :: iorb, ispin
    logical, intent(in) :: create
    real(c_double) :: coef2(2)
    integer(c_int32_t) :: cr(1), io(1), sp(1)
    type(c_ptr) :: p1, p2
    if (.not. c_associated(gpu_comm)) stop "gpu_apply_op_mpi_d: no communicator"
    coef2 = [1d0, 0d0]
    cr(1) = -1; if (create) cr(1) = 1
    io(1) = int(iorb - 1, c_int32_t); sp(1) = int(ispin - 1, c_int32_t)
    p1 = c_null_ptr; if (size(v_shard) > 0) p1 = c_loc(v_shard)
    p2 = c_null_ptr; if (size(vv_shard) > 0) p2 = c_loc(vv_shard)
    call gpu_check(edigpu_apply_cops_sharded(hsrc, hdst, gpu_comm, p1, p2, 1_c_int, coef2, cr, io, sp), "gpu_apply_op_mpi_d")
  end subroutine gpu_apply_op_mpi_d

  !> delete_Hv_sector_* counterpart (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:212-279)
  subroutine gpu_delete_sector()
    if (c_associated(gpu_sector)) then
       if (.not. gpu_sector_borrowed) call gpu_check(edigpu_destroy(gpu_sector), "gpu_delete_sector")
       gpu_sector = c_null_ptr
       gpu_sector_borrowed = .false.
    end if
  end subroutine gpu_delete_sector

  !> per-solve cache of sector handles (row f2): the reference builds the sector Hamiltonian anew in every
  !! tridiag_Hv_sector_* (build_Hv_sector_* ... delete_Hv_sector_*); with the cache a repeated request for the same
  !! (model, sector) returns the handle that is already on the device.  max_mb: device-memory budget.
  subroutine gpu_cache_create(max_mb)
    integer, intent(in) :: max_mb
    if (c_associated(gpu_cache)) return
    call gpu_check(edigpu_cache_create(gpu_cache, int(max_mb, c_int64_t) * 1048576_c_int64_t), "gpu_cache_create")
  end subroutine gpu_cache_create

  !> in place of gpu_build_normal / gpu_build_flat / gpu_build_normal_cmplx inside build_Hv_sector_*: kind 0 normal
  !! (q1, q2 = Nup, Ndw), 1 stored superc / nonsu2 (q1 = Sz / Ntot), 2 on-the-fly, 3 normal with complex algebra.  The live
  !! handle then belongs to the cache: delete_Hv_sector_* (gpu_delete_sector) only lets go of it.
  subroutine gpu_build_cached(m, kind, q1, q2)
    type(edigpu_model_t), intent(in) :: m
    integer, intent(in) :: kind, q1, q2
    if (.not. c_associated(gpu_cache)) stop "gpu_build_cached: call gpu_cache_create first"
    if (c_associated(gpu_sector)) stop "gpu_build_cached: a sector is already allocated"
    call gpu_check(edigpu_cache_get(gpu_cache, m, int(kind, c_int), int(q1, c_int), int(q2, c_int), gpu_sector), &
         "gpu_build_cached")
    gpu_sector_borrowed = .true.
  end subroutine gpu_build_cached

  subroutine gpu_cache_stats(hits, misses, evictions)
    integer, intent(out) :: hits, misses, evictions
    integer(c_int64_t) :: st(5)
    call gpu_check(edigpu_cache_stats(gpu_cache, st), "gpu_cache_stats")
    hits = int(st(1)); misses = int(st(2)); evictions = int(st(3))
  end subroutine gpu_cache_stats

  !> a new bath (the next DMFT iteration) makes every cached sector stale: drop them
  subroutine gpu_cache_clear()
    if (c_associated(gpu_cache)) call gpu_check(edigpu_cache_clear(gpu_cache), "gpu_cache_clear")
  end subroutine gpu_cache_clear

  subroutine gpu_cache_destroy()
    if (c_associated(gpu_cache)) call gpu_check(edigpu_cache_destroy(gpu_cache), "gpu_cache_destroy")
    gpu_cache = c_null_ptr
  end subroutine gpu_cache_destroy

end module EDIGPU_SHIM
