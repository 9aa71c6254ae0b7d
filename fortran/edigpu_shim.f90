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

  integer, parameter, public :: EDIGPU_MAXORB = 5, EDIGPU_MAXBATH = 16
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
  end type edigpu_model_t

  interface
     function edigpu_last_error() bind(C, name="edigpu_last_error") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function edigpu_last_error
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
  public :: gpu_build_normal, gpu_build_normal_cmplx, gpu_build_flat, gpu_build_orbs
  public :: spMatVec_gpu_d, spMatVec_gpu_c
  public :: gpu_lanc_tridiag_d, gpu_lanc_tridiag_c
  public :: flatten_rows_count

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

  !> delete_Hv_sector_* counterpart (ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:212-279)
  subroutine gpu_delete_sector()
    if (c_associated(gpu_sector)) then
       call gpu_check(edigpu_destroy(gpu_sector), "gpu_delete_sector")
       gpu_sector = c_null_ptr
    end if
  end subroutine gpu_delete_sector

end module EDIGPU_SHIM
