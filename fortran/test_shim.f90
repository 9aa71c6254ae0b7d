!> Fortran-side check of the drop-in boundary: a Fortran host hands a small Kronecker sector
!! (Hd, Hup, Hdw as rows flattened the way the reference's sparse_matrix_csr rows would be) and a
!! complex flat CSR to libedigpu.so through EDIGPU_SHIM, calls the dd_sparse_HxV / cc_sparse_HxV
!! compatible procedures through procedure pointers (as spHtimesV_p / spHtimesV_cc are used), and
!! compares with a plain Fortran product.  Run on the GPU box (tests/test_fortran_shim.py).
program test_shim
  use, intrinsic :: iso_c_binding
  use EDIGPU_SHIM
  implicit none

  abstract interface
     subroutine dd_sparse_HxV(Nloc, v, Hv)
       integer :: Nloc
       real(8), dimension(Nloc) :: v, Hv
     end subroutine dd_sparse_HxV
     subroutine cc_sparse_HxV(Nloc, v, Hv)
       integer :: Nloc
       complex(8), dimension(Nloc) :: v, Hv
     end subroutine cc_sparse_HxV
  end interface
  procedure(dd_sparse_HxV), pointer :: spHtimesV_p => null()
  procedure(cc_sparse_HxV), pointer :: spHtimesV_cc => null()

  integer, parameter :: DimUp = 6, DimDw = 5, N = DimUp*DimDw
  real(8) :: hup(DimUp,DimUp), hdw(DimDw,DimDw), hd(N), v(N), hv(N), ref(N)
  integer(c_int64_t) :: up_rp(0:DimUp), dw_rp(0:DimDw), nd_rp(0:N)
  integer(c_int32_t), allocatable :: up_col(:), dw_col(:), nd_col(:)
  real(8), allocatable :: up_val(:), dw_val(:), nd_val(:)
  real(8) :: alanc(8), blanc(8), err
  integer :: i, j, iup, idw, k, nfail
  character(len=64) :: arg1, arg2, arg3, arg4

  nfail = 0
  call gpu_init(0)
  ! `test_shim mpi <rank> <world> <name>`: one rank of a Fortran + "MPI" host whose ranks share this GPU (the
  ! communicator is the library's shared-memory transport; tests/test_fortran_shim.py starts the ranks)
  if (command_argument_count() >= 4) then
     call get_command_argument(1, arg1); call get_command_argument(2, arg2)
     call get_command_argument(3, arg3); call get_command_argument(4, arg4)
     if (trim(arg1) == "mpi") then
        read(arg2, *) i
        read(arg3, *) j
        call test_mpi_rank(i, j, trim(arg4), nfail)
        if (nfail == 0) then
           write(*,"(A,I2)") "FORTRAN SHIM MPI OK rank", i
        else
           write(*,"(A,I3)") "FORTRAN SHIM MPI FAILED checks:", nfail
           stop 1
        end if
        stop
     end if
  end if

  ! ---- a small symmetric Kronecker problem ----
  hup = 0d0; hdw = 0d0
  do i = 1, DimUp-1
     hup(i,i+1) = 0.3d0 + 0.1d0*i; hup(i+1,i) = hup(i,i+1)
  end do
  hup(1,DimUp) = -0.7d0; hup(DimUp,1) = -0.7d0
  do i = 1, DimDw-1
     hdw(i,i+1) = -0.2d0*i; hdw(i+1,i) = hdw(i,i+1)
  end do
  do i = 1, N
     hd(i) = 0.05d0*i - 1d0
     v(i) = sin(0.37d0*i)
  end do
  call dense_to_rows(hup, up_rp, up_col, up_val)
  call dense_to_rows(hdw, dw_rp, dw_col, dw_val)
  nd_rp = 0; allocate(nd_col(1), nd_val(1)); nd_col = 0; nd_val = 0d0
  call gpu_set_normal(DimUp, DimDw, 0, DimDw, hd, up_rp, up_col, up_val, dw_rp, dw_col, dw_val, &
       .false., nd_rp, nd_col, nd_val)
  spHtimesV_p => spMatVec_gpu_d
  call spHtimesV_p(N, v, hv)
  ! reference: spMatVec_normal_main loops (ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:517-650)
  ref = hd*v
  do idw = 1, DimDw
     do iup = 1, DimUp
        i = iup + (idw-1)*DimUp
        do j = 1, DimDw
           ref(i) = ref(i) + hdw(idw,j)*v(iup + (j-1)*DimUp)
        end do
        do j = 1, DimUp
           ref(i) = ref(i) + hup(iup,j)*v(j + (idw-1)*DimUp)
        end do
     end do
  end do
  err = maxval(abs(hv-ref))/maxval(abs(ref))
  write(*,"(A,ES10.2)") "normal H*v through spHtimesV_p: max rel err =", err
  if (err > 1d-13) nfail = nfail + 1
  ! device-resident sp_lanc_tridiag replacement: alpha_1 = <v|H|v>/<v|v>
  call gpu_lanc_tridiag_d(v, alanc, blanc)
  err = abs(alanc(1) - dot_product(v, ref)/dot_product(v, v))
  write(*,"(A,ES10.2)") "lanczos alpha(1) vs <v|H|v>/<v|v>: abs err =", err
  if (err > 1d-12) nfail = nfail + 1
  call gpu_delete_sector()

  ! ---- a small Hermitian complex CSR through spHtimesV_cc ----
  call test_complex(nfail)

  ! ---- library-built sector from the model globals (no ed_buildh_* on the host) ----
  call test_model_build(nfail)

  ! ---- SURVEY.md 8(f) rows f1 / f3 from the Fortran host: eigensolvers, device-resident c^+ and tridiagonalisation ----
  call test_next_rows(nfail)

  ! ---- the N > 1 entry points with a world of one (RCCL communicator) ----
  call test_comm_world_of_one(nfail)
  call test_sector_cache(nfail)

  if (nfail == 0) then
     write(*,"(A)") "FORTRAN SHIM OK"
  else
     write(*,"(A,I3)") "FORTRAN SHIM FAILED checks:", nfail
     stop 1
  end if

contains

  subroutine dense_to_rows(a, rp, col, val)
    real(8), intent(in) :: a(:,:)
    integer(c_int64_t), intent(out) :: rp(0:)
    integer(c_int32_t), allocatable, intent(out) :: col(:)
    real(8), allocatable, intent(out) :: val(:)
    integer :: sizes(size(a,1)), ii, jj, kk
    do ii = 1, size(a,1)
       sizes(ii) = count(a(ii,:) /= 0d0)
    end do
    call flatten_rows_count(sizes, rp)
    allocate(col(max(1,sum(sizes))), val(max(1,sum(sizes))))
    kk = 0
    do ii = 1, size(a,1)
       do jj = size(a,2), 1, -1          ! unsorted on purpose: insertion order is arbitrary upstream
          if (a(ii,jj) /= 0d0) then
             kk = kk + 1
             col(kk) = jj - 1            ! 0-based for the C side
             val(kk) = a(ii,jj)
          end if
       end do
    end do
  end subroutine dense_to_rows

  subroutine test_complex(nfail)
    integer, intent(inout) :: nfail
    integer, parameter :: M = 7
    complex(8) :: h(M,M), x(M), y(M), yref(M)
    integer(c_int64_t) :: rp(0:M)
    integer(c_int32_t), allocatable :: col(:)
    complex(8), allocatable :: val(:)
    integer :: sizes(M), ii, jj, kk
    real(8) :: e
    h = (0d0,0d0)
    do ii = 1, M
       h(ii,ii) = cmplx(0.5d0*ii, 0d0, 8)
       if (ii < M) then
          h(ii,ii+1) = cmplx(0.1d0*ii, -0.3d0, 8); h(ii+1,ii) = conjg(h(ii,ii+1))
       end if
       x(ii) = cmplx(cos(1d0*ii), sin(2d0*ii), 8)
    end do
    do ii = 1, M
       sizes(ii) = count(h(ii,:) /= (0d0,0d0))
    end do
    call flatten_rows_count(sizes, rp)
    allocate(col(sum(sizes)), val(sum(sizes)))
    kk = 0
    do ii = 1, M
       do jj = 1, M
          if (h(ii,jj) /= (0d0,0d0)) then
             kk = kk + 1; col(kk) = jj - 1; val(kk) = h(ii,jj)
          end if
       end do
    end do
    call gpu_set_csr_c(M, M, 0, rp, col, val)
    spHtimesV_cc => spMatVec_gpu_c
    call spHtimesV_cc(M, x, y)
    yref = matmul(h, x)
    e = maxval(abs(y-yref))/maxval(abs(yref))
    write(*,"(A,ES10.2)") "complex CSR H*v through spHtimesV_cc: max rel err =", e
    if (e > 1d-13) nfail = nfail + 1
    call gpu_delete_sector()
  end subroutine test_complex

  !> single-orbital Anderson impurity with one bath level, sector (Nup,Ndw) = (1,1): 4 x 4, known in
  !! closed form.  Basis index i = iup + (idw-1)*2 with iup/idw = 1: electron on the impurity,
  !! 2: electron on the bath level (ascending integers, ED_SECTOR.f90:217-242).
  subroutine test_model_build(nfail)
    integer, intent(inout) :: nfail
    type(edigpu_model_t) :: m
    real(8), parameter :: ed = -0.4d0, eb = 0.7d0, vhyb = 0.35d0, u = 2.0d0
    real(8) :: h(4,4), x(4), y(4), yref(4), e
    real(8) :: ebath(1,1,1), vbath(1,1,1), zero2(1,1)
    complex(8) :: hloc(1,1,1,1)
    integer :: ii
    m%ed_mode = 0; m%bath_type = 0; m%norb = 1; m%nbath = 1; m%nspin = 1; m%hfmode = 0; m%xmu = 0d0
    zero2 = 0d0
    call gpu_model_set_kanamori(m, [u], zero2, zero2, zero2, zero2)
    hloc = cmplx(ed, 0d0, 8)
    call gpu_model_set_hloc(m, hloc)
    ebath = eb; vbath = vhyb
    call gpu_model_set_bath(m, ebath, vbath)
    call gpu_build_normal(m, 1, 1, 0, -1)
    h = 0d0
    h(1,1) = 2d0*ed + u; h(2,2) = eb + ed; h(3,3) = ed + eb; h(4,4) = 2d0*eb
    h(1,2) = vhyb; h(2,1) = vhyb; h(3,4) = vhyb; h(4,3) = vhyb      ! up hop, idw fixed
    h(1,3) = vhyb; h(3,1) = vhyb; h(2,4) = vhyb; h(4,2) = vhyb      ! down hop, iup fixed
    do ii = 1, 4
       x(ii) = 0.3d0*ii - 0.5d0
    end do
    spHtimesV_p => spMatVec_gpu_d
    call spHtimesV_p(4, x, y)
    yref = matmul(h, x)
    e = maxval(abs(y-yref))/maxval(abs(yref))
    write(*,"(A,ES10.2)") "model-built Anderson sector (1,1) H*v: max rel err =", e
    if (e > 1d-13) nfail = nfail + 1
    call gpu_delete_sector()
  end subroutine test_model_build

  !> Norb = 1, Nbath = 3 Anderson model (Ns = 4) with fixed parameters
  subroutine anderson_model(m)
    type(edigpu_model_t), intent(out) :: m
    real(8) :: ebath(1,1,3), vbath(1,1,3), zero2(1,1)
    complex(8) :: hloc(1,1,1,1)
    m%ed_mode = 0; m%bath_type = 0; m%norb = 1; m%nbath = 3; m%nspin = 1; m%hfmode = 0; m%xmu = 0d0
    zero2 = 0d0
    call gpu_model_set_kanamori(m, [2.0d0], zero2, zero2, zero2, zero2)
    hloc = cmplx(-0.9d0, 0d0, 8)
    call gpu_model_set_hloc(m, hloc)
    ebath(1,1,:) = [-0.8d0, 0.1d0, 0.9d0]; vbath(1,1,:) = [0.3d0, 0.45d0, 0.25d0]
    call gpu_model_set_bath(m, ebath, vbath)
  end subroutine anderson_model

  subroutine test_next_rows(nfail)
    integer, intent(inout) :: nfail
    type(edigpu_model_t) :: m
    type(c_ptr) :: hgs, hexc, vgs_dev, seed_dev
    real(8), allocatable :: evec(:), hv(:), seed(:), hseed(:), basis(:,:)
    real(8) :: e0, evals(3), al(12), bl(12), norm2, e
    integer :: n, n2
    call anderson_model(m)
    call gpu_build_normal(m, 2, 2, 0, -1)              ! sector (2,2): 6 x 6
    n = gpu_sector_dim(gpu_sector_handle())
    allocate(evec(n), hv(n), basis(n,3))
    ! row f1: sp_lanc_eigh and sp_eigh replacements
    call gpu_sp_lanc_eigh_d(e0, evec, 200, 1d-14)
    spHtimesV_p => spMatVec_gpu_d
    call spHtimesV_p(n, evec, hv)
    e = maxval(abs(hv - e0*evec))
    write(*,"(A,F14.9,A,ES10.2)") "gpu_sp_lanc_eigh_d: E0 =", e0, "  residual =", e
    if (e > 1d-9 .or. abs(dot_product(evec, evec) - 1d0) > 1d-12) nfail = nfail + 1
    call gpu_sp_eigh_d(evals, basis, 20, 300, 1d-18)        ! the reference's default lanc_tolerance
    e = abs(evals(1) - e0)
    call spHtimesV_p(n, basis(:,2), hv)
    e = max(e, maxval(abs(hv - evals(2)*basis(:,2))))
    write(*,"(A,3F12.7,A,ES10.2)") "gpu_sp_eigh_d: evals =", evals, "  err =", e
    if (e > 1d-9 .or. evals(2) < evals(1) .or. evals(3) < evals(2)) nfail = nfail + 1
    ! row f3: ground state left on the device -> c^+_up -> tridiagonalisation, nothing but alpha/beta/norm2 comes back
    vgs_dev = gpu_vec_alloc(n)
    call gpu_sp_lanc_eigh_dev(e0, vgs_dev, 200, 1d-14)
    hgs = c_null_ptr
    call gpu_sector_swap(hgs)                           ! hgs = the (2,2) sector, no live sector now
    call gpu_build_normal(m, 3, 2, 0, -1)               ! the sector c^+_up leads to
    hexc = gpu_sector_handle()
    n2 = gpu_sector_dim(hexc)
    seed_dev = gpu_vec_alloc(n2)
    call gpu_apply_op(hgs, hexc, vgs_dev, seed_dev, 1, 1, .true.)
    call gpu_lanc_tridiag_dev(seed_dev, al, bl, norm2)
    allocate(seed(n2), hseed(n2))
    call gpu_vec_download_d(seed, seed_dev)
    call spHtimesV_p(n2, seed, hseed)
    e = abs(norm2 - dot_product(seed, seed)) + abs(al(1) - dot_product(seed, hseed)/dot_product(seed, seed))
    write(*,"(A,F12.8,A,ES10.2)") "device-resident c^+|gs> -> tridiag: norm2 =", norm2, "  err =", e
    if (e > 1d-11 .or. norm2 <= 0d0 .or. norm2 >= 1d0) nfail = nfail + 1
    ! apply_Cops with one term must reproduce apply_op
    call gpu_apply_cops(hgs, hexc, vgs_dev, seed_dev, [1d0], [1], [1], [1])
    call gpu_vec_download_d(hseed, seed_dev)
    if (maxval(abs(hseed - seed)) > 1d-15) nfail = nfail + 1
    call gpu_vec_free(seed_dev); call gpu_vec_free(vgs_dev)
    call gpu_sector_destroy(hgs)
    call gpu_delete_sector()
  end subroutine test_next_rows

  subroutine test_comm_world_of_one(nfail)
    integer, intent(inout) :: nfail
    type(edigpu_model_t) :: m
    character(kind=c_char) :: id(128)
    real(8), allocatable :: x(:), y(:), yref(:)
    real(8) :: a1(10), b1(10), a2(10), b2(10), n2, e
    integer :: n, first, count, ii
    call anderson_model(m)
    call gpu_build_normal(m, 2, 2, 0, -1)
    n = gpu_sector_dim(gpu_sector_handle())
    allocate(x(n), y(n), yref(n))
    do ii = 1, n
       x(ii) = cos(0.7d0*ii) + 0.1d0*ii
    end do
    call gpu_comm_unique_id(id)
    call gpu_comm_create(0, 1, id)
    call gpu_shard_plan(6, 0, 1, first, count)
    if (first /= 0 .or. count /= 6) nfail = nfail + 1
    spHtimesV_p => spMatVec_gpu_d
    call spHtimesV_p(n, x, yref)
    spHtimesV_p => spMatVec_mpi_gpu_d
    call spHtimesV_p(n, x, y)
    e = maxval(abs(y - yref))/maxval(abs(yref))
    call gpu_lanc_tridiag_d(x, a1, b1)
    call gpu_lanc_tridiag_mpi_d(x, a2, b2, n2)
    e = max(e, maxval(abs(a1 - a2)), maxval(abs(b1 - b2)), abs(n2 - dot_product(x, x))/n2)
    write(*,"(A,ES10.2)") "world-of-one RCCL communicator, spMatVec_mpi_gpu_d + gpu_lanc_tridiag_mpi_d: err =", e
    if (e > 1d-11) nfail = nfail + 1
    call gpu_comm_destroy()
    call gpu_delete_sector()
  end subroutine test_comm_world_of_one

  !> the per-solve sector cache in the shape of the Green's-function loop: build_Hv_sector(jsector) ... delete_Hv_sector
  !! for the same few sectors again and again; a repeated request must return the handle that is already there
  subroutine test_sector_cache(nfail)
    integer, intent(inout) :: nfail
    type(edigpu_model_t) :: m
    type(c_ptr) :: h1, h2
    real(8) :: a1(6), b1(6), a2(6), b2(6)
    real(8), allocatable :: x(:)
    integer :: hits, misses, evictions, n, ii, k
    call anderson_model(m)
    call gpu_cache_create(256)
    do k = 1, 3
       call gpu_build_cached(m, 0, 2, 2)
       if (k == 1) h1 = gpu_sector_handle()
       h2 = gpu_sector_handle()
       n = gpu_sector_dim(h2)
       if (.not. allocated(x)) then
          allocate(x(n))
          do ii = 1, n
             x(ii) = sin(0.3d0*ii) + 0.05d0*ii
          end do
       end if
       if (k == 1) call gpu_lanc_tridiag_d(x, a1, b1)
       if (k == 3) call gpu_lanc_tridiag_d(x, a2, b2)
       call gpu_delete_sector()                 ! lets go of the borrowed handle, the cache keeps it
       call gpu_build_cached(m, 0, 3, 2)        ! the sector c^+ leads to
       call gpu_delete_sector()
    end do
    call gpu_cache_stats(hits, misses, evictions)
    write(*,"(A,3I4)") "sector cache (hits, misses, evictions):", hits, misses, evictions
    if (.not. c_associated(h1, h2)) nfail = nfail + 1
    if (hits /= 4 .or. misses /= 2 .or. evictions /= 0) nfail = nfail + 1
    if (maxval(abs(a1 - a2)) > 0d0 .or. maxval(abs(b1 - b2)) > 0d0) nfail = nfail + 1
    call gpu_cache_clear()
    call gpu_cache_destroy()
  end subroutine test_sector_cache

  !> one rank of `world`: whole-sector handle (transposed exchange), the product and the tridiagonalisation on this
  !! rank's down rows against the single-process results computed first with the same handle
  subroutine test_mpi_rank(rank, world, name, nfail)
    integer, intent(in) :: rank, world
    character(len=*), intent(in) :: name
    integer, intent(inout) :: nfail
    type(edigpu_model_t) :: m
    real(8), allocatable :: x(:), y(:), yref(:)
    real(8) :: a1(10), b1(10), a2(10), b2(10), n2, e
    integer :: n, first, count, ii, lo, hi, dimup
    call anderson_model(m)
    call gpu_build_normal(m, 2, 2, 0, -1)               ! DimUp = DimDw = 6
    dimup = 6
    n = gpu_sector_dim(gpu_sector_handle())
    allocate(x(n), yref(n))
    do ii = 1, n
       x(ii) = cos(0.7d0*ii) + 0.1d0*ii
    end do
    spHtimesV_p => spMatVec_gpu_d
    call spHtimesV_p(n, x, yref)
    call gpu_lanc_tridiag_d(x, a1, b1)
    call gpu_comm_create_shm(rank, world, name, 1048576_c_int64_t)
    call gpu_shard_plan(6, rank, world, first, count)
    lo = first*dimup + 1; hi = (first + count)*dimup
    allocate(y(max(1, hi - lo + 1)))
    spHtimesV_p => spMatVec_mpi_gpu_d
    call spHtimesV_p(hi - lo + 1, x(lo:hi), y)
    e = 0d0
    if (hi >= lo) e = maxval(abs(y(1:hi-lo+1) - yref(lo:hi)))/maxval(abs(yref))
    call gpu_lanc_tridiag_mpi_d(x(lo:hi), a2, b2, n2)
    e = max(e, maxval(abs(a1 - a2)), maxval(abs(b1 - b2)), abs(n2 - dot_product(x, x))/n2)
    write(*,"(A,I2,A,I2,A,ES10.2)") "rank", rank, " of", world, ": sharded product + tridiagonalisation err =", e
    if (e > 1d-11) nfail = nfail + 1
    ! the spectrum solve on shards (sp_eigh with MpiComm, ED_DIAG_NORMAL.f90:221-242) against the single-process solver,
    ! then the Green's-function seed c_{1,up} |gs> on shards (apply_op_C + scatter_vector_MPI, ED_GF_NORMAL.f90:141-175)
    ! against the product's own identity <seed|seed> = <gs| n_{1,up} |gs> = occupation summed over the ranks' shards
    call test_mpi_eigh(rank, world, m, x, lo, hi, nfail)
    call gpu_comm_destroy()
    call gpu_delete_sector()
  end subroutine test_mpi_rank

  subroutine test_mpi_eigh(rank, world, m, x, lo, hi, nfail)
    integer, intent(in) :: rank, world, lo, hi
    type(edigpu_model_t), intent(in) :: m
    real(8), intent(in) :: x(:)
    integer, intent(inout) :: nfail
    real(8) :: ev1(2), ev2(2), e, ov
    real(8), allocatable :: b1(:,:), b2(:,:), seed(:), seedref(:), full(:)
    type(c_ptr) :: hsrc, hdst, dsrc, ddst
    integer :: n, nloc, first2, count2, lo2, hi2, n2
    n = size(x); nloc = max(0, hi - lo + 1)
    allocate(b1(n, 2), b2(max(1, nloc), 2))
    call gpu_sp_eigh_d(ev1, b1, 12, 300, 1d-12)
    call gpu_sp_eigh_mpi_d(ev2, b2(1:nloc, :), 12, 300, 1d-12)
    e = maxval(abs(ev1 - ev2))
    ! the ground-state vectors agree up to a sign (the sector's ground state is not degenerate)
    if (nloc > 0) then
       ov = dot_product(b1(lo:hi, 1), b2(1:nloc, 1))
       e = max(e, min(maxval(abs(b1(lo:hi, 1) - b2(1:nloc, 1))), maxval(abs(b1(lo:hi, 1) + b2(1:nloc, 1)))))
    end if
    write(*,"(A,I2,A,ES10.2)") "rank", rank, ": eigenpairs on shards vs one GPU err =", e
    if (e > 1d-9) nfail = nfail + 1
    ! seed: c_{1,up} gs, sector (2,2) -> (1,2): DimUp 4, DimDw 6 (the same down rows: the shard plan is the same)
    hsrc = c_null_ptr
    call gpu_sector_swap(hsrc)                      ! hsrc = the (2,2) sector, no sector live
    call gpu_build_normal(m, 1, 2, 0, -1)
    hdst = gpu_sector_handle()
    n2 = gpu_sector_dim(hdst)
    call gpu_shard_plan(6, rank, world, first2, count2)
    lo2 = first2*4 + 1; hi2 = (first2 + count2)*4
    allocate(seed(max(1, hi2 - lo2 + 1)), seedref(n2))
    call gpu_apply_op_mpi_d(hsrc, hdst, b2(1:nloc, 1), seed(1:max(0, hi2 - lo2 + 1)), 1, 1, .false.)
    ! reference: the whole vector through the single-GPU entry point
    dsrc = gpu_vec_alloc(n); ddst = gpu_vec_alloc(n2)
    call gpu_vec_upload_d(dsrc, b1(:, 1))
    call gpu_apply_op(hsrc, hdst, dsrc, ddst, 1, 1, .false.)
    call gpu_vec_download_d(seedref, ddst)
    call gpu_vec_free(dsrc); call gpu_vec_free(ddst)
    e = 0d0
    if (hi2 >= lo2) e = min(maxval(abs(seed(1:hi2-lo2+1) - seedref(lo2:hi2))), maxval(abs(seed(1:hi2-lo2+1) + seedref(lo2:hi2))))
    write(*,"(A,I2,A,ES10.2)") "rank", rank, ": c_1up |gs> on shards vs one GPU err =", e
    if (e > 1d-9) nfail = nfail + 1
    call gpu_delete_sector()
    call gpu_sector_swap(hsrc)
  end subroutine test_mpi_eigh

end program test_shim
