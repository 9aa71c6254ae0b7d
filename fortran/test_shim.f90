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

  nfail = 0
  call gpu_init(0)

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

end program test_shim
