// kernels_trl.hip -- multi-vector kernels of the thick-restart Lanczos eigensolver (edigpu_lanczos_eigh_multi).
//
// The reference's default spectrum path is ARPACK through SciFortran's sp_eigh (LANC_METHOD=arpack,
// ED_NORMAL/ED_DIAG_NORMAL.f90:179-196, ncv = lanc_ncv_factor*Neigen + lanc_ncv_add): an ncv-dimensional
// Krylov basis, full re-orthogonalisation, restarts that keep the wanted Ritz vectors.  The basis lives on
// the device as ncv+1 contiguous vectors; what ARPACK does with BLAS-2 on the host (classical Gram-Schmidt
// against the basis, basis rotation at a restart) are the three streaming kernels below, NC basis vectors
// per pass so that w is read once per NC instead of once per vector.
// Complex sectors use the complex inner product (the real view of a Hermitian matrix has every eigenvalue
// twice: v and i*v).
#include "kernels.hpp"

namespace edigpu {

constexpr int kTrlNT = 256;

// two consecutive doubles at 8-byte alignment (the compiler emits one 16-byte load / store for it)
struct alignas(8) trl_d2 {
  double x, y;
};
constexpr int kTrlNC = 16;  // basis vectors per sweep launch (w is read once per group; 8 per launch measured 4 % slower)

// partial[(c*2+q) * gridDim + block] = sum over the block's elements of conj(Q_c) * w  (q: re, im)
template <bool CPLX>
__global__ void __launch_bounds__(kTrlNT)
    trl_mdot_kernel(int64_t n, int nc, const double* __restrict__ Q, int64_t ldq, const double* __restrict__ w,
                    double* __restrict__ partial, const int* __restrict__ skip) {
  __shared__ double red[kTrlNT / 64][2 * kTrlNC];
  if (skip && *skip) return;  // second Gram-Schmidt pass not needed (trl_decide_kernel)
  double sr[kTrlNC], si[kTrlNC];
#pragma unroll
  for (int c = 0; c < kTrlNC; c++) sr[c] = si[c] = 0.0;
  const int64_t nit = CPLX ? n : (n + 1) / 2;  // complex: one element per step; real: a pair
  for (int64_t i = (int64_t)blockIdx.x * kTrlNT + threadIdx.x; i < nit; i += (int64_t)gridDim.x * kTrlNT) {
    if (CPLX) {
      const double2 x = reinterpret_cast<const double2*>(w)[i];
#pragma unroll
      for (int c = 0; c < kTrlNC; c++)
        if (c < nc) {
          const double2 q = reinterpret_cast<const double2*>(Q + c * ldq)[i];
          sr[c] += q.x * x.x + q.y * x.y;
          si[c] += q.x * x.y - q.y * x.x;
        }
    } else if (2 * i + 1 < n) {
      // real vectors: the loop runs over pairs of elements (16-byte accesses), see the loop bound below
      const trl_d2 x = reinterpret_cast<const trl_d2*>(w)[i];
#pragma unroll
      for (int c = 0; c < kTrlNC; c++)
        if (c < nc) {
          const trl_d2 q = *reinterpret_cast<const trl_d2*>(Q + c * ldq + 2 * i);
          sr[c] += q.x * x.x + q.y * x.y;
        }
    } else {  // odd tail element
      const double x = w[2 * i];
#pragma unroll
      for (int c = 0; c < kTrlNC; c++)
        if (c < nc) sr[c] += Q[c * ldq + 2 * i] * x;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < kTrlNC; c++) {
    double a = sr[c], b = si[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off, 64);
      b += __shfl_down(b, off, 64);
    }
    if (lane == 0) {
      red[wv][2 * c] = a;
      red[wv][2 * c + 1] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * kTrlNC) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kTrlNT / 64; k++) t += red[k][threadIdx.x];
    partial[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = t;
  }
}

// h[2*c+q] = sum_b partial[(2*c+q) * nb + b]
__global__ void __launch_bounds__(256)
    trl_msum_kernel(const double* __restrict__ partial, int nb, double* __restrict__ h, const int* __restrict__ skip) {
  __shared__ double sh[256];
  if (skip && *skip) return;
  const int k = blockIdx.x;  // 0 .. 2*NC-1
  double s = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) s += partial[(int64_t)k * nb + b];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) h[k] = sh[0];
}

// w -= sum_c h_c * Q_c
template <bool CPLX>
__global__ void __launch_bounds__(kTrlNT)
    trl_maxpy_kernel(int64_t n, int nc, const double* __restrict__ Q, int64_t ldq, const double* __restrict__ h,
                     double* __restrict__ w, const int* __restrict__ skip) {
  double hr[kTrlNC], hi[kTrlNC];
  if (skip && *skip) return;
  bool any = false;
#pragma unroll
  for (int c = 0; c < kTrlNC; c++) {
    hr[c] = c < nc ? h[2 * c] : 0.0;
    hi[c] = c < nc ? h[2 * c + 1] : 0.0;
    any = any || hr[c] != 0.0 || hi[c] != 0.0;
  }
  if (!any) return;  // every coefficient of this group was filtered out (trl_decide_kernel): w stays as it is
  const int64_t nit = CPLX ? n : (n + 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * kTrlNT + threadIdx.x; i < nit; i += (int64_t)gridDim.x * kTrlNT) {
    if (CPLX) {
      double2 x = reinterpret_cast<double2*>(w)[i];
#pragma unroll
      for (int c = 0; c < kTrlNC; c++)
        if (c < nc && (hr[c] != 0.0 || hi[c] != 0.0)) {  // uniform: a filtered vector is not even read
          const double2 q = reinterpret_cast<const double2*>(Q + c * ldq)[i];
          x.x -= hr[c] * q.x - hi[c] * q.y;
          x.y -= hr[c] * q.y + hi[c] * q.x;
        }
      reinterpret_cast<double2*>(w)[i] = x;
    } else if (2 * i + 1 < n) {
      trl_d2 x = reinterpret_cast<trl_d2*>(w)[i];
#pragma unroll
      for (int c = 0; c < kTrlNC; c++)
        if (c < nc && hr[c] != 0.0) {
          const trl_d2 q = *reinterpret_cast<const trl_d2*>(Q + c * ldq + 2 * i);
          x.x -= hr[c] * q.x;
          x.y -= hr[c] * q.y;
        }
      reinterpret_cast<trl_d2*>(w)[i] = x;
    } else {
      double x = w[2 * i];
#pragma unroll
      for (int c = 0; c < kTrlNC; c++)
        if (c < nc && hr[c] != 0.0) x -= hr[c] * Q[c * ldq + 2 * i];
      w[2 * i] = x;
    }
  }
}

// out_c = sum_j Y[j][c] * Q_j  for c < nc (Y real, row-major m x ldy): basis rotation at a restart
__global__ void __launch_bounds__(kTrlNT)
    trl_rotate_kernel(int64_t len, int m, int nc, const double* __restrict__ Q, int64_t ldq,
                      const double* __restrict__ Y, int ldy, int c0, double* __restrict__ out, int64_t ldo) {
  // pairs of doubles per step (16-byte accesses), a single element at an odd tail
  const int64_t nit = (len + 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * kTrlNT + threadIdx.x; i < nit; i += (int64_t)gridDim.x * kTrlNT) {
    const bool pair = 2 * i + 1 < len;
    double ax[kTrlNC], ay[kTrlNC];
#pragma unroll
    for (int c = 0; c < kTrlNC; c++) ax[c] = ay[c] = 0.0;
    for (int j = 0; j < m; j++) {
      trl_d2 q;
      if (pair) {
        q = *reinterpret_cast<const trl_d2*>(Q + j * ldq + 2 * i);
      } else {
        q.x = Q[j * ldq + 2 * i];
        q.y = 0.0;
      }
#pragma unroll
      for (int c = 0; c < kTrlNC; c++)
        if (c < nc) {
          const double y = Y[j * ldy + c0 + c];
          ax[c] = fma(y, q.x, ax[c]);
          ay[c] = fma(y, q.y, ay[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < kTrlNC; c++)
      if (c < nc) {
        if (pair) {
          trl_d2 r;
          r.x = ax[c];
          r.y = ay[c];
          *reinterpret_cast<trl_d2*>(out + c * ldo + 2 * i) = r;
        } else {
          out[c * ldo + 2 * i] = ax[c];
        }
      }
  }
}

__global__ void __launch_bounds__(kTrlNT) trl_scale_kernel(int64_t len, double* __restrict__ v, double f) {
  const int64_t nit = (len + 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * kTrlNT + threadIdx.x; i < nit; i += (int64_t)gridDim.x * kTrlNT) {
    if (2 * i + 1 < len) {
      trl_d2 x = reinterpret_cast<trl_d2*>(v)[i];
      x.x *= f;
      x.y *= f;
      reinterpret_cast<trl_d2*>(v)[i] = x;
    } else {
      v[2 * i] *= f;
    }
  }
}

static inline int trl_grid(int64_t n) {
  int64_t nb = (n + kTrlNT - 1) / kTrlNT;
  return (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}

// h (device, 2*nvec doubles: re, im) = Q[0..nvec)^H w ; then w -= Q h.  n = complex or real element count.
// Decisions after the first classical Gram-Schmidt sweep of a Lanczos step.  h: nvec coefficient pairs followed
// by <w_old|w_old> (the sweep dots w with itself as one more column).
//  * "twice is enough" (Kahan / Parlett): a second pass is needed only when the first removed most of the vector,
//    |w_new|^2 = |w_old|^2 - |h|^2 < eta^2 |w_old|^2  ->  skip = 0.
//  * in exact arithmetic only the last two coefficients are non-zero (three-term recurrence; after a restart also
//    those of the kept Ritz vectors).  The others measure the loss of orthogonality; while they are below
//    thr * |w_new| they are not subtracted -- hf gets exact zeros there and the subtraction kernel does not read
//    those basis vectors -- which keeps the basis orthogonal to thr (1e-11 by default, far below the 1e-8 of
//    semi-orthogonality schemes) at about half the traffic.  When a second pass is needed nothing is filtered.
__global__ void trl_decide_kernel(const double* __restrict__ h, int nvec, double eta2, double thr2,
                                  double* __restrict__ hf, int* __restrict__ skip) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double hh = 0.0;
  for (int c = 0; c < nvec; c++) hh += h[2 * c] * h[2 * c] + h[2 * c + 1] * h[2 * c + 1];
  const double ww = h[2 * nvec];
  const bool enough = ww - hh >= eta2 * ww;
  *skip = enough ? 1 : 0;
  const double cut = enough ? thr2 * (ww - hh) : -1.0;
  for (int c = 0; c < nvec; c++) {
    const double a = h[2 * c], b = h[2 * c + 1];
    const bool keep = c >= nvec - 2 || a * a + b * b > cut;
    hf[2 * c] = keep ? a : 0.0;
    hf[2 * c + 1] = keep ? b : 0.0;
  }
}

// Selective second pass: h = nvec coefficient pairs followed by <w|w> (w counted as one more column of the sweep).
// Coefficients below thr * |w| are set to exactly zero IN PLACE (the subtraction kernel then does not read those
// basis vectors; when all of a group are zero it does not touch w either); the <w|w> slot is cleared.
__global__ void trl_filter_kernel(double* __restrict__ h, int nvec, double thr2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double cut = thr2 * h[2 * nvec];
  for (int c = 0; c < nvec; c++) {
    const double a = h[2 * c], b = h[2 * c + 1];
    if (!(a * a + b * b > cut)) h[2 * c] = h[2 * c + 1] = 0.0;
  }
  h[2 * nvec] = h[2 * nvec + 1] = 0.0;
}

int trl_filter(double* h_dev, int nvec, double thr2, hipStream_t st) {
  hipLaunchKernelGGL(trl_filter_kernel, dim3(1), dim3(64), 0, st, h_dev, nvec, thr2);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// h_dev[2c..2c+1] = <Q_c|w> for c < ndot (Q_c = Q + c*ldq; the caller may count w itself as the last column)
int trl_dots(int cplx, int64_t n, int ndot, const double* Q, int64_t ldq, const double* w, double* h_dev,
             double* partial, hipStream_t st, const int* skip) {
  const int nb = trl_grid(n);
  for (int c0 = 0; c0 < ndot; c0 += kTrlNC) {
    const int nc = ndot - c0 < kTrlNC ? ndot - c0 : kTrlNC;
    const double* q0 = Q + (int64_t)c0 * ldq;
    if (cplx)
      hipLaunchKernelGGL((trl_mdot_kernel<true>), dim3(nb), dim3(kTrlNT), 0, st, n, nc, q0, ldq, w, partial, skip);
    else
      hipLaunchKernelGGL((trl_mdot_kernel<false>), dim3(nb), dim3(kTrlNT), 0, st, n, nc, q0, ldq, w, partial, skip);
    hipLaunchKernelGGL(trl_msum_kernel, dim3(2 * kTrlNC), dim3(256), 0, st, partial, nb, h_dev + 2 * c0, skip);
  }
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// w -= sum_c h_c Q_c for c < nvec; basis vectors with an exactly zero coefficient are not read
int trl_subtract(int cplx, int64_t n, int nvec, const double* Q, int64_t ldq, const double* h_dev, double* w,
                 hipStream_t st, const int* skip) {
  const int nb = trl_grid(n);
  for (int c0 = 0; c0 < nvec; c0 += kTrlNC) {
    const int nc = nvec - c0 < kTrlNC ? nvec - c0 : kTrlNC;
    const double* q0 = Q + (int64_t)c0 * ldq;
    if (cplx)
      hipLaunchKernelGGL((trl_maxpy_kernel<true>), dim3(nb), dim3(kTrlNT), 0, st, n, nc, q0, ldq, h_dev + 2 * c0, w,
                         skip);
    else
      hipLaunchKernelGGL((trl_maxpy_kernel<false>), dim3(nb), dim3(kTrlNT), 0, st, n, nc, q0, ldq, h_dev + 2 * c0, w,
                         skip);
  }
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// w -= Q (Q^H w) over nvec basis vectors; h_dev receives the coefficients; skip: device flag, set = no-op
int trl_orthogonalize(int cplx, int64_t n, int nvec, const double* Q, int64_t ldq, double* w, double* h_dev,
                      double* partial, hipStream_t st, const int* skip) {
  if (trl_dots(cplx, n, nvec, Q, ldq, w, h_dev, partial, st, skip)) return 1;
  return trl_subtract(cplx, n, nvec, Q, ldq, h_dev, w, st, skip);
}

int trl_decide(const double* h_dev, int nvec, double eta2, double thr2, double* hf_dev, int* skip, hipStream_t st) {
  hipLaunchKernelGGL(trl_decide_kernel, dim3(1), dim3(64), 0, st, h_dev, nvec, eta2, thr2, hf_dev, skip);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// h_dev[0] = <w|w>
int trl_norm2(int cplx, int64_t n, const double* w, double* h_dev, double* partial, hipStream_t st) {
  const int nb = trl_grid(n);
  if (cplx)
    hipLaunchKernelGGL((trl_mdot_kernel<true>), dim3(nb), dim3(kTrlNT), 0, st, n, 1, w, (int64_t)0, w, partial, nullptr);
  else
    hipLaunchKernelGGL((trl_mdot_kernel<false>), dim3(nb), dim3(kTrlNT), 0, st, n, 1, w, (int64_t)0, w, partial,
                       nullptr);
  hipLaunchKernelGGL(trl_msum_kernel, dim3(2 * kTrlNC), dim3(256), 0, st, partial, nb, h_dev, nullptr);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int trl_partial_doubles(void) { return 2 * kTrlNC * 1024; }

// out[0..k) = Q[0..m) * Y[:, 0..k)   (len = doubles per vector)
int trl_rotate_basis(int64_t len, int m, int k, const double* Q, int64_t ldq, const double* Y_dev, int ldy,
                     double* out, int64_t ldo, hipStream_t st) {
  const int nb = trl_grid(len);
  for (int c0 = 0; c0 < k; c0 += kTrlNC) {
    const int nc = k - c0 < kTrlNC ? k - c0 : kTrlNC;
    hipLaunchKernelGGL(trl_rotate_kernel, dim3(nb), dim3(kTrlNT), 0, st, len, m, nc, Q, ldq, Y_dev, ldy, c0,
                       out + (int64_t)c0 * ldo, ldo);
  }
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int trl_scale(int64_t len, double* v, double f, hipStream_t st) {
  hipLaunchKernelGGL(trl_scale_kernel, dim3(trl_grid(len)), dim3(kTrlNT), 0, st, len, v, f);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

}  // namespace edigpu
