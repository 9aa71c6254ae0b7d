// kernels_lanczos.hip -- device-resident three-term recurrence for gfx950.
//
// Keeps the Lanczos vectors in HBM between H*v products; replaces the host loop
// of SciFortran's sp_lanc_tridiag / sp_lanc_eigh that the reference drives through
// spHtimesV_p (call sites: reference ED_NORMAL/ED_HAMILTONIAN_NORMAL.f90:360-365,
// ED_NORMAL/ED_DIAG_NORMAL.f90:206-214).  alpha/beta stay on the device (scal[]), so
// a whole tridiagonalisation is enqueued without a host synchronisation.
//
// Complex vectors are processed as 2n reals: only Re<a|b> and norms are needed
// because H is Hermitian (alpha, beta real).
//
// Reductions are deterministic: a fixed grid writes one partial per workgroup
// (wave shuffle + LDS), a single-workgroup kernel adds the partials in order.
#include "exchange_index.hpp"
#include "kernels.hpp"
#include "lz_finalize.hpp"

namespace edigpu {

constexpr int kLzNT = 256;

__device__ inline double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

__device__ inline double block_sum(double x) {
  __shared__ double ws[kLzNT / 64];
  x = wave_sum(x);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) ws[w] = x;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kLzNT / 64; i++) t += ws[i];
  }
  __syncthreads();
  return t;  // valid on thread 0
}

// ---- norm of the start vector, then scale (first step of lanczos_iteration) ----
__global__ void __launch_bounds__(kLzNT)
    k_sumsq(const double* __restrict__ v, int64_t n, double* __restrict__ partial) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT)
    s += v[i] * v[i];
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// mode 0: scal[SC_NORM] = sqrt(sum) ; stop if zero
// mode 1: alpha -> scal[SC_ALPHA], scal[SC_AB+iter]
// mode 2: beta = sqrt(sum) -> scal[SC_BETA], scal[SC_AB+nlanc+iter+1] ; stop if < thr
__global__ void __launch_bounds__(1024)
    k_finalize(const double* __restrict__ partial, int np, double* __restrict__ scal, int mode,
               int iter, int nlanc) {
  __shared__ double sh[1024];
  if (mode != 0 && scal[SC_STOP] != 0.0) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 1024) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double tot = sh[0];
    if (mode == 0) {
      const double nrm = sqrt(tot);
      scal[SC_NORM] = nrm;
      scal[SC_STOP] = (nrm == 0.0) ? 1.0 : 0.0;
      scal[SC_NDONE] = 0.0;
    } else if (mode == 1) {
      scal[SC_ALPHA] = tot;
      scal[SC_AB + iter] = tot;
      scal[SC_NDONE] = (double)(iter + 1);
    } else {
      const double b = sqrt(tot);
      scal[SC_BETA] = b;
      // breakdown: |beta| below the threshold, or an exact zero / NaN whatever the threshold is (a zero threshold
      // must not let 1/beta through: seed = eigenvector, diagonal H, Nlanc = Dim)
      if (!(fabs(b) > 0.0) || fabs(b) < scal[SC_THR])
        scal[SC_STOP] = 1.0;
      else if (iter + 1 < nlanc)
        scal[SC_AB + nlanc + iter + 1] = b;
    }
  }
}

__global__ void __launch_bounds__(kLzNT)
    k_scale_by_norm(double* __restrict__ v, int64_t n, const double* __restrict__ scal) {
  if (scal[SC_STOP] != 0.0) return;
  const double inv = 1.0 / scal[SC_NORM];
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT)
    v[i] *= inv;
}

// (vin, vout) <- (vout/beta, -beta*vin)
__global__ void __launch_bounds__(kLzNT)
    k_rotate(double* __restrict__ vin, double* __restrict__ vout, int64_t n,
             const double* __restrict__ scal) {
  if (scal[SC_STOP] != 0.0) return;
  const double b = scal[SC_BETA], ib = 1.0 / b;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double t = vin[i];
    vin[i] = vout[i] * ib;
    vout[i] = -b * t;
  }
}

// vout += tmp ; partial alpha = <vin|vout>
__global__ void __launch_bounds__(kLzNT)
    k_alpha(const double* __restrict__ vin, double* __restrict__ vout,
            const double* __restrict__ tmp, int64_t n, double* __restrict__ partial,
            const double* __restrict__ scal) {
  if (scal[SC_STOP] != 0.0) return;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double w = vout[i] + tmp[i];
    vout[i] = w;
    s += vin[i] * w;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// vout -= alpha*vin ; partial beta^2 = <vout|vout>
__global__ void __launch_bounds__(kLzNT)
    k_beta(const double* __restrict__ vin, double* __restrict__ vout, int64_t n,
           double* __restrict__ partial, const double* __restrict__ scal) {
  if (scal[SC_STOP] != 0.0) return;
  const double a = scal[SC_ALPHA];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double w = vout[i] - a * vin[i];
    vout[i] = w;
    s += w * w;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// acc += coef * vin   (Ritz-vector accumulation in the second pass); skipped once iter >= ndone
__global__ void __launch_bounds__(kLzNT)
    k_axpy(double* __restrict__ acc, const double* __restrict__ vin, int64_t n, double coef,
           const double* __restrict__ scal, int iter) {
  if ((double)iter >= scal[SC_NDONE]) return;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT)
    acc[i] += coef * vin[i];
}

__device__ inline uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void __launch_bounds__(kLzNT)
    k_fill_random(double* __restrict__ v, int64_t n, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const uint64_t h = splitmix(seed ^ (uint64_t)i * 0xD1B54A32D192ED03ull);
    v[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

static inline dim3 red_grid(int64_t n) {
  int64_t nb = (n + kLzNT - 1) / kLzNT;
  if (nb > kRedBlocks) nb = kRedBlocks;
  if (nb < 1) nb = 1;
  return dim3((unsigned)nb);
}

static inline dim3 ew_grid(int64_t n) {
  int64_t nb = (n + kLzNT - 1) / kLzNT;
  if (nb > 256 * 16) nb = 256 * 16;
  if (nb < 1) nb = 1;
  return dim3((unsigned)nb);
}

int lz_norm_begin(double* vin, int64_t n, double* partial, double* scal, hipStream_t st) {
  dim3 g = red_grid(n);
  hipLaunchKernelGGL(k_sumsq, g, dim3(kLzNT), 0, st, vin, n, partial);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, st, partial, (int)g.x, scal, 0, 0, 0);
  hipLaunchKernelGGL(k_scale_by_norm, ew_grid(n), dim3(kLzNT), 0, st, vin, n, scal);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

__global__ void __launch_bounds__(kLzNT)
    k_rotate_lazy(double* __restrict__ P, double* __restrict__ Q, int64_t n, const double* __restrict__ scal) {
  if (scal[SC_STOP] != 0.0) return;
  const double a = scal[SC_ALPHA], b = scal[SC_BETA], ib = 1.0 / b;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double p = P[i];
    P[i] = (Q[i] - a * p) * ib;
    Q[i] = -b * p;
  }
}

int lz_rotate_lazy(double* P, double* Q, int64_t n, const double* scal, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_rotate_lazy, ew_grid(n), dim3(kLzNT), 0, st, P, Q, n, scal);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// X <- (Q - alpha P) / beta: the next Lanczos vector into a third buffer (alpha = 0 unless the axpy is still pending).
// The step of the impurity-block image with rows staged in halves (kernels_ib.hip): what its rows kernel does while
// it stages a whole row has to be its own pass there.
__global__ void __launch_bounds__(kLzNT)
    k_next_vector(const double* __restrict__ P, const double* __restrict__ Q, double* __restrict__ X, int64_t n,
                  const double* __restrict__ scal, int lazy) {
  if (scal[SC_STOP] != 0.0) return;
  const double a = lazy ? scal[SC_ALPHA] : 0.0, ib = 1.0 / scal[SC_BETA];
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT)
    X[i] = (Q[i] - a * P[i]) * ib;
}

int lz_next_vector(const double* P, const double* Q, double* X, int64_t n, const double* scal, bool lazy, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_next_vector, ew_grid(n), dim3(kLzNT), 0, st, P, Q, X, n, scal, lazy ? 1 : 0);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// Q += tmp (the product H P that was written to its own buffer) with the three sums of the one-reduction recurrence:
// <P|Q>, sum (Q - sg P)^2 about sg = the previous alpha, <P|P> (k_finalize_ab).  For the sectors whose product has no
// fused epilogue (ed_total_ud = F, phonon branches, complex normal mode, rows staged in column parts).
__global__ void __launch_bounds__(kLzNT)
    k_add_dot3(const double* __restrict__ P, double* __restrict__ Q, const double* __restrict__ tmp, int64_t n,
               const double* __restrict__ scal, double* __restrict__ partial) {
  __shared__ double red[3 * (kLzNT / 64)];
  double a = 0.0, q = 0.0, nn = 0.0;
  if (scal[SC_STOP] == 0.0) {
    const double sg = scal[SC_ALPHA];
    for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
      const double p = P[i], w = Q[i] + tmp[i];
      Q[i] = w;
      const double d = w - sg * p;
      a += p * w;
      q += d * d;
      nn += p * p;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    q += __shfl_down(q, off, 64);
    nn += __shfl_down(nn, off, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave] = a;
    red[kLzNT / 64 + wave] = q;
    red[2 * (kLzNT / 64) + wave] = nn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tq = 0.0, tn = 0.0;
#pragma unroll
    for (int i = 0; i < kLzNT / 64; i++) {
      ta += red[i];
      tq += red[kLzNT / 64 + i];
      tn += red[2 * (kLzNT / 64) + i];
    }
    partial[blockIdx.x] = ta;
    partial[gridDim.x + blockIdx.x] = tq;
    partial[2 * gridDim.x + blockIdx.x] = tn;
  }
}

int lz_add_dot3(const double* P, double* Q, const double* tmp, int64_t n, const double* scal, double* partial,
                int* np, hipStream_t st) {
  dim3 g = red_grid(n);
  hipLaunchKernelGGL(k_add_dot3, g, dim3(kLzNT), 0, st, P, Q, tmp, n, scal, partial);
  EDIGPU_HIP(hipGetLastError());
  *np = (int)g.x;
  return 0;
}

int lz_rotate(double* vin, double* vout, int64_t n, const double* scal, hipStream_t st) {
  hipLaunchKernelGGL(k_rotate, ew_grid(n), dim3(kLzNT), 0, st, vin, vout, n, scal);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int lz_alpha(const double* vin, double* vout, const double* tmp, int64_t n, double* partial,
             double* scal, int iter, int nlanc, hipStream_t st) {
  dim3 g = red_grid(n);
  hipLaunchKernelGGL(k_alpha, g, dim3(kLzNT), 0, st, vin, vout, tmp, n, partial, scal);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, st, partial, (int)g.x, scal, 1, iter, nlanc);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// alpha and beta from the sweep's partials.  The sweeps accumulate qq = sum (Q - sg v)^2 about sg = the previous
// alpha (still in scal[SC_ALPHA] when they run) and vv = <v|v>, so beta^2 = |Q - alpha v|^2 = qq - 2 d (alpha - sg vv)
// + d^2 vv with d = alpha - sg: a
// spectrum far from zero (large xmu / Hartree shifts: |alpha| >> beta on every step) no longer cancels.  When the difference still loses
// more than ~3 digits to cancellation (near-invariant subspace, rare) the same single workgroup recomputes
// beta^2 = |Q - alpha v|^2 directly by sweeping the two vectors (Q is left untouched: the axpy stays pending).
// One launch per step: the former separate fallback kernel cost ~4.7 us per step while idle.
__global__ void __launch_bounds__(1024)
    k_finalize_ab(const double* __restrict__ partial, int np, const double* __restrict__ P,
                  const double* __restrict__ Q, int64_t n, double* __restrict__ scal, int iter, int nlanc) {
  __shared__ double sh[3 * 1024];
  if (scal[SC_STOP] != 0.0) return;
  lz_finalize_ab_device<1024>(partial, np, P, Q, n, scal, iter, nlanc, sh);
}

int lz_finalize_alpha_beta(const double* P, const double* Q, int64_t n, double* partial, int np,
                           double* scal, int iter, int nlanc, hipStream_t st) {
  hipLaunchKernelGGL(k_finalize_ab, dim3(1), dim3(1024), 0, st, partial, np, P, Q, n, scal, iter, nlanc);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int lz_finalize_alpha(const double* partial, int np, double* scal, int iter, int nlanc, hipStream_t st) {
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, st, partial, np, scal, 1, iter, nlanc);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int lz_beta(const double* vin, double* vout, int64_t n, double* partial, double* scal, int iter,
            int nlanc, hipStream_t st) {
  dim3 g = red_grid(n);
  hipLaunchKernelGGL(k_beta, g, dim3(kLzNT), 0, st, vin, vout, n, partial, scal);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, st, partial, (int)g.x, scal, 2, iter, nlanc);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int lz_axpy_coef(double* acc, const double* vin, int64_t n, double coef, const double* scal,
                 int iter, hipStream_t st) {
  hipLaunchKernelGGL(k_axpy, ew_grid(n), dim3(kLzNT), 0, st, acc, vin, n, coef, scal, iter);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

__global__ void __launch_bounds__(256) k_to_blocked(const double* __restrict__ src, double* __restrict__ dst, int64_t dim_up,
                                                    int64_t dim_dw, int shift, int64_t n) {
  const int64_t ps = dim_dw << shift;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t panel = i / ps, rem = i - panel * ps;
    const int64_t row = rem >> shift, col = (panel << shift) + (rem & (((int64_t)1 << shift) - 1));
    dst[i] = col < dim_up ? src[row * dim_up + col] : 0.0;
  }
}

__global__ void __launch_bounds__(256) k_from_blocked(const double* __restrict__ src, double* __restrict__ dst, int64_t dim_up,
                                                      int64_t dim_dw, int shift, int64_t n) {
  const int64_t ps = dim_dw << shift;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / dim_up, col = i - row * dim_up;
    dst[i] = src[(col >> shift) * ps + (row << shift) + (col & (((int64_t)1 << shift) - 1))];
  }
}

int vec_to_blocked(const double* src, double* dst, int64_t dim_up, int64_t dim_dw, int shift, hipStream_t st) {
  const int64_t n = ((dim_up + ((int64_t)1 << shift) - 1) >> shift) * (dim_dw << shift);
  const unsigned g = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
  hipLaunchKernelGGL(k_to_blocked, dim3(g), dim3(256), 0, st, src, dst, dim_up, dim_dw, shift, n);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_from_blocked(const double* src, double* dst, int64_t dim_up, int64_t dim_dw, int shift, hipStream_t st) {
  const int64_t n = dim_up * dim_dw;
  const unsigned g = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
  hipLaunchKernelGGL(k_from_blocked, dim3(g), dim3(256), 0, st, src, dst, dim_up, dim_dw, shift, n);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int lz_fill_random(double* v, int64_t n, uint64_t seed, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_random, ew_grid(n), dim3(kLzNT), 0, st, v, n, seed);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// ---- stand-alone vector kernels with explicit device scalars (sharded N>1 loop) ----
__global__ void __launch_bounds__(kLzNT)
    kv_rotate(int64_t n, double* __restrict__ vin, double* __restrict__ vout, const double* __restrict__ beta2) {
  const double b = sqrt(beta2[0]), ib = 1.0 / b;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double t = vin[i];
    vin[i] = vout[i] * ib;
    vout[i] = -b * t;
  }
}

__global__ void __launch_bounds__(kLzNT)
    kv_add_dot(int64_t n, const double* __restrict__ vin, double* __restrict__ vout,
               const double* __restrict__ tmp, double* __restrict__ partial) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double w = vout[i] + tmp[i];
    vout[i] = w;
    s += vin[i] * w;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kLzNT)
    kv_axpy_nrm2(int64_t n, const double* __restrict__ vin, double* __restrict__ vout,
                 const double* __restrict__ alpha, double* __restrict__ partial) {
  const double a = alpha[0];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double w = vout[i] - a * vin[i];
    vout[i] = w;
    s += w * w;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(1024) kv_sum(const double* __restrict__ partial, int np, double* __restrict__ out) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 1024) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

__global__ void __launch_bounds__(kLzNT) kv_scale(int64_t n, double* __restrict__ v, const double* __restrict__ nrm2) {
  const double inv = 1.0 / sqrt(nrm2[0]);
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) v[i] *= inv;
}

// ---- fused vector updates of the transposed-exchange Lanczos step (normal mode, N > 1) ----
// ab = (<v|w>, <w|w>) of the previous step, already summed over the ranks: alpha = ab[0], beta^2 = ab[1] - alpha^2.
// One pass: the pending axpy w - alpha v, the rotate (v, w) <- (w/beta, -beta v), and the new v written straight
// into the all-to-all send buffer (layout of transpose_pack_kernel, kernels_ops.hip; halo columns are written
// into every block that holds them; positions never written stay zero from the allocation).
__global__ void __launch_bounds__(kLzNT)
    kv_rotate_pack(int first, int64_t dim_up, int64_t nrows, int64_t q, int world, int64_t pcol, int halo,
                   double* __restrict__ vin, double* __restrict__ vout, const double* __restrict__ ab,
                   double* __restrict__ send) {
  double a = 0.0, b = 1.0, ib = 1.0;
  if (!first) {
    a = ab[0];
    b = sqrt(ab[1] - a * a);
    ib = 1.0 / b;
  }
  const int64_t n = nrows * dim_up;
  for (int64_t e = (int64_t)blockIdx.x * kLzNT + threadIdx.x; e < n; e += (int64_t)gridDim.x * kLzNT) {
    const int64_t i = e / dim_up, col = e - i * dim_up;
    double x = vin[e];
    if (!first) {
      const double t = x;
      x = (vout[e] - a * t) * ib;
      vin[e] = x;
      vout[e] = -b * t;
    }
    int64_t clo, chi;
    xch_blocks_of(col, pcol, halo, world, clo, chi);
    for (int64_t c = clo; c <= chi; c++) send[xch_send_slot(c, i, col, q, pcol, halo)] = x;
  }
}

// w += tmp + (the down half received from the column shards); partials of <v|w> and <w|w>
__global__ void __launch_bounds__(kLzNT)
    kv_unpack_add_dot2(int64_t dim_up, int64_t nrows, int64_t q, int64_t pcol, int halo,
                       const double* __restrict__ vin, double* __restrict__ vout, const double* __restrict__ tmp,
                       const double* __restrict__ back, double* __restrict__ partial) {
  const int64_t n = nrows * dim_up;
  double s = 0.0, qq = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * kLzNT + threadIdx.x; e < n; e += (int64_t)gridDim.x * kLzNT) {
    const int64_t i = e / dim_up, col = e - i * dim_up;
    const double w = vout[e] + tmp[e] + back[xch_back_slot(i, col, q, pcol, halo)];
    vout[e] = w;
    s += vin[e] * w;
    qq += w * w;
  }
  s = block_sum(s);
  qq = block_sum(qq);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s;
    partial[kRedBlocks / 2 + blockIdx.x] = qq;
  }
}

__global__ void __launch_bounds__(1024) kv_sum2(const double* __restrict__ partial, int np, double* __restrict__ out) {
  __shared__ double sh[2][1024];
  double s = 0.0, t = 0.0;
  for (int i = threadIdx.x; i < np; i += 1024) {
    s += partial[i];
    t += partial[kRedBlocks / 2 + i];
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = t;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sh[0][0];
    out[1] = sh[1][0];
  }
}

// the same fused updates for any row-sharded vector (flat modes, all-gather form); n counts doubles
__global__ void __launch_bounds__(kLzNT)
    kv_rotate_lazy(int64_t n, double* __restrict__ vin, double* __restrict__ vout, const double* __restrict__ ab) {
  const double a = ab[0], b = sqrt(ab[1] - a * a), ib = 1.0 / b;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double t = vin[i];
    vin[i] = (vout[i] - a * t) * ib;
    vout[i] = -b * t;
  }
}

__global__ void __launch_bounds__(kLzNT)
    kv_add_dot2(int64_t n, const double* __restrict__ vin, double* __restrict__ vout, const double* __restrict__ tmp,
                double* __restrict__ partial) {
  double s = 0.0, qq = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kLzNT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLzNT) {
    const double w = vout[i] + tmp[i];
    vout[i] = w;
    s += vin[i] * w;
    qq += w * w;
  }
  s = block_sum(s);
  qq = block_sum(qq);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s;
    partial[kRedBlocks / 2 + blockIdx.x] = qq;
  }
}

int vec_rotate_lazy(int64_t n, double* vin, double* vout, const double* ab, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(kv_rotate_lazy, ew_grid(n), dim3(kLzNT), 0, st, n, vin, vout, ab);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_add_dot2(int64_t n, const double* vin, double* vout, const double* tmp, double* out2, double* work,
                 hipStream_t st) {
  int64_t nb = (n + kLzNT - 1) / kLzNT;
  if (nb > kRedBlocks / 2) nb = kRedBlocks / 2;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(kv_add_dot2, dim3((unsigned)nb), dim3(kLzNT), 0, st, n, vin, vout, tmp, work);
  hipLaunchKernelGGL(kv_sum2, dim3(1), dim3(1024), 0, st, work, (int)nb, out2);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_rotate_pack(int first, int64_t dim_up, int64_t nrows, int64_t q, int world, int64_t pcol, int halo,
                    double* vin, double* vout, const double* ab, double* send, hipStream_t st) {
  if (nrows * dim_up <= 0) return 0;
  hipLaunchKernelGGL(kv_rotate_pack, ew_grid(nrows * dim_up), dim3(kLzNT), 0, st, first, dim_up, nrows, q, world, pcol,
                     halo, vin, vout, ab, send);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_unpack_add_dot2(int64_t dim_up, int64_t nrows, int64_t q, int64_t pcol, int halo, const double* vin,
                        double* vout, const double* tmp, const double* back, double* out2, double* work,
                        hipStream_t st) {
  const int64_t n = nrows * dim_up;
  int64_t nb = (n + kLzNT - 1) / kLzNT;
  if (nb > kRedBlocks / 2) nb = kRedBlocks / 2;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(kv_unpack_add_dot2, dim3((unsigned)nb), dim3(kLzNT), 0, st, dim_up, nrows, q, pcol, halo, vin,
                     vout, tmp, back, work);
  hipLaunchKernelGGL(kv_sum2, dim3(1), dim3(1024), 0, st, work, (int)nb, out2);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_rotate(int64_t n, double* vin, double* vout, const double* beta2, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(kv_rotate, ew_grid(n), dim3(kLzNT), 0, st, n, vin, vout, beta2);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_add_dot(int64_t n, const double* vin, double* vout, const double* tmp, double* out, double* work,
                hipStream_t st) {
  dim3 g = red_grid(n > 0 ? n : 1);
  hipLaunchKernelGGL(kv_add_dot, g, dim3(kLzNT), 0, st, n, vin, vout, tmp, work);
  hipLaunchKernelGGL(kv_sum, dim3(1), dim3(1024), 0, st, work, (int)g.x, out);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_axpy_nrm2(int64_t n, const double* vin, double* vout, const double* alpha, double* out, double* work,
                  hipStream_t st) {
  dim3 g = red_grid(n > 0 ? n : 1);
  hipLaunchKernelGGL(kv_axpy_nrm2, g, dim3(kLzNT), 0, st, n, vin, vout, alpha, work);
  hipLaunchKernelGGL(kv_sum, dim3(1), dim3(1024), 0, st, work, (int)g.x, out);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

int vec_scale(int64_t n, double* v, const double* nrm2, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(kv_scale, ew_grid(n), dim3(kLzNT), 0, st, n, v, nrm2);
  EDIGPU_HIP(hipGetLastError());
  return 0;
}

// ---- streaming ceilings of this device (bench.py reports them next to the kernel's achieved rate) ----
__global__ void __launch_bounds__(512) k_bw_read(const double2* __restrict__ p, int64_t n, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 512) s += p[i].x + p[i].y;
  if (s == 1.2345e300) out[0] = s;
}
__global__ void __launch_bounds__(512) k_bw_copy(const double2* __restrict__ p, double2* __restrict__ q, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 512) q[i] = p[i];
}
__global__ void __launch_bounds__(512) k_bw_triad(const double2* __restrict__ a, const double2* __restrict__ b,
                                                  double2* __restrict__ c, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 512) {
    const double2 x = a[i], y = b[i];
    c[i] = make_double2(x.x + 3.0 * y.x, x.y + 3.0 * y.y);
  }
}

// out[0..2] = read, copy (read + write), triad (2 reads + 1 write) in GB/s on three buffers of `bytes` each
int measure_membw(int64_t bytes, double out[3]) {
  const int64_t n = bytes / 16;
  double2 *a = nullptr, *b = nullptr, *c = nullptr;
  double* sink = nullptr;
  EDIGPU_HIP(hipMalloc((void**)&a, (size_t)n * 16));
  EDIGPU_HIP(hipMalloc((void**)&b, (size_t)n * 16));
  EDIGPU_HIP(hipMalloc((void**)&c, (size_t)n * 16));
  EDIGPU_HIP(hipMalloc((void**)&sink, 8));
  EDIGPU_HIP(hipMemset(a, 0, (size_t)n * 16));
  EDIGPU_HIP(hipMemset(b, 0, (size_t)n * 16));
  hipEvent_t e0, e1;
  EDIGPU_HIP(hipEventCreate(&e0));
  EDIGPU_HIP(hipEventCreate(&e1));
  const dim3 g(2048), blk(512);
  const int reps = 10;
  for (int w = 0; w < 3; w++) {
    for (int r = 0; r < reps + 2; r++) {
      if (r == 2) EDIGPU_HIP(hipEventRecord(e0, 0));
      if (w == 0) hipLaunchKernelGGL(k_bw_read, g, blk, 0, 0, a, n, sink);
      if (w == 1) hipLaunchKernelGGL(k_bw_copy, g, blk, 0, 0, a, c, n);
      if (w == 2) hipLaunchKernelGGL(k_bw_triad, g, blk, 0, 0, a, b, c, n);
    }
    EDIGPU_HIP(hipEventRecord(e1, 0));
    EDIGPU_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    EDIGPU_HIP(hipEventElapsedTime(&ms, e0, e1));
    out[w] = (double)(w + 1) * (double)n * 16.0 * reps / ((double)ms * 1e6);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipFree(c);
  (void)hipFree(sink);
  return 0;
}

}  // namespace edigpu
