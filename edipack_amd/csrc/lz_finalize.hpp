// lz_finalize.hpp -- alpha and beta of one Lanczos step from the per-workgroup partial sums, as device code shared by
// the stand-alone finalize kernel (kernels_lanczos.hip: k_finalize_ab) and the sweeps that do it themselves in their
// last workgroup (kernels_panel.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace edigpu {

// Called by ALL threads of one workgroup of blockDim.x = NT threads (a power of two <= 1024); sh: 3 * NT doubles of LDS.
// partial = [np] <P|Q> | [np] sum (Q - sg P)^2 | [np] <P|P> with sg = scal[SC_ALPHA] (the previous alpha).
// beta^2 = |Q - alpha P|^2 = qq - 2 d (alpha - sg vv) + d^2 vv, d = alpha - sg -- exact for any sg, so a spectrum far
// from zero does not cancel; when the difference still loses more than ~3 digits (near-invariant subspace, rare) the
// workgroup recomputes |Q - alpha P|^2 by sweeping the two vectors.  iter < 0: the step index is scal[SC_NDONE].
template <int NT>
__device__ inline void lz_finalize_ab_device(const double* __restrict__ partial, int np, const double* __restrict__ P,
                                             const double* __restrict__ Q, int64_t n, double* __restrict__ scal, int iter,
                                             int nlanc, double* sh) {
  double* sa = sh;
  double* sq = sh + NT;
  double* sn = sh + 2 * NT;
  if (iter < 0) iter = (int)scal[SC_NDONE];
  double a = 0.0, q = 0.0, nn = 0.0;
  for (int i = threadIdx.x; i < np; i += NT) {
    a += partial[i];
    q += partial[np + i];
    nn += partial[2 * np + i];
  }
  sa[threadIdx.x] = a;
  sq[threadIdx.x] = q;
  sn[threadIdx.x] = nn;
  __syncthreads();
  for (int off = NT / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sa[threadIdx.x] += sa[threadIdx.x + off];
      sq[threadIdx.x] += sq[threadIdx.x + off];
      sn[threadIdx.x] += sn[threadIdx.x + off];
    }
    __syncthreads();
  }
  const double alpha = sa[0], qq = sq[0], vv = sn[0], sg = scal[SC_ALPHA];
  const double d = alpha - sg;
  double b2 = qq - 2.0 * d * (alpha - sg * vv) + d * d * vv;
  const bool exact = b2 < 1e-3 * qq;  // the same value in every thread
  __syncthreads();
  if (exact) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += NT) {
      const double w = Q[i] - alpha * P[i];
      s += w * w;
    }
    sa[threadIdx.x] = s;
    __syncthreads();
    for (int off = NT / 2; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sa[threadIdx.x] += sa[threadIdx.x + off];
      __syncthreads();
    }
    b2 = sa[0];
  }
  if (threadIdx.x == 0) {
    scal[SC_ALPHA] = alpha;
    scal[SC_AB + iter] = alpha;
    scal[SC_NDONE] = (double)(iter + 1);
    scal[SC_EXACT] = 0.0;
    const double b = sqrt(b2 > 0.0 ? b2 : 0.0);
    scal[SC_BETA] = b;
    // breakdown: |beta| below the threshold, or an exact zero / NaN whatever the threshold is
    if (!(fabs(b) > 0.0) || fabs(b) < scal[SC_THR])
      scal[SC_STOP] = 1.0;
    else if (iter + 1 < nlanc)
      scal[SC_AB + nlanc + iter + 1] = b;
  }
}

// Last-workgroup epilogue of a sweep that has just written its three partials (thread 0 of every workgroup): the
// workgroup that arrives last at the counter finalizes the step, so that no separate kernel has to be launched.  All
// threads of the workgroup call this; returns after the finalize (if this workgroup did it).  counter must be zero at
// launch; the finalizing workgroup resets it.  sh: 3 * NT doubles + one int, free for use at this point.
template <int NT>
__device__ inline void lz_finalize_if_last(unsigned int* counter, const double* partial, const double* P, const double* Q,
                                           int64_t n, double* scal, int nlanc, double* sh) {
  int* flag = reinterpret_cast<int*>(sh + 3 * NT);
  if (scal[SC_STOP] != 0.0) return;  // recurrence already terminated (uniform; nobody counts)
  __syncthreads();  // this workgroup's partials are written (thread 0) and sh is free
  if (threadIdx.x == 0) {
    __threadfence();  // release: the partials are visible device-wide before the arrival is counted
    const unsigned int prev = atomicAdd(counter, 1u);
    *flag = prev == gridDim.x - 1u;
    if (*flag) __threadfence();  // acquire: the other workgroups' partials
  }
  __syncthreads();
  if (!*flag) return;
  __threadfence();
  lz_finalize_ab_device<NT>(partial, (int)gridDim.x, P, Q, n, scal, -1, nlanc, sh);
  if (threadIdx.x == 0) *counter = 0u;
}

}  // namespace edigpu
