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
// COHERENT: the partials were written by other workgroups of the SAME launch with device-scope stores; read them with
// device-scope loads (they bypass whatever this XCD's L2 still holds of the previous step's partials)
template <int NT, bool COHERENT = false>
__device__ inline void lz_finalize_ab_device(const double* partial, int np, const double* __restrict__ P,
                                             const double* __restrict__ Q, int64_t n, double* __restrict__ scal, int iter,
                                             int nlanc, double* sh) {
  double* sa = sh;
  double* sq = sh + NT;
  double* sn = sh + 2 * NT;
  if (iter < 0) iter = (int)scal[SC_NDONE];
  double a = 0.0, q = 0.0, nn = 0.0;
  for (int i = threadIdx.x; i < np; i += NT) {
    if (COHERENT) {
      a += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q += __hip_atomic_load(partial + np + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nn += __hip_atomic_load(partial + 2 * np + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      a += partial[i];
      q += partial[np + i];
      nn += partial[2 * np + i];
    }
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
    if (COHERENT) __threadfence();  // (rare) the sweep's result vector, written by the other workgroups of this launch
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

// Last-workgroup epilogue of a sweep (thread 0 of every workgroup holds its three sums): the workgroup that arrives
// last at the counter finalizes the step, so that no separate kernel has to be launched.  All threads of the workgroup
// call this.  No fences: a release fence per workgroup writes back its XCD's whole L2 on this part (measured 2x slower
// than the separate kernel); instead the three partials are written with device-scope stores (write-through), the wave
// waits for them to complete, and only then counts itself with a device-scope atomic; the last workgroup reads the
// partials with device-scope loads.  counter must be zero at launch; the finalizing workgroup resets it.  sh: 3 * NT
// doubles + one int, free for use at this point.
template <int NT>
__device__ inline void lz_finalize_if_last(unsigned int* counter, double* partial, double t, double q, double n_,
                                           const double* P, const double* Q, int64_t n, double* scal, int nlanc, double* sh) {
  int* flag = reinterpret_cast<int*>(sh + 3 * NT);
  if (scal[SC_STOP] != 0.0) return;  // recurrence already terminated (uniform; nobody counts)
  __syncthreads();                   // sh is free
  if (threadIdx.x == 0) {
    __hip_atomic_store(partial + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(partial + gridDim.x + blockIdx.x, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(partial + 2 * gridDim.x + blockIdx.x, n_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);   // the stores have completed before this workgroup counts as arrived
    const unsigned int prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = prev == gridDim.x - 1u;
  }
  __syncthreads();
  if (!*flag) return;
  lz_finalize_ab_device<NT, true>(partial, (int)gridDim.x, P, Q, n, scal, -1, nlanc, sh);
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace edigpu
