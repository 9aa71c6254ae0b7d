// exchange_index.hpp -- where an element of a row shard sits in the buffers of the transposed exchange (the reference's
// vector_transpose_MPI, ED_NORMAL/ED_HAMILTONIAN_NORMAL_COMMON.f90:66-167).
//
// A rank owns q down rows of V[idw][iup].  For the column half of the product every rank needs ALL rows of a block of
// pcol up columns, plus `halo` columns on both sides for Hnd: block (r -> c) = rows of rank r, columns
// [c pcol - halo, (c + 1) pcol + halo).  send[c][i][j] is laid out so that one equal-split all-to-all delivers
// recv[r][i][j] = rows r q + i in order, i.e. the column shard with row stride pw = pcol + 2 halo and no unpacking.
//
// Plain C++ shared by every kernel that packs or unpacks (kernels_ops.hip, kernels_lanczos.hip, edigpu_shard.hip) and by
// the host-only entry points edigpu_exchange_send_map / edigpu_exchange_back_map, through which the CPU suite drives
// the product's own index arithmetic between gloo ranks (tests/test_sharding_gloo.py).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define XCH_HD __host__ __device__ inline
#else
#define XCH_HD inline
#endif

namespace edigpu {

// the column blocks clo .. chi that hold column col (its own block and, inside the halo, the neighbours)
XCH_HD void xch_blocks_of(int64_t col, int64_t pcol, int halo, int world, int64_t& clo, int64_t& chi) {
  clo = col >= halo ? (col - halo) / pcol : 0;
  chi = (col + halo) / pcol;
  if (chi > world - 1) chi = world - 1;
}

// slot of (row i of the shard, column col) in block c of the send buffer
XCH_HD int64_t xch_send_slot(int64_t c, int64_t i, int64_t col, int64_t q, int64_t pcol, int halo) {
  return (c * q + i) * (pcol + 2 * halo) + (col - c * pcol + halo);
}

// what send slot e holds: the element index i * dim_up + col of the shard, or -1 (a zero: rows past the shard's nrows,
// columns outside [0, dim_up))
XCH_HD int64_t xch_send_source(int64_t e, int64_t dim_up, int64_t nrows, int64_t q, int64_t pcol, int halo) {
  const int64_t pw = pcol + 2 * halo;
  const int64_t j = e % pw, i = (e / pw) % q, c = e / (pw * q);
  const int64_t col = c * pcol - halo + j;
  return (i < nrows && col >= 0 && col < dim_up) ? i * dim_up + col : -1;
}

// the way back delivers back[c][i][halo + j] = the column half of H*v for (row i of this rank, column c pcol + j)
XCH_HD int64_t xch_back_slot(int64_t i, int64_t col, int64_t q, int64_t pcol, int halo) {
  const int64_t c = col / pcol, j = col - c * pcol;
  return (c * q + i) * (pcol + 2 * halo) + halo + j;
}

}  // namespace edigpu
