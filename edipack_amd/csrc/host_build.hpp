// host_build.hpp -- host-side construction of sector bases and sector Hamiltonians.
//
// Product code (not the oracle).  Functionally it takes the place of the reference's
// build_sector (ED_SECTOR.f90:165-373) and ed_buildh_*_main (ED_NORMAL/
// ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:26-267, ED_SUPERC/..:29-293, ED_NONSU2/..:29-175),
// but it is organised differently: every spin species is described by a one-body
// matrix A(p,q) (coefficient of c^+_p c_q), states are ranked with two lookup tables
// instead of a binary search, fermionic signs come from popcounts of bit masks, and
// CSR rows are emitted directly (no insert-and-search container).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/edigpu.h"

namespace edigpu {

struct HostCsr {
  int64_t nrow = 0, ncol = 0;
  bool is_complex = false;
  std::vector<int64_t> rowptr;
  std::vector<int32_t> col;
  std::vector<double> val;  // nnz or 2*nnz
  int64_t nnz() const { return (int64_t)col.size(); }
};

// Fixed-particle-number basis of `nbits` levels, ascending integer order
// (the order build_sector produces, ED_SECTOR.f90:217-242), with O(1) ranking.
struct CombBasis {
  int nbits = 0, npart = 0, hbits = 0;
  std::vector<int32_t> states;   // index -> bit pattern
  std::vector<int32_t> off_hi;   // [2^(nbits-hbits)]
  std::vector<int32_t> rank_lo;  // [2^hbits]
  void init(int nbits_, int npart_);
  int64_t size() const { return (int64_t)states.size(); }
  inline int32_t rank(uint32_t m) const {
    return off_hi[m >> hbits] + rank_lo[m & ((1u << hbits) - 1u)];
  }
};

// Factored image of the diagonal and of the non-local block (what the library generates when it
// builds the sector itself; nothing in the reference stores it this way):
//   Hd(iup,idw)  = eux[impd(idw)][iup] + ed[idw]
//   Hnd          = sum_t coef[t] * (Pdw_t (x) Pup_t), P = signed partial permutations
// Partner entries: bits 0..30 partner index, bit 31 set = negative sign, 0xFFFFFFFF = no partner.
struct HostFactored {
  bool valid = false;
  int nimp = 0;                 // 2^norb
  std::vector<double> eux;      // nimp * dim_up
  std::vector<double> ed;       // dim_dw
  std::vector<uint8_t> impd;    // dim_dw
  int nterms = 0;
  std::vector<double> coef;     // nterms
  std::vector<uint32_t> jup;    // nterms * dim_up
  std::vector<uint32_t> jdw;    // nterms * dim_dw
};

struct HostNormal {
  HostFactored fac;
  int ns = 0, nup = 0, ndw = 0;
  int64_t dim_up = 0, dim_dw = 0, dw_first = 0, dw_count = 0;
  CombBasis bup, bdw;
  std::vector<double> hd;  // local rows (explicit_arrays only)
  HostCsr up, dw, nd;      // nd over local rows, global columns (explicit_arrays only)
  bool has_nd = false;
  int64_t nd_nnz = 0;      // entries of Hnd on the local rows (always set)
  // what the sector was built from, for the impurity-block image (host_ib.cpp): per species (0 up, 1 down) the
  // ns x ns hop matrix A(p,q) and the level energies; xt[(imp_up << norb) | imp_dw] = inter-spin impurity interaction
  // + constant; samespin[a * norb + b] = (Ust - Jh)(a,b), a < b.  Empty for sectors that did not come from a model.
  int norb = 0;
  std::vector<double> ob_a[2], ob_eps[2];
  std::vector<double> xt, samespin;
};

struct HostFlat {
  int ns = 0;
  int64_t dim = 0, row_first = 0, row_count = 0;
  std::vector<int32_t> states;
  HostCsr h;  // complex, local rows, global columns
};

// On-the-fly ("direct", ed_sparse_H=F) image of a superc / nonsu2 sector: nothing of H is stored, only
// the sector map, two ranking tables and the operator-term list from which every H*v regenerates the
// matrix elements (the reference's directMatVec_*_main, ED_NONSU2/ED_HAMILTONIAN_NONSU2_DIRECT_HxV.f90:22-252).
struct DirectTerm {          // coef * (-1)^{popc(s & sign_mask) + csign} at column rank(s ^ flip),
  uint32_t need_set;         // applicable iff (s & need_set) == need_set && (s & need_clear) == 0
  uint32_t need_clear;
  uint32_t flip;
  uint32_t sign_mask;
  int32_t csign;
  int32_t pair;              // 1: a hop merged with its reverse: applicable iff exactly one of the two
  double cre, cim;           //    levels in `flip` is occupied; (cre,cim) if (s & need_set) != 0,
  double c2re, c2im;         //    (c2re,c2im) otherwise; csign2 in the upper half-word of csign
};

struct HostDirect {
  int ns = 0, norb = 0;
  int64_t dim = 0, row_first = 0, row_count = 0;
  std::vector<int32_t> states;          // local rows
  std::vector<int32_t> off_dw, rk_up;   // rank(s) = off_dw[s >> ns] + rk_up[s & (2^ns - 1)]
  std::vector<DirectTerm> terms;
  std::vector<double> dtab;             // 4 * 256: sum of one-body energies per byte of s
  std::vector<double> xtab;             // 2^(2 norb): interaction energy of the impurity bits (+ constant)
};

int model_ns(const edigpu_model& m);
int64_t binomial(int n, int k);

// ed_total_ud = F ("orbs") image of a normal-mode sector with per-orbital quantum numbers: the vector is the
// tensor [iup_1..iup_Norb, idw_1..idw_Norb] (first index fastest); H = Hd + one small factor per axis
// (reference ed_buildh_normal_orbs, ED_NORMAL/ED_HAMILTONIAN_NORMAL_STORED_HxV.f90:273-496).
struct HostOrbs {
  int naxes = 0;                      // 2 * Norb: up factors then down factors
  int64_t dim = 0;
  std::vector<int64_t> dims;          // [naxes]
  std::vector<HostCsr> fac;           // [naxes] dims[k] x dims[k], real
  std::vector<double> hd;             // explicit diagonal (hand-over / export)
  // factored diagonal (library-built): Hd(i) = sum_k eax[k][idx_k] + xtab[imp bits], imp bit of axis k =
  // impbit[k][idx_k] at bit position k
  bool factored = false;
  std::vector<std::vector<double>> eax;
  std::vector<std::vector<uint8_t>> impbit;
  std::vector<double> xtab;           // 2^naxes
};

// returns "" on success, else an error message
std::string build_orbs(const edigpu_model& m, const int* nups, const int* ndws, HostOrbs& out,
                       bool explicit_diag = false);
// explicit_arrays = false skips the O(Dim) images (hd, the Hnd CSR): the factored tables are all the
// kernels need; the arrays are only materialised for export (edigpu_normal_export).
std::string build_normal(const edigpu_model& m, int nup, int ndw, int64_t dw_first,
                         int64_t dw_count, HostNormal& out, bool explicit_arrays = true);
// jz_basis: the nonsu2 sector (Ntot = sector, twoJz) of Jz_basis=T (ED_SECTOR.f90:289-350)
std::string build_flat(const edigpu_model& m, int sector, int64_t row_first, int64_t row_count,
                       HostFlat& out, bool jz_basis = false, int twojz = 0);
std::string sector_map_jz(const edigpu_model& m, int ntot, int twojz, std::vector<int32_t>& out);
// jz_basis: the nonsu2 sector (Ntot = sector, twoJz) of Jz_basis=T; refuses models whose terms change twoJz
std::string build_direct(const edigpu_model& m, int sector, int64_t row_first, int64_t row_count,
                         HostDirect& out, bool jz_basis = false, int twojz = 0);
// change of twoJz when level p (0 .. 2 Ns - 1, up levels first) is filled: +-1 from the spin, 2 Lzdiag(iorb) from the
// orbital (levels labelled iorb + Norb * ibath; Lzdiag = [-1, +1, 0], ED_VARS_GLOBAL.f90:283)
int twojz_of_level(int p, int ns, int norb);
// Hand-over images: the factored tables recovered from spH0d / spH0nd as the reference built them (local rows of a
// dw-shard, global columns).  true: fac reproduces the diagonal within 8 ulp of max|Hd| and Hnd entry by entry with at
// most max_terms terms; false: not of that form (the caller keeps the explicit image).
bool factor_handover(int64_t dim_up, int64_t dim_dw, int64_t dw_first, int64_t dw_count, const double* hd,
                     const int64_t* nd_rowptr, const int32_t* nd_col, const double* nd_val, int max_terms,
                     HostFactored& fac);
edigpu_model imag_part_model(const edigpu_model& m, bool& any);
// the complex (_CMPLX_NORMAL) sector as one real sector on the doubled up index 2 iup + (re | im); "" on success
std::string build_normal_doubled(const edigpu_model& m, int nup, int ndw, HostNormal& out, int max_terms);
bool eph_offdiagonal(const edigpu_model& m);
edigpu_model eph_operator_model(const edigpu_model& m);
std::string sector_dim(const edigpu_model& m, int q1, int q2, int64_t& dim);
std::string sector_map(const edigpu_model& m, int q1, int q2, int which, std::vector<int32_t>& out);

}  // namespace edigpu
