// host_ib.hpp -- "impurity-block" image of a normal-mode sector (host side).
//
// The basis of one spin species is the ascending list of Ns-bit words with N bits set, impurity levels in the
// lowest Norb bits (ED_SETUP.f90:605-622).  The words that share their BATH part b = word >> Norb are therefore
// adjacent: they form a block of C(Norb, n) states, n = N - popcount(b), one per impurity pattern with n bits set.
// Every off-diagonal one-body term of H_up / H_dw (stored/H_up.f90:59-79: hybridisation c+_a c_k + h.c.; :8-24:
// impurity hops; H_non_local.f90 for the two-body terms) moves one electron between an impurity level and ONE bath
// level k, or between two impurity levels: it couples block b to block b ^ (1 << k) (or to itself) through a small
// matrix on the impurity patterns whose entries are +/- V(a,k) with a sign that factorises into (impurity bits above
// a) x (bath bits below k).  A hop between TWO bath levels (replica / general baths, H_up.f90:26-50) couples block b to
// b ^ (1 << k1 | 1 << k2) element by element (IbSide::pmask).  The kernels of kernels_ib.hip work on whole blocks:
//   rows kernel   a lane owns a block of COLUMNS of one staged row; per bath level one table look-up gives the
//                 partner block, its <= C(Norb, n+-1) adjacent words come from the LDS
//   columns kernel a lane group owns a block of ROWS x 16 columns; the row blocks of a chunk (rows that share their
//                 HIGH bath bits: closed under the hops to the low bath levels) are staged in the LDS, only the hops
//                 to the high bath levels read other chunks through the L2
// This file builds the tables both kernels need from the one-body matrices the sector builder already has.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "host_build.hpp"

namespace edigpu {

constexpr int kIbPanel = 16;     // columns per panel of the vector layout (one 128-byte line per row and panel)
constexpr int kIbMaxNorb = 3;    // impurity levels per block the kernels are instantiated for
constexpr int kIbMaxBath = 14;   // bath levels per species (block tables have 2^nb entries of 16 bits)
constexpr int kIbMaxTerms = 16;  // factored Hnd terms
constexpr int kIbMaxPairs = 64;  // bath-bath hops per species (replica / general baths)

struct IbSide {                  // one spin species
  int ns = 0, npart = 0, norb = 0, nb = 0;
  int64_t dim = 0;
  // first state of block b (index in the species' basis), 0xFFFF = no such block in this sector; [2^nb]
  std::vector<uint16_t> first;
  std::vector<double> vtab;      // [nb][4]: [k][a] amplitude between impurity level a (< norb) and bath level k; [k][3] = its energy
  std::vector<double> timp;      // [norb][norb], symmetric, zero diagonal: impurity-impurity hops
  std::vector<double> ebath;     // [2^nb]: one-body energy of the bath part of a block
  std::vector<double> eimp;      // [2^norb]: one-body + same-spin density-density energy of an impurity pattern
  // bath-bath hops (bath_type replica / general, stored/H_up.f90:26-50: the off-diagonal elements of a replica's matrix
  // between the levels of two orbitals): t (c+_k1 c_k2 + h.c.) couples block b to block b ^ mask -- same class, same
  // impurity patterns -- with the sign of the occupied bath levels strictly between the two.
  // pmask[q] = (1 << k1 | 1 << k2) | (levels strictly between) << 16, pt[q] = t
  std::vector<uint32_t> pmask;
  std::vector<double> pt;
};

// Rows longer than the LDS: the row image is built for ONE value of the top bath bit at a time ("half").  The hops over
// the other nb - 1 levels stay inside a half; the hop over the top level reads the partner block's words from the
// vector itself (kernels_ib.hip, TOP).  A half starts on a panel edge, so its pieces of a row are whole.
struct IbUpHalf {
  int panel0 = 0, npanels = 0;       // this half's panels of the padded row
  std::vector<uint16_t> ublist;      // LOW bath word (without the top bit) | kIbSkip, classes padded to 64 as in HostIb
  int ucls[kIbMaxNorb + 2] = {0};
  std::vector<uint16_t> rmap;        // [npanels * 16]: image word of a position counted from panel0 * 16
  int rcb[5] = {0, 0, 0, 0, 0}, rcs[5] = {0, 0, 0, 0, 0};
  int rimg_len = 0;
  std::vector<uint16_t> utop;        // [ublist.size()]: position (whole padded row) of the first column of the block
                                     // with the top bit toggled, kIbNone when the sector has no such block
};

struct HostIb {
  bool valid = false;
  std::string why;               // when not valid: what the image cannot express
  int norb = 0;
  IbSide up, dw;
  // ---- up side: column positions (blocks never straddle a panel when the sector has Hnd terms) ----
  int npanels = 0;               // panels of kIbPanel columns
  std::vector<int32_t> pos;      // [dim_up]: position of column iup in the padded row
  std::vector<uint16_t> upos;    // [2^nb_up]: position of the first column of block b (0xFFFF none)
  // rows kernel: blocks sorted by class n (ascending bath word inside a class), every class padded to a multiple of
  // 64 entries with copies of its first block marked kIbSkip (computed, never written)
  std::vector<uint16_t> ublist;
  int ucls[kIbMaxNorb + 2] = {0};  // ublist range of class n: [ucls[n], ucls[n+1])
  // rows kernel, LDS image of a staged row (ib_core.hpp RowImage): class by class, word by word.  urank[b] = rank of
  // bath word b inside its class (= its index in the class's part of ublist); class n starts at rcb[n + 1], its word j
  // of block i at rcb[n + 1] + j * rcs[n + 1] + i; rimg_len words in all (slack for over-reads, last word = the zero
  // every padding position maps to).  rmap[position] = word of the image (padding: rimg_len - 1).
  std::vector<uint16_t> urank, rmap;
  int rcb[5] = {0, 0, 0, 0, 0}, rcs[5] = {0, 0, 0, 0, 0};
  int rimg_len = 0;
  // split rows (see IbUpHalf): nhalf = 2, half[h] = the blocks with top bit h; urank_low[w] = rank of the (nb - 1)-bit
  // word w among the words of equal occupation -- the position of a block inside its class in EITHER half (a half holds
  // every low word of an occupation or none)
  int nhalf = 1;
  IbUpHalf half[2];
  std::vector<uint16_t> urank_low;
  // diagonal: Hd(iup, idw) = up.ebath[b_up] + xu[impd(idw)][p_up] + ed[idw]
  std::vector<double> xu;        // [2^norb (impurity pattern of the down word)][2^norb (of the up word)]
  std::vector<double> ed;        // [dim_dw]
  std::vector<uint8_t> impd;     // [dim_dw]
  // ---- down side: row chunks ----
  int lowbits = 0;               // bath levels 0 .. lowbits-1 stay inside a chunk
  std::vector<int32_t> chunk_row;   // [nchunks + 1] first row of a chunk
  std::vector<int32_t> chunk_blk;   // [nchunks + 1] first entry of a chunk in dblist
  // per chunk: blocks sorted by class, classes padded to multiples of 8 entries; entry = bath word | kIbSkip
  std::vector<uint16_t> dblist;
  std::vector<int32_t> dcls;        // [nchunks][kIbMaxNorb + 2] class ranges relative to chunk_blk
  int max_chunk_rows = 0;
  // per bath word of the down species: meta[b][k] = first row of block b ^ (1 << k) (0xFFFF none), k < nb;
  // meta[b][14] = first row of block b itself; meta[b][15] = bit k set when the bath levels below k hold an odd number
  // of electrons; [2^nb][16]
  std::vector<uint16_t> dmeta;
  // ---- factored Hnd in block-relative form ----
  int nterms = 0;
  std::vector<double> ndcoef;       // [nterms]
  // [nterms][norb + 1][4]: for the j-th row of a class-n block: partner row j' | 0x80 sign, 0xFF none
  std::vector<uint8_t> nd_dw;
  // [nterms][npanels * 16]: for a column position: (partner position - position + 8) | 0x80 sign, 0xFF none
  std::vector<uint8_t> nd_up;
};

constexpr uint16_t kIbSkip = 0x8000u;
constexpr uint16_t kIbNone = 0xFFFFu;

// Builds the image of the whole sector held by hn (made by build_normal, which leaves the one-body data in hn).
// max_chunk_rows: rows a chunk of the columns kernel may hold (LDS budget / 128 bytes).  lds_budget > 0 (bytes): a row whose
// image and tables need more is staged in two halves (nhalf = 2); < 0: always (tests); 0: never.  out.valid = false with out.why set when the sector is not of this
// form (the caller keeps the generic kernels).
void build_ib(const HostNormal& hn, int max_chunk_rows, HostIb& out, int lds_budget = 0);

}  // namespace edigpu
