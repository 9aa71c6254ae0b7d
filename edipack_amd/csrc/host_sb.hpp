// host_sb.hpp -- "local-block" tables of a normal-mode sector (host side of kernels_sb.hip, round 4).
//
// Same vector layout, same two-kernel product and the same chunk idea as the impurity-block image (host_ib.hpp), but a
// block now holds the states that share the word w of the WALKED bath levels only: the nb0 lowest bath levels join the
// impurity levels as "local" levels (nloc = norb + nb0), a block has C(nloc, n) states (up to 10 for nloc = 5) and the
// hops among the local levels are compile-time register arithmetic inside the block (sb_core.hpp).  One walk step --
// partner look-up, sign, amplitudes -- then serves up to 10 states instead of 1-3, which is what the round-3 counters
// asked for (134 / 100 vector instructions per element for ~21 multiply-adds).
//
// build_sb derives its tables from the impurity-block image of the same sector (HostIb: column positions, padded panels,
// diagonal tables, Hnd column table) and the one-body data of the sector builder (HostNormal).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "host_ib.hpp"

namespace edigpu {

constexpr int kSbMaxLoc = 6;   // local levels (sb_core.hpp kMaxLoc)
constexpr int kSbNdStride = 32;

struct SbSide {  // one spin species
  int ns = 0, npart = 0, norb = 0, nloc = 0, nbw = 0;
  int64_t dim = 0;
  std::vector<uint16_t> first;  // [2^nbw] first state of block w, kIbNone: no such block in this sector
  std::vector<double> vtab;     // [nbw][4]: [k][a] amplitude impurity level a <-> walked level k; [k][3] = its energy
  std::vector<double> tloc;     // [nloc][nloc] hops among the local levels (symmetric, zero diagonal)
  std::vector<uint32_t> korb;   // [norb] walked levels that couple to impurity level a
  bool single = false;          // every walked level couples to at most one impurity level
};

// Rows longer than the LDS (the impurity-block image has nhalf = 2: the first block with the top bath bit set starts a
// panel): the rows kernel stages ONE value of the top walked bit at a time.  Inside a half the walk runs over the nbw - 1
// lower levels and the LOW word; the hop over the top level reads the partner block's words (same low word, other half)
// from the vector itself.
struct SbUpHalf {
  int panel0 = 0, npanels = 0;   // this half's panels of the padded row
  std::vector<uint16_t> ublist;  // as HostSb::ublist, with the low word (top bit cleared)
  std::vector<int32_t> uslot;
  // per ublist entry: position (whole padded row) of the first column of the block with the top bit toggled, kIbNone
  // when the sector has no such block; and where that block's columns are NOT consecutive (a panel edge between two of
  // its impurity blocks): ugap = (first column after the gap) | (positions skipped << 4), 0x0F = none
  std::vector<uint16_t> utop;
  std::vector<uint8_t> ugap;
  std::vector<uint16_t> rmap;    // [npanels * 16]: image word of a position counted from panel0 * 16
  std::vector<double> ebw;       // [2^(nbw-1)] one-body energy of the low word (+ the top level's when this half has it)
};

struct HostSb {
  bool valid = false;
  std::string why;
  int norb = 0, nb0 = 0, nloc = 0;
  int amode = 0;  // 1: both species `single` and norb > 1 (sb_core.hpp AMODE)
  SbSide up, dw;
  // ---- rows kernel ----
  // A wave-slot = 64 consecutive blocks of one class (classes padded to multiples of 64 with copies of their first block
  // marked kIbSkip).  Thread t of wave v holds, for s < rows_nbt, the block ublist[(s * nw + v) * 64 + t % 64]; uslot[s *
  // nw + v] = class | (index of the slot's first block inside its class << 8), -1: no slot.  The slots are dealt to the
  // waves longest first (a class-5-of-10 slot costs ten times a class-0 slot).
  int rows_nt = 0, rows_nbt = 0;
  std::vector<uint16_t> ublist;
  std::vector<int32_t> uslot;
  // LDS image of a staged row: word j of block i of class n at (wbase(n) + j) * rcs + i (sb_core.hpp RowImage), rcs = 1 +
  // a power of two >= the largest class, the same for every class; urank[w] = i; rmap[position] = image word (padding
  // positions: the zero word rimg_len - 1)
  std::vector<uint16_t> urank, rmap;
  int rcs = 0;
  int rimg_len = 0;
  std::vector<double> e0;   // [2^nb0] one-body energy of the low bath bits of the up species
  std::vector<double> ebw;  // [2^nbw] one-body energy of the walked levels of the up species a word occupies
  // split rows (SbUpHalf): nhalf = 2, the tables above that describe a whole row (ublist, uslot, rmap, ebw) are empty, urank
  // holds the rank of a LOW word among the low words of equal occupation (2^(nbw-1) entries), rcs / rimg_len serve both halves
  int nhalf = 1;
  SbUpHalf half[2];
  // ---- columns kernel ----
  // chunk = rows that share the walked levels >= lowbits.  A wave-slot = cols_gs blocks of one class AND one high word w >>
  // lowbits (padded to multiples of cols_gs per such group: the hops over the levels >= lowbits are then uniform in a slot); wave v of a workgroup takes the slots v, v + cols_nw, ... of a chunk: dslot[chunk_slot[c] + q] =
  // class or -1, blocks dblist[(chunk_slot[c] + q) * cols_gs + g].
  int cols_nw = 0, cols_gs = 8;  // waves per workgroup; blocks per wave-slot (8: a lane holds two columns, 4: one)
  int lowbits = 0, max_chunk_rows = 0, max_chunk_slots = 0;
  std::vector<int32_t> chunk_row, chunk_slot;
  std::vector<int32_t> dslot;
  std::vector<uint16_t> dblist;
  std::vector<uint16_t> dmeta;  // [2^nbw][16] as HostIb::dmeta, on walked words
  // what the columns kernel stages per chunk, packed so that it arrives by ONE contiguous copy: chunk c's bytes
  // [cdesc_off[c], cdesc_off[c + 1]) = the 32-byte dmeta records of its blocks in slot order (cols_gs per slot), their dblist
  // entries, the slot classes (4 bytes per slot); every part 16-byte aligned
  std::vector<uint8_t> cdesc;
  std::vector<int32_t> cdesc_off;
  // ---- Hnd ----
  std::vector<uint32_t> nd_dw;  // [nterms][nloc + 1][kSbNdStride]: partner state inside the block | 0x80 sign, 0xFF none
};

// nb0: low bath levels folded into the blocks; rows_nt / rows_nbt: threads per workgroup and blocks per thread of the
// rows kernel; cols_nw: waves per workgroup of the columns kernel.  out.valid = false with out.why set when the sector
// is not of this form (the caller keeps the impurity-block kernels).
void build_sb(const HostNormal& hn, const HostIb& ib, int nb0, int max_chunk_rows, int rows_nt, int rows_nbt, int cols_nw,
              HostSb& out, int cols_gs = 8);
// bytes of a chunk's packed descriptors for nsl wave-slots of gs blocks
inline uint32_t sb_desc_bytes(int nsl, int gs) {
  return (uint32_t)nsl * gs * 32 + (((uint32_t)nsl * gs * 2 + 15) & ~15u) + (((uint32_t)nsl * 4 + 15) & ~15u);
}

// wave-slots of the rows kernel a sector needs with nb0 low levels folded in (to choose rows_nt / rows_nbt); -1: not of
// the form.  split: rows staged in halves -- the larger of the two halves' counts
int sb_rows_slots(const HostNormal& hn, int nb0, bool split = false);

}  // namespace edigpu
