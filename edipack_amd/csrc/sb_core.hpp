// sb_core.hpp -- per-block arithmetic of the LOCAL-block kernels (kernels_sb.hip; host_sb.hpp explains the decomposition).
//
// Round 4.  ib_core.hpp works on blocks of C(Norb, n) states that share their whole bath word and pays one walk step
// (partner look-up, sign, amplitudes) per hop for 1-3 states: 134 / 100 vector instructions per element for ~21 useful
// multiply-adds.  Here the NB0 LOWEST bath levels are folded into the block: a species' state is (w << NLOC) | p with
// NLOC = NIMP + NB0 "local" levels (p = local pattern: impurity bits below, low bath bits above) and w the word of the
// nbw = Ns - NLOC WALKED bath levels.  A block is the C(NLOC, n) adjacent states of one w (n = N - popcount(w)):
//   * hops among the local levels (impurity-impurity, impurity <-> low bath) stay inside the block and are unrolled at
//     compile time on registers: no index work at all;
//   * a hop over a walked level k couples block w to block w ^ (1 << k) through the same compile-time pattern as before
//     (only the NIMP impurity levels take part; the low bath bits sit between impurity and level k and enter the sign),
//     so ONE walk step now serves up to C(5,2) = 10 states.
// Plain C++17 shared by the kernels and by the CPU shim tests/host_sb.cpp.
#pragma once
#include "ib_core.hpp"

// tuning variants of the columns kernel (scripts/ab_build.sh)
#ifndef SB_V_HOPS
#define SB_V_HOPS 2
#endif
#ifndef SB_V_H0
#define SB_V_H0 1
#endif
#ifndef SB_V_TOUCH
#define SB_V_TOUCH 0
#endif

namespace edigpu {
namespace sb {

using ib::binom;
using ib::ctz32;
using ib::flip;
using ib::FmaD;
using ib::FmaP;
using ib::idx;
using ib::Pair;
using ib::pat;
using ib::popc;
using ib::popc32;
using ib::sfor;

constexpr int kMaxLoc = 6;     // local levels per block the kernels may be instantiated for
constexpr int kNdStride = 32;  // entries per (term, class) line of the Hnd row table

// f(std::integral_constant<int, N>) for the class n (0 .. NLOC) given at run time (uniform over a wave)
template <int NLOC, class F>
IB_HD void for_class(int n, F&& f) {
  ib::for_class<NLOC>(n, f);
}

// ---- coupling of one block to one partner block over a walked bath level -------------------------------------------
// acc[j] += sum over the impurity levels a in AMASK the hop can use: s * V(a) * xp[partner of state j]
// DOWN = true : the level is EMPTY in our block -> the partner has it occupied and one LOCAL electron less (class N-1):
//               a runs over the occupied impurity levels of state j
// DOWN = false: the level is OCCUPIED -> partner class N+1, a over the empty impurity levels
// sign: every local level above a lies between a and the walked level -> parity of the local bits above a (compile
// time); the walked bits below the level are the caller's (folded into v).
template <int NLOC, int NIMP, int AMASK, int N, bool DOWN, class T, class FMA>
IB_HD void couple(const double* v /* [NIMP] */, const T* xp, T* acc, FMA&& fma_) {
  constexpr int M = binom(NLOC, N);
  sfor<0, M>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = pat(NLOC, N, j);
    sfor<0, NIMP>([&](auto A) {
      constexpr int a = decltype(A)::value;
      if constexpr (((AMASK >> a) & 1) != 0) {
        constexpr bool occ = ((p >> a) & 1) != 0;
        if constexpr (occ == DOWN) {
          constexpr int p2 = p ^ (1 << a);
          constexpr int j2 = idx(p2);
          constexpr bool neg = (popc((unsigned)(p >> (a + 1))) & 1) != 0;
          fma_(acc[j], neg ? -v[a] : v[a], xp[j2]);
        }
      }
    });
  });
}

// hops among the local levels: acc[j] += +/- t(a1,a2) x[j'], p_j' = p_j with the electron moved.  BB = false: pairs of
// two bath levels are left out at compile time (normal / hybrid baths have no bath-bath hops).
template <int NLOC, int NIMP, bool BB, int N, class T, class FMA, class TP>
IB_HD void couple_loc(TP tloc /* [NLOC][NLOC] */, const T* x, T* acc, FMA&& fma_) {
  constexpr int M = binom(NLOC, N);
  sfor<0, NLOC>([&](auto A1) {
    constexpr int a1 = decltype(A1)::value;
    sfor<a1 + 1, NLOC>([&](auto A2) {
      constexpr int a2 = decltype(A2)::value;
      if constexpr (BB || a1 < NIMP) {
        const double t = tloc[a1 * NLOC + a2];
        if (t != 0.0) {  // uniform
          sfor<0, M>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int p = pat(NLOC, N, j);
            if constexpr ((((p >> a1) ^ (p >> a2)) & 1) != 0) {
              constexpr int p2 = p ^ (1 << a1) ^ (1 << a2);
              constexpr int btw = ((1 << a2) - 1) & ~((1 << (a1 + 1)) - 1);
              constexpr bool neg = (popc((unsigned)(p & btw)) & 1) != 0;
              fma_(acc[j], neg ? -t : t, x[idx(p2)]);
            }
          });
        }
      }
    });
  });
}

// ---- rows kernel: one block of columns of the staged row -----------------------------------------------------------
// The staged row is held class by class, word by word, every class with the same stride cs (a power of two >= the
// largest class): word j of the i-th block of class n sits at (wbase(n) + j) * cs + i, wbase(n) = number of local
// patterns with fewer than n bits.  With cs a compile-time constant every word of a block is ONE address plus an
// immediate offset.  (The hosts pick cs = 2^k + 1.)
constexpr int wbase(int nloc, int n) {
  int s = 0;
  for (int q = 0; q < n; q++) s += binom(nloc, q);
  return s;
}

struct RowImage {
  const double* row;
  const uint16_t* rank;  // [2^nbw] rank of a walked word inside its class
  const double* ebath;   // [2^nbw] one-body energy of the walked levels a word occupies
  int cs;                // class stride (used when the template parameter CS is 0: host)
};

// AMODE 0: every impurity level couples to every walked level (hybrid baths)
// AMODE 1: a walked level couples to ONE impurity level (bath_type normal): korb[a] = the levels of impurity level a; a
//          lane walks its bits orbital by orbital and only the multiply-adds of that orbital are issued
// vtab : [nbw][vs]: amplitudes V(a,k), a < NIMP.  vs = 4 in the host tables; the kernels' LDS copy has vs = 10 (80 bytes:
//        16-byte aligned rows that start in different banks -- a lane reads the row of ITS level, and with vs = 4 the
//        eleven rows share four bank groups)
// tloc : [NLOC][NLOC] hops among the local levels (uniform)
// xuc  : [2^NIMP] diagonal part that depends on the block's impurity pattern (and on the row); e0: [2^NB0] energy of the
//        low bath bits; dconst: the rest of the row's diagonal
// TP / KP: pointers to (uniform) read-only tables; on the device they are constant-address-space pointers, so that the
// loads are scalar (a plain global pointer is read with vector loads once the kernel stores anything)
template <int NIMP, int NB0, int AMODE, int N, int CS, class TP, class KP>
IB_HD void rows_block(const RowImage& im, uint32_t w, uint32_t i, int nbw, const double* vtab, int vs, KP korb,
                      TP tloc, double dconst, TP xuc, TP e0, double* acc) {
  constexpr int NLOC = NIMP + NB0;
  constexpr int M = binom(NLOC, N);
  constexpr int IMPM = (1 << NIMP) - 1;
  constexpr int ALL = IMPM;
  const int cs = CS > 0 ? CS : im.cs;
  const double* row = im.row;
  const double* own = row + wbase(NLOC, N) * cs + i;
  {
    // the whole diagonal first: the block's own words are not needed after the local hops
    const double d = dconst + im.ebath[w];
    double x[M];
    sfor<0, M>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = pat(NLOC, N, j);
      x[j] = own[j * cs];
      acc[j] = ((xuc[p & IMPM] + e0[p >> NIMP]) + d) * x[j];
    });
    if constexpr (NLOC > 1) couple_loc<NLOC, NIMP, false, N, double>(tloc, x, acc, FmaD{});
  }
  const uint32_t wm = (1u << nbw) - 1u;
  auto walk = [&](auto DOWNC, auto AMC, uint32_t m) {
    constexpr bool DOWN = decltype(DOWNC)::value;
    constexpr int AM = decltype(AMC)::value;
    constexpr int NP = DOWN ? N - 1 : N + 1;
    constexpr int MP = binom(NLOC, NP);  // 0: no such class
    if constexpr (MP > 0) {
      const double* pbase = row + wbase(NLOC, NP) * cs;
      while (m) {
        const int k = ctz32(m);
        m &= m - 1u;
        const uint32_t bit = 1u << k;
        const double* pp = pbase + im.rank[w ^ bit];
        const uint32_t neg = (uint32_t)popc32(w & (bit - 1u)) & 1u;
        double xp[MP], v[NIMP];
        sfor<0, MP>([&](auto J) { xp[decltype(J)::value] = pp[decltype(J)::value * cs]; });
        sfor<0, NIMP>([&](auto A) {
          constexpr int a = decltype(A)::value;
          if constexpr (((AM >> a) & 1) != 0) v[a] = flip(vtab[k * vs + a], neg);
          else v[a] = 0.0;
        });
        couple<NLOC, NIMP, AM, N, DOWN, double>(v, xp, acc, FmaD{});
      }
    }
  };
  if constexpr (AMODE == 0) {
    walk(std::true_type{}, std::integral_constant<int, ALL>{}, ~w & wm);
    walk(std::false_type{}, std::integral_constant<int, ALL>{}, w);
  } else {
    sfor<0, NIMP>([&](auto A) {
      constexpr int a = decltype(A)::value;
      walk(std::true_type{}, std::integral_constant<int, (1 << a)>{}, ~w & wm & korb[a]);
    });
    sfor<0, NIMP>([&](auto A) {
      constexpr int a = decltype(A)::value;
      walk(std::false_type{}, std::integral_constant<int, (1 << a)>{}, w & korb[a]);
    });
  }
}

// ---- rows kernel, rows staged in halves (host_sb.hpp SbUpHalf): the hop over the TOP walked level ----------------------
// The image holds the blocks with one value of the top walked bit; rows_block walks the other levels with nbw - 1 and the
// LOW word.  xp: the words of the partner block (same low word, top bit toggled), read from the vector itself.  TOPSET = the
// top level is occupied in our block: partner class N + 1, else N - 1.  Every walked level below the top one is in wlow, so
// the sign is its parity.
template <int NIMP, int NB0, int N, bool TOPSET>
IB_HD void rows_top(uint32_t wlow, const double* vtop /* [NIMP]: V(a, top) */, const double* xp, double* acc) {
  constexpr int NLOC = NIMP + NB0;
  constexpr int ALL = (1 << NIMP) - 1;
  const uint32_t neg = (uint32_t)popc32(wlow) & 1u;
  double v[NIMP];
  sfor<0, NIMP>([&](auto A) { v[decltype(A)::value] = flip(vtop[decltype(A)::value], neg); });
  couple<NLOC, NIMP, ALL, N, !TOPSET, double>(v, xp, acc, FmaD{});
}
// words of the partner block of a class-N block over the top level (0: no such class)
template <int NLOC, int N, bool TOPSET>
constexpr int rows_top_words() {
  return TOPSET ? (N < NLOC ? binom(NLOC, N + 1) : 0) : (N >= 1 ? binom(NLOC, N - 1) : 0);
}

// ---- columns kernel: one block of rows x CW adjacent columns (T = Pair: two, T = double: one) ----------------------------
template <class T> struct FmaOf;
template <> struct FmaOf<double> { using type = FmaD; };
template <> struct FmaOf<Pair> { using type = FmaP; };
// chunk : the chunk's rows of the panel in the LDS, [row - chunk_row0][16] doubles; col = even column in the panel
// meta  : the 16 entries of the block's walked word (host_sb.hpp dmeta): [k] first row of block w ^ (1 << k), [14] own
//         first row, [15] bit k set when the walked levels below k hold an odd number of electrons
// gload(row): pointer to the two columns of a global row of the panel (partner blocks over the levels >= low live in
//         other chunks)
// acc[M] is added to: + (Hdw (x) 1) v.  The caller starts the accumulators (zero, or the rows kernel's part).
// The levels >= low are taken in level order with the partner rows of TWO levels in flight (buffers ga / gb); the
// levels < low, whose partners are in the LDS, are walked between the first requests and their use.
// whigh = w >> low, the SAME for the eight blocks of a wave-slot (host_sb.cpp deals the slots per (class, high word)): the
// direction of a hop over a level >= low, and with it the number of partner rows and the coupling code, is uniform.
// mid(): called between the LDS work and the hops over the levels >= low
template <int NIMP, int NB0, int AMODE, int N, class T, bool LINEAR = true, class GLoad, class TP, class KP, class Mid>
IB_HD void cols_block(const double* chunk, int chunk_row0, uint32_t w, uint32_t whigh, int own_row, const uint16_t* meta, int nbw, int low,
                      const double* vtab, int vs, KP korb, TP tloc, int col, GLoad&& gload, T* acc, Mid&& mid) {
  using Fma = typename FmaOf<T>::type;
  constexpr int NLOC = NIMP + NB0;
  constexpr int M = binom(NLOC, N);
  constexpr int MPD = binom(NLOC, N - 1), MPU = binom(NLOC, N + 1);
  constexpr int MPX = (MPD > MPU ? MPD : MPU) > 0 ? (MPD > MPU ? MPD : MPU) : 1;
  constexpr int ALL = (1 << NIMP) - 1;
  const uint32_t sbits = meta[15];
  auto lds_pair = [&](int row_rel) -> T { return *reinterpret_cast<const T*>(chunk + row_rel * 16 + col); };
  // one hop over walked level k with the partner rows in xp (amplitudes from the LDS table; uniform k or not)
  auto use = [&](auto DOWNC, auto AMC, int k, const T* xp) {
    constexpr bool DOWN = decltype(DOWNC)::value;
    constexpr int AM = decltype(AMC)::value;
    constexpr int MP = DOWN ? MPD : MPU;
    if constexpr (MP > 0) {
      const uint32_t neg = (sbits >> k) & 1u;
      double v[NIMP];
      sfor<0, NIMP>([&](auto A) {
        constexpr int a = decltype(A)::value;
        if constexpr (((AM >> a) & 1) != 0) v[a] = flip(vtab[k * vs + a], neg);
        else v[a] = 0.0;
      });
      couple<NLOC, NIMP, AM, N, DOWN, T>(v, xp, acc, Fma{});
    }
  };
  // which impurity levels level k couples to: all (AMODE 0), or the one whose mask holds it -- the multiply-adds of the
  // others are skipped by a branch on the mask (uniform for the levels >= low of an unmerged chunk)
  auto use_any = [&](auto DOWNC, int k, const T* xp) {
    if constexpr (AMODE == 0) {
      use(DOWNC, std::integral_constant<int, ALL>{}, k, xp);
    } else {
      sfor<0, NIMP>([&](auto A) {
        constexpr int a = decltype(A)::value;
        if ((korb[a] >> k) & 1u) use(DOWNC, std::integral_constant<int, (1 << a)>{}, k, xp);
      });
    }
  };
  const int nhigh = nbw - low;
  // partner rows of the block over level low + h (uniform direction: see whigh)
  auto load_hop = [&](int h, T* xg) {
    // LINEAR: the partner block's rows follow its first row 16 doubles apart (one address + immediate offsets); else every
    // row has its own address (row shards: a block may straddle the rows of two ranks)
    const int r2 = (int)meta[low + h];
    const auto g0 = gload(r2);
    const int n2 = ((whigh >> h) & 1u) ? MPU : MPD;
    sfor<0, MPX>([&](auto J) {
      constexpr int j2 = decltype(J)::value;
      if (j2 < n2) {  // uniform
        if constexpr (LINEAR) xg[j2] = *reinterpret_cast<const T*>(g0 + j2 * 16);
        else xg[j2] = *reinterpret_cast<const T*>(gload(r2 + j2));
      }
    });
  };
  auto use_hop = [&](int h, const T* xg) {
    if ((whigh >> h) & 1u)
      use_any(std::false_type{}, low + h, xg);
    else
      use_any(std::true_type{}, low + h, xg);
  };
  // hops among the local levels and the levels < low: everything in the LDS
  if constexpr (NLOC > 1) {
    T x[M];
    sfor<0, M>([&](auto J) { x[decltype(J)::value] = lds_pair(own_row - chunk_row0 + decltype(J)::value); });
    couple_loc<NLOC, NIMP, false, N, T>(tloc, x, acc, Fma{});
  }
  const uint32_t lowmask = (1u << low) - 1u;
  auto low_walk = [&](auto DOWNC, auto AMC, uint32_t m) {
    constexpr bool DOWN = decltype(DOWNC)::value;
    constexpr int MP = DOWN ? MPD : MPU;
    if constexpr (MP > 0) {
      while (m) {
        const int k = ctz32(m);
        m &= m - 1u;
        const int r2 = (int)meta[k] - chunk_row0;
        T xp[MP];
        sfor<0, MP>([&](auto J) { xp[decltype(J)::value] = lds_pair(r2 + decltype(J)::value); });
        use(DOWNC, AMC, k, xp);
      }
    }
  };
  if constexpr (AMODE == 0) {
    low_walk(std::true_type{}, std::integral_constant<int, ALL>{}, ~w & lowmask);
    low_walk(std::false_type{}, std::integral_constant<int, ALL>{}, w & lowmask);
  } else {
    sfor<0, NIMP>([&](auto A) {
      constexpr int a = decltype(A)::value;
      low_walk(std::true_type{}, std::integral_constant<int, (1 << a)>{}, ~w & lowmask & korb[a]);
      low_walk(std::false_type{}, std::integral_constant<int, (1 << a)>{}, w & lowmask & korb[a]);
    });
  }
  mid();
  // The levels >= low, two at a time: the partner rows of both are requested, then used.  Every pass of the loop is
  // self-contained on purpose -- with requests that cross the loop edge the compiler's wait-count bookkeeping turns
  // conservative (measured: a vmcnt(0) at the loop head, i.e. every hop paid a full round trip and the result rows'
  // HBM latency on top).
  T ga[MPX], gb[MPX];
  int h = 0;
#if SB_V_HOPS == 0
  if (nhigh > 0) load_hop(0, ga);
  for (; h + 1 < nhigh; h += 2) {
    load_hop(h + 1, gb);
    use_hop(h, ga);
    if (h + 2 < nhigh) load_hop(h + 2, ga);
    use_hop(h + 1, gb);
  }
  if (h < nhigh) use_hop(h, ga);
#elif SB_V_HOPS == 1
  for (; h + 1 < nhigh; h += 2) {
    load_hop(h, ga);
    load_hop(h + 1, gb);
    use_hop(h, ga);
    use_hop(h + 1, gb);
  }
  if (h < nhigh) {
    load_hop(h, ga);
    use_hop(h, ga);
  }
#else
  for (; h < nhigh; h++) {
    load_hop(h, ga);
    use_hop(h, ga);
  }
  (void)gb;
#endif
}

// ---- columns kernel: the factored Hnd terms of one block of rows x two columns -------------------------------------
// nd_dw: [nterms][NLOC + 1][kNdStride] (32-bit entries: scalar loads on the device): for state j of a class-n block:
//        partner state j' | 0x80 sign, 0xFF none
// ndu   : this panel's slice of nd_up, term t at ndu + t * ustride: (partner column - column + 8) | 0x80 sign, 0xFF none
// The terms only move impurity electrons: the partner rows belong to the same block, the partner columns to the same
// panel (small impurity blocks of columns never straddle one), so every operand is in the staged chunk.
template <int NIMP, int NB0, int N, class T, class UP, class TP>
IB_HD void cols_block_nd(const double* chunk, int own_rel, int col, int nterms, TP ndcoef, UP nd_dw,
                         const uint8_t* ndu, int ustride, T* acc) {
  constexpr int NLOC = NIMP + NB0;
  constexpr int M = binom(NLOC, N);
  constexpr int CW = (int)(sizeof(T) / sizeof(double));
  for (int t = 0; t < nterms; t++) {
    uint32_t u[CW];
    bool any = false;
    sfor<0, CW>([&](auto C) {
      u[decltype(C)::value] = ndu[t * ustride + col + decltype(C)::value];
      any = any || u[decltype(C)::value] != 0xFFu;
    });
    if (!any) continue;  // none of the lane's columns takes part in this term
    const auto dd = nd_dw + (t * (NLOC + 1) + N) * kNdStride;
    const double c = ndcoef[t];
    sfor<0, M>([&](auto J) {
      constexpr int j = decltype(J)::value;
      const uint32_t d = dd[j];
      if (d != 0xFFu) {  // uniform: a function of (term, class, state)
        const double* prow = chunk + (own_rel + (int)(d & 0x7Fu)) * 16 + col;
        double* aj = reinterpret_cast<double*>(&acc[j]);
        sfor<0, CW>([&](auto C) {
          constexpr int cc = decltype(C)::value;
          if (u[cc] != 0xFFu) aj[cc] = __builtin_fma(flip(c, ((d ^ u[cc]) >> 7) & 1u), prow[cc + (int)(u[cc] & 0x7Fu) - 8], aj[cc]);
        });
      }
    });
  }
}

}  // namespace sb
}  // namespace edigpu
