// ib_core.hpp -- per-block arithmetic of the impurity-block kernels (host_ib.hpp explains the decomposition).
//
// Plain C++17, usable from device code (kernels_ib.hip) and from the host (tests/host_ib.cpp runs the same routines
// on ordinary arrays, so the CPU suite checks the tables and the index / sign logic without a GPU).
//
// Conventions.  A species' state is (b << NORB) | p: p = impurity pattern, b = bath word.  Class n = popcount(p); the
// j-th state of a class-n block is the j-th NORB-bit word with n bits set, ascending.  A hop between impurity level a
// and bath level k has the matrix element V(a,k) * (-1)^(imp bits above a + bath bits below k)
// (ED_AUX_FUNX.f90:334-384 c / cdg: the sign counts the occupied levels between the two).
#pragma once
#include <cstdint>
#include <type_traits>

#if defined(__HIPCC__)
#define IB_HD __host__ __device__ inline
#else
#define IB_HD inline
#endif

// a value that is the same on every lane of a wave (device: moved to a scalar register, so loops on it are scalar loops)
#if defined(__HIP_DEVICE_COMPILE__)
#define IB_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define IB_UNIFORM(x) (x)
#endif

namespace edigpu {
namespace ib {

constexpr int binom(int n, int k) {
  if (k < 0 || k > n) return 0;
  int r = 1;
  for (int i = 1; i <= k; i++) r = r * (n - k + i) / i;
  return r;
}
constexpr int popc(unsigned x) {
  int c = 0;
  for (; x; x &= x - 1) c++;
  return c;
}
// j-th word (ascending) with n of its low `norb` bits set
constexpr int pat(int norb, int n, int j) {
  int c = 0;
  for (int p = 0; p < (1 << norb); p++)
    if (popc((unsigned)p) == n) {
      if (c == j) return p;
      c++;
    }
  return 0;
}
// rank of p among the words with the same popcount
constexpr int idx(int p) {
  int c = 0;
  for (int q = 0; q < p; q++)
    if (popc((unsigned)q) == popc((unsigned)p)) c++;
  return c;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
IB_HD void sfor(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    sfor<B + 1, E>(f);
  }
}

// f(std::integral_constant<int, N>) for the class n (0 .. NORB) given at run time (uniform over a wave).  An else-chain
// on purpose: with independent tests the compiler merges what the bodies write with vector selects.
template <int NORB, int C = 0, class F>
IB_HD void for_class(int n, F&& f) {
  if constexpr (C == NORB) {
    f(std::integral_constant<int, C>{});
  } else {
    if (n == C)
      f(std::integral_constant<int, C>{});
    else
      for_class<NORB, C + 1>(n, f);
  }
}

struct alignas(16) Pair {
  double x, y;
};

IB_HD int popc32(uint32_t v) { return __builtin_popcount(v); }
IB_HD int ctz32(uint32_t v) { return __builtin_ctz(v); }  // v != 0
IB_HD double flip(double x, uint32_t neg) {  // neg = 0 or 1: one xor on the sign bit
  return __builtin_bit_cast(double, __builtin_bit_cast(uint64_t, x) ^ ((uint64_t)neg << 63));
}

// ---- the couplings of one block to one partner block, shared by both kernels --------------------------------------
// acc[j] += sum over the impurity levels a that the hop can use:  s * V(a) * xp[partner of row j]
// DOWN = true : the bath level is EMPTY in our block  -> the partner has it occupied and one impurity electron less
//               (class N-1): a runs over the occupied impurity levels of row j
// DOWN = false: the bath level is OCCUPIED in our block -> partner class N+1, a over the empty impurity levels
// T: double (rows kernel) or Pair (columns kernel: two adjacent columns)
template <int NORB, int N, bool DOWN, class T, class FMA>
IB_HD void couple(const double* v /* [NORB] amplitudes V(a,k), bath sign already applied */, const T* xp, T* acc, FMA&& fma_) {
  constexpr int M = binom(NORB, N);
  sfor<0, M>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = pat(NORB, N, j);
    sfor<0, NORB>([&](auto A) {
      constexpr int a = decltype(A)::value;
      constexpr bool occ = ((p >> a) & 1) != 0;
      if constexpr (occ == DOWN) {
        constexpr int p2 = p ^ (1 << a);
        constexpr int j2 = idx(p2);
        constexpr bool neg = (popc((unsigned)(p >> (a + 1))) & 1) != 0;
        fma_(acc[j], neg ? -v[a] : v[a], xp[j2]);
      }
    });
  });
}

// impurity-impurity hops inside a block: acc[j] += +/- t(a1,a2) x[j'], p_j' = p_j with the electron moved
template <int NORB, int N, class T, class FMA>
IB_HD void couple_imp(const double* timp /* [NORB][NORB] */, const T* x, T* acc, FMA&& fma_) {
  constexpr int M = binom(NORB, N);
  sfor<0, NORB>([&](auto A1) {
    constexpr int a1 = decltype(A1)::value;
    sfor<a1 + 1, NORB>([&](auto A2) {
      constexpr int a2 = decltype(A2)::value;
      const double t = timp[a1 * NORB + a2];
      if (t != 0.0) {  // uniform
        sfor<0, M>([&](auto J) {
          constexpr int j = decltype(J)::value;
          constexpr int p = pat(NORB, N, j);
          if constexpr ((((p >> a1) ^ (p >> a2)) & 1) != 0) {
            constexpr int p2 = p ^ (1 << a1) ^ (1 << a2);
            constexpr int btw = ((1 << a2) - 1) & ~((1 << (a1 + 1)) - 1);
            constexpr bool neg = (popc((unsigned)(p & btw)) & 1) != 0;
            fma_(acc[j], neg ? -t : t, x[idx(p2)]);
          }
        });
      }
    });
  });
}

struct FmaD {
  IB_HD void operator()(double& a, double c, double x) const { a = __builtin_fma(c, x, a); }
};
struct FmaP {
  IB_HD void operator()(Pair& a, double c, const Pair& x) const {
    a.x = __builtin_fma(c, x.x, a.x);
    a.y = __builtin_fma(c, x.y, a.y);
  }
};

// ---- rows kernel: one block of columns of the staged row -----------------------------------------------------------
// The staged row is held CLASS BY CLASS, word by word: word j of the i-th block of class n sits at
// cb[n] + j * cs[n] + i (i = rank of the block's bath word inside its class, ascending).  Lanes of a wave hold
// consecutive blocks of one class, and the partner block b ^ (1 << k) of consecutive blocks has (almost) consecutive
// ranks in ITS class: a gather instruction reads (almost) consecutive words -- few bank conflicts, where the position
// order of the vector scatters the 64 reads over the whole row (measured: 8.9 LDS cycles per instruction, half of
// them conflicts, the LDS pipe busy 75 % of the kernel).
struct RowImage {
  const double* row;     // the staged row (LDS)
  const uint16_t* rank;  // [2^nb] rank of a bath word inside its class (LDS)
  int cb[5], cs[5];      // first word / stride between the words of a block, per class (index n + 1: cb[0] = class -1)
};

// vtab  : [nb][4] (LDS): amplitudes V(a,k), a < NORB, and in [k][3] the level energy eps_k of bath level k
// timp  : [NORB][NORB] (uniform: scalar loads on the device)
// dconst: ed[idw];  xu: [2^NORB] diagonal part that depends on this block's impurity pattern (+ that of the down word)
// acc[M] receives (Hd + 1 (x) Hup) v for the block's columns
// np, pmask, pt: the bath-bath hops (host_ib.hpp IbSide::pmask; np = 0 for normal / hybrid baths), uniform
template <int NORB, int N>
IB_HD void rows_block(const RowImage& im, uint32_t b, uint32_t i, int nb, const double* vtab, const double* timp,
                      double dconst, const double* xu, double* acc, int np = 0, const uint32_t* pmask = nullptr,
                      const double* pt = nullptr) {
  constexpr int M = binom(NORB, N);
  const double* row = im.row;
  const double* own = row + im.cb[N + 1] + i;
  const int so = im.cs[N + 1];
  double x[M];
  sfor<0, M>([&](auto J) {
    constexpr int j = decltype(J)::value;
    x[j] = own[j * so];
    acc[j] = 0.0;
  });
  if constexpr (NORB > 1) couple_imp<NORB, N, double>(timp, x, acc, FmaD{});
  // A lane walks the clear bits of its own bath word (hops that fill the level: partner class N - 1), then the set
  // bits (partner class N + 1): no dead slots, no divergence -- every block of a class has the same number of each --
  // and only the partner's real words are read.  The price: the level differs from lane to lane, so the amplitudes
  // V(a,k) come from the LDS copy of vtab (rows of 32 bytes: at most 14 distinct addresses per instruction).
  // The walk over the set bits also sums the bath part of the diagonal, eps_k of the occupied levels (a per-block
  // table of it would cost two registers per block for the whole kernel).
  double ebath = 0.0;
  if constexpr (N >= 1) {
    constexpr int MP = binom(NORB, N - 1);
    const double* pbase = row + im.cb[N];
    const int st2 = im.cs[N];
    uint32_t m0 = ~b & ((1u << nb) - 1u);
    while (m0) {
      const int k = ctz32(m0);
      m0 &= m0 - 1u;
      const double* pp = pbase + im.rank[b | (1u << k)];
      const uint32_t neg = (uint32_t)popc32(b & ((1u << k) - 1u)) & 1u;
      double xp[MP], v[NORB];
      sfor<0, MP>([&](auto J) { xp[decltype(J)::value] = flip(pp[decltype(J)::value * st2], neg); });
      sfor<0, NORB>([&](auto A) { v[decltype(A)::value] = vtab[k * 4 + decltype(A)::value]; });
      couple<NORB, N, true, double>(v, xp, acc, FmaD{});
    }
  }
  {
    constexpr int MP = binom(NORB, N + 1);  // 0 for the top class: only the energies are summed
    const double* pbase = row + im.cb[N < NORB ? N + 2 : N + 1];
    const int st2 = im.cs[N < NORB ? N + 2 : N + 1];
    uint32_t m1 = b;
    while (m1) {
      const int k = ctz32(m1);
      m1 &= m1 - 1u;
      ebath += vtab[k * 4 + 3];
      if constexpr (N < NORB) {
        const double* pp = pbase + im.rank[b & ~(1u << k)];
        const uint32_t neg = (uint32_t)popc32(b & ((1u << k) - 1u)) & 1u;
        double xp[MP > 0 ? MP : 1], v[NORB];
        sfor<0, MP>([&](auto J) { xp[decltype(J)::value] = flip(pp[decltype(J)::value * st2], neg); });
        sfor<0, NORB>([&](auto A) { v[decltype(A)::value] = vtab[k * 4 + decltype(A)::value]; });
        couple<NORB, N, false, double>(v, xp, acc, FmaD{});
      }
    }
  }
  // bath-bath hops: the partner block has the same class and the same impurity patterns, word j pairs with word j
  for (int q = 0; q < np; q++) {  // uniform
    const uint32_t pm = pmask[q], mk = pm & 0xFFFFu, occ = b & mk;
    if (occ != 0u && occ != mk) {  // exactly one of the two levels is occupied
      const double* pp = row + im.cb[N + 1] + im.rank[b ^ mk];
      const double ts = flip(pt[q], (uint32_t)popc32(b & (pm >> 16)) & 1u);
      sfor<0, M>([&](auto J) {
        constexpr int j = decltype(J)::value;
        acc[j] = __builtin_fma(ts, pp[j * so], acc[j]);
      });
    }
  }
  sfor<0, M>([&](auto J) {
    constexpr int j = decltype(J)::value;
    acc[j] = __builtin_fma(ebath + dconst + xu[pat(NORB, N, j)], x[j], acc[j]);
  });
}

// ---- rows kernel, split rows: the hop over the TOP bath level ----------------------------------------------------------
// The image holds the blocks with one value of the top bath bit (host_ib.hpp IbUpHalf); rows_block walks the other
// levels with nb - 1 and the LOW bath word.  xp: the words of the partner block (top bit toggled), read from the vector.
// TOPSET = the top level is occupied in our block: partner class N + 1, else N - 1.  Every level below the top one is
// in blow, so the bath sign is its parity.
template <int NORB, int N, bool TOPSET>
IB_HD void rows_top(uint32_t blow, const double* vtop /* [NORB]: V(a, top) */, const double* xp, double* acc) {
  const uint32_t neg = (uint32_t)popc32(blow) & 1u;
  double v[NORB];
  sfor<0, NORB>([&](auto A) { v[decltype(A)::value] = flip(vtop[decltype(A)::value], neg); });
  couple<NORB, N, !TOPSET, double>(v, xp, acc, FmaD{});
}
// words of the partner block of a class-N block over the top level (0: no such class)
template <int NORB, int N, bool TOPSET>
constexpr int rows_top_words() {
  return TOPSET ? (N < NORB ? binom(NORB, N + 1) : 0) : (N >= 1 ? binom(NORB, N - 1) : 0);
}

// ---- columns kernel: one block of rows x two adjacent columns ------------------------------------------------------
// chunk : the chunk's rows of the panel in the LDS, [row - chunk_row0][16] doubles; col = even column in the panel
// meta  : the 16 entries of the block's bath word (host_ib.hpp dmeta): partner first rows, sign bits in [15]
// gload(row): the two columns of a global row of the panel (rows outside the chunk: hops to the bath levels >= low)
// acc[M] is added to: + (Hdw (x) 1) v
constexpr int kHB = 3;  // high bath levels whose partner rows are in flight together

// np, pmask, pt: the bath-bath hops; dmeta = the whole table (the first row of the partner block b ^ mask is its entry 14)
template <int NORB, int N, class GLoad>
IB_HD void cols_block(const double* chunk, int chunk_row0, uint32_t b, int own_row, const uint16_t* meta, int nb, int low,
                      const double* vtab, const double* timp, int col, GLoad&& gload, Pair* acc, int np = 0,
                      const uint32_t* pmask = nullptr, const double* pt = nullptr, const uint16_t* dmeta = nullptr) {
  constexpr int M = binom(NORB, N);
  constexpr int MPD = binom(NORB, N - 1), MPU = binom(NORB, N + 1);
  constexpr int MPX = MPD > MPU ? MPD : MPU;
  const uint32_t sbits = meta[15];
  auto lds_pair = [&](int row_rel) -> Pair { return *reinterpret_cast<const Pair*>(chunk + row_rel * 16 + col); };
  auto use = [&](int k, const Pair* xp) {
    const uint32_t neg = (sbits >> k) & 1u;
    double v[NORB];
    sfor<0, NORB>([&](auto A) { v[decltype(A)::value] = flip(vtab[k * 4 + decltype(A)::value], neg); });
    if ((b >> k) & 1u) {
      if constexpr (N < NORB) couple<NORB, N, false, Pair>(v, xp, acc, FmaP{});
    } else {
      if constexpr (N >= 1) couple<NORB, N, true, Pair>(v, xp, acc, FmaP{});
    }
  };
  // rows of the partner block the hop over level k reads (0 when the partner class does not exist)
  auto nrows = [&](int k) -> int { return ((b >> k) & 1u) ? MPU : MPD; };
  // high levels, first batch: issue the loads before the LDS work
  Pair xg[kHB][MPX > 0 ? MPX : 1];
  const int nhigh = nb - low;
  auto issue = [&](int h0) {
    sfor<0, kHB>([&](auto H) {
      constexpr int h = decltype(H)::value;
      const int k = low + h0 + h;
      if (h0 + h < nhigh) {
        const int r2 = meta[k], n2 = nrows(k);
        sfor<0, MPX>([&](auto J) {
          constexpr int j2 = decltype(J)::value;
          if (j2 < n2) xg[h][j2] = gload(r2 + j2);
        });
      }
    });
  };
  auto consume = [&](int h0) {
    sfor<0, kHB>([&](auto H) {
      constexpr int h = decltype(H)::value;
      if (h0 + h < nhigh) use(low + h0 + h, xg[h]);
    });
  };
  issue(0);
  // impurity-impurity hops and the low levels: everything in the LDS.  The low levels are walked in as many parts as
  // there are batches of high levels, one part behind each batch's loads.
  if constexpr (NORB > 1) {
    Pair x[M];
    sfor<0, M>([&](auto J) { x[decltype(J)::value] = lds_pair(own_row - chunk_row0 + decltype(J)::value); });
    couple_imp<NORB, N, Pair>(timp, x, acc, FmaP{});
  }
  // Low levels: a lane walks the CLEAR low bits of its own bath word (partner class N - 1), then the SET ones (class
  // N + 1), as the rows kernel does.  A loop over k with a branch on the bit would run BOTH couplings for every level
  // (the eight blocks of a wave differ in their low bits), and read partner rows that are then thrown away.
  const uint32_t lowmask = (1u << low) - 1u;
  auto low_walk = [&](auto DOWNC, uint32_t m) {
    constexpr bool DOWN = decltype(DOWNC)::value;
    constexpr int MP = DOWN ? MPD : MPU;
    if constexpr (MP > 0) {
      while (m) {
        const int k = ctz32(m);
        m &= m - 1u;
        const int r2 = (int)meta[k] - chunk_row0;
        Pair xp[MP];
        sfor<0, MP>([&](auto J) { xp[decltype(J)::value] = lds_pair(r2 + decltype(J)::value); });
        const uint32_t neg = (sbits >> k) & 1u;
        double v[NORB];
        sfor<0, NORB>([&](auto A) { v[decltype(A)::value] = flip(vtab[k * 4 + decltype(A)::value], neg); });
        couple<NORB, N, DOWN, Pair>(v, xp, acc, FmaP{});
      }
    }
  };
  const int nbatch = nhigh > kHB ? 2 : 1;  // (the host keeps nhigh <= 2 kHB)
  low_walk(std::true_type{}, ~b & lowmask);
  if (nbatch == 2) {
    consume(0);
    issue(kHB);
  }
  low_walk(std::false_type{}, b & lowmask);
  consume(nbatch == 2 ? kHB : 0);
  // bath-bath hops: rows of the partner block, in the chunk when both levels are low ones
  for (int q = 0; q < np; q++) {  // uniform
    const uint32_t pm = pmask[q], mk = pm & 0xFFFFu, occ = b & mk;
    if (occ != 0u && occ != mk) {
      const int r2 = dmeta[(size_t)(b ^ mk) * 16 + 14];
      const double ts = flip(pt[q], (uint32_t)popc32(b & (pm >> 16)) & 1u);
      const bool inside = (mk >> low) == 0u;  // uniform
      sfor<0, M>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const Pair xp = inside ? lds_pair(r2 - chunk_row0 + j) : gload(r2 + j);
        FmaP{}(acc[j], ts, xp);
      });
    }
  }
}

// ---- columns kernel: the factored Hnd terms of one block of rows x two columns -------------------------------------
// nd_dw: [nterms][NORB + 1][4] partner row inside the block | 0x80 sign, 0xFF none (host_ib.hpp)
// ndu   : this panel's slice of nd_up, term t at ndu + t * ustride: (partner column - column + 8) | 0x80 sign, 0xFF none
// The partner rows belong to the same block (the terms only move impurity electrons) and the partner columns to the
// same panel (blocks of columns never straddle one), so every operand is in the staged chunk.
template <int NORB, int N>
IB_HD void cols_block_nd(const double* chunk, int own_rel, int col, int nterms, const double* ndcoef, const uint8_t* nd_dw,
                         const uint8_t* ndu, int ustride, Pair* acc) {
  constexpr int M = binom(NORB, N);
  for (int t = 0; t < nterms; t++) {
    const uint8_t* dd = nd_dw + (t * (NORB + 1) + N) * 4;
    const double c = ndcoef[t];
    const uint32_t u0 = ndu[t * ustride + col], u1 = ndu[t * ustride + col + 1];
    sfor<0, M>([&](auto J) {
      constexpr int j = decltype(J)::value;
      const uint32_t d = dd[j];
      if (d != 0xFFu) {  // uniform over the lanes of a class
        const double* prow = chunk + (own_rel + (int)(d & 0x7Fu)) * 16 + col;
        if (u0 != 0xFFu) acc[j].x = __builtin_fma(flip(c, ((d ^ u0) >> 7) & 1u), prow[(int)(u0 & 0x7Fu) - 8], acc[j].x);
        if (u1 != 0xFFu) acc[j].y = __builtin_fma(flip(c, ((d ^ u1) >> 7) & 1u), prow[1 + (int)(u1 & 0x7Fu) - 8], acc[j].y);
      }
    });
  }
}

}  // namespace ib
}  // namespace edigpu
