// kernels_sb3.hip -- the local-block kernels (kernels_sb_impl.hpp) for 3 impurity level(s) per species, 2 low bath
// levels folded into the blocks (5 local levels).  One translation unit per orbital count: they compile in parallel.
#include "kernels_sb_impl.hpp"

namespace edigpu {

int sb_rows_3(const IbDev* d, const SbArgs& a, int fuse, const double* P, double* Q, double* X, hipStream_t st) {
  if (d->sb->amode) return sb_launch_rows<3, 2, 1>(d, a, fuse, P, Q, X, st);
  return sb_launch_rows<3, 2, 0>(d, a, fuse, P, Q, X, st);
}

int sb_cols_3(const IbDev* d, const SbArgs& a, int mode, const double* v, double* hv, hipStream_t st, int* nblocks) {
  if (d->sb->amode) return sb_launch_cols<3, 2, 1>(d, a, mode, v, hv, st, nblocks);
  return sb_launch_cols<3, 2, 0>(d, a, mode, v, hv, st, nblocks);
}

}  // namespace edigpu
