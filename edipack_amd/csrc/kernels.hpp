// kernels.hpp -- launch wrappers of the gfx950 kernels (definitions in kernels_*.hip).
#pragma once
#include <functional>

#include "edigpu_internal.hpp"

namespace edigpu {

// device-resident Lanczos scalars (kernels_lanczos.hip)
enum { SC_ALPHA = 0, SC_BETA = 1, SC_STOP = 2, SC_NDONE = 3, SC_NORM = 4, SC_THR = 5, SC_EXACT = 6, SC_AB = 8 };
constexpr int kRedBlocks = 1024;
constexpr int kMaxPartials = 1 << 16;  // minimum capacity of the per-workgroup partial buffer (edigpu_sector::partial_cap)

// opt a kernel into more than 48 KiB of dynamic LDS, once per (device, kernel, size) instead of on every launch
int ensure_dynamic_lds(const void* kernel, size_t bytes);
// workgroups of a kernel that stay resident on one CU (cached occupancy query; < 1 on error) and the CU count
int resident_blocks(const void* kernel, int threads, size_t dyn_lds);
int device_cu_count();

// ---- normal mode (kernels_normal.hip) ----
// phase: 3 = fused (local+remote, overwrite), 1 = local only (overwrite), 2 = remote only (accumulate)
// v_local : first element of the shard's own rows, v_full : element 0 of the whole vector.
int launch_normal(const edigpu_sector* s, const double* v_local, const double* v_full, double* hv,
                  int phase, hipStream_t st);
int normal_pick_rows_per_block(int64_t dim_up, int64_t dw_count);
// transposed exchange: the row half and the column half of the product on a whole-sector handle
bool normal_transposable(const edigpu_sector* s);
bool normal_transposable_el(const edigpu_sector* s);  // ignoring the phonon blocks
int launch_normal_rows(const edigpu_sector* s, int64_t dw_first, int64_t dw_count, const double* v_rows,
                       double* hv_rows, hipStream_t st);
int launch_normal_cols(const edigpu_sector* s, int64_t col_first, int64_t ncol, int64_t stride, int halo,
                       const double* w, double* hv, hipStream_t st);
// electron-phonon ladder: up[i] += c_up t[i], dn[i] += c_dn t[i] (either target may be NULL); n doubles
int launch_eph_scatter(int64_t n, const double* t, double* up, double c_up, double* dn, double c_dn, hipStream_t st);
// _CMPLX_NORMAL: interleaved complex <-> planar, y = (yr - t1) + i (yi + t2)  (kernels_ops.hip)
int launch_deinterleave(int64_t n, const double* z, double* re, double* im, hipStream_t st);
int launch_combine_interleave(int64_t n, const double* yr, const double* yi, const double* t1, const double* t2,
                              double* z, hipStream_t st);
// row shard <-> all-to-all buffers (kernels_ops.hip)
int launch_transpose_pack(int64_t dim_up, int64_t nrows, int64_t q, int world, int64_t pcol, int halo,
                          const double* v_rows, double* send, hipStream_t st);
int launch_transpose_unpack_add(int64_t dim_up, int64_t nrows, int64_t q, int world, int64_t pcol, int halo,
                                const double* recv, double* hv_rows, hipStream_t st);
// fused Lanczos step (normal, single shard): P = Lanczos vector, Q = work vector, see kernels_normal.hip
bool normal_lanczos_fusable(const edigpu_sector* s);
// nlanc / finalized: when the sweep finalizes the step in its last workgroup (*finalized = true) no finalize kernel must follow
// X / in_x: the impurity-block image writes the new Lanczos vector of a later step to X instead of P (*in_x = true: the
// caller swaps its P and X buffers); other images ignore them
int launch_normal_lanczos(const edigpu_sector* s, double* P, double* Q, const double* scal,
                          double* partial, int64_t partial_cap, bool first, bool lazy_axpy, hipStream_t st, int* npartial,
                          int nlanc = 0, bool* finalized = nullptr, double* X = nullptr, bool* in_x = nullptr);
// alpha = sum(partial[0:np]), beta = sqrt(sum(partial[np:2np]) - alpha^2) with an exact fallback pass
int lz_finalize_alpha_beta(const double* P, const double* Q, int64_t n, double* partial, int np,
                           double* scal, int iter, int nlanc, hipStream_t st);
int lz_finalize_alpha(const double* partial, int np, double* scal, int iter, int nlanc, hipStream_t st);

// ---- flat CSR (kernels_csr.hip) ----
// y[0:nrow] (=|+=) A x ; T = double (cplx=0) or (re,im) pairs (cplx=1)
int launch_csr(const DevCsr& a, int cplx, const double* x, double* y, int accumulate,
               hipStream_t st);
int launch_zero(double* y, int64_t n, hipStream_t st);

// ---- direct / on-the-fly (kernels_direct.hip): hv[local rows] = H v_full ----
// sig: device scalar the <Q|Q> partial is accumulated about (the previous alpha, scal + SC_ALPHA; see k_finalize_ab)
// cap: capacity of `partial` in doubles, checked BEFORE anything is enqueued (2 partials per workgroup)
int launch_direct_lanczos(const edigpu_sector* s, const double* v_full, double* q, double* partial, int64_t cap, int* np,
                          const double* sig, hipStream_t st);
bool csr_lanczos_fusable(const DevCsr& a);
int launch_csr_lanczos(const DevCsr& a, int cplx, const double* x, double* y, double* partial, int64_t cap, int* np,
                       const double* sig, hipStream_t st);
// (P, Q) <- ((Q - alpha*P)/beta, -beta*P): rotate with the pending axpy of the fused step folded in
int lz_rotate_lazy(double* P, double* Q, int64_t n, const double* scal, hipStream_t st);
int lz_next_vector(const double* P, const double* Q, double* X, int64_t n, const double* scal, bool lazy, hipStream_t st);
int launch_direct(const edigpu_sector* s, const double* v_full, double* hv, hipStream_t st);

// ---- Lanczos vector kernels (kernels_lanczos.hip); n counts doubles ----
int lz_norm_begin(double* vin, int64_t n, double* partial, double* scal, hipStream_t st);
int lz_rotate(double* vin, double* vout, int64_t n, const double* scal, hipStream_t st);
int lz_alpha(const double* vin, double* vout, const double* tmp, int64_t n, double* partial,
             double* scal, int iter, int nlanc, hipStream_t st);
int lz_beta(const double* vin, double* vout, int64_t n, double* partial, double* scal, int iter,
            int nlanc, hipStream_t st);
int lz_axpy_coef(double* acc, const double* vin, int64_t n, double coef, const double* scal,
                 int iter, hipStream_t st);
int lz_add_dot3(const double* P, double* Q, const double* tmp, int64_t n, const double* scal, double* partial, int* np,
                hipStream_t st);
int lz_fill_random(double* v, int64_t n, uint64_t seed, hipStream_t st);
// natural layout (idw * DimUp + iup) <-> panel-major layout of the Lanczos loop (normal_args.hpp: blk_shift); the
// padding columns of the last panel are written as zeros
int vec_to_blocked(const double* src, double* dst, int64_t dim_up, int64_t dim_dw, int shift, hipStream_t st);
int vec_from_blocked(const double* src, double* dst, int64_t dim_up, int64_t dim_dw, int shift, hipStream_t st);
int launch_normal_blocked(const edigpu_sector* s, const double* v, double* hv, hipStream_t st);
// impurity-block image (kernels_ib.hip): layout conversion, plain product and fused Lanczos step on its vectors
int vec_to_ib(const IbDev* ib, const double* src, double* dst, hipStream_t st);
int vec_from_ib(const IbDev* ib, const double* src, double* dst, hipStream_t st);
int launch_ib(const edigpu_sector* s, const double* v, double* hv, hipStream_t st);
int launch_ib_lanczos(const edigpu_sector* s, const double* P, double* Q, double* X, const double* scal, double* partial,
                      int64_t partial_cap, bool first, bool lazy_axpy, hipStream_t st, int* npartial);
// local-block kernels on the same layout (kernels_sb.hip, round 4); launch_ib / launch_ib_lanczos take them when IbDev::sb is set
int launch_sb(const edigpu_sector* s, const double* v, double* hv, hipStream_t st);
// generic LDS row kernel on the padded panel layout in position order (kernels_normal.hip; IbDev::pr)
int launch_normal_rows_pos(const edigpu_sector* s, const double* P, double* Q, double* X, int what, const double* scal, hipStream_t st);
int launch_sb_lanczos(const edigpu_sector* s, const double* P, double* Q, double* X, const double* scal, double* partial,
                      int64_t partial_cap, bool first, bool lazy_axpy, hipStream_t st, int* npartial);
// row shards on the padded panel layout (kernels_sb.hip): see there for the shard form of the layout
bool sb_shardable(const edigpu_sector* s);
int sb_shard_panels(const edigpu_sector* s, int world);
int sb_shard_to_panels(const edigpu_sector* s, const double* rows, double* dst, int64_t count, int64_t q, int world, hipStream_t st);
int sb_shard_from_panels_add(const edigpu_sector* s, const double* a, const double* b, double* rows, int64_t count, int64_t q, hipStream_t st);
int launch_sb_rows_shard(const edigpu_sector* s, int64_t row0, int64_t count, int64_t q, const double* v, double* hv, hipStream_t st);
int launch_sb_cols_shard(const edigpu_sector* s, int p0, int np, int64_t q, int npmax, const double* v, double* out, hipStream_t st);
int sb_nb0(int norb);
int sb_cols_waves();
int sb_cols_gs();
size_t sb_rows_lds(int nbw, int rimg_len, bool top = false);
// one half of the rows kernel's product on a sector whose rows are staged in halves (h = 0 / 1: top walked level empty /
// occupied); scal: the recurrence's scalars (stop flag) or null
int launch_sb_rows_half(const edigpu_sector* s, int h, const double* v, double* hv, const double* scal, hipStream_t st);
size_t sb_cols_lds(int nbw, int nloc, int max_chunk_rows, int max_chunk_slots, int gs);
bool sb_rows_config(int norb, int slots, int plen, int cs, int* nt_out, int* nbt_out);
size_t ib_rows_lds_bytes(int nb, int rimg_len);
size_t ib_cols_lds_bytes(int nb, int max_chunk_rows, int max_chunk_blocks);
bool ib_rows_config(int norb, int nb, int nlist, int plen, int rimg_len, int* nt_out, int* nbt_out, bool split = false);
int measure_membw(int64_t bytes, double out[3]);
// stand-alone vector kernels with explicit device scalars (sharded loop)
int vec_rotate(int64_t n, double* vin, double* vout, const double* beta2, hipStream_t st);
int vec_add_dot(int64_t n, const double* vin, double* vout, const double* tmp, double* out, double* work, hipStream_t st);
int vec_axpy_nrm2(int64_t n, const double* vin, double* vout, const double* alpha, double* out, double* work, hipStream_t st);
int vec_scale(int64_t n, double* v, const double* nrm2, hipStream_t st);
// one-reduction recurrence on any sharded vector: (v, w) <- ((w - alpha v)/beta, -beta v) from ab = (<v|w>, <w|w>),
// and w += tmp with this rank's (<v|w>, <w|w>) (work: kRedBlocks doubles)
int vec_rotate_lazy(int64_t n, double* vin, double* vout, const double* ab, hipStream_t st);
int vec_add_dot2(int64_t n, const double* vin, double* vout, const double* tmp, double* out2, double* work,
                 hipStream_t st);
// fused vector updates of the transposed-exchange Lanczos step (work: kRedBlocks doubles)
int vec_rotate_pack(int first, int64_t dim_up, int64_t nrows, int64_t q, int world, int64_t pcol, int halo,
                    double* vin, double* vout, const double* ab, double* send, hipStream_t st);
int vec_unpack_add_dot2(int64_t dim_up, int64_t nrows, int64_t q, int64_t pcol, int halo, const double* vin,
                        double* vout, const double* tmp, const double* back, double* out2, double* work,
                        hipStream_t st);

// ---- thick-restart Lanczos driver (edigpu_capi.hip), shared by edigpu_lanczos_eigh_multi and its sharded twin ----
struct TrlOps {
  std::function<int(const double* in, double* out, hipStream_t st)> apply;  // out = H in on this rank's elements
  std::function<int(double* dev, size_t n, hipStream_t st)> allreduce;     // sum over the ranks, in place; empty: one rank
};
int trl_solve(int device, hipStream_t st, int cplx, int64_t n, int64_t len, int64_t nglobal, const TrlOps& ops, int neigen, int ncv,
              double tol, int maxrestart, const double* v0, uint64_t seed_offset, double* evals, double* evecs,
              int* nconv_out, int* nmatvec_out);

// ---- thick-restart Lanczos multi-vector kernels (kernels_trl.hip) ----
// h_dev (2 doubles per basis vector: re, im) = Q^H w, then w -= Q h; n counts complex or real elements
int trl_orthogonalize(int cplx, int64_t n, int nvec, const double* Q, int64_t ldq, double* w, double* h_dev,
                      double* partial, hipStream_t st, const int* skip = nullptr);
int trl_dots(int cplx, int64_t n, int ndot, const double* Q, int64_t ldq, const double* w, double* h_dev,
             double* partial, hipStream_t st, const int* skip = nullptr);
int trl_subtract(int cplx, int64_t n, int nvec, const double* Q, int64_t ldq, const double* h_dev, double* w,
                 hipStream_t st, const int* skip = nullptr);
// after the first sweep (nvec coefficients + <w|w> in h_dev): skip <- 1 when at least eta2 of |w|^2 is left (no second
// pass); hf_dev <- the coefficients with those below thr * |w_new| set to exactly zero (all kept when skip = 0)
int trl_decide(const double* h_dev, int nvec, double eta2, double thr2, double* hf_dev, int* skip, hipStream_t st);
int trl_norm2(int cplx, int64_t n, const double* w, double* h_dev, double* partial, hipStream_t st);
// coefficients (nvec pairs + <w|w>) below sqrt(thr2) * |w| -> exact zeros, in place
int trl_filter(double* h_dev, int nvec, double thr2, hipStream_t st);
int trl_partial_doubles(void);
int trl_rotate_basis(int64_t len, int m, int k, const double* Q, int64_t ldq, const double* Y_dev, int ldy,
                     double* out, int64_t ldo, hipStream_t st);
int trl_scale(int64_t len, double* v, double f, hipStream_t st);

// ---- c / c^+ between normal-mode sectors (kernels_ops.hip) ----
int launch_apply_op_normal(int64_t dst_dimup, int64_t dst_dimdw, int64_t src_dimup, int spin_down,
                           const uint32_t* part, const double* src, double* dst, hipStream_t st, double coef = 1.0,
                           int accumulate = 0);

// one electronic block of a superc / nonsu2 phonon handle, phase 1 (own rows) / 2 (gathered block)  (edigpu_capi.hip)
int apply_flat_block(edigpu_sector* s, const double* v_local, const double* v_full, double* hv, int phase, hipStream_t st);
int launch_phonon(const edigpu_sector* s, const double* v, double* hv, hipStream_t st);
// the same pass on a shard of down rows [dw_first, dw_first + dw_count) of a normal-mode sector: (Nph + 1) blocks of
// dw_count * DimUp elements (the layout of spMatVec_mpi_normal_main); density couplings only
int launch_phonon_rows(const edigpu_sector* s, int64_t dw_first, int64_t dw_count, const double* v, double* hv,
                       hipStream_t st);
int launch_apply_op_flat(int64_t ndst, int ns, uint32_t bit, int create, const int32_t* dst_states,
                         const int32_t* src_offdw, const int32_t* src_rkup, const double* src, double* dst,
                         hipStream_t st, double cre = 1.0, double cim = 0.0, int accumulate = 0);
// apply_Cops on a window of destination units (edigpu_capi.hip): v_dst_rows = sum_s coef_s O_s v_src for the
// destination's down rows (normal mode) / rows (superc, nonsu2) [first, first + count); v_src_full holds the WHOLE source
// vector in the reference's layout; coef2 = (re, im) per operator (normal mode: real).  Returns after the stream finished.
int apply_cops_rows(edigpu_sector* src, edigpu_sector* dst, const double* v_src_full, double* v_dst_rows, int64_t first,
                    int64_t count, int nops, const double* coef2, const int32_t* create, const int32_t* iorb,
                    const int32_t* ispin, hipStream_t st, const char* who);

// ---- ed_total_ud = F sectors (kernels_orbs.hip) ----
int launch_orbs(const edigpu_sector* s, const double* v, double* hv, hipStream_t st);

// ---- on-device construction of the stored flat image (kernels_build.hip) ----
struct BuildArgs {
  int64_t nrow, row_first;  // local rows, first global row
  int64_t lo, hi;           // column window of the loc block (global indices)
  int ns, norb, nterms;
  const int32_t* states;
  const int32_t* off_dw;
  const int32_t* rk_up;
  const DirectTerm* terms;
  const uint8_t* vid;  // [2 * nterms]: dictionary id of +coef for (term, forward / reverse); id ^ 1 = -coef
  const double* dtab;
  const double* xtab;
};
// widths per 64-row slice of the loc / non-local block; totals = {entries loc, entries non-local,
// longest row loc, longest row non-local}
int launch_build_count(const BuildArgs& a, int32_t* width_loc, int32_t* width_non, unsigned long long* totals,
                       hipStream_t st);
int launch_build_fill(const BuildArgs& a, int which, int maxlen, const int32_t* sptr, uint32_t* pk, double* diag,
                      hipStream_t st);

}  // namespace edigpu
