// vsweep.hip -- plane sweep for grid operators with VARIABLE coefficients (MK_SPMV_VSWEEP).
//
// Stands in for hypre_ParCSRMatrixMatvec on matrices like the authors' real inputs: a grid structure, every row its own
// values.  The constant-coefficient sweep (spmv.hip spmv_sweep_kernel) reads no CSR array at all; here the values are
// what the product is bound by (8 B per entry), and everything else is arranged so that nothing but them, one 2-byte
// pattern id, one 4-byte row pointer, x once and y once crosses the HBM interface per row:
//   - x: block (chunk of C = 2T rows of a plane, z-segment) walks plane by plane, the x entries of planes k-1, k, k+1
//     around its rows sit in an LDS ring of four slabs filled by coalesced 16-byte loads that run three planes ahead in
//     registers (as in spmv_sweep_kernel): every gather is an LDS read, HBM sees x once;
//   - values: the rows of a chunk are consecutive, so their values are ONE contiguous span val[rowptr[g0] ..
//     rowptr[g0 + C]): it is loaded with coalesced 16-byte loads three steps ahead, staged in LDS, and every row reads
//     its entries from there at rowptr[row] - rowptr[g0] + (number of earlier entries of the base pattern the row has);
//   - column indices: never read.  Row r's pattern id selects a bit mask over the base pattern (offsets a*P + b), whose
//     LDS ring offsets come eight at a time from a table by scalar loads.
// Rows are summed left to right over the same (value, x) pairs as the CSR arrays hold, so the result equals the serial
// reference's bit for bit (same rounding sequence).  Algorithmic bytes: SURVEY 8(d)'s 12 nnz + 20 n + 4; read here:
// 8 nnz + 22 n.
#include <cfloat>

#include "mk_internal.h"

namespace mk {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte access that is only 8-byte aligned
typedef int i2_t __attribute__((ext_vector_type(2)));
constexpr int kNS = 4;   // ring slots: planes k-1, k, k+1 and the one being written

template <bool NT, typename V>
__device__ __forceinline__ void vs_st(V *ptr, V v) {
  if (NT) __builtin_nontemporal_store(v, ptr); else *ptr = v;
}
template <bool NT, typename V>
__device__ __forceinline__ V vs_ld(const V *ptr) {
  return NT ? __builtin_nontemporal_load(ptr) : *ptr;
}

template <bool NT>
__device__ __forceinline__ d2u_t vs_ld_u(const d2u_t *ptr) {   // keeps the 8-byte alignment of the type (no template deduction on it)
  if (NT) return __builtin_nontemporal_load(ptr);
  return *ptr;
}

// MODE 0: y = alpha A x + beta y.  1: y = A x, partials of <y,x>, <x,x>.  2: Jacobi sweep y = x + D^-1 (rhs - extra - A x),
// partial of ||rhs - extra - A x||^2, D picked up from the row's own entry with offset 0 (the last one stored).
// MODE 3 (CG, first kernel of every step but the first): K3 rides in the window fill, as in spmv_sweep_kernel MODE 5.
//   Prologue = K3's (gamma', hypre's convergence test, beta from the partials in slab_in).  The window is filled with
//   p' = z + beta x instead of x (x = the old p; z = rhs, the residual, times `extra` = 1/diag per row when the Jacobi
//   preconditioner is on: a third window stream), the block's own rows of p' go to pnew, then as MODE 1 on p':
//   y = A p', partials of <y,p'>, <p',p'>.  flags: bit 0 two_norm, bit 1 rel_change.
//   With xupd (the solution vector) the step's solution update rides here too, one kernel late: the fill has the old p in
//   registers, so the block's own rows do xupd += alpha_prev * p_old (alpha_prev = the alpha of the step before, pending in
//   the state record) and leave partials of <xupd,xupd>, <xupd,p'> from which the next kernel derives <x,x> after ITS update
//   (hypre's rel_change test needs it); K2 then no longer reads p or touches x: 8 B per row less.
// NL = 16-byte ring loads per thread per slab = ceil(sw / 2T); NVL = 16-byte value loads per thread per plane-step
// (NVL * 2T >= len * C entries).
// RPL = rows per lane.  2: rows of at most 8 entries (C = 2T rows per chunk, the values of three plane-steps in flight in
// registers).  1: LONG rows, up to 28 entries (27-point operators: 216 B of values per row): C = T = 512 rows per chunk, one
// block of 8 waves per CU whose stage holds len * 512 values (110 KB for 27 points) beside the window ring, ONE plane-step
// of values in flight in registers (a step of 512 long rows outlasts the memory latency several times over).  Same order
// of operations per row either way.
template <int MODE, int NL, int NVL, int T, bool NT, bool PRE, int RPL = 2>
__global__ __launch_bounds__(T, 2) void spmv_vsweep_kernel(
    const SweepArgs S, const SweepEntry *__restrict__ etab, int64_t nrows, int64_t ncols, int64_t nnz,
    const unsigned short *__restrict__ pat, const unsigned *__restrict__ pmask, int npat, const int *__restrict__ rowptr,
    const int *__restrict__ rowpair, const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, double alpha, double beta,
    double *__restrict__ slab, const int *__restrict__ done, const double *__restrict__ rhs, const double *__restrict__ extra,
    PcgState *st, const double *__restrict__ slab_in, int nblocks_in, int stride_in, double *__restrict__ pnew, int flags,
    double *__restrict__ norms, double *__restrict__ xupd) {
  if (done != nullptr && *done != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // ring slots: planes k-1, k, k+1 and the one being written -- which, with this kernel's two barriers per step, may be the
  // slot plane k-1 leaves (the write sits between the barriers, when nobody reads): long rows take three slots and give
  // the LDS to the value stage
  constexpr int C = RPL * T, NS = (RPL == 1) ? 3 : kNS, VCAP = NVL * 2 * T;
  // register stages of values: three plane-steps ahead for short rows; long rows two (a plane-step of 512 rows x 27 values is
  // 110 KB per CU), re-loaded during the products (ILV below)
  constexpr int VST = (RPL == 1) ? 2 : 3;
  const int sw = S.sw;
  // entries of the value stage: what the loads cover (RPL 2), or exactly one plane-step of the chunk's values (RPL 1: the
  // last pieces of the last load fall behind it and are not written -- the stage fills the LDS to the last KB)
  const int vcap = (RPL == 1) ? ((S.len * C + 1) & ~1) : VCAP;
  double *ring = reinterpret_cast<double *>(smem_raw);          // NS slabs of sw entries
  double *sval = ring + NS * sw;                                 // vcap + 8 entries: the values of one plane-step
  unsigned *smask = reinterpret_cast<unsigned *>(sval + vcap + 8);   // npat
  const int tid = threadIdx.x;
  if (MODE == 3) {   // K3's prologue (krylov_kernels.h pcg_update_p_kernel), every block the same bits
    if (block_flag_raised(&st->done)) return;
    const bool leader = (blockIdx.x == 0 && tid == 0);
    double d[3];
    reduce_slab<3, T>(slab_in, nblocks_in, stride_in, d, ring);
    const double gamma_new = d[0];
    const double i_prod = (flags & 1) ? d[1] : d[0];
    const double xx = d[2];
    const double bi_prod = st->bi_prod, eps = st->eps, al = st->alpha, pp = st->pp, gamma_old = st->gamma_old;
    __syncthreads();
    bool conv = (i_prod / bi_prod < eps);
    if (conv && (flags & 2) && i_prod > 0.0) {
      const double ratio = al * al * pp / xx;
      if (ratio >= eps) conv = false;
    }
    const bool tiny_gamma = (gamma_new < DBL_MIN) && (-gamma_new < DBL_MIN);
    if (leader) {
      st->gamma = gamma_new;
      st->i_prod = i_prod;
      if (norms) norms[st->iter] = sqrt(i_prod / bi_prod);
      if (conv) { st->converged = 1; st->done = 1; }
      else if (tiny_gamma) { st->breakdown = 3; st->done = 2; }
    }
    if (conv || tiny_gamma) return;
    beta = gamma_new / gamma_old;
    if (xupd != nullptr) alpha = st->x_pending ? al : 0.0;   // (alpha has no other use in this mode)
    __syncthreads();   // ring was reduction scratch
  }
  for (int i = tid; i < npat; i += T) smask[i] = pmask[i];
  const int nb = gridDim.x;   // multiple of 8; blocks of one XCD take neighbouring chunks of a plane (shared halos in its L2)
  const int lid = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  if (lid < S.ncp * S.nseg) {
    const int nr = (int)nrows;
    const int cp = lid % S.ncp, seg = lid / S.ncp;
    const int k0 = (int)((int64_t)seg * S.K / S.nseg), k1 = (int)((int64_t)(seg + 1) * S.K / S.nseg);
    const int r0 = cp * C;
    const int rows_here = min(C, S.P - r0);
    const int lr = RPL * tid;   // the lane's row(s) in the chunk
    const int lw = 2 * tid;     // its place in the window fill (two entries per 16-byte piece, pieces 2T apart)
    // every load that runs ahead is unconditional, with its address clamped into the arrays (see spmv_sweep_kernel)
    const int xlast = ((int)ncols - 1) & ~1;
    const int rlast = (RPL == 2) ? ((nr - 1) & ~1) : nr - 1;
    const int elast = (int)max((int64_t)0, nnz - 1);   // val[elast + 1] is allocation slack (dev_alloc): never used by a row

    // the one window piece of this lane that holds rows of the block's own chunk (pieces lie 2 T >= C apart), and its offset
    // in the chunk: the rows whose p' the lane stores and, with xupd, whose solution entries it updates
    int oi = -1, ownoff = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int own = lw + i * 2 * T - S.wlo;
      if (own >= 0 && own < rows_here && lw + i * 2 * T < sw) { oi = i; ownoff = own; }
    }
    const bool defx = (MODE == 3) && xupd != nullptr;
    auto load_x = [&](int j) -> d2_t {
      const int g = min(max(min(j, k1) * S.P + r0 + ownoff, 0), (nr - 1) & ~1);   // (plane k0 - 1 of the first segment lies before the array)
      return vs_ld<NT>(reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(xupd) + ((unsigned)g << 3)));
    };
    constexpr int NLR = (MODE == 3) ? NL : 1;              // second window stream (the residual) only in MODE 3
    constexpr int NLD = (MODE == 3 && PRE) ? NL : 1;       // third (1/diag per row) with the Jacobi preconditioner
    auto load_slab = [&](int j, d2_t(&reg)[NL], d2_t(&regr)[NLR], d2_t(&regd)[NLD], d2_t &regx) {   // window of plane j
      if (defx) regx = load_x(j);
      const int g0 = min(j, k1) * S.P + r0 - S.wlo + lw;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int g = min(max(g0 + i * 2 * T, 0), xlast);
        reg[i] = *reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(x) + ((unsigned)g << 3));
        if (MODE == 3) regr[i] = *reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(rhs) + ((unsigned)g << 3));
        if (MODE == 3 && PRE) regd[i] = *reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(extra) + ((unsigned)g << 3));
      }
    };
    auto write_slab = [&](int slot, int j, const d2_t(&reg)[NL], const d2_t(&regr)[NLR], const d2_t(&regd)[NLD], const d2_t &regx) {
      double *dst = ring + slot * sw + lw;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int wi = lw + i * 2 * T;   // position in the window
        if (wi < sw) {
          d2_t v = reg[i];
          if (MODE == 3) {
            d2_t z = regr[i];
            if (PRE) { z.x *= regd[i].x; z.y *= regd[i].y; }
            v.x = beta * v.x + z.x;   // K3: p = beta p + z
            v.y = beta * v.y + z.y;
            const int own = wi - S.wlo;   // the block's own rows of its own planes go to the new p
            if (j >= k0 && j < k1 && own >= 0 && own < rows_here) {
              const int g = j * S.P + r0 + own;
              if (g + 1 < nr) vs_st<NT>(reinterpret_cast<d2_t *>(reinterpret_cast<char *>(pnew) + ((unsigned)g << 3)), v);
              else if (g < nr) *reinterpret_cast<double *>(reinterpret_cast<char *>(pnew) + ((unsigned)g << 3)) = v.x;
              if (defx && i == oi && g < nr) {   // x += alpha_prev p_old on the same rows; partials of <x,x>, <x,p'>
                d2_t xv = regx;
                xv.x += alpha * reg[i].x;
                xv.y += alpha * reg[i].y;
                acc2 += xv.x * xv.x;
                acc3 += xv.x * v.x;
                if (g + 1 < nr) {
                  acc2 += xv.y * xv.y;
                  acc3 += xv.y * v.y;
                  vs_st<NT>(reinterpret_cast<d2_t *>(reinterpret_cast<char *>(xupd) + ((unsigned)g << 3)), xv);
                } else {
                  *reinterpret_cast<double *>(reinterpret_cast<char *>(xupd) + ((unsigned)g << 3)) = xv.x;
                }
              }
            }
          }
          *reinterpret_cast<d2_t *>(dst + i * 2 * T) = v;
        }
      }
    };
    // first value of the chunk's rows in plane j (uniform over the block: scalar loads)
    auto span_begin = [&](int j) -> int { return rowptr[min(min(j, k1) * S.P + r0, nr)]; };
    auto load_vals = [&](int vb, d2u_t(&reg)[NVL]) {   // val[vb + 2 (tid + i T)], two at a time
#pragma unroll
      for (int i = 0; i < NVL; i++) {
        const int e = min(vb + 2 * (tid + i * T), elast);
        reg[i] = vs_ld_u<NT>(reinterpret_cast<const d2u_t *>(reinterpret_cast<const char *>(val) + ((size_t)(unsigned)e << 3)));
      }
    };
    // LDS stage of the values, swizzled: entry e lives at e ^ ((e >> 5) & 1).  A lane reads the values of its two
    // rows at a stride of 2 * len entries from its neighbour's: an even stride uses only every other pair of banks and
    // 64-bit reads collide two ways (measured: bank-conflict cycles were 70 % of the LDS busy cycles); flipping the low
    // bit in every other block of 32 entries makes the 32 lanes of a pass hit 32 different bank pairs for len 3, 5, 7.
    // A thread's write positions 2 (tid + i T) all have bit 5 = bit 4 of tid: it swaps its pairs or never does.
    // (Only for even strides: with one row per lane and 27 entries per row the stride is odd, 16 lanes already hit 16 different
    // bank pairs, and flipping bit 0 of some of them would create the two-way conflicts the swizzle is there to remove.)
    const int swm = ((RPL * S.len) & 1) ? 0 : 1;
    const bool swp = swm && ((tid >> 4) & 1);
    auto write_vals = [&](const d2u_t(&reg)[NVL]) {
#pragma unroll
      for (int i = 0; i < NVL; i++) {
        d2_t v;
        v.x = swp ? reg[i].y : reg[i].x;
        v.y = swp ? reg[i].x : reg[i].y;
        if (RPL == 2 || 2 * (tid + i * T) < vcap) *reinterpret_cast<d2_t *>(sval + 2 * (tid + i * T)) = v;
      }
    };
    auto stage_at = [&](int idx) -> double { return sval[idx ^ ((idx >> 5) & swm)]; };
    auto row_of = [&](int j) -> int { return j * S.P + r0 + lr; };
    auto live = [&](int j) -> bool { return j < k1 && lr < rows_here && row_of(j) < nr; };
    auto load_ids = [&](int j) -> unsigned {   // pattern ids of rows g, g+1 (g even); the id behind the last row is slack
      const int g = min(row_of(j), rlast);
      if (RPL == 1) return (unsigned)vs_ld<NT>(reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(pat) + ((unsigned)g << 1)));
      return vs_ld<NT>(reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(pat) + ((unsigned)g << 1)));
    };
    // first value of the lane's row pair: rowpair[g / 2] = rowptr[g] (g even), 2 B per row instead of 4; the second row's values
    // follow the first row's, whose length is the popcount of its mask
    auto load_rp = [&](int j) -> int {
      const int g = min(row_of(j), rlast);
      if (RPL == 1) return vs_ld<NT>(reinterpret_cast<const int *>(reinterpret_cast<const char *>(rowptr) + ((unsigned)g << 2)));   // (4 B per row of 200)
      return vs_ld<NT>(reinterpret_cast<const int *>(reinterpret_cast<const char *>(rowpair) + ((unsigned)g << 1)));
    };
    auto load_pair = [&](const double *p, int j) -> d2_t {
      const int g = min(row_of(j), rlast);
      if (RPL == 1) {
        d2_t v;
        v.x = vs_ld<NT>(reinterpret_cast<const double *>(reinterpret_cast<const char *>(p) + ((unsigned)g << 3)));
        v.y = 0.0;
        return v;
      }
      return vs_ld<NT>(reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(p) + ((unsigned)g << 3)));
    };
    const double *aux = (MODE == 2) ? rhs : (MODE == 0 && beta != 0.0) ? y : nullptr;
    constexpr bool kExtra = (MODE == 2);

    // register stages, used in place: ring stage (k mod 3) holds slab k+2 at step k; value stage ((k+1) mod 3) holds the
    // values of plane k+1 at step k, is written to LDS behind the step's first barrier and re-loaded with plane k+4
    d2_t st0[NL], st1[NL], st2[NL], rt0[NLR], rt1[NLR], rt2[NLR], dt0[NLD], dt1[NLD], dt2[NLD];
    d2_t xt0 = {0.0, 0.0}, xt1 = xt0, xt2 = xt0;
    constexpr int NV1 = (VST >= 2) ? NVL : 1, NV2 = (VST == 3) ? NVL : 1;   // (stages the variant does not have shrink to one register)
    d2u_t vs0[NVL], vs1[NV1], vs2[NV2];
    int vb0 = span_begin(k0), vb1 = span_begin(k0 + 1), vb2 = span_begin(k0 + 2);
    load_slab(k0 - 1, st0, rt0, dt0, xt0);
    load_slab(k0, st1, rt1, dt1, xt1);
    load_slab(k0 + 1, st2, rt2, dt2, xt2);
    load_vals(vb0, vs0);
    write_slab(0, k0 - 1, st0, rt0, dt0, xt0);
    write_slab(1, k0, st1, rt1, dt1, xt1);
    write_slab(2, k0 + 1, st2, rt2, dt2, xt2);
    write_vals(vs0);
    int vbcur = vb0;                // span begin of the plane whose values sit in LDS
    int vbn;                        // begin of the span loaded next (plane k + VST + 1 at step k), fetched one step early
    if constexpr (VST == 3) {
      load_vals(vb1, reinterpret_cast<d2u_t(&)[NVL]>(vs1));
      load_vals(vb2, reinterpret_cast<d2u_t(&)[NVL]>(vs2));
      vb0 = span_begin(k0 + 3);
      load_vals(vb0, vs0);
      vbn = span_begin(k0 + 4);
    } else if constexpr (VST == 2) {
      vb0 = vb1;                    // stage 0: plane k0 + 1; stage 1 is loaded with plane k0 + 2 during the first step's products
      load_vals(vb0, vs0);
      vbn = vb2;
    } else {
      vb0 = vb1;                    // the one stage holds plane k0 + 1 now
      load_vals(vb0, vs0);
      vbn = vb2;
    }
    d2_t ax0 = {0.0, 0.0}, ax1 = ax0, ax2 = ax0, ex0 = ax0, ex1 = ax0, ex2 = ax0;
    unsigned id0 = load_ids(k0);
    int rp0 = load_rp(k0);
    if (aux) ax0 = load_pair(aux, k0);
    if (kExtra && extra) ex0 = load_pair(extra, k0);
    load_slab(k0 + 2, st0, rt0, dt0, xt0);
    unsigned id1 = load_ids(k0 + 1);
    int rp1 = load_rp(k0 + 1);
    if (aux) ax1 = load_pair(aux, k0 + 1);
    if (kExtra && extra) ex1 = load_pair(extra, k0 + 1);
    load_slab(k0 + 3, st1, rt1, dt1, xt1);
    unsigned id2 = load_ids(k0 + 2);
    int rp2 = load_rp(k0 + 2);
    if (aux) ax2 = load_pair(aux, k0 + 2);
    if (kExtra && extra) ex2 = load_pair(extra, k0 + 2);
    load_slab(k0 + 4, st2, rt2, dt2, xt2);
    __syncthreads();

    // vbcur: span begin of plane k (its values are in sval now).  vsn / vbs: value stage of plane k+1 and its span begin,
    // re-loaded with plane k+4 whose span begin is vbn (fetched during the previous step).
    // s3 = (k - k0) mod 3, a constant at every call (the loops below advance by multiples of three)
    // ILV (long rows, two value stages): the stage written to LDS in the step before (vso) is re-loaded DURING this step's
    // products, a few loads in front of every batch of entries, instead of in one burst behind the LDS writes: a CU keeps only
    // so many misses in flight, a burst of 14 loads per lane held all its waves in the issue until memory had taken most of
    // them, and loads and products took turns (27-point 256^3: 0.60 ms without the products, 0.45 ms without the loads, 0.885 ms
    // together).
    constexpr bool ILV = (RPL == 1 && VST == 2);
    constexpr int NB = ILV ? (2 * NVL + 7) / 8 : 1, QL = (NVL + NB - 1) / NB;
    auto step = [&](int k, int s3, d2_t(&st)[NL], d2_t(&rt)[NLR], d2_t(&dt)[NLD], d2_t &xt, unsigned &idr, int &rpr, d2_t &axr, d2_t &exr, d2u_t(&vsn)[NVL], int &vbs,
                    d2u_t(&vso)[NVL], int &vbo) {
      const int vbk = vbcur;
      const unsigned idc = idr;
      const int rpc = rpr;
      const d2_t axc = axr, exc = exr;
      const int jb = (NS == 4) ? ((k - k0 + 1) & 3) : (s3 + 1) % 3;   // ring slot of plane k
      const bool is_live = live(k);
      {
        const int g = row_of(k);
        const bool two = (RPL == 2) && g + 1 < nr;
        const unsigned mA = smask[min(idc & 0xffffu, (unsigned)npat - 1u)];
        const unsigned mB = two ? smask[min(idc >> 16, (unsigned)npat - 1u)] : 0u;
        const char *wl = reinterpret_cast<const char *>(ring) + lr * 8;
        // positions of the two rows' first values in the stage; a row without a live partner reads its own
        const int vA = rpc - vbk;
        const int vB = two ? vA + __popc(mA) : vA;
        double s0 = 0.0, s1 = 0.0;
        // long rows: a wave whose rows all hold the whole base pattern (every wave away from the grid's faces) finds entry u
        // of a row at u -- no population counts, no selects; same products added in the same order
        const unsigned fullm = S.len >= 32 ? 0xffffffffu : (1u << S.len) - 1u;
        const bool wave_full = (RPL == 1) && __all(!is_live || mA == fullm);
        auto batch = [&](int u0) {
          typedef int i16_t __attribute__((ext_vector_type(16)));
          const i16_t *eb = reinterpret_cast<const i16_t *>(etab + jb * 32 + u0);
          const i16_t b0 = eb[0], b1 = eb[1];
          if (RPL == 1 && wave_full) {
            const int offf[8] = {b0[0], b0[4], b0[8], b0[12], b1[0], b1[4], b1[8], b1[12]};
            double pf[8], cf[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              pf[u] = *reinterpret_cast<const double *>(wl + offf[u]);
              cf[u] = stage_at(vA + u0 + u);
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
              if (u0 + u < S.len) s0 = s0 + cf[u] * pf[u];
            return;
          }
          const int off[8] = {b0[0], b0[4], b0[8], b0[12], b1[0], b1[4], b1[8], b1[12]};
          const unsigned ka = mA >> u0, kb = mB >> u0;
          const int va0 = vA + __popc(mA & ((1u << u0) - 1u));
          const int vb0p = vB + __popc(mB & ((1u << u0) - 1u));
          double p0[8], p1[8], c0[8], c1[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const double *w = reinterpret_cast<const double *>(wl + off[u]);
            p0[u] = w[0];
            if (RPL == 2) p1[u] = w[1];
            // the u-th entry of the base pattern is the row's (number of earlier entries it has)-th value
            c0[u] = stage_at(va0 + __popc(ka & ((1u << u) - 1u)));
            if (RPL == 2) c1[u] = stage_at(vb0p + __popc(kb & ((1u << u) - 1u)));
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const double t0 = s0 + c0[u] * p0[u];
            s0 = ((ka >> u) & 1u) ? t0 : s0;
            if (RPL == 2) {
              const double t1 = s1 + c1[u] * p1[u];
              s1 = ((kb >> u) & 1u) ? t1 : s1;
            }
          }
        };
        if constexpr (ILV) {
          const int vbnew = vbn;   // span begin of plane k + 2 (fetched during the step before)
#pragma unroll
          for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int i = b * QL; i < (b + 1) * QL; i++)
              if (i < NVL) {
                const int e = min(vbnew + 2 * (tid + i * T), elast);
                vso[i] = vs_ld_u<NT>(reinterpret_cast<const d2u_t *>(reinterpret_cast<const char *>(val) + ((size_t)(unsigned)e << 3)));
              }
            if (is_live && b * 8 < S.len) batch(b * 8);
          }
          vbo = vbnew;
        } else {
          if (is_live)
            for (int u0 = 0; u0 < S.len; u0 += 8) batch(u0);
        }
        if (is_live) {
        d2_t out;
        if (MODE == 0) {
          out.x = (beta == 0.0) ? alpha * s0 : alpha * s0 + beta * axc.x;
          out.y = (beta == 0.0) ? alpha * s1 : alpha * s1 + beta * axc.y;
        } else {
          const double *wc = reinterpret_cast<const double *>(wl) + jb * sw + S.wlo;   // x[g], x[g+1]
          const double xa = wc[0], xb = (RPL == 2) ? wc[1] : 0.0;
          if (MODE == 2) {
            const double ra = (axc.x - exc.x) - s0, rb = (axc.y - exc.y) - s1;
            double da = 0.0, db = 0.0;
            if (S.diag >= 0) {
              const unsigned below = (1u << S.diag) - 1u;
              if ((mA >> S.diag) & 1u) da = stage_at(vA + __popc(mA & below));
              if (RPL == 2 && ((mB >> S.diag) & 1u)) db = stage_at(vB + __popc(mB & below));
            }
            out.x = xa + ra / da;
            out.y = xb + rb / db;
            acc0 += ra * ra;
            if (two) acc0 += rb * rb;
          } else {
            out.x = s0;
            out.y = s1;
            acc0 += s0 * xa;
            acc1 += xa * xa;
            if (two) { acc0 += s1 * xb; acc1 += xb * xb; }
          }
        }
        if (two) vs_st<NT>(reinterpret_cast<d2_t *>(reinterpret_cast<char *>(y) + ((unsigned)g << 3)), out);
        else if (RPL == 1) vs_st<NT>(reinterpret_cast<double *>(reinterpret_cast<char *>(y) + ((unsigned)g << 3)), out.x);
        else *reinterpret_cast<double *>(reinterpret_cast<char *>(y) + ((unsigned)g << 3)) = out.x;
        }
      }
      __syncthreads();   // everybody has read this plane's values
      // slab k+2 (loaded three steps ago) into the slot plane k-2 occupied; the values of plane k+1 into the stage
      write_slab((NS == 4) ? ((k - k0 + 3) & 3) : s3, k + 2, st, rt, dt, xt);
      write_vals(vsn);
      // look-ahead loads into the registers just consumed: ids / row pointers / aux of plane k+3, slab k+5, values k+4
      idr = load_ids(k + 3);
      rpr = load_rp(k + 3);
      if (aux) axr = load_pair(aux, k + 3);
      if (kExtra && extra) exr = load_pair(extra, k + 3);
      load_slab(k + 5, st, rt, dt, xt);
      vbcur = vbs;
      if constexpr (ILV) {
        vbn = span_begin(k + VST + 1);   // (the stage is re-loaded during the next step's products)
      } else {
        vbs = vbn;
        load_vals(vbn, vsn);
        vbn = span_begin(k + VST + 2);
      }
      __syncthreads();
    };
    if constexpr (VST == 2) {
      auto &vA = vs0;
      auto &vB = reinterpret_cast<d2u_t(&)[NVL]>(vs1);   // (NV1 == NVL here)
      for (int k = k0; k < k1; k += 6) {   // three ring stages x two value stages
        step(k, 0, st0, rt0, dt0, xt0, id0, rp0, ax0, ex0, vA, vb0, vB, vb1);
        step(k + 1, 1, st1, rt1, dt1, xt1, id1, rp1, ax1, ex1, vB, vb1, vA, vb0);
        step(k + 2, 2, st2, rt2, dt2, xt2, id2, rp2, ax2, ex2, vA, vb0, vB, vb1);
        step(k + 3, 0, st0, rt0, dt0, xt0, id0, rp0, ax0, ex0, vB, vb1, vA, vb0);
        step(k + 4, 1, st1, rt1, dt1, xt1, id1, rp1, ax1, ex1, vA, vb0, vB, vb1);
        step(k + 5, 2, st2, rt2, dt2, xt2, id2, rp2, ax2, ex2, vB, vb1, vA, vb0);
      }
    } else {
    for (int k = k0; k < k1; k += 3) {   // straight-line body: steps at or behind k1 compute nothing (live() is false)
      if constexpr (VST == 3) {
        auto &v1 = reinterpret_cast<d2u_t(&)[NVL]>(vs1);
        auto &v2 = reinterpret_cast<d2u_t(&)[NVL]>(vs2);
        step(k, 0, st0, rt0, dt0, xt0, id0, rp0, ax0, ex0, v1, vb1, v1, vb1);
        step(k + 1, 1, st1, rt1, dt1, xt1, id1, rp1, ax1, ex1, v2, vb2, v2, vb2);
        step(k + 2, 2, st2, rt2, dt2, xt2, id2, rp2, ax2, ex2, vs0, vb0, vs0, vb0);
      } else {
        step(k, 0, st0, rt0, dt0, xt0, id0, rp0, ax0, ex0, vs0, vb0, vs0, vb0);
        step(k + 1, 1, st1, rt1, dt1, xt1, id1, rp1, ax1, ex1, vs0, vb0, vs0, vb0);
        step(k + 2, 2, st2, rt2, dt2, xt2, id2, rp2, ax2, ex2, vs0, vb0, vs0, vb0);
      }
    }
    }
  }
  if (MODE != 0) {
    __syncthreads();
    double w[4] = {acc0, acc1, acc2, acc3};
    block_sum<4, T>(w, ring);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = w[0];
      if (MODE == 1 || MODE == 3) slab[1 * kSlabStride + blockIdx.x] = w[1];
      if (MODE == 3 && xupd != nullptr) {
        slab[2 * kSlabStride + blockIdx.x] = w[2];
        slab[3 * kSlabStride + blockIdx.x] = w[3];
      }
    }
  }
}

// rowpair[i] = rowptr[2 i]: where the values of rows 2 i, 2 i + 1 (one lane's pair) begin
__global__ void rowpair_kernel(int64_t nrows, const int *__restrict__ rowptr, int *__restrict__ rowpair) {
  const int64_t np = (nrows + 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (int64_t)gridDim.x * blockDim.x) rowpair[i] = rowptr[2 * i];
}
int build_vsweep_rowpair(mk_matrix *A) {
  if (A->rowpair || !A->sweep || !A->sweep_streamed) return MK_OK;
  const int64_t np = (A->nrows + 1) / 2;
  MK_TRY(dev_alloc(&A->rowpair, (size_t)np + 1));
  hipLaunchKernelGGL(rowpair_kernel, dim3((unsigned)std::min<int64_t>((np + 255) / 256, 8192)), dim3(256), 0, ctx().stream, A->nrows, A->rowptr, A->rowpair);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// CG arguments of MODE 3 (unused by the other modes)
struct VsCg {
  PcgState *st = nullptr;
  const double *slab_in = nullptr;
  int nblocks_in = 0, stride_in = kSlabStride;
  double *pnew = nullptr;
  int flags = 0;
  double *norms = nullptr;
  double *xupd = nullptr;
};

template <int MODE, int NL, int NVL, int T, bool PRE, int RPL = 2>
static int launch_vsweep_t(const mk_matrix *A, const SweepArgs &S, size_t lds, int blocks, bool nt, double alpha, const double *x, double beta,
                           double *y, double *slab, const int *done, const double *rhs, const double *extra, const VsCg &cg) {
  Context &c = ctx();
  auto kern = nt ? spmv_vsweep_kernel<MODE, NL, NVL, T, true, PRE, RPL> : spmv_vsweep_kernel<MODE, NL, NVL, T, false, PRE, RPL>;
  // dynamic LDS above 48 KB needs an opt-in per kernel (no allocation).  The attribute belongs to the kernel, not to the
  // matrix: it is only ever raised, so a matrix with a smaller stage cannot lower it under one with a larger stage.
  static size_t lds_max[2] = {48 * 1024, 48 * 1024};   // per template instance and cache policy
  if (lds > lds_max[nt ? 1 : 0]) {
    MK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_max[nt ? 1 : 0] = lds;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(T), lds, c.stream, S, A->etab, A->nrows, A->ncols, A->nnz, A->pat, A->pmask,
                     A->npat, A->rowptr, A->rowpair, A->val, x, y, alpha, beta, slab, done, rhs, extra, cg.st, cg.slab_in, cg.nblocks_in,
                     cg.stride_in, cg.pnew, cg.flags, cg.norms, cg.xupd);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

template <int MODE, bool PRE = false>
static int launch_vsweep_m(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab, const int *done,
                           int *nblocks, const double *rhs, const double *extra, const VsCg &cg = VsCg()) {
  Context &c = ctx();
  if (!A->sweep || !A->sweep_streamed || !A->pat || !A->rowpair)
    return fail(MK_ERR_UNSUPPORTED, "spmv: MK_SPMV_VSWEEP needs rows that are sub-lists, by column offset, of one grid pattern of at most 28 entries");
  SweepArgs S = *A->sweep;
  const bool long_rows = (S.rpl == 1);   // one row per lane (plan_sweep: rows of more than 8 entries)
  const int T = long_rows ? S.C : S.C / 2;
  const int nvl = long_rows ? (S.len <= 14 ? 7 : 14) : (S.len <= 3 ? 3 : S.len <= 5 ? 5 : S.len <= 7 ? 7 : 8);
  const size_t lds = (size_t)(long_rows ? 3 : kNS) * S.sw * 8 + (long_rows ? (size_t)((S.len * S.C + 1) & ~1) + 8 : (size_t)nvl * S.C + 8) * 8 +
                     (size_t)A->npat * 4;
  if ((long_rows ? (S.len > 28 || (T != 512 && T != 256)) : S.len > 8) || lds > 160 * 1024 - 64)
    return fail(MK_ERR_UNSUPPORTED, "spmv: vsweep stage of %zu bytes (rows of %d entries) does not fit the LDS", lds, S.len);
  const int per_cu = std::max(1, (int)((160 * 1024) / lds));
  // segments: two rounds of resident blocks (each segment re-reads two halo planes: at least 8 planes per segment)
  static const int env_nseg = getenv("MK_SWEEP_NSEG") ? atoi(getenv("MK_SWEEP_NSEG")) : 0;
  const int64_t target = (int64_t)c.num_cus * per_cu;
  int nseg = env_nseg > 0 ? env_nseg : (int)std::max<int64_t>(1, (2 * target + S.ncp - 1) / S.ncp);
  nseg = std::min(nseg, std::max(1, S.K / 8));
  nseg = std::max(1, std::min(nseg, S.K));
  while ((int64_t)S.ncp * nseg > kSlabStride - 8 && nseg > 1) nseg--;
  if ((int64_t)S.ncp * nseg > kSlabStride - 8) return fail(MK_ERR_UNSUPPORTED, "spmv: sweep grid of %d chunks per plane is too large", S.ncp);
  S.nseg = nseg;
  const int blocks = (S.ncp * nseg + 7) / 8 * 8;
  if (nblocks) *nblocks = blocks;
  if (dry_launch()) return MK_OK;
  const int nl = std::max(2, (S.sw + 2 * T - 1) / (2 * T));
  const bool nt = nt_rows(A->nrows) || nt_matrix((double)A->nnz * 8.0);
#define MK_VS1(NLL, NVV, TT) return launch_vsweep_t<MODE, NLL, NVV, TT, PRE, 1>(A, S, lds, blocks, nt, alpha, x, beta, y, slab, done, rhs, extra, cg);
  if (long_rows) {
    // (no CG mode for long rows: its window fill holds three streams, the register file then leaves room for ONE value stage,
    // and the step with K3 folded in measured slower than K1 + K2 + K3 with two stages: 27-point 256^3 1.073 against 1.023 ms)
    if constexpr (MODE == 3) return fail(MK_ERR_UNSUPPORTED, "spmv: the vsweep kernel has no fused CG mode for rows of more than 8 entries");
    else {
    if (T == 512 && nl == 2) { if (nvl == 7) MK_VS1(2, 7, 512) else MK_VS1(2, 14, 512) }
    if (T == 256 && nl == 2) { if (nvl == 7) MK_VS1(2, 7, 256) else MK_VS1(2, 14, 256) }
    if (T == 256 && nl == 3) { if (nvl == 7) MK_VS1(3, 7, 256) else MK_VS1(3, 14, 256) }
    return fail(MK_ERR_UNSUPPORTED, "spmv: vsweep window of %d entries for chunks of %d long rows", S.sw, S.C);
    }
  }
#undef MK_VS1
#define MK_VS(NLL, NVV, TT) return launch_vsweep_t<MODE, NLL, NVV, TT, PRE>(A, S, lds, blocks, nt, alpha, x, beta, y, slab, done, rhs, extra, cg);
#define MK_VS_NV(NLL, TT)                \
  switch (nvl) {                         \
    case 3: MK_VS(NLL, 3, TT)            \
    case 5: MK_VS(NLL, 5, TT)            \
    case 7: MK_VS(NLL, 7, TT)            \
    default: MK_VS(NLL, 8, TT)           \
  }
  if (T == 256 && nl == 2) { MK_VS_NV(2, 256) }
  if (T == 256 && nl == 3) { MK_VS_NV(3, 256) }
  if (T == 512 && nl == 2) { MK_VS_NV(2, 512) }
  if (T == 512 && nl == 3) { MK_VS_NV(3, 512) }
#undef MK_VS_NV
#undef MK_VS
  return fail(MK_ERR_UNSUPPORTED, "spmv: vsweep window of %d entries for chunks of %d rows", S.sw, S.C);
}

int launch_vsweep(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done,
                  int *nblocks, const double *rhs, const double *extra) {
  if (mode == 0) return launch_vsweep_m<0>(A, alpha, x, beta, y, slab, done, nblocks, rhs, extra);
  if (mode == 1) return launch_vsweep_m<1>(A, alpha, x, beta, y, slab, done, nblocks, rhs, extra);
  if (mode == 2) return launch_vsweep_m<2>(A, alpha, x, beta, y, slab, done, nblocks, rhs, extra);
  return fail(MK_ERR_INVALID, "launch_vsweep: mode %d", mode);
}

// CG step on vsweep matrices, first kernel from the second iteration on: p_new = z + beta p_old formed in the window fill
// (z = r, or r * dinv per row), s = A p_new, partials of <s, p_new>, <p_new, p_new> into slab_out; prologue = K3's
int launch_vsweep_cg_pdots(const mk_matrix *A, const double *r, const double *p_old, double *p_new, const double *dinv, double *s,
                           const double *slab_in, int nblocks_in, int stride_in, double *slab_out, PcgState *st, int two_norm,
                           int rel_change, double *norms, int *nblocks, double *xupd) {
  VsCg cg;
  cg.xupd = xupd;
  cg.st = st; cg.slab_in = slab_in; cg.nblocks_in = nblocks_in; cg.stride_in = stride_in; cg.pnew = p_new;
  cg.flags = (two_norm ? 1 : 0) | (rel_change ? 2 : 0); cg.norms = norms;
  if (dinv) return launch_vsweep_m<3, true>(A, 1.0, p_old, 0.0, s, slab_out, nullptr, nblocks, r, dinv, cg);
  return launch_vsweep_m<3, false>(A, 1.0, p_old, 0.0, s, slab_out, nullptr, nblocks, r, nullptr, cg);
}

}  // namespace mk
