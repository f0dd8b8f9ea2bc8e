// dist_internal.h -- what the row-block solvers outside dist.hip (BiCGSTAB, GMRES) need of the distributed operator.
#pragma once
#include <rccl/rccl.h>

#include <vector>

#include "mk_internal.h"

namespace mk {
struct LocalWorld;
}

struct mk_comm {
  int nranks = 1, rank = 0;
  bool local = false;
  mk::LocalWorld *world = nullptr;
  ncclComm_t coll = nullptr, p2p = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_halo = nullptr;
  // matrices created on this communicator; mk_comm_destroy() on a communicator that still has users only marks it, the
  // last matrix to go tears it down (a host language's finaliser order is not ours to choose)
  int users = 0;
  bool dead = false;
};

struct mk_dist_matrix {
  mk_comm *comm = nullptr;
  int nranks = 1, rank = 0;
  int64_t n_global = 0, row_begin = 0, row_end = 0, n_local = 0, n_ghost = 0;
  std::vector<int64_t> row_start;
  mk_matrix diag;              // n_local x n_local, local column ids
  mk_matrix offd;              // n_offd_rows x n_ghost (compressed rows)
  int *offd_rows = nullptr;    // local row of each offd row
  int n_offd_rows = 0;
  int gap0 = 0, gap1 = 0;      // the longest run of local rows [gap0, gap1) that touch no ghost (a z-slab: all but its end planes)
  int offd_lanes = 1;
  int *ghost_global = nullptr; // device, sorted global ids of the ghosts
  std::vector<int> h_ghost;
  std::vector<int> recv_cnt, recv_off, send_cnt, send_off;
  int n_send = 0;
  int *send_idx = nullptr;     // device: local rows to pack, grouped by destination
  double *sendbuf = nullptr;
  bool plan_ready = false;
  // transposed product (CGNR), built at first use: diag^T, offd^T with the columns mapped back to local rows
  // (n_ghost x n_local: row g = what this rank's rows contribute to ghost column g), its result, and what the
  // neighbours contribute to the rows of the send plan
  mk_matrix *diagT = nullptr, *offdT = nullptr;
  double *ghost_out = nullptr, *recvbuf = nullptr;
};


namespace mk {

// every rank this process drives: plan complete, local worlds whole and in rank order, one rank per process over RCCL
int dist_check_ranks(const char *who, int R, mk_dist_matrix **A);
// y[r] = alpha A x[r] + beta y[r] on every rank (collective): halo of x (x[r] holds n_local owned + n_ghost ghost entries) beside
// the diag-block product, then the off-rank block; `skip` (per rank, may be null): device flags, nonzero -> that rank's kernels
// return at once (the exchange itself always runs: it is collective)
int dist_spmv(int R, mk_dist_matrix **A, double **x, double **y, double alpha, double beta, const int *const *skip);
// y[r] = alpha A^T x[r] + beta y[r] on every rank (collective; x[r], y[r]: n_local entries).  The transpose of a row block
// is a column block: diag^T x gives the owned part, offd^T x what this rank's rows contribute to the rows of OTHER ranks;
// those go back along the exchange plan in reverse (the ghosts' owners receive and add, neighbour by neighbour in rank
// order: fixed summation order)
int dist_spmv_transpose(int R, mk_dist_matrix **A, double **x, double **y, double alpha, double beta, const int *const *skip);
// sum over all ranks of bufs[r][0 .. count), in place (RCCL all-reduce, or one kernel in a local world)
int dist_all_reduce(int R, mk_dist_matrix **A, double **bufs, int count);

}  // namespace mk
