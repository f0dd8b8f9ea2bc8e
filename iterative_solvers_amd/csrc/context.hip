// context.hip -- library life cycle, error text, vectors, timers, host-side partitioner.
#include <map>
#include <mutex>
#include <unordered_map>
#include "mk_internal.h"

#include <algorithm>

namespace mk {

static thread_local char g_err[1024] = "no error";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ------------------------------------------------------------------ caching allocator (see mk_internal.h)
namespace {
struct AllocCache {
  std::mutex mu;
  std::unordered_map<void *, size_t> live;        // every block handed out -> its size
  std::multimap<size_t, void *> idle;             // blocks waiting for reuse, by size
  size_t idle_bytes = 0, cap = 0;
  long long alloc_calls = 0, free_calls = 0;
  bool enabled = false;
};
AllocCache g_cache;
constexpr size_t kCacheMinBlock = (size_t)1 << 20;
}  // namespace

void cache_release_all() {
  std::lock_guard<std::mutex> lk(g_cache.mu);
  for (auto &b : g_cache.idle) (void)hipFree(b.second);
  g_cache.idle.clear();
  g_cache.idle_bytes = 0;
}

void *cached_alloc(size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  void *p = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    g_cache.alloc_calls++;
    if (g_cache.enabled && bytes >= kCacheMinBlock) {
      auto it = g_cache.idle.lower_bound(bytes);
      if (it != g_cache.idle.end() && it->first <= bytes + bytes / 8) {
        p = it->second;
        g_cache.idle_bytes -= it->first;
        g_cache.live[p] = it->first;
        g_cache.idle.erase(it);
        return p;
      }
    }
  }
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {   // give back what the cache holds and try once more
    (void)hipGetLastError();
    cache_release_all();
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  std::lock_guard<std::mutex> lk(g_cache.mu);
  g_cache.live[p] = bytes;
  return p;
}

// Streams other than ctx().stream that library work runs on (RCCL communication streams, the local world's side stream).
// A block that goes back to the free list may be handed to a new user at once, whose work is ordered behind the old
// user's only on ctx().stream; so before a block is kept for reuse every side stream must have drained.  The solvers
// join their side streams back into ctx().stream by events before they return, so the query below normally finds
// them idle (a few microseconds per free, nothing on the single-GPU path: no side stream exists there).
static std::vector<hipStream_t> g_side_streams;
void register_side_stream(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_cache.mu);
  g_side_streams.push_back(s);
}
void unregister_side_stream(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_cache.mu);
  for (size_t i = 0; i < g_side_streams.size(); i++)
    if (g_side_streams[i] == s) { g_side_streams.erase(g_side_streams.begin() + i); break; }
}

void cached_free(void *p) {
  size_t bytes = 0;
  {
    std::vector<hipStream_t> side;
    {
      std::lock_guard<std::mutex> lk(g_cache.mu);
      if (g_cache.enabled) side = g_side_streams;
    }
    for (hipStream_t s : side)
      if (hipStreamQuery(s) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); }
  }
  {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    g_cache.free_calls++;
    auto it = g_cache.live.find(p);
    if (it != g_cache.live.end()) {
      bytes = it->second;
      g_cache.live.erase(it);
    }
    if (g_cache.enabled && bytes >= kCacheMinBlock && g_cache.idle_bytes + bytes <= g_cache.cap) {
      g_cache.idle.emplace(bytes, p);
      g_cache.idle_bytes += bytes;
      return;
    }
  }
  (void)hipFree(p);
}

int &dry_launch() {
  static int flag = 0;
  return flag;
}

Context &ctx() {
  static Context c;
  return c;
}

__global__ void fill_kernel(double *p, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

}  // namespace mk

using namespace mk;

extern "C" {

const char *mk_last_error_string(void) { return g_err; }

int mk_init(int device_id) {
  Context &c = ctx();
  if (c.ready) {
    if (c.device == device_id) return MK_OK;
    return fail(MK_ERR_INVALID, "mk_init: already initialised on device %d", c.device);
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(MK_ERR_NO_DEVICE, "mk_init: no HIP device visible (%s)", hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev)
    return fail(MK_ERR_INVALID, "mk_init: device %d out of range (0..%d)", device_id, ndev - 1);
  MK_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  MK_HIP(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MK_ERR_NO_DEVICE, "mk_init: device %d is %s; this library is built for gfx950 only",
                device_id, prop.gcnArchName);
  c.device = device_id;
  c.num_cus = prop.multiProcessorCount;
  MK_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  MK_HIP(hipEventCreate(&c.ev0));
  MK_HIP(hipEventCreate(&c.ev1));
  {
    size_t free_b = 0, total_b = 0;
    MK_HIP(hipMemGetInfo(&free_b, &total_b));
    const char *cap_mb = getenv("MK_ALLOC_CACHE_MB");
    std::lock_guard<std::mutex> lk(g_cache.mu);
    g_cache.cap = cap_mb ? (size_t)atoll(cap_mb) << 20 : total_b / 4;
    g_cache.enabled = !(getenv("MK_NO_ALLOC_CACHE") && atoi(getenv("MK_NO_ALLOC_CACHE")));
  }
  c.pinned_bytes = 1 << 16;
  MK_HIP(hipHostMalloc(&c.pinned, c.pinned_bytes, hipHostMallocDefault));
  MK_TRY(dev_alloc(&c.slab, (size_t)kMaxParts * kSlabStride));
  MK_TRY(dev_alloc(&c.scalars, 8));
  MK_TRY(init_static_scratch(device_id));   // device_id is current: every GPU has its own copy of the array
  c.ready = true;
  return MK_OK;
}

int mk_finalize(void) {
  Context &c = ctx();
  if (!c.ready) return MK_OK;
  (void)hipStreamSynchronize(c.stream);
  (void)hipEventDestroy(c.ev0);
  (void)hipEventDestroy(c.ev1);
  (void)hipHostFree(c.pinned);
  dev_free(c.slab);
  dev_free(c.scalars);
  {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    g_cache.enabled = false;   // objects destroyed after this point free their memory directly
  }
  cache_release_all();
  (void)hipStreamDestroy(c.stream);
  c = Context();
  return MK_OK;
}

void *mk_get_stream(void) { return ctx().ready ? (void *)ctx().stream : nullptr; }

int mk_get_device_pci_bus_id(int device_id, char *out, int capacity) {
  if (!out || capacity < 16) return fail(MK_ERR_INVALID, "mk_get_device_pci_bus_id: buffer of at least 16 bytes needed");
  if (device_id < 0) {
    if (!ctx().ready) return fail(MK_ERR_NOT_INIT, "mk_init() not called");
    device_id = ctx().device;
  }
  MK_HIP(hipDeviceGetPCIBusId(out, capacity, device_id));
  return MK_OK;
}

int mk_memory_info(long long *in_use_bytes, long long *idle_bytes) {
  MK_REQUIRE_INIT();
  std::lock_guard<std::mutex> lk(g_cache.mu);
  long long live = 0;
  for (auto &b : g_cache.live) live += (long long)b.second;
  if (in_use_bytes) *in_use_bytes = live;
  if (idle_bytes) *idle_bytes = (long long)g_cache.idle_bytes;
  return MK_OK;
}

int mk_memory_calls(long long *alloc_calls, long long *free_calls) {
  MK_REQUIRE_INIT();
  std::lock_guard<std::mutex> lk(g_cache.mu);
  if (alloc_calls) *alloc_calls = g_cache.alloc_calls;
  if (free_calls) *free_calls = g_cache.free_calls;
  return MK_OK;
}

int mk_memory_trim(void) {
  MK_REQUIRE_INIT();
  MK_HIP(hipStreamSynchronize(ctx().stream));
  cache_release_all();
  return MK_OK;
}

int mk_synchronize(void) {
  MK_REQUIRE_INIT();
  MK_HIP(hipStreamSynchronize(ctx().stream));
  return MK_OK;
}

int mk_timer_start(void) {
  MK_REQUIRE_INIT();
  MK_HIP(hipEventRecord(ctx().ev0, ctx().stream));
  return MK_OK;
}

int mk_timer_stop(float *elapsed_ms) {
  MK_REQUIRE_INIT();
  if (!elapsed_ms) return fail(MK_ERR_INVALID, "mk_timer_stop: null output");
  MK_HIP(hipEventRecord(ctx().ev1, ctx().stream));
  MK_HIP(hipEventSynchronize(ctx().ev1));
  MK_HIP(hipEventElapsedTime(elapsed_ms, ctx().ev0, ctx().ev1));
  return MK_OK;
}

// ------------------------------------------------------------------------- vectors

int mk_vector_create(int64_t n, mk_vector **v) {
  MK_REQUIRE_INIT();
  if (!v || n < 0) return fail(MK_ERR_INVALID, "mk_vector_create: bad arguments");
  mk_vector *p = new mk_vector();
  p->n = n;
  p->cap = n;
  int rc = dev_alloc(&p->d, (size_t)n);
  if (rc != MK_OK) {
    delete p;
    return rc;
  }
  *v = p;
  return MK_OK;
}

int mk_vector_set(mk_vector *v, const double *host_values) {
  MK_REQUIRE_INIT();
  if (!v || (!host_values && v->n)) return fail(MK_ERR_INVALID, "mk_vector_set: null argument");
  if (v->n) {
    MK_HIP(hipMemcpyAsync(v->d, host_values, sizeof(double) * v->n, hipMemcpyHostToDevice, ctx().stream));
    MK_HIP(hipStreamSynchronize(ctx().stream));
  }
  return MK_OK;
}

int mk_vector_fill(mk_vector *v, double value) {
  MK_REQUIRE_INIT();
  if (!v) return fail(MK_ERR_INVALID, "mk_vector_fill: null vector");
  if (v->n) {
    int blocks = (int)((v->n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, ctx().stream, v->d, v->n, value);
    MK_HIP(hipGetLastError());
  }
  return MK_OK;
}

int mk_vector_get(const mk_vector *v, double *host_values) {
  MK_REQUIRE_INIT();
  if (!v || (!host_values && v->n)) return fail(MK_ERR_INVALID, "mk_vector_get: null argument");
  if (v->n) {
    MK_HIP(hipMemcpyAsync(host_values, v->d, sizeof(double) * v->n, hipMemcpyDeviceToHost, ctx().stream));
    MK_HIP(hipStreamSynchronize(ctx().stream));
  }
  return MK_OK;
}

int mk_vector_destroy(mk_vector *v) {
  if (!v) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  dev_free(v->d);
  delete v;
  return MK_OK;
}

int64_t mk_vector_size(const mk_vector *v) { return v ? v->n : -1; }

// --------------------------------------------------------------------- partitioner
// solvers.c:214-231 walks the row-sorted COO: start at nnz/np, run to the end of the row
// that index falls in, record, step by nnz/np again, force the last part to the end.  Here
// the same walk runs on the row-pointer array with exclusive ends and 64-bit offsets, and
// rows without entries between two parts stay with the earlier part.
int mk_partition_rows(int64_t nrows, const int64_t *rowptr, int nparts, int64_t *row_start) {
  if (!rowptr || !row_start || nparts < 1 || nrows < 0)
    return fail(MK_ERR_INVALID, "mk_partition_rows: bad arguments");
  const int64_t nnz = rowptr[nrows];
  const int64_t step = nnz / nparts;
  int64_t cursor = step;
  row_start[0] = 0;
  for (int part = 0; part < nparts; part++) {
    int64_t end_row;
    if (part == nparts - 1 || cursor >= nnz) {
      end_row = nrows;
    } else {
      int64_t lo = 0, hi = nrows;  // largest row with rowptr[row] <= cursor
      while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (rowptr[mid] <= cursor) lo = mid; else hi = mid;
      }
      end_row = lo + 1;
      cursor = rowptr[end_row];
      while (end_row < nrows && rowptr[end_row + 1] == rowptr[end_row]) end_row++;
    }
    if (end_row < row_start[part]) end_row = row_start[part];
    row_start[part + 1] = end_row;
    cursor += step;
  }
  row_start[nparts] = nrows;
  return MK_OK;
}

// ------------------------------------------------------------- exchange planning (host only)
// The host half of what HYPRE's ParCSR communication package holds (behind HYPRE_IJMatrixAssemble, solvers.c:362):
// which x entries a rank needs from which owner, and what it has to send.  mk_dist_matrix_create runs exactly these
// functions (dist.hip split_block / build_plan_*); they need no GPU, so the multi-process CPU test drives them too.

int mk_halo_check_row_start(int nranks, const int64_t *row_start, int64_t n_global) {
  if (!row_start || nranks < 1) return fail(MK_ERR_INVALID, "row_start: null or no ranks");
  if (n_global > 0x7fffffffLL) return fail(MK_ERR_UNSUPPORTED, "more than 2^31-1 global columns");
  if (row_start[0] != 0 || row_start[nranks] != n_global) return fail(MK_ERR_INVALID, "row_start must run from 0 to n_global");
  for (int q = 0; q < nranks; q++)
    if (row_start[q + 1] < row_start[q]) return fail(MK_ERR_INVALID, "row_start decreases");
  return MK_OK;
}

// ghosts = sorted, distinct global column ids outside [row_begin, row_end) that the block refers to
int mk_halo_ghost_columns(int64_t nrows, const int *rowptr, const int *col, int64_t row_begin, int64_t row_end, int64_t n_global,
                          int *ghosts, int64_t capacity, int64_t *n_ghost) {
  if (!rowptr || !n_ghost || nrows < 0 || (rowptr[nrows] > 0 && !col)) return fail(MK_ERR_INVALID, "mk_halo_ghost_columns: bad argument");
  std::vector<int> g;
  for (int64_t k = 0; k < rowptr[nrows]; k++) {
    const int c = col[k];
    if (c < 0 || c >= n_global) return fail(MK_ERR_INVALID, "mk_halo_ghost_columns: column %d outside [0,%lld)", c, (long long)n_global);
    if (c < row_begin || c >= row_end) g.push_back(c);
  }
  std::sort(g.begin(), g.end());
  g.erase(std::unique(g.begin(), g.end()), g.end());
  *n_ghost = (int64_t)g.size();
  if (ghosts) {
    if ((int64_t)g.size() > capacity) return fail(MK_ERR_INVALID, "mk_halo_ghost_columns: %zu ghosts, room for %lld", g.size(), (long long)capacity);
    std::copy(g.begin(), g.end(), ghosts);
  }
  return MK_OK;
}

// owners are contiguous row blocks and the ghosts are sorted: the ghosts of owner q are one contiguous run
int mk_halo_plan_recv(int nranks, int rank, const int64_t *row_start, const int *ghosts, int64_t n_ghost, int *recv_cnt, int *recv_off) {
  if (!row_start || !recv_cnt || !recv_off || nranks < 1 || rank < 0 || rank >= nranks || n_ghost < 0 || (n_ghost && !ghosts))
    return fail(MK_ERR_INVALID, "mk_halo_plan_recv: bad argument");
  recv_off[0] = 0;
  for (int q = 0; q < nranks; q++) {
    const int lo_v = (int)std::min<int64_t>(row_start[q], 0x7fffffff), hi_v = (int)std::min<int64_t>(row_start[q + 1], 0x7fffffff);
    const int *lo = std::lower_bound(ghosts, ghosts + n_ghost, lo_v);
    const int *hi = std::lower_bound(ghosts, ghosts + n_ghost, hi_v);
    recv_cnt[q] = (int)(hi - lo);
    recv_off[q + 1] = recv_off[q] + recv_cnt[q];
  }
  if (recv_cnt[rank] != 0) return fail(MK_ERR_INVALID, "exchange plan: own rows listed as ghosts");
  if (recv_off[nranks] != n_ghost) return fail(MK_ERR_INVALID, "exchange plan: %lld ghosts, %d of them inside the row ranges", (long long)n_ghost, recv_off[nranks]);
  return MK_OK;
}

// recv_cnt_all[q * nranks + r] = what rank q receives from rank r (the all-gathered recv_cnt rows)
int mk_halo_plan_send(int nranks, int rank, const int *recv_cnt_all, int *send_cnt, int *send_off) {
  if (!recv_cnt_all || !send_cnt || !send_off || nranks < 1 || rank < 0 || rank >= nranks) return fail(MK_ERR_INVALID, "mk_halo_plan_send: bad argument");
  send_off[0] = 0;
  for (int q = 0; q < nranks; q++) {
    send_cnt[q] = recv_cnt_all[(size_t)q * nranks + rank];
    if (send_cnt[q] < 0) return fail(MK_ERR_INVALID, "exchange plan: negative count from rank %d", q);
    send_off[q + 1] = send_off[q] + send_cnt[q];
  }
  if (send_cnt[rank] != 0) return fail(MK_ERR_INVALID, "exchange plan: a rank sends to itself");
  return MK_OK;
}

}  // extern "C"
