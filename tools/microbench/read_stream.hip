// read_stream.hip -- what a hand-written streaming READ reaches on this GPU, by launch shape (context for the rooflines of the
// kernels that only read a matrix stream: DESIGN 4.1, band kernel).  Sums 2 GiB of doubles; every variant reads each byte once.
//   T      threads per block (256 / 512 / 1024)
//   BPC    blocks per CU
//   ILP    16-byte loads a thread has in flight per step
//   NT     nontemporal loads
//   ORDER  0: grid-stride (step s of block b reads tile s * blocks + b: neighbouring blocks read neighbouring tiles)
//          1: every block streams through its own contiguous 1 / blocks of the array
//          2: the blocks of an XCD (blockIdx % 8) share a contiguous eighth, tiles dealt round-robin inside it
// build: hipcc --offload-arch=gfx950 -O3 -o read_stream read_stream.hip ; run: ./read_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int T, int ILP, bool NT, int ORDER>
__global__ __launch_bounds__(T) void k_read(const d2 *__restrict__ a, size_t nvec, double *__restrict__ out) {
  const size_t tile = (size_t)T * ILP;   // vectors per tile
  const size_t ntiles = nvec / tile;
  double acc = 0.0;
  size_t first, count, stride;
  if (ORDER == 0) { first = blockIdx.x; stride = gridDim.x; count = (ntiles - first + stride - 1) / stride; }
  else if (ORDER == 1) { const size_t per = ntiles / gridDim.x; first = blockIdx.x * per; stride = 1; count = per; }
  else { const size_t per = ntiles / 8, slots = gridDim.x / 8; first = (blockIdx.x & 7) * per + (blockIdx.x >> 3); stride = slots;
         count = (per - (blockIdx.x >> 3) + slots - 1) / slots; }
  for (size_t s = 0; s < count; s++) {
    const d2 *p = a + (first + s * stride) * tile + threadIdx.x;
    d2 v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = NT ? __builtin_nontemporal_load(p + (size_t)i * T) : p[(size_t)i * T];
#pragma unroll
    for (int i = 0; i < ILP; i++) acc += v[i].x + v[i].y;
  }
  if (acc == 12345.678) out[0] = acc;   // (never true: keeps the loads alive)
}

// Two streams read side by side as the CSR kernels read val[] and col[]: per step ILP 16-byte loads from a and ILP loads of
// W bytes (8 or 16) per lane from b, every block streaming its own contiguous ranges of both (ORDER 1).  W = 16 reads b in
// tiles half as long, so that both forms move 1.5 x the bytes of a alone.  PAIR32: the 16-byte loads of a leave as two per
// lane from ADJACENT addresses (lane stride 32 bytes) instead of lane-consecutive ones.
typedef int i2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
template <int T, int ILP, int W, bool PAIR32>
__global__ __launch_bounds__(T) void k_two(const d2 *__restrict__ a, const int *__restrict__ b, size_t nvec, double *__restrict__ out) {
  const size_t tile = (size_t)T * ILP;   // 16-byte vectors of a per step
  const size_t per = nvec / tile / gridDim.x;
  double acc = 0.0;
  for (size_t s = 0; s < per; s++) {
    const size_t t0 = ((size_t)blockIdx.x * per + s) * tile;   // first vector of the tile in a; b holds 2 ints per vector of a
    d2 v[ILP];
    int c[ILP * 2];
    if (PAIR32) {
#pragma unroll
      for (int i = 0; i < ILP; i += 2) {
        const d2 *p = a + t0 + (size_t)i * T + 2 * threadIdx.x;
        v[i] = __builtin_nontemporal_load(p);
        v[i + 1] = __builtin_nontemporal_load(p + 1);
      }
    } else {
#pragma unroll
      for (int i = 0; i < ILP; i++) v[i] = __builtin_nontemporal_load(a + t0 + (size_t)i * T + threadIdx.x);
    }
    if (W == 8) {
#pragma unroll
      for (int i = 0; i < ILP; i++) {
        const i2 q = __builtin_nontemporal_load(reinterpret_cast<const i2 *>(b) + t0 + (size_t)i * T + threadIdx.x);
        c[2 * i] = q.x; c[2 * i + 1] = q.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < ILP / 2; i++) {
        const i4 q = __builtin_nontemporal_load(reinterpret_cast<const i4 *>(b) + t0 / 2 + (size_t)i * T + threadIdx.x);
        c[4 * i] = q.x; c[4 * i + 1] = q.y; c[4 * i + 2] = q.z; c[4 * i + 3] = q.w;
      }
    }
#pragma unroll
    for (int i = 0; i < ILP; i++) acc += v[i].x * c[2 * i] + v[i].y * c[2 * i + 1];
  }
  if (acc == 12345.678) out[0] = acc;
}

// ... and with the LDS traffic and the two barriers per step of the band kernel beside the streams: every step stores its
// products in LDS (16 bytes per lane and load), barrier, every lane reads GATHER doubles back from scattered addresses, barrier.
template <int T, int ILP, int GATHER>
__global__ __launch_bounds__(T) void k_two_lds(const d2 *__restrict__ a, const int *__restrict__ b, size_t nvec, double *__restrict__ out) {
  __shared__ d2 stage[T * ILP];
  const size_t tile = (size_t)T * ILP;
  const size_t per = nvec / tile / gridDim.x;
  double acc = 0.0;
  const double *sd = reinterpret_cast<const double *>(stage);
  for (size_t s = 0; s < per; s++) {
    const size_t t0 = ((size_t)blockIdx.x * per + s) * tile;
    d2 v[ILP];
    i2 c[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = __builtin_nontemporal_load(a + t0 + (size_t)i * T + threadIdx.x);
#pragma unroll
    for (int i = 0; i < ILP; i++) c[i] = __builtin_nontemporal_load(reinterpret_cast<const i2 *>(b) + t0 + (size_t)i * T + threadIdx.x);
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      d2 p;
      p.x = v[i].x * c[i].x; p.y = v[i].y * c[i].y;
      stage[i * T + threadIdx.x] = p;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GATHER; g++) acc += sd[((threadIdx.x * 37u + g * 613u + (unsigned)s * 11u) % (unsigned)(2 * T * ILP))];
    __syncthreads();
  }
  if (acc == 12345.678) out[0] = acc;
}
// ... and closer still to the band kernel: the LDS reads are GATHERS whose addresses come from the loaded columns (a ring of
// M doubles), their products go to the stage, and behind the barrier G lanes per "row" of ROWLEN entries add them up, a
// shuffle tree finishes the row, one lane per row stores y (nontemporal) -- fixed row length, everything else as in band.hip.
template <int T, int ILP, int G, int MRING>
__global__ __launch_bounds__(T) void k_band_like(const d2 *__restrict__ a, const int *__restrict__ b, size_t nvec, double *__restrict__ y, double *__restrict__ out) {
  extern __shared__ double sm[];
  double *xw = sm;                 // MRING doubles
  double *stage = sm + MRING;      // 2 T ILP doubles
  for (int i = threadIdx.x; i < MRING; i += T) xw[i] = 1.0;
  __syncthreads();
  const size_t tile = (size_t)T * ILP;
  const size_t per = nvec / tile / gridDim.x;
  constexpr int R = T / G;                       // rows per step
  constexpr int ROWLEN = 2 * ILP * G;            // entries per row (all of the tile's 2 T ILP entries)
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  double acc = 0.0;
  for (size_t s = 0; s < per; s++) {
    const size_t t0 = ((size_t)blockIdx.x * per + s) * tile;
    d2 v[ILP];
    i2 c[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = __builtin_nontemporal_load(a + t0 + (size_t)i * T + threadIdx.x);
#pragma unroll
    for (int i = 0; i < ILP; i++) c[i] = __builtin_nontemporal_load(reinterpret_cast<const i2 *>(b) + t0 + (size_t)i * T + threadIdx.x);
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      const unsigned s0 = (unsigned)(c[i].x + (int)(threadIdx.x * 37u + i * 101u)) % (unsigned)MRING;
      const unsigned s1 = (unsigned)(c[i].y + (int)(threadIdx.x * 53u + i * 211u)) % (unsigned)MRING;
      d2 p;
      p.x = v[i].x * xw[s0]; p.y = v[i].y * xw[s1];
      reinterpret_cast<d2 *>(stage)[i * T + threadIdx.x] = p;
    }
    __syncthreads();
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < 2 * ILP; t++) sum += stage[rl * ROWLEN + g + t * G];
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_down(sum, off, G);
    if (g == 0) __builtin_nontemporal_store(sum, y + ((size_t)blockIdx.x * per + s) * R + rl);
    acc += sum;
    __syncthreads();
  }
  if (acc == 12345.678) out[0] = acc;
}
template <int T, int ILP, int G, int MRING>
static void run_band_like(const d2 *a, const int *b, size_t nvec, double *y, double *out, int cus, int bpc) {
  const int blocks = cus * bpc;
  const size_t lds = ((size_t)MRING + 2 * T * ILP) * 8;
  auto kern = k_band_like<T, ILP, G, MRING>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(T), lds, 0, a, b, nvec, y, out);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(T), lds, 0, a, b, nvec, y, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("band-like (ring gathers by loaded columns, stage, row pass with %d lanes per row, y stored): T %4d bpc %d ilp %d ring %d: %7.1f us  %5.2f TB/s\n", G, T, bpc, ILP, MRING,
         ms * 100.0, (double)nvec * 24.0 / (ms * 1e-4) / 1e12);
  fflush(stdout);
}

template <int T, int ILP, int GATHER>
static void run_two_lds(const d2 *a, const int *b, size_t nvec, double *out, int cus, int bpc) {
  const int blocks = cus * bpc;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_two_lds<T, ILP, GATHER>), dim3(blocks), dim3(T), 0, 0, a, b, nvec, out);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k_two_lds<T, ILP, GATHER>), dim3(blocks), dim3(T), 0, 0, a, b, nvec, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("two streams + LDS stage + 2 barriers per step: T %4d bpc %d ilp %d, %d scattered LDS reads per lane: %7.1f us  %5.2f TB/s\n", T, bpc, ILP, GATHER,
         ms * 100.0, (double)nvec * 24.0 / (ms * 1e-4) / 1e12);
  fflush(stdout);
}

template <int T, int ILP, int W, bool PAIR32>
static void run_two(const d2 *a, const int *b, size_t nvec, double *out, int cus, int bpc) {
  const int blocks = cus * bpc;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_two<T, ILP, W, PAIR32>), dim3(blocks), dim3(T), 0, 0, a, b, nvec, out);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k_two<T, ILP, W, PAIR32>), dim3(blocks), dim3(T), 0, 0, a, b, nvec, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("two streams: T %4d bpc %d ilp %d, b in %2d-byte loads, a %s: %7.1f us  %5.2f TB/s\n", T, bpc, ILP, W, PAIR32 ? "in adjacent pairs (lane stride 32)" : "lane-consecutive",
         ms * 100.0, (double)nvec * 24.0 / (ms * 1e-4) / 1e12);
  fflush(stdout);
}

template <int T, int ILP, bool NT, int ORDER>
static void run(const d2 *a, size_t nvec, double *out, int cus, int bpc) {
  const int blocks = cus * bpc;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_read<T, ILP, NT, ORDER>), dim3(blocks), dim3(T), 0, 0, a, nvec, out);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k_read<T, ILP, NT, ORDER>), dim3(blocks), dim3(T), 0, 0, a, nvec, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("T %4d bpc %d ilp %d nt %d order %d: %7.1f us  %5.2f TB/s\n", T, bpc, ILP, (int)NT, ORDER, ms * 100.0, (double)nvec * 16.0 / (ms * 1e-4) / 1e12);
  fflush(stdout);
}

int main() {
  int dev = 0, cus = 0;
  CK(hipSetDevice(dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t nvec = (size_t)1 << 27;   // 2 GiB
  d2 *a; double *out;
  CK(hipMalloc(&a, nvec * 16)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, nvec * 16));
  printf("%d CUs, 2 GiB\n", cus);
  run<256, 4, false, 0>(a, nvec, out, cus, 8);
  run<256, 4, true, 0>(a, nvec, out, cus, 8);
  run<256, 8, true, 0>(a, nvec, out, cus, 8);
  run<256, 4, true, 0>(a, nvec, out, cus, 4);
  run<512, 4, true, 0>(a, nvec, out, cus, 2);
  run<512, 8, true, 0>(a, nvec, out, cus, 2);
  run<512, 4, true, 0>(a, nvec, out, cus, 4);
  run<1024, 4, true, 0>(a, nvec, out, cus, 2);
  run<512, 4, true, 1>(a, nvec, out, cus, 2);
  run<512, 8, true, 1>(a, nvec, out, cus, 2);
  run<512, 4, false, 1>(a, nvec, out, cus, 2);
  run<256, 4, true, 1>(a, nvec, out, cus, 8);
  run<512, 4, true, 2>(a, nvec, out, cus, 2);
  run<512, 8, true, 2>(a, nvec, out, cus, 2);
  run<256, 4, true, 2>(a, nvec, out, cus, 8);
  run<256, 2, true, 2>(a, nvec, out, cus, 8);
  run<256, 1, true, 0>(a, nvec, out, cus, 8);
  int *bb;
  CK(hipMalloc(&bb, nvec * 8));
  CK(hipMemset(bb, 0, nvec * 8));
  run_two<512, 4, 8, false>(a, bb, nvec, out, cus, 2);
  run_two<512, 4, 16, false>(a, bb, nvec, out, cus, 2);
  run_two<512, 4, 16, true>(a, bb, nvec, out, cus, 2);
  run_two<512, 4, 8, true>(a, bb, nvec, out, cus, 2);
  run_two_lds<512, 4, 0>(a, bb, nvec, out, cus, 2);
  run_two_lds<512, 4, 8>(a, bb, nvec, out, cus, 2);
  run_two_lds<512, 4, 16>(a, bb, nvec, out, cus, 2);
  run_two_lds<256, 4, 8>(a, bb, nvec, out, cus, 4);
  double *yy;
  CK(hipMalloc(&yy, nvec / 4 * 8 + 4096));
  run_band_like<512, 4, 8, 6072>(a, bb, nvec, yy, out, cus, 2);
  run_band_like<512, 3, 8, 6072>(a, bb, nvec, yy, out, cus, 2);
  run_band_like<512, 2, 8, 6072>(a, bb, nvec, yy, out, cus, 2);
  // ... and with both streams starting 40 / 20 bytes off a cache line, as a chunk of a CSR matrix does (the last tile is left out)
  printf("misaligned by 40 / 20 bytes:\n");
  run_two<512, 4, 8, false>(reinterpret_cast<const d2 *>(reinterpret_cast<const char *>(a) + 40), reinterpret_cast<const int *>(reinterpret_cast<const char *>(bb) + 20),
                            nvec - (size_t)512 * 4 * 512, out, cus, 2);
  run_two<256, 4, 8, false>(a, bb, nvec, out, cus, 8);
  run_two<256, 4, 16, false>(a, bb, nvec, out, cus, 8);
  return 0;
}
