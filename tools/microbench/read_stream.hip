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
  return 0;
}
