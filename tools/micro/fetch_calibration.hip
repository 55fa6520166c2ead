// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS code's access patterns: kernels that move a
// known number of bytes, one kernel name per pattern, each over a footprint far above the Infinity Cache (1 GiB) and one
// below it (96 MiB, swept repeatedly -- the counters sit on the fabric side of L2 and count Infinity-Cache hits, which
// this pair of footprints shows).  MI355X_MICROARCH.md states FETCH_SIZE = 1/2 of the bytes for a 16-byte-per-lane
// streaming read and says "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern before trusting an absolute".  Patterns (what the E0 kernels issue):
//   read4 / read8 / read16   coalesced 4-, 8-, 16-byte loads per lane (cw / landmark-slot words, lmrec entries, uv rows)
//   gather32                 32-byte records at random 32-byte-aligned positions (cm_gram's landmark gathers, q4c)
//   gather168                168 contiguous bytes per lane at random 192-byte-strided records (e0_ck's camera records)
//   write8 / write16         coalesced 8- / 16-byte stores per lane (partial records, mirrors)
//   scatter32                32-byte stores at random positions (q of cold observations)
// Build: hipcc --offload-arch=gfx950 -O3 -o build/fetch_calibration tools/micro/fetch_calibration.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -- build/fetch_calibration
//        rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -- build/fetch_calibration     (separate passes)
// and tools/micro/fetch_calibration_table.py out/f out/w prints bytes per pattern against the counters.
// The program prints the byte count of every launch itself ("pattern footprint bytes").
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                  \
  do {                                                                            \
    hipError_t r_ = (e);                                                          \
    if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); std::exit(1); } \
  } while (0)

template <class T>
__global__ __launch_bounds__(256) void read_big(const T* in, size_t n, double* sink) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = in[i];
    acc += *reinterpret_cast<const float*>(&v);
  }
  if (acc == 123.456) sink[0] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void read_small(const T* in, size_t n, double* sink) {  // same code, other name
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = in[i];
    acc += *reinterpret_cast<const float*>(&v);
  }
  if (acc == 123.456) sink[0] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void write_big(T* out, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
template <class T>
__global__ __launch_bounds__(256) void write_small(T* out, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void gather32_big(const double4* in, const uint32_t* idx, size_t n, double* sink) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[idx[i]].x;
  if (acc == 123.456) sink[0] = acc;
}
__global__ __launch_bounds__(256) void gather32_small(const double4* in, const uint32_t* idx, size_t n, double* sink) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[idx[i]].x;
  if (acc == 123.456) sink[0] = acc;
}
__global__ __launch_bounds__(256) void gather168_small(const double2* in, const uint32_t* idx, size_t n, double* sink) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double2* r = in + (size_t)idx[i] * 12;  // 192-byte records, 168 bytes read: ten 16-byte loads + one 8-byte
#pragma unroll
    for (int j = 0; j < 10; ++j) acc += r[j].x;
    acc += reinterpret_cast<const double*>(r + 10)[0];
  }
  if (acc == 123.456) sink[0] = acc;
}
__global__ __launch_bounds__(256) void scatter32_big(double4* out, const uint32_t* idx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[idx[i]] = make_double4(1, 2, 3, 4);
}

int main() {
  const size_t BIG = (size_t)1 << 30, SMALL = (size_t)96 << 20;
  char* buf;
  double* sink;
  CHECK(hipMalloc((void**)&buf, BIG));
  CHECK(hipMalloc((void**)&sink, 64));
  CHECK(hipMemset(buf, 0, BIG));
  // random 32-byte-record indices over the big and the small footprint; camera-record indices over 1 778 records
  const size_t NG = (size_t)8 << 20;
  std::vector<uint32_t> hb(NG), hs(NG), hc(NG);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (size_t i = 0; i < NG; ++i) {
    hb[i] = (uint32_t)(rnd() % (BIG / 32));
    hs[i] = (uint32_t)(rnd() % (SMALL / 32));
    hc[i] = (uint32_t)(rnd() % 1778);
  }
  uint32_t *ib, *is, *ic;
  CHECK(hipMalloc((void**)&ib, NG * 4)); CHECK(hipMalloc((void**)&is, NG * 4)); CHECK(hipMalloc((void**)&ic, NG * 4));
  CHECK(hipMemcpy(ib, hb.data(), NG * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(is, hs.data(), NG * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(ic, hc.data(), NG * 4, hipMemcpyHostToDevice));
  const int grid = 256 * 8;
  const int REP_SMALL = 8;
  std::printf("pattern footprint bytes_per_launch\n");
#define RUN_READ(T, name)                                                                                         \
  hipLaunchKernelGGL(read_big<T>, dim3(grid), dim3(256), 0, 0, (const T*)buf, BIG / sizeof(T), sink);            \
  std::printf("%s big %zu\n", name, BIG);                                                                          \
  for (int r = 0; r < REP_SMALL; ++r)                                                                             \
    hipLaunchKernelGGL(read_small<T>, dim3(grid), dim3(256), 0, 0, (const T*)buf, SMALL / sizeof(T), sink);       \
  std::printf("%s small %zu\n", name, SMALL);
  RUN_READ(float, "read4")
  RUN_READ(double, "read8")
  RUN_READ(double2, "read16")
#define RUN_WRITE(T, name, v)                                                                                     \
  hipLaunchKernelGGL(write_big<T>, dim3(grid), dim3(256), 0, 0, (T*)buf, BIG / sizeof(T), v);                     \
  std::printf("%s big %zu\n", name, BIG);                                                                          \
  for (int r = 0; r < REP_SMALL; ++r) hipLaunchKernelGGL(write_small<T>, dim3(grid), dim3(256), 0, 0, (T*)buf, SMALL / sizeof(T), v); \
  std::printf("%s small %zu\n", name, SMALL);
  RUN_WRITE(double, "write8", 1.0)
  RUN_WRITE(double2, "write16", make_double2(1, 2))
  hipLaunchKernelGGL(gather32_big, dim3(grid), dim3(256), 0, 0, (const double4*)buf, ib, NG, sink);
  std::printf("gather32 big %zu (+ %zu index bytes)\n", NG * 32, NG * 4);
  for (int r = 0; r < REP_SMALL; ++r) hipLaunchKernelGGL(gather32_small, dim3(grid), dim3(256), 0, 0, (const double4*)buf, is, NG, sink);
  std::printf("gather32 small %zu (+ %zu index bytes)\n", NG * 32, NG * 4);
  for (int r = 0; r < REP_SMALL; ++r) hipLaunchKernelGGL(gather168_small, dim3(grid), dim3(256), 0, 0, (const double2*)buf, ic, NG / 8, sink);
  std::printf("gather168 small %zu (+ %zu index bytes; 341 KB of records: L2-resident)\n", NG / 8 * 168, NG / 8 * 4);
  hipLaunchKernelGGL(scatter32_big, dim3(grid), dim3(256), 0, 0, (double4*)buf, ib, NG);
  std::printf("scatter32 big %zu (+ %zu index bytes)\n", NG * 32, NG * 4);
  CHECK(hipDeviceSynchronize());
  return 0;
}
