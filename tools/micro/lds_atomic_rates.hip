// Microbenchmark: throughput of LDS atomic adds on gfx950 (one 1024-thread workgroup per CU, every lane adding to
// its own address: bank-conflict free), cycles per wave-instruction per CU.  Build:
//   hipcc --offload-arch=gfx950 -O3 -o build/lds_atomic_rates tools/micro/lds_atomic_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, int iters, int stride) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double* p = lds + ((threadIdx.x >> 6) * 64 + lane * stride % 64) + (threadIdx.x >> 6) * 0;
  double v = 1.0 + threadIdx.x;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      double* q = p + u * 1024;
      if (KIND == 0) __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 1) __hip_atomic_fetch_add((unsigned long long*)q, (unsigned long long)threadIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 2) __hip_atomic_fetch_add((float*)q, (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 3) __hip_atomic_fetch_add((unsigned*)q, threadIdx.x + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 4) { volatile double* vq = q; *vq = v; }           // plain ds_write_b64
      if (KIND == 5) { volatile double* vq = q; v += *vq; }          // plain ds_read_b64
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = lds[threadIdx.x] + v;
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
  const char* names[] = {"ds_add_f64", "ds_add_u64", "ds_add_f32", "ds_add_u32", "ds_write_b64", "ds_read_b64"};
  void (*ks[])(double*, long long*, int, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>};
  const int iters = 2000;
  for (int stride : {1, 2}) {
    for (int kind = 0; kind < 6; ++kind) {
      hipFuncSetAttribute((const void*)ks[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
      hipLaunchKernelGGL(ks[kind], dim3(256), dim3(1024), 16384 * 8, 0, out, cyc, iters, stride);
      hipDeviceSynchronize();
      long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
      // 16 waves x iters x 12 wave-instructions per CU
      printf("%-13s lane stride %d: %.2f cycles (clock64 ticks) per wave-instruction per CU\n", names[kind], stride, avg / (16.0 * iters * 12));
    }
  }
  return 0;
}
