// Device check of the fixed-point conversions of e0_ck<DET> (povar_kernels_ck.hpp: ck_fix / ck_unfix / ck_xp):
// hipcc --offload-arch=gfx950 -O3 -I../../povar_amd/csrc fix_check.hip -o /tmp/fix_check && /tmp/fix_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "povar_kernels_ck.hpp"
using namespace povar;
__global__ void k_fix(const double* x, const int* e, unsigned long long* f, double* back, int* xp, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  f[i] = ck_fix(x[i], e[i]);
  back[i] = ck_unfix(f[i], e[i]);
  xp[i] = ck_xp(x[i]);
}
__global__ void k_sum(const double* x, int e, int n, double* out) {
  __shared__ double acc[1];
  if (threadIdx.x == 0) acc[0] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) ck_add_fix(acc, x[i], e);
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ck_unfix(__double_as_longlong(acc[0]), e);
}
int main() {
  const int n = 1 << 16;
  std::mt19937_64 g(1);
  std::normal_distribution<double> nd;
  std::uniform_real_distribution<double> ud(-20, 3);
  std::vector<double> x(n);
  std::vector<int> e(n);
  for (int i = 0; i < n; ++i) {
    x[i] = nd(g) * std::pow(10.0, ud(g));
    int xe; std::frexp(x[i], &xe);
    e[i] = 61 - xe - (int)(g() % 20);
  }
  double *dx, *db, *ds; int *de, *dp; unsigned long long* df;
  hipMalloc(&dx, n * 8); hipMalloc(&db, n * 8); hipMalloc(&de, n * 4); hipMalloc(&dp, n * 4); hipMalloc(&df, n * 8); hipMalloc(&ds, 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(de, e.data(), n * 4, hipMemcpyHostToDevice);
  k_fix<<<n / 256, 256>>>(dx, de, df, db, dp, n);
  std::vector<double> back(n); std::vector<int> xp(n); std::vector<unsigned long long> f(n);
  hipMemcpy(back.data(), db, n * 8, hipMemcpyDeviceToHost); hipMemcpy(xp.data(), dp, n * 4, hipMemcpyDeviceToHost); hipMemcpy(f.data(), df, n * 8, hipMemcpyDeviceToHost);
  int bad = 0, badx = 0;
  for (int i = 0; i < n; ++i) {
    const double want = std::nearbyint(std::ldexp(x[i], e[i]));
    if (std::fabs((double)(long long)f[i] - want) > 1.0 || std::fabs(back[i] - x[i]) > std::ldexp(1.0, -e[i])) { if (bad++ < 5) std::printf("bad %g e %d fix %lld want %.0f back %g\n", x[i], e[i], (long long)f[i], want, back[i]); }
    int xe; std::frexp(x[i], &xe);
    if (xe != xp[i]) { if (badx++ < 5) std::printf("xp %g: %d want %d\n", x[i], xp[i], xe); }
  }
  std::vector<double> y(4096);
  double ref = 0;
  for (auto& v : y) { v = nd(g); }
  for (auto& v : y) ref += v;
  hipMemcpy(dx, y.data(), 4096 * 8, hipMemcpyHostToDevice);
  k_sum<<<1, 1024>>>(dx, 61 - 14, 4096, ds);
  double s; hipMemcpy(&s, ds, 8, hipMemcpyDeviceToHost);
  std::printf("conversions bad %d of %d, frexp bad %d; LDS sum %.17g ref %.17g diff %.3g\n", bad, n, badx, s, ref, s - ref);
  return bad || badx;
}
