// fetchcal.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the normal-mode
// kernels: every kernel below touches EVERY byte of a 1.2 GB buffer exactly once (no reuse, far beyond the 256 MiB
// Infinity Cache), so known bytes / counter bytes is the correction factor of that shape ("weak" item 8).
//   stream16  : contiguous, 16 B per lane            (row kernel staging: MI355X_MICROARCH.md says FETCH_SIZE = 1/2)
//   stream8   : contiguous,  8 B per lane
//   seg1k     : a wave reads one 1024-byte row segment (16 B per lane) of a [rows x cols] matrix, consecutive waves of a
//               workgroup take consecutive ROWS of the same column panel -- the two-column / tiled panel sweep
//   seg512    : the same with 512-byte segments, 8 B per lane -- the one-column panel sweep
//   seg1k_rmw : seg1k as read-modify-write of a second matrix (the result round trip of the sweep)
// Build: hipcc -O3 --offload-arch=gfx950 fetchcal.hip -o fetchcal ; run under rocprofv3 --pmc FETCH_SIZE (WRITE_SIZE ...)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void __launch_bounds__(512) stream16(const double2* __restrict__ p, size_t n, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i].x + p[i].y;
  if (s == 1.2345e300) out[0] = s;
}
__global__ void __launch_bounds__(512) stream8(const double* __restrict__ p, size_t n, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 1.2345e300) out[0] = s;
}
// panel sweep shape: panel = blockIdx % npanels, rows strided over the workgroups of a panel; one row segment per wave
template <int LANE_DOUBLES>  // 2: 1024-byte segments (16 B / lane), 1: 512-byte segments (8 B / lane)
__global__ void __launch_bounds__(512) seg(const double* __restrict__ v, double* __restrict__ hv, int rows, int cols, int rmw,
                                           double* out) {
  constexpr int W = 64 * LANE_DOUBLES;
  const int npanels = cols / W, panel = blockIdx.x % npanels, chunk = blockIdx.x / npanels, nchunks = gridDim.x / npanels;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0;
  for (int r = chunk * 8 + wave; r < rows; r += nchunks * 8) {
    const size_t o = (size_t)r * cols + (size_t)panel * W + (size_t)lane * LANE_DOUBLES;
    if (LANE_DOUBLES == 2) {
      double2 x = *reinterpret_cast<const double2*>(v + o);
      if (rmw) {
        double2 y = *reinterpret_cast<const double2*>(hv + o);
        y.x += x.x; y.y += x.y;
        *reinterpret_cast<double2*>(hv + o) = y;
      } else s += x.x + x.y;
    } else {
      double x = v[o];
      if (rmw) hv[o] += x; else s += x;
    }
  }
  if (s == 1.2345e300) out[0] = s;
}

int main() {
  const int rows = 12288, cols = 12288;   // 1.208 GB per matrix, cols a multiple of 128
  const size_t n = (size_t)rows * cols;
  double *v, *hv, *out; CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&hv, n * 8)); CK(hipMalloc(&out, 8));
  CK(hipMemset(v, 0, n * 8)); CK(hipMemset(hv, 0, n * 8)); CK(hipDeviceSynchronize());
  printf("bytes per kernel: %.0f (read shapes), rmw: read 2x, write 1x that\n", (double)n * 8);
  for (int rep = 0; rep < 2; rep++) {
    stream16<<<2048, 512>>>(reinterpret_cast<const double2*>(v), n / 2, out);
    stream8<<<2048, 512>>>(v, n, out);
    seg<2><<<(cols / 128) * 32, 512>>>(v, hv, rows, cols, 0, out);
    seg<1><<<(cols / 64) * 16, 512>>>(v, hv, rows, cols, 0, out);
    seg<2><<<(cols / 128) * 32, 512>>>(v, hv, rows, cols, 1, out);
    CK(hipDeviceSynchronize());
  }
  return 0;
}
