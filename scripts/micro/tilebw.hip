// tilebw.hip -- micro-benchmark: bandwidth of a 2D-tiled sweep of a [rows x cols] fp64 matrix where a workgroup
// reads (and optionally rewrites) a tile of R scattered rows x SEG bytes (the access pattern an LDS-tiled
// down-term kernel would have): how much HBM efficiency do short row segments cost?
// Build: hipcc -O3 --offload-arch=gfx950 tilebw.hip -o tilebw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

// tile t: column block cb = t % ncb, row group rg = t / ncb; rows of a group: rg + k * ngroups (scattered)
template <int SEGD>  // doubles per row segment
__global__ void __launch_bounds__(512) tile_rw(const double* __restrict__ v, double* __restrict__ hv, int rows, int cols,
                                               int R, int rmw, double* out) {
  const int ncb = cols / SEGD, ngroups = rows / R;
  double s = 0;
  for (int t = blockIdx.x; t < ncb * ngroups; t += gridDim.x) {
    const int cb = t % ncb, rg = t / ncb;
    for (int i = threadIdx.x; i < R * SEGD; i += 512) {
      const int k = i / SEGD, c = i % SEGD;
      const size_t idx = (size_t)(rg + k * ngroups) * cols + (size_t)cb * SEGD + c;
      double x = v[idx];
      if (rmw) hv[idx] = hv[idx] + x; else s += x;
    }
  }
  if (s == 1.2345e300) out[0] = s;
}

int main() {
  const int rows = 3432, cols = 3432 - 3432 % 128;  // config 2 shape, cols multiple of 128
  const size_t n = (size_t)rows * cols;
  double *v, *hv, *out; CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&hv, n * 8)); CK(hipMalloc(&out, 8));
  CK(hipMemset(v, 0, n * 8)); CK(hipMemset(hv, 0, n * 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rmw = 0; rmw < 2; rmw++)
    for (int seg : {8, 16, 32, 64, 128}) {
      for (int R : {35, 1225}) {
        float ms = 0;
        for (int w = 0; w < 2; w++) {
          CK(hipEventRecord(a));
          for (int r = 0; r < 20; r++) {
            switch (seg) {
              case 8: tile_rw<8><<<1024, 512>>>(v, hv, rows, cols, R, rmw, out); break;
              case 16: tile_rw<16><<<1024, 512>>>(v, hv, rows, cols, R, rmw, out); break;
              case 32: tile_rw<32><<<1024, 512>>>(v, hv, rows, cols, R, rmw, out); break;
              case 64: tile_rw<64><<<1024, 512>>>(v, hv, rows, cols, R, rmw, out); break;
              default: tile_rw<128><<<1024, 512>>>(v, hv, rows, cols, R, rmw, out); break;
            }
          }
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        const double bytes = (double)n * 8 * (rmw ? 3 : 1) * 20;
        printf("%s  segment %4d B  rows/tile %4d : %7.0f GB/s\n", rmw ? "v read + hv rmw" : "v read only    ", seg * 8, R,
               bytes / ms / 1e6);
      }
    }
  return 0;
}
