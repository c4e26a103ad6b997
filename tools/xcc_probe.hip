// tools/xcc_probe.hip — which XCD does workgroup `id` of a 1-D grid run on? Reads HW_REG_XCC_ID per workgroup and compares with id % 8
// for several grid / block sizes (the premise of the XCD-banded tile order of gemm_glds.h and of the XCD-local kron fusion). In this otherwise
// idle process workgroup 0 always lands on XCD 0; inside bench.py (other queues active before) the same probe starts at XCD 5: only the PERIOD
// (ids equal modulo 8 share an XCD) can be relied on, and that is what the library checks (dense.hip: xcd_map_ok).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/xcc_probe tools/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int ID>
__global__ void probe(unsigned *out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg((8 - 1) << 11 | 0 << 6 | ID);
}
int main() {
  unsigned *d;
  CK(hipMalloc(&d, 4096 * 4));
  std::vector<unsigned> h(4096);
  for (int block : {128, 256, 512}) {
    for (int grid : {8, 32, 64, 256, 512, 1024}) {
      CK(hipMemset(d, 0xff, 4096 * 4));
      hipLaunchKernelGGL(probe<20>, dim3(grid), dim3(block), 0, 0, d);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost));
      int ok = 0;
      for (int i = 0; i < grid; ++i) ok += ((h[i] & 15u) == (unsigned)(i & 7));
      printf("block %3d grid %4d: XCC_ID[3:0] == id %% 8 for %4d of %4d workgroups; first 16 raw:", block, grid, ok, grid);
      for (int i = 0; i < 16 && i < grid; ++i) printf(" %x", h[i]);
      printf("\n");
    }
  }
  // back-to-back launches: does the starting XCD of a launch depend on the previous one?
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(probe<20>, dim3(12), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(probe<20>, dim3(32), dim3(256), 0, 0, d + 64);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 96 * 4, hipMemcpyDeviceToHost));
    printf("rep %d: grid 12 ->", rep);
    for (int i = 0; i < 12; ++i) printf(" %x", h[i] & 15u);
    printf("  | then grid 32 ->");
    for (int i = 0; i < 16; ++i) printf(" %x", h[64 + i] & 15u);
    printf("\n");
  }
  return 0;
}
