// which XCD does a workgroup run on?  s_getreg_b32 HW_REG_XCC_ID (gfx940+: hardware register 20, bits 3:0)
// build: hipcc -O3 --offload-arch=gfx950 tools/dev/xcc_id.hip -o tools/dev/xcc_id.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
}
int main() {
  int* d; hipMalloc(&d, 1024 * 4);
  for (int threads : {64, 1024}) {
    k<<<64, threads>>>(d);
    int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%4d threads per workgroup, XCC_ID of workgroups 0..63:", threads);
    for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
    printf("\n");
  }
  return 0;
}
