// Probe of ds_read_b64_tr_b16 semantics (gfx950): prints which LDS elements each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out){
  __shared__ __attribute__((aligned(16))) short lds[64*64];
  for (int i = threadIdx.x; i < 64*64; i += 64) lds[i] = (short)i;   // element (r,c) = r*64+c
  __syncthreads();
  const int l = threadIdx.x, grp = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
  const short* addr = lds + (4*grp + q) * 64 + 4*p;                  // lane 4q+p: row 4grp+q, cols 4p..4p+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[l*4+e] = v[e];
}
int main(){
  short* d; hipMalloc(&d, 64*4*2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    int g = l >> 4, i = l & 15;
    for (int e = 0; e < 4; ++e) { int want = (4*g+e)*64 + i; if (h[l*4+e] != want) ok = 0; }
  }
  printf("expected mapping (lane i of group g gets column i of rows 4g..4g+3): %s\n", ok ? "CONFIRMED" : "DIFFERENT");
  for (int l = 0; l < 64; l += 5) printf("lane %2d: %d %d %d %d  (r,c)=(%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l*4],h[l*4+1],h[l*4+2],h[l*4+3],
     h[l*4]/64,h[l*4]%64,h[l*4+1]/64,h[l*4+1]%64,h[l*4+2]/64,h[l*4+2]%64,h[l*4+3]/64,h[l*4+3]%64);
  return 0;
}
