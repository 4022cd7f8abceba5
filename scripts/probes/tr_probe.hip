// probe: semantics of ds_read_b64_tr_b16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint16_t* out){
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for(int i=threadIdx.x;i<1024;i+=64) lds[i]=i;
  __syncthreads();
  uint32_t addr = (uint32_t)(uintptr_t)(lds) + threadIdx.x*8;  // lane t -> elements 4t..4t+3
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[threadIdx.x*4+0]=r.x&0xffff; out[threadIdx.x*4+1]=r.x>>16; out[threadIdx.x*4+2]=r.y&0xffff; out[threadIdx.x*4+3]=r.y>>16;
}
int main(){
  uint16_t* d; hipMalloc(&d, 64*4*2); k<<<1,64>>>(d); uint16_t h[256]; hipMemcpy(h,d,512,hipMemcpyDeviceToHost);
  for(int l=0;l<64;l++){ printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4],h[l*4+1],h[l*4+2],h[l*4+3]); }
  return 0;
}
