#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../p264decoder_amd/csrc/hip/kernel_inter.h"
template<int VAR> __device__ __forceinline__ void tapx(uint32_t n0, uint32_t n1, uint32_t n2, int t[4]) {
    const int C0 = 0x1414fb01, C1 = 0x000001fb;
    n0 ^= 0x80808080u; n1 ^= 0x80808080u; n2 ^= 0x80808080u;
    uint32_t A[4] = { n0, alignbyte(n1,n0,1), alignbyte(n1,n0,2), alignbyte(n1,n0,3) };
    uint32_t B[4] = { n1, alignbyte(n2,n1,1), alignbyte(n2,n1,2), alignbyte(n2,n1,3) };
    for (int i=0;i<4;i++) {
        if (VAR==1) { int d0 = __builtin_amdgcn_sdot4((int)A[i], C0, 0, false), d1 = __builtin_amdgcn_sdot4((int)B[i], C1, 0, false); t[i] = d0 + d1 + 4096; }
        else if (VAR==2) { int d1 = __builtin_amdgcn_sdot4((int)B[i], C1, 4096, false); asm volatile("s_nop 7" ::: "memory"); t[i] = __builtin_amdgcn_sdot4((int)A[i], C0, d1, false); asm volatile("s_nop 7" ::: "memory"); }
        else { t[i] = __builtin_amdgcn_sdot4((int)A[i], C0, __builtin_amdgcn_sdot4((int)B[i], C1, 4096, false), false); }
    }
}
template<int VAR> __device__ __forceinline__ void hvx(const uint32_t *w, int r, int b, int acc[4]) {
    acc[0]=acc[1]=acc[2]=acc[3]=512;
    const int cv[6] = { 1, -5, 20, 20, -5, 1 };
#pragma unroll
    for (int k = 0; k < 6; k++) { uint32_t n0,n1,n2; int t[4]; row12(w, r-2+k, b-2, n0,n1,n2); tapx<VAR>(n0,n1,n2,t);
#pragma unroll
        for (int i=0;i<4;i++) acc[i] += cv[k]*t[i]; }
}
__global__ void k(const uint32_t* win, int* out) {
  __shared__ uint32_t w[56];
  if (threadIdx.x < 56) w[threadIdx.x] = win[blockIdx.x*56 + threadIdx.x];
  __syncthreads();
  int lane = threadIdx.x; int r = 2 + (lane>>3), b = 2 + (lane & 7);
  int a0[4],a1[4],a2[4]; hvx<0>(w,r,b,a0); hvx<1>(w,r,b,a1); hvx<2>(w,r,b,a2);
  const uint8_t* wb = (const uint8_t*)w;
  auto f = [&](int x,int y){ return (int)wb[y*16+x]; };
  auto th = [&](int x,int y){ return f(x-2,y)-5*f(x-1,y)+20*(f(x,y)+f(x+1,y))-5*f(x+2,y)+f(x+3,y); };
  for (int i=0;i<4;i++) { int x=b+i; int tt = 512 + th(x,r-2)-5*th(x,r-1)+20*(th(x,r)+th(x,r+1))-5*th(x,r+2)+th(x,r+3);
     int ok = (b+i+3<=15);
     int* o = out + ((blockIdx.x*64+lane)*4+i)*4; o[0]=ok?tt:0; o[1]=ok?a0[i]:0; o[2]=ok?a1[i]:0; o[3]=ok?a2[i]:0; }
}
int main(){ const int NB=64; uint32_t* h=(uint32_t*)malloc(NB*56*4); srand(1); for(int i=0;i<NB*56;i++) h[i]=rand()*65536u+rand(); uint32_t* d; int* o; hipMalloc(&d,NB*56*4); hipMalloc(&o,NB*64*16*4); hipMemcpy(d,h,NB*56*4,hipMemcpyHostToDevice);
 k<<<NB,64>>>(d,o); int* ho=(int*)malloc(NB*64*16*4); hipMemcpy(ho,o,NB*64*16*4,hipMemcpyDeviceToHost); int bad[3]={0,0,0}; int shown=0;
 for(int i=0;i<NB*64*4;i++){ for(int v=0;v<3;v++) if(ho[i*4+1+v]!=ho[i*4]) { bad[v]++; if(shown<10 && v==0){ printf("idx %d (lane %d i %d) want %d got %d diff %d\n", i, (i/4)%64, i%4, ho[i*4], ho[i*4+1], ho[i*4+1]-ho[i*4]); shown++; } } }
 printf("mismatches: nested-acc %d  separate-add %d  nops %d  of %d\n", bad[0],bad[1],bad[2],NB*64*4); return 0; }
