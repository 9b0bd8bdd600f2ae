#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../p264decoder_amd/csrc/hip/kernel_inter.h"
__global__ void k(const uint32_t* win, int* out) {
  __shared__ uint32_t w[56];
  if (threadIdx.x < 56) w[threadIdx.x] = win[threadIdx.x];
  __syncthreads();
  int lane = threadIdx.x; int r = 2 + (lane>>3), b = 2 + (lane & 7);   // b up to 9
  const uint8_t* wb = (const uint8_t*)w;
  auto f = [&](int x,int y){ return (int)wb[y*16+x]; };
  auto th = [&](int x,int y){ return f(x-2,y)-5*f(x-1,y)+20*(f(x,y)+f(x+1,y))-5*f(x+2,y)+f(x+3,y); };
  uint32_t n0,n1,n2; int t[4]; row12(w, r, b-2, n0,n1,n2); tap_h4(n0,n1,n2,t);
  int bad=0; for(int i=0;i<4;i++) if (b+i+3<=15 && t[i]!=th(b+i,r)) bad|=1<<i;
  uint32_t hv = hv4(w, r, b);
  int badhv=0;
  for (int i=0;i<4;i++) { if (b+i+3>15) continue; int x=b+i; int tt = th(x,r-2)-5*th(x,r-1)+20*(th(x,r)+th(x,r+1))-5*th(x,r+2)+th(x,r+3); int v = clip255((tt+512)>>10); if (((hv>>(8*i))&255)!=(unsigned)v) badhv|=1<<i; }
  out[lane*2]=bad; out[lane*2+1]=badhv;
  if (lane==0) { for (int k=0;k<6;k++){ uint32_t a0,a1,a2; int tt[4]; row12(w, r-2+k, b-2, a0,a1,a2); tap_h4(a0,a1,a2,tt); out[128+k*2]=tt[2]; out[128+k*2+1]=th(b+2,r-2+k);} out[140]=hv; int x=b+2; int tt = th(x,r-2)-5*th(x,r-1)+20*(th(x,r)+th(x,r+1))-5*th(x,r+2)+th(x,r+3); out[141]=tt; out[142]=(tt+512)>>10; }
}
int main(){ uint32_t h[56]; srand(1); for(int i=0;i<56;i++) h[i]=rand()*65536u+rand(); uint32_t* d; int* o; hipMalloc(&d,sizeof h); hipMalloc(&o,160*4); hipMemcpy(d,h,sizeof h,hipMemcpyHostToDevice);
 k<<<1,64>>>(d,o); int ho[160]; hipMemcpy(ho,o,sizeof ho,hipMemcpyDeviceToHost); int nb=0,nh=0; for(int i=0;i<64;i++){ if(ho[2*i]) nb++; if(ho[2*i+1]) nh++; } printf("bad tap lanes %d bad hv lanes %d\n",nb,nh); for(int i=0;i<16;i++) printf("%d:%x/%x ",i,ho[2*i],ho[2*i+1]); printf("\n"); for(int k=0;k<6;k++) printf("row%d t=%d ref=%d\n",k,ho[128+2*k],ho[129+2*k]); printf("hv=%08x sum=%d val=%d\n",ho[140],ho[141],ho[142]); return 0; }
