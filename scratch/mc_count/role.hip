// static instruction count of one role of k_mc for one key (scratch/mc_count/run.sh): -DROLE=0..3 -DEXPM_FORCE_KEY=k -DP264AMD_TIMING_BUILD
#include "kernel_mc.h"
#ifndef PASSB
#define PASSB false
#endif
__global__ __launch_bounds__(256, 4)
void k_role(const PicDev *__restrict__ pics, const uint32_t *__restrict__ mc_all, Geom g, McLayout ml, int sub, int wgs)
{
    __shared__ __attribute__((aligned(16))) uint8_t images[MC_IMAGE_BYTES];
    __shared__ uint32_t ref_tab[2 * P264HIP_MAX_REFS];
    if (threadIdx.x < 32) ref_tab[threadIdx.x] = pics->ref_off[threadIdx.x & 15];
    __syncthreads();
#if ROLE == 0
    mc_luma_body<true, PASSB>(images, ref_tab, pics, mc_all, g, ml, sub, wgs);
#elif ROLE == 1
    mc_luma_body<false, PASSB>(images, ref_tab, pics, mc_all, g, ml, sub, wgs);
#elif ROLE == 2
    mc_chroma_body<true, PASSB>(images, ref_tab, pics, mc_all, g, ml, sub, wgs);
#else
    mc_chroma_body<false, PASSB>(images, ref_tab, pics, mc_all, g, ml, sub, wgs);
#endif
}
