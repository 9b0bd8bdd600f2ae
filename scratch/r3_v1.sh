#!/bin/bash
bash scratch/variants_run.sh "base stnt base:P264AMD_MC_BAND_LOG2=3 base:P264AMD_MC_BAND_LOG2=5 stnt:P264AMD_MC_BAND_LOG2=3 base:P264AMD_MC_WGS_PER_PIC=36 base:P264AMD_MC_WGS_PER_PIC=64" | tee gpurun_out/r3_v1.txt
