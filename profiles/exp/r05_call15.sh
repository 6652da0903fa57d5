#!/bin/bash
# round 5, call 15: SQ counters of the matrix-pipe wave kernels, against the fp32-FMA form
bash profiles/exp/pmc_wave.sh gpurun_out/r05p/mx 2>&1 | tee gpurun_out/r05p_mx.txt | cut -c1-600
COSY_TUNE_LIB=$PWD/cosypose_amd/lib/r05_nomx.so bash profiles/exp/pmc_wave.sh gpurun_out/r05p/nomx 2>&1 | tee gpurun_out/r05p_nomx.txt | cut -c1-600
rm -rf gpurun_out/r05p/*/pmc_*/
