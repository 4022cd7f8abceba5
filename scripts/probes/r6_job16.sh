#!/bin/bash
# HBM-bound kernels of the path in GB/s (north_star: "achieved HBM GB/s ... against chip peak"), VAE tests after the descriptor-lock patch
set -x
R=$PWD
O=$R/gpurun_out/r6_job16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 300 python $R/scripts/bench_hbm_kernels.py 2>/dev/null | grep -v amdgpu > $O/hbm_kernels.txt; cat $O/hbm_kernels.txt
timeout 900 python -m pytest $R/tests/test_gpu_vae.py $R/tests/test_gpu_rollout.py -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
