#!/bin/bash
set -x
R=$PWD
O=$R/gpurun_out/r6_job18; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 900 python -m pytest $R/tests/test_gpu_vae.py -m gpu -x -q -s -k "f16x1" > $O/tests.txt 2>&1; tail -3 $O/tests.txt; grep -a 'uint8 pixels\|vs the fp32' $O/tests.txt
