#!/bin/bash
# round 6, fourth GPU job: the block-backward C entry and the f16x1 decoder mode -- their tests, the G micro-step with / without the entry, a bench line with the new leg
set -x
R=$PWD
O=$R/gpurun_out/r6_job4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 1200 python -m pytest $R/tests/test_gpu_train.py $R/tests/test_gpu_vae.py $R/tests/test_gpu_trainer.py -m gpu -x -q -s > $O/tests.txt 2>&1
tail -4 $O/tests.txt; grep -a 'vs the fp32 oracle at' $O/tests.txt
timeout 300 python $R/scripts/bench_gstep.py 2>/dev/null | grep -v amdgpu > $O/gstep.txt
timeout 300 python - > $O/gstep_python_sequencing.txt 2>/dev/null <<'PY'
import runpy, sys
from adv_grpo_amd import mmdit_train
mmdit_train.SD3TransformerLoRA.c_block_bwd = False
sys.argv = ["bench_gstep.py"]
runpy.run_path(sys.argv and __import__("os").environ["PYTHONPATH"] + "/scripts/bench_gstep.py", run_name="__main__")
PY
cat $O/gstep.txt $O/gstep_python_sequencing.txt
timeout 900 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 400 $O/bench_c2.err
ls -la $O
