#!/bin/bash
# config 5's artefacts again (the first pass of take_profiles.sh r6 hit an AttributeError in the inherited arithmetic()), the qwen vae tests, the new trainer test
set -x
R=$PWD
O=$R/gpurun_out/prof_r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 900 python -m pytest $R/tests/test_gpu_qwen_vae.py $R/tests/test_gpu_trainer.py -m gpu -x -q -k "qwen or side_streams or in_flight" > $O/tests_after_fix.txt 2>&1
tail -3 $O/tests_after_fix.txt
timeout 1500 python $R/bench.py --config c5 --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
tail -c 300 $O/bench_c5.err
rocprofv3 --kernel-trace --stats -d $O/kt_c5 -o x -- python $R/bench.py --config c5 --steps 1 --warmup 1 --no-pricing --no-epoch > $O/bench_c5_prof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $O/kt_c5/x_results.db $O/kernel_stats_c5.md > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_c5_$c -o x -- python $R/bench.py --config c5 --steps 1 --warmup 1 --no-epoch --no-cpu-baseline --no-pricing --schedule serial > /dev/null 2>&1
done
mkdir -p $O/pmc_c5 && cp -r $O/pmc_c5_FETCH_SIZE $O/pmc_c5_WRITE_SIZE $O/pmc_c5/
python $R/scripts/pmc_traffic.py $O/pmc_c5 $O/pmc_traffic_c5.json > $O/pmc_traffic_c5.txt
rm -rf $O/kt_c5 $O/pmc_c5_* $O/pmc_c5
ls -la $O | head -5
