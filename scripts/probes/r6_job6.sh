#!/bin/bash
# round 6, sixth GPU job: the VAE decode entry with the overlapped workspace -- bit identity again, config 4's epoch (time + peak memory), decode on one stream under two groups in flight
set -x
R=$PWD
O=$R/gpurun_out/r6_job6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 900 python -m pytest $R/tests/test_gpu_vae.py -m gpu -x -q -k "c_entry or f16x1 or two_streams or 1024" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 900 python $R/bench.py --config c4 --steps 2 --warmup 1 --no-pricing > $O/bench_c4.json 2> $O/bench_c4.err
tail -c 300 $O/bench_c4.err
for i in 1 2; do
  timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline > $O/bench_c2_two_stream_decode_$i.json 2>/dev/null
  timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-epoch --no-pricing --no-cpu-baseline --vae-one-stream > $O/bench_c2_one_stream_decode_$i.json 2>/dev/null
done
ls -la $O
