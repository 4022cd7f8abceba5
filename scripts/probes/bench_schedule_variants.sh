mkdir -p gpurun_out/r5
F="--steps 8 --warmup 2 --no-epoch --no-cpu-baseline --no-pricing"
for rep in 1 2; do
for v in "0 0" "1 0" "0 -1" "1 -1"; do set -- $v
  python bench.py $F --decode-in-future $1 --rollout-priority $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('decode_in_future=$1 prio=$2', d['ms_per_step'], d['frac_of_bf16_mfma_peak'], d['roofline']['frac'])"
done; done
