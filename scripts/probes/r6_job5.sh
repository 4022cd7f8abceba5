#!/bin/bash
# round 6, fifth GPU job: the VAE decode C entry (bit-identity tests, users of the decoder), f16x1 without the lo requests, config 4's micro-steps in order
set -x
R=$PWD
O=$R/gpurun_out/r6_job5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 1500 python -m pytest $R/tests/test_gpu_vae.py $R/tests/test_gpu_rollout.py $R/tests/test_gpu_goldens.py $R/tests/test_abi.py -m gpu -x -q -s > $O/tests.txt 2>&1
tail -4 $O/tests.txt; grep -a 'vs the fp32 oracle at' $O/tests.txt
timeout 300 python - > $O/vae_modes.txt 2>/dev/null <<'PY'
import time, torch
from adv_grpo_amd import synthetic
from adv_grpo_amd.model_configs import VaeConfig
from adv_grpo_amd.vae import AutoencoderKLDecoder
cfg = VaeConfig()
W = synthetic.vae_decoder_weights(cfg, 4321, fp16_checkpoint=True)
lat = torch.randn(8, 16, 64, 64, device="cuda").to(torch.bfloat16)
for name, kw, c in (("f16x2 (default), C entry", {}, True), ("f16x2 (default), Python sequencing", {}, False), ("f16x1 (TF32-class), C entry", dict(f16_single=True), True)):
    dec = AutoencoderKLDecoder(W, cfg, "cuda", mode="bf16x3", **kw)
    dec.c_decode = c
    for _ in range(3): dec.decode_to_image(lat)
    torch.cuda.synchronize()
    ts = []
    for r in range(3):
        t0 = time.perf_counter()
        for _ in range(5): dec.decode_to_image(lat)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"{name}: {min(ts):.2f} ms per 8 x 512^2 decode (min of 3 x 5)   [{' '.join(f'{t:.2f}' for t in ts)}]")
PY
cat $O/vae_modes.txt
timeout 900 python $R/bench.py --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
tail -c 300 $O/bench_c4.err
ls -la $O
