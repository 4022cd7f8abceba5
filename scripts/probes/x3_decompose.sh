#!/bin/bash
# What the k loop of the split-bf16 convolution spends its time on: the 8 x 512^2 decode with the kernel's DMA (1), MFMAs (2) or
# fragment reads (3) removed in turn (results are WRONG in those modes).  Needs the experiments build: make EXPERIMENTS=1.
for d in 0 1 2 3 0; do
  echo "== ADVGRPO_X3_DBG=$d"
  ADVGRPO_X3_DBG=$d ADVGRPO_LIB=$PWD/adv_grpo_amd/libadvgrpo_experiments.so python scripts/bench_vae_modes.py 2>&1 | grep x3
done
