#!/bin/bash
# Timing-only ablations of attention_fwd_pipe_kernel (ATT_ABL bits, csrc/attention_pipe.hip): builds one library per variant
# next to the product one; run `python scripts/bench_attention.py` with ADVGRPO_LIB pointing at each on the GPU box.
#   scripts/ablate_attention.sh build 1 2 4 ...      (here, no GPU)      scripts/ablate_attention.sh run 1 2 4 ...   (GPU box)
set -e
cd "$(dirname "$0")/../adv_grpo_amd/csrc"
mode=$1; shift
for v in "$@"; do
  if [ "$mode" = build ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DATT_ABL=$v -c attention_pipe.hip -o obj/attention_pipe_abl$v.o
    objs=$(ls obj/*.o | grep -v attention_pipe)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libadvgrpo_abl$v.so $objs obj/attention_pipe_abl$v.o
  else
    echo "== ATT_ABL=$v"; ADVGRPO_LIB=$PWD/../libadvgrpo_abl$v.so python ../../scripts/bench_attention.py
  fi
done
