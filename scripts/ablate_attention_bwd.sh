#!/bin/bash
# Builds one library per BP_ABL value (attention_bwd_pipe.hip's timing-only ablations; results are wrong) next to the product
# library:   scripts/ablate_attention_bwd.sh 1 2 4 8   ->  adv_grpo_amd/libadvgrpo_abl_bwd_<v>.so   (git-ignored; they travel with gpurun)
# on the GPU box:  for v in ...; do ADVGRPO_LIB=adv_grpo_amd/libadvgrpo_abl_bwd_$v.so python scripts/bench_attention_bwd.py; done
# a value of the form a<N> builds BP_AHEAD=N instead
set -e
cd "$(dirname "$0")/../adv_grpo_amd/csrc"
make -j8 > /dev/null
OTHERS=$(ls obj/*.o | grep -v attention_bwd_pipe.o)
for v in "$@"; do
  case $v in a*) DEF="-DBP_AHEAD=${v#a}";; w*) DEF="-DBP_WAVES=${v#w}";; d*) DEF="-DBP_DMA_SLOT=${v#d}";; *) DEF="-DBP_ABL=$v";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $DEF -c attention_bwd_pipe.hip -o /tmp/att_bwd_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libadvgrpo_abl_bwd_$v.so $OTHERS /tmp/att_bwd_$v.o
  echo built $v
done
