#!/bin/bash
# Builds one library per D128_VAR value (attention_d128.hip's timing experiments) next to the product library:
#   scripts/ablate_d128.sh 1 2 4 8   ->  adv_grpo_amd/libadvgrpo_abl_d128_<v>.so   (git-ignored; they travel with gpurun)
# and on the GPU box:  for v in ...; do ADVGRPO_LIB=adv_grpo_amd/libadvgrpo_abl_d128_$v.so python scripts/bench_attention_d128.py; done
set -e
cd "$(dirname "$0")/../adv_grpo_amd/csrc"
make -j8 > /dev/null
OTHERS=$(ls obj/*.o | grep -v attention_d128.o)
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DD128_VAR=$v -c attention_d128.hip -o /tmp/att_d128_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libadvgrpo_abl_d128_$v.so $OTHERS /tmp/att_d128_$v.o
  echo built $v
done
