#!/bin/bash
# A/B of two builds on the SAME box: scripts/ab.sh <cmd...>  runs cmd with libadvgrpo_base.so and libadvgrpo_hip.so alternately
for i in 1 2; do
  for lib in base hip; do
    echo "== $lib"; ADVGRPO_LIB=$PWD/adv_grpo_amd/libadvgrpo_$lib.so "$@"
  done
done
