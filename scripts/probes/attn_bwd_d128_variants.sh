#!/bin/bash
# A/B of the head-dim-128 attention backward kernel variants (ADVGRPO_ATTN_BWD_D128 = dQ variant << 4 | dK/dV variant, see
# attention_bwd_d128.hip's launcher): parity test, then per-kernel average time from a rocprofv3 kernel trace on the same box.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  echo "== variant $v"
  ADVGRPO_ATTN_BWD_D128=$v timeout 100 python -m pytest $R/tests/test_gpu_qwen_train.py -q -x -k "attention_backward_d128" 2>&1 | tail -1
  rm -rf /tmp/p$v
  ADVGRPO_ATTN_BWD_D128=$v PYTHONPATH=$R timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/p$v -o x -- python $R/scripts/probes/attn_bwd_d128_time.py > /dev/null 2>&1
  timeout 30 python $R/scripts/rocpd_stats.py /tmp/p$v/x_results.db /tmp/p$v/s.md > /dev/null 2>&1
  grep "attn_bwd_d128" /tmp/p$v/s.md < /dev/null | sed -e 's/void advgrpo::(anonymous namespace):://' -e 's/(advgrpo::AttnBwdParams)//' | cut -d'|' -f2,5
done
