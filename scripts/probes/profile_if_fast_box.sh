#!/bin/bash
# Boxes of the pool differ by up to 8 % (377 - 408 ms per serial config-2 step in round 5, LABNOTES.md 6): gauge the lease with the 20-second attention
# benchmark first and take the profile set only on a box at or above the round's median (S = 1229 attention forward <= LIMIT us).
LIMIT=${1:-169}
us=$(python scripts/bench_attention.py 2>/dev/null | grep "S=1229" | sed 's/.*min \([0-9.]*\) us.*/\1/')
echo "attention forward at 16x24x1229 on this box: $us us (limit $LIMIT)"
if [ -n "$us" ] && python -c "import sys; sys.exit(0 if float('$us') <= float('$LIMIT') else 1)"; then
  bash scripts/take_profiles.sh r5 > gpurun_out/take_profiles_r5.log 2>&1
  echo "profile set taken"
else
  echo "skipped"
fi
