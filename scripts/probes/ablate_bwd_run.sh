cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ $v = base ]; then L=$R/adv_grpo_amd/libadvgrpo_hip.so; else L=$R/adv_grpo_amd/libadvgrpo_abl_bwd_$v.so; fi
  rm -rf /tmp/bp_$v
  ADVGRPO_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp_$v -o x -- python $R/scripts/bench_attention_bwd.py > /dev/null 2>&1
  python - $v <<'PY'
import csv,glob,sys,collections
f=glob.glob('/tmp/bp_%s/**/*kernel_trace.csv'%sys.argv[1],recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'attn_bwd_pipe' in n or 'attn_bwd_dq' in n or 'attn_bwd_dkdv' in n:
        d[('dkdv' if ('<true>' in n or 'dkdv' in n) else 'dq', r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print(sys.argv[1], ' '.join('%s/%s:%.0f'%(k[0],k[1],sorted(v)[len(v)//2]) for k,v in sorted(d.items())))
PY
done
