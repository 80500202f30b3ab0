#!/bin/bash
# k_sketch_super on config 5's shard: LDS counters (what the step loop waits for), one rocprofv3 --pmc pass per group
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
mkdir -p $R/gpurun_out/r05s
i=0
for grp in "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/r05s/g$i -- python3 $R/bench.py --workload c5_aa --steps 1 --warmup 0 --no-cpu-baseline --no-host-leg --no-parity > $R/gpurun_out/r05s/g$i.log 2>&1 || { echo "group $i ($grp) failed"; tail -2 $R/gpurun_out/r05s/g$i.log; continue; }
  python3 - $R/gpurun_out/r05s/g$i <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_sketch_super' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(acc): print('k_sketch_super %-24s %.4g per launch (%d samples)' % (k, acc[k]/max(1,n[k])*1, n[k]))
PY
  find $R/gpurun_out/r05s/g$i -name "*counter_collection.csv" -delete
done
