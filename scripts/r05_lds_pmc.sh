#!/bin/bash
# LDS / wait counters of the kernels of one bench workload, one rocprofv3 --pmc pass per group.
# usage (GPU box): scripts/r05_lds_pmc.sh <workload> <tag> [kernel substrings, default: the count unit's]  -> gpurun_out/<tag>_lds_counters.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
wl=$1; tag=$2; shift 2
kernels=${*:-"k_part_scatter1 k_arr_scatter_seg k_part_build_q"}
O=$R/gpurun_out/$tag; mkdir -p $O; out=$R/gpurun_out/${tag}_lds_counters.txt; : > $out
i=0
# PMC_GROUPS="A B C|D E": other counter groups than the default six, one pass each
DEFAULT_GROUPS="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT|SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES|SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE|SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_WR|SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
IFS="|" read -ra G <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
for grp in "${G[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-host-leg --no-parity --no-configs > $O/g$i.log 2>&1 || { echo "group $i ($grp) failed" >> $out; tail -2 $O/g$i.log >> $out; continue; }
  python3 - $O/g$i $kernels >> $out <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for k in sys.argv[2:]:
            if k in r['Kernel_Name']:
                acc[(k,r['Counter_Name'])]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k in sorted(acc): print('%-18s %-24s %.4g per launch (%d launches)' % (k[0], k[1], acc[k]/max(1,n[k]), n[k]))
PY
  find $O/g$i -name "*counter_collection.csv" -delete
done
cat $out
