#!/bin/bash
# Where k_pmh_points' instructions go: a diagnostic build (-DKMU_DIAG=1) with parts of the kernel switched off
# (KMU_PMH_ABLATE: 512 = no pass 2; 8192 = the keys that pass the cheap test are not worked off; 16384 = no cheap test either: the
# list walk alone), SQ_INSTS_VALU and the launch time of each.  Sketch-only headline workload; the rows of the ablated runs are wrong.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
mkdir -p $R/gpurun_out/r05p
for abl in 0 512 8704 25088; do
  KMU_LIB=$R/kmerutils_amd/libkmu_diag.so KMU_PMH_ABLATE=$abl timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/r05p/a$abl -- \
    python3 $R/bench.py --workload ont_k31_sketch --steps 1 --warmup 0 --no-cpu-baseline --no-host-leg --no-parity > $R/gpurun_out/r05p/a$abl.log 2>&1 || { echo "ablate $abl failed"; tail -3 $R/gpurun_out/r05p/a$abl.log; continue; }
  python3 - $R/gpurun_out/r05p/a$abl $abl <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float); n=collections.Counter(); dur=[]
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_pmh_points' in r['Kernel_Name'] and 'short' not in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for f in glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_pmh_points' in r['Kernel_Name'] and 'short' not in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
keys=4357236706
print('KMU_PMH_ABLATE=%-6s k_pmh_points: VALU %.4g wave-instructions per launch = %.1f lane-instructions per key, SALU %.4g, %s ms' % (
    sys.argv[2], acc['SQ_INSTS_VALU']/max(1,n['SQ_INSTS_VALU']), acc['SQ_INSTS_VALU']/max(1,n['SQ_INSTS_VALU'])*64/keys, acc['SQ_INSTS_SALU']/max(1,n['SQ_INSTS_SALU']), ['%.2f'%d for d in dur]))
PY
  find $R/gpurun_out/r05p/a$abl -name "*counter_collection.csv" -delete -o -name "*kernel_trace.csv" -delete
done
