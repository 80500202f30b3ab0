#!/bin/bash
# VERDICT r02 #3's mark: ten consecutive processes of the same command, k_part_scatter1's mean launch time in each
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03d_scatter1_runs.txt
echo "# ten consecutive processes: python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-host-leg" > $out
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 200 python bench.py --workload ont_k31_count --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-host-leg > gpurun_out/ten_$i.json 2> gpurun_out/ten_$i.err || { tail -3 gpurun_out/ten_$i.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ten_$i.json').read().strip().splitlines()[-1]);k=d['kernels']
print('run %2d  k_part_scatter1 %.2f  k_arr_scatter %.2f  k_part_build_q %.2f  step %.2f ms' % ($i, k['k_part_scatter1']['avg_ms'], k['k_arr_scatter']['avg_ms'], k['k_part_build_q']['avg_ms'], d['ms_per_step']))" | tee -a $out
done
