#!/bin/bash
# k_pmh_points: long reads taken by whole workgroups -- sketch tests, then the A/B on the bench (device leg + host leg)
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py tests/test_golden_signatures.py -x -q -m gpu -k "probminhash or points or sketch or pmh or two_kernel or golden or smallk or short" > gpurun_out/t_ptslong.log 2>&1
rc=$?
tail -4 gpurun_out/t_ptslong.log
grep -q "Memory access fault" gpurun_out/t_ptslong.log && { echo GPU FAULT; exit 1; }
[ $rc -eq 0 ] || exit 1
fi
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/ab_pl_$label.json 2> gpurun_out/ab_pl_$label.err
  rc=$?
  if grep -q "Memory access fault" gpurun_out/ab_pl_$label.err; then echo "GPU FAULT in $label"; exit 1; fi
  [ $rc -eq 0 ] || { echo "$label failed rc=$rc"; tail -3 gpurun_out/ab_pl_$label.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_pl_$label.json').read().strip().splitlines()[-1])
print('$label', 'dev ms',round(d['ms_per_step'],2),'host ms', round(d['host_to_host']['ms_per_step'],2), {k:round(v['avg_ms'],2) for k,v in d['kernels'].items() if 'pmh' in k and '+' not in k}, d['checks']['sig_checksum'], d['checks'].get('host_leg_equals_device_leg'))"
}
run dflt KMU_X=1
run taper KMU_PIPE_TAPER=1
run c384 KMU_PIPE_CHUNK_MB=384
run dflt_b KMU_X=1
run taper_b KMU_PIPE_TAPER=1
run c640 KMU_PIPE_CHUNK_MB=640
run c384_taper KMU_PIPE_CHUNK_MB=384 KMU_PIPE_TAPER=1
