#!/bin/bash
# rocprofv3 passes over the default bench: kernel trace + FETCH_SIZE + WRITE_SIZE + instruction counts (separate passes), summarised
# into profiles/<tag>_*.  usage: scripts/pmc_bench.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
O=$R/gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs "$@" > ${O}_trace.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${O}_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-host-leg --no-parity "$@" > ${O}_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${O}_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-host-leg --no-parity "$@" > ${O}_write.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d ${O}_insts -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-host-leg --no-parity "$@" > ${O}_insts.log 2>&1 || exit 1
grep '^{' ${O}_trace.log > $R/gpurun_out/${tag}_bench.json
# PMC_NO_LATEST=1: another workload than the headline's -- profiles/pmc_latest.json (what bench.py reads) stays as it is
cd $R && python3 scripts/summarize_prof.py $tag ${O}_trace ${O}_fetch ${O}_write ${O}_insts --bench $R/gpurun_out/${tag}_bench.json ${PMC_NO_LATEST:+--no-latest} > $R/gpurun_out/prof_$tag.summary 2>&1
cp $R/profiles/${tag}_* $R/profiles/pmc_latest.json $R/gpurun_out/ 2>/dev/null
# drop the bulky raw traces, keep the stats
find ${O}_trace ${O}_fetch ${O}_write ${O}_insts -name "*kernel_trace.csv" -delete; find ${O}_fetch ${O}_write ${O}_insts -name "*counter_collection.csv" -delete
cat $R/gpurun_out/prof_$tag.summary
