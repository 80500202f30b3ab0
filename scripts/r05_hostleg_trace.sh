#!/bin/bash
# the host-to-host leg's kernel timeline: where the GPU idles between kernels (rocprofv3 --kernel-trace of one bench run with the host leg; the last
# kmu_sketch_count call's kernels).  usage (GPU box): scripts/r05_hostleg_trace.sh  -> gpurun_out/r05_hostleg_trace.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
O=$R/gpurun_out/r05hl; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-parity > $O/bench.json 2> $O/bench.err || { echo "trace run failed"; tail -3 $O/bench.err; exit 1; }
python3 - $O <<'PY' > $R/gpurun_out/r05_hostleg_trace.txt
import csv, glob, sys, json
rows = []
for f in glob.glob(sys.argv[1] + '/trace/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# the host leg's steps: runs of kernels that contain k_unpack2b; take the last such call (from its first k_unpack2b back to ... its build)
idx = [i for i, r in enumerate(rows) if 'k_unpack2b' in r[2]]
if not idx: print('no k_unpack2b in the trace'); sys.exit(0)
# split unpack launches into calls: a gap of more than 50 ms between unpacks starts a new call
calls = [[idx[0]]]
for a, b in zip(idx, idx[1:]):
    if rows[b][0] - rows[a][0] > 60e6: calls.append([b])
    else: calls[-1].append(b)
last = calls[-1]
i0 = last[0]
# the call ends with the first k_part_build_q behind its last unpack
i1 = next(i for i in range(last[-1], len(rows)) if 'k_part_build_q' in rows[i][2])
while i1 + 1 < len(rows) and ('k_count_add_spill' in rows[i1 + 1][2]): i1 += 1
seg = rows[i0:i1 + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy = 0; cur_end = t0; gaps = []
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, n))
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print('host leg, last call: first kernel to last kernel %.2f ms; GPU busy %.2f ms; idle %.2f ms in %d gaps; %d kernel launches' % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), len(seg)))
print('gaps over 0.1 ms (ms, at ms from the start, before kernel):')
for g, at, n in sorted(gaps, reverse=True)[:25]:
    if g > 1e5: print('  %.3f at %.2f before %s' % (g / 1e6, at / 1e6, n[:60]))
by = {}
for s, e, n in seg:
    k = n.split('(')[0][-40:]; by.setdefault(k, [0, 0]); by[k][0] += 1; by[k][1] += e - s
print('kernel time by name (ms, launches):')
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:16]: print('  %-42s %8.2f %4d' % (k, t / 1e6, c))
PY
cat $R/gpurun_out/r05_hostleg_trace.txt
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench under the tracer: device leg', round(d['ms_per_step'],2), 'host to host', round(d['host_to_host']['ms_per_step'],2))" >> $R/gpurun_out/r05_hostleg_trace.txt
find $O/trace -name "*_kernel_trace.csv" -delete
