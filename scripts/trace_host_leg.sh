#!/bin/bash
# timeline of the host-to-host leg: kernel trace + memory-copy trace of one run, summarised as (start, end, what) rows of the last leg
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_host; rm -rf $O; mkdir -p $O; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kmu::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + n[:40]))
for f in glob.glob(d + "/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s" % r.get("Direction", "?").replace("MEMORY_COPY_", "")))
ev.sort()
# the last host leg: from the last big H2D burst backwards ~200 ms
big = [e for e in ev if e[2].startswith("C") and "HOST_TO_DEVICE" in e[2] and e[1] - e[0] > 500_000]
if not big: print("no big uploads found"); print(ev[-5:]); sys.exit(0)
t_end = ev[-1][1]
# find the start of the last sequence of big uploads (gap > 50 ms before)
starts = [big[0][0]]
for a, b in zip(big, big[1:]):
    if b[0] - a[1] > 50_000_000: starts.append(b[0])
t0 = starts[-1]
rows = [e for e in ev if e[0] >= t0 - 1_000_000]
out = open(d + "/timeline.txt", "w")
for s, e, w in rows:
    if e - s < 150_000 and not w.startswith("C"): continue  # kernels under 0.15 ms are left out
    out.write("%9.3f %9.3f %8.3f  %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, w))
out.close()
print(open(d + "/timeline.txt").read()[:6000])
PY
find $O -name "*.csv" -size +1M -delete
