#!/bin/bash
# rocprofv3 kernel trace of the prefill legs (bench.py's seven GEMMs of one layer at M = 2048, int4 and int8 weights):
# per-kernel durations and the gaps between consecutive kernels.  scripts/prof_prefill.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02q}
mkdir -p gpurun_out/$TAG; rm -rf gpurun_out/$TAG/pprof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/pprof -o p -- python scripts/prefill_bench.py 2048 > gpurun_out/$TAG/prefill_prof.json 2>/dev/null
python - "$TAG" <<'PY'
import csv,glob,collections,sys
tag=sys.argv[1]
f=glob.glob("gpurun_out/%s/pprof/**/p_kernel_trace.csv"%tag, recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out=[]
agg=collections.defaultdict(list)
prev=None
for r in rows:
    n=r["Kernel_Name"]
    if "gemm" not in n and "reduce" not in n and "cvt_a16" not in n: prev=None; continue
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    key=(n.split("(")[0].replace("void ns::","")[:40], r["Grid_Size_X"], r["Grid_Size_Y"])
    agg[key].append(((e-s)/1e3, (s-prev)/1e3 if prev else 0.0))
    prev=e
with open("gpurun_out/%s/prefill_kernels.txt"%tag,"w") as o:
    for k,v in agg.items():
        d=sorted(x[0] for x in v); g=sorted(x[1] for x in v)
        line="%-42s grid %7s x %s  n %3d  dur med %.1f us (min %.1f)  gap-before med %.1f us" % (k[0],k[1],k[2],len(v),d[len(d)//2],d[0],g[len(g)//2])
        print(line); o.write(line+"\n")
PY
cat gpurun_out/$TAG/prefill_prof.json | tail -1
find gpurun_out/$TAG/pprof -name "*.csv" -size +5M -delete
