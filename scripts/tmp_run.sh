cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/g3prof
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g3prof -o g3 -- python scripts/gemm_shapes_bench.py 2048 > gpurun_out/g3prof_out.json 2>/dev/null
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/g3prof/**/g3_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    if "gemm3" in r["Kernel_Name"]:
        agg[(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in agg.items():
    v=sorted(v); print(k, len(v), "min %.1f med %.1f max %.1f us" % (v[0]/1e3, v[len(v)//2]/1e3, v[-1]/1e3))
PY
cat gpurun_out/g3prof_out.json | tail -1
find gpurun_out/g3prof -name "*.csv" -size +5M -delete
