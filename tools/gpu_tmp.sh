cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r05_v25_c2prof; mkdir -p $O; export TMPDIR=/tmp
python tools/perf_conv_igemm.py 1000 2>&1 | grep -v amdgpu.ids | tee $O/layers.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c2 -- python tools/perf_c2.py 3000 1000 winograd > $O/prof.log 2>&1; tail -1 $O/prof.log
python - <<EOF2
import csv,glob
f=glob.glob("gpurun_out/r05_v25_c2prof/prof/**/*kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "naive_conv" not in r["Name"]]
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total (without MIOpen's find-mode kernel) %.1f ms" % (tot/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:12]:
    print("%6.2f%% %8.2f ms %6s  avg %8.3f ms  %s" % (100*float(r["TotalDurationNs"])/tot, float(r["TotalDurationNs"])/1e6, r["Calls"], float(r["AverageNs"])/1e6, r["Name"][:90]))
EOF2
rm -rf $O/prof
