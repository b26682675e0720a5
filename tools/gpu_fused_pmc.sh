#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/fpmc
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/fpmc -o a -- python $R/tools/exp_fused_pmc.py 2>&1 | grep waves
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/fpmc -o b -- python $R/tools/exp_fused_pmc.py 2>&1 | grep waves
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/fpmc/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "fused" in r["Kernel_Name"]:
            k = "pipe" if "pipe" in r["Kernel_Name"] else "wg4"
            acc[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, d in acc.items():
        for c, v in d.items():
            byd = collections.defaultdict(float)
            for i, x in v: byd[i] += x
            vals = sorted(byd.values())
            print(f"{k:5s} {c:28s} {vals[len(vals)//2]:.4g}")
PY
