#!/usr/bin/env python
"""One entry of profiles/pmc_by_kernel.json from a tools/pmc_kernel.sh summary:
    python tools/pmc_summary_to_table.py <summary.json> <table key> <shape string> <algorithmic bytes> <source note>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, key, shape, alg, note = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
s = json.load(open(src))
c = s["counters_median_per_dispatch"]
path = os.path.join(ROOT, "profiles", "pmc_by_kernel.json")
table = json.load(open(path))
table[key] = {
    "traffic_bytes": s["l2_miss_fabric_bytes"], "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "algorithmic_bytes": alg, "traffic_over_algorithmic": s["l2_miss_fabric_bytes"] / alg, "l2_hit_rate": s.get("l2_hit_rate"),
    "mfma_busy_frac": s.get("mfma_busy_frac"), "effective_clock_GHz": s.get("effective_clock_GHz"),
    "lds_bank_conflict_frac": s.get("lds_bank_conflict_frac"), "kernel_ms_traced": s.get("kernel_ms_traced_median"),
    "match": {"shape": shape}, "source": note}
json.dump(table, open(path, "w"), indent=1)
print(key, json.dumps(table[key])[:400])
