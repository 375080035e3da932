#!/bin/bash
# config 3 (10 000 mixed sites) with the items of a chunk longest first (the tree) against tools/variants/lib_base.so (site order)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out; : > gpurun_out/lpt_ab.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workflow.py tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_gpu_klib.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1
for round in 1 2; do for v in base tree; do
  if [ $v = base ]; then export PG_LIB=$R/tools/variants/lib_base.so; else unset PG_LIB; fi
  timeout 300 python bench.py --workload config3 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | V=$v python -c '
import json,sys,os
d=json.loads(sys.stdin.readline())
print(json.dumps({"variant": os.environ["V"], "sites_per_s": round(d["sites"]["sites_per_s"],1), "Mreads_s": round(d["value"]/1e6,3), "ms_per_step": round(d["ms_per_step"],3), "fill_ms": round(d["roofline"]["avg_launch_ms"],3), "launches": d["roofline"]["launches"], "verified": d["sites"].get("verified",{}).get("mismatches")}))' | tee -a gpurun_out/lpt_ab.jsonl
done; done
