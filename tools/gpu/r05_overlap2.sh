#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5w; mkdir -p $O
export TMPDIR=/tmp
for mode in one bothlow; do
  case $mode in one) E="PG_FILL_STREAMS=1";; bothlow) E="PG_FILL_STREAMS=2 PG_FILLS_LOW=1";; esac
  (cd /tmp && env $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$mode -o t -- python $R/bench.py --steps 6 --warmup 2 --reads 200000 --workspace-gib 8 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --e2e-steps 0 > $O/bench_$mode.json 2> $O/bench_$mode.err)
  f=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  echo "$mode $(python tools/overlap_check.py $f) value=$(python -c "import json;print(round(json.loads([l for l in open('$O/bench_$mode.json') if l.startswith('{')][-1])['value']))")" | tee -a $O/overlap.txt
done
for rep in 1 2 3; do for mode in one bothlow onelow; do
  case $mode in one) E="PG_FILL_STREAMS=1";; bothlow) E="PG_FILL_STREAMS=2 PG_FILLS_LOW=1";; onelow) E="PG_FILL_STREAMS=1 PG_FILLS_LOW=1";; esac
  env $E python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'mode': '$mode', 'rep': $rep, 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/e2e_ab.jsonl
done; done
for mode in one bothlow; do
  case $mode in one) E="PG_FILL_STREAMS=1";; bothlow) E="PG_FILL_STREAMS=2 PG_FILLS_LOW=1";; esac
  env $E python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --e2e-steps 0 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'mode': '$mode', 'headline': round(d['value']), 'ms_per_step': round(d['ms_per_step'],2), 'launches': d['roofline']['launches']}))" | tee -a $O/headline_ab.jsonl
done
