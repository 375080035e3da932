#!/bin/bash
# A/B of library variants on ONE box (boxes differ by 2-3 %): tools/gpu/ab_variants.sh name1 name2 ...
# (tools/variants/lib_<name>.so from tools/build_variant.sh / build_patched_variant.sh / build_nonop_variant.sh).  Three
# interleaved rounds of the fill + traceback alone (tools/fill_probe.py), then one short bench per variant (the step with the
# traceback under the next fill, verification against the reference's gssw.c included).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/ab; mkdir -p $O; : > $O/ab.jsonl
for round in 1 2 3; do for v in "$@"; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a $O/ab.jsonl
done; done
for v in "$@"; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --sites-steps 0 --stream-batches 0 --e2e-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "
import json; d=json.loads(open('$O/bench_$v.json').readline()); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('verified'))"
done
