#!/bin/bash
# round 4, GPU call m: final verification -- whole GPU suite, smoke, one-rank RCCL rows with the corrected stream picker, default bench line
cd /root/repo; out=gpurun_out/r4m; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -30 > $out/pytest.log; tail -4 $out/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke\|SMOKE" > $out/smoke.log; cat $out/smoke.log
row() { label=$1; shift; env "$@" MASTER_ADDR=127.0.0.1 MASTER_PORT=29587 timeout 300 python bench.py --no_cpu_baseline --mode ${MODE:-graph} > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$label.json').read().strip().splitlines()[-1]); c=d['config']; print('%-26s'%'$label', d['value'], 'img/s', d['ms_per_step'], 'ms/step  mode', c['mode'], ' rccl_ranks', c['rccl_ranks'], ' distinct queues', c['distinct_hw_queues_found'], ' capture_fallback', c['capture_fallback'], ' host ms', c['host_enqueue_ms_per_step'])
except Exception as e: print('$label failed', e)
PY
}
{ row no_group DD_X=0
row rccl1_end DD_BENCH_FORCE_DIST=1 DD_SEG_REDUCE=end
row rccl1_overlap DD_BENCH_FORCE_DIST=1 DD_SEG_REDUCE=overlap
MODE=eager row no_group_eager DD_X=0
MODE=eager row rccl1_eager DD_BENCH_FORCE_DIST=1; } | tee $out/rccl_rows.txt
timeout 400 python bench.py > $out/r04_bench_line_default.json 2> $out/bench_default.err; cut -c1-400 $out/r04_bench_line_default.json; grep "bench " $out/bench_default.err
