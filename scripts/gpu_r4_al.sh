#!/bin/bash
# round 4, GPU call al: the whole GPU suite + smoke + default bench on the last commit
cd /root/repo; out=/root/repo/gpurun_out/r4al; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 1500 python -u -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1 < /dev/null; echo "rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | grep "smoke\|SMOKE" > $out/smoke.log; tail -1 $out/smoke.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err < /dev/null; cut -c1-200 $out/bench_default.json
