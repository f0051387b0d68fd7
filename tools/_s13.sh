cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/bench_live.json 2> gpurun_out/bench_live.err
tail -3 gpurun_out/bench_live.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_live.json').read().strip().splitlines()[-1])
r=d['roofline']; print({k:r.get(k) for k in ('traffic','traffic_source','traffic_detail','traffic_over_algorithmic','traffic_live_failed','kernel_ms','frac')})
PY
