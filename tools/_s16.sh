cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IPK_BENCH_NO_LIVE_PMC=1 python bench.py --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print({k:r.get(k) for k in ('kernel_ms','frac','ceiling_ms','ceiling_frac_of_peak','frac_of_ceiling','copy_ceiling_GBps','mix_ceiling_GBps','mix_ceiling_frac_of_peak','frac_of_mix_ceiling')})"
