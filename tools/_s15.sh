cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wh in "10000 10000" "2560 39062" "5120 19531" "20480 4882" "40960 2441" "6000 4000" "12000 8000" "16384 16384"; do set -- $wh
  python bench.py --no-cpu-baseline --no-check --width $1 --height $2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; mp=$1*$2/1e6
print('%6d x %6d  %.0f MP  kernel %.4f ms (%.3f ns/MP)  skeleton %.4f ms (%.3f ns/MP, %.3f of peak)' % ($1,$2,mp,r['kernel_ms'],r['kernel_ms']*1e3/mp,r['ceiling_ms'],r['ceiling_ms']*1e3/mp,r['ceiling_frac_of_peak']))"
done
