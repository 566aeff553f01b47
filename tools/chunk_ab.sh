QUICK="--no-cpu-baseline --no-other-leg --no-parity --no-configs"
for cs in 0 16 21 11 0; do
  python bench.py --steps 3 --warmup 1 $QUICK --chunk-seqs $cs 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d = json.loads(l[-1]); k = d.get('kernel_ms_per_step', {})
print('chunk $cs', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 40})"
done
