for r in 1 2; do
for kw in "20 5" "40 8" "20 8" "40 5"; do
  set -- $kw
  out=$(timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
  echo "steps $1 warmup $2: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
done
