for v in old new old new; do
  cp ab_tmp/libkws_$v.so ei-keyword-spotting_amd/libkws_mi355x.so
  echo -n "$v: "; python bench.py --no-cpu-baseline | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], [x['kernel_ms'] for x in d['also']])"
done
cp ab_tmp/libkws_new.so ei-keyword-spotting_amd/libkws_mi355x.so
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | head -3
