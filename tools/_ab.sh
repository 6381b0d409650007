for i in 1 2; do timeout 300 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['value'], r['ms_per_step'])"
done
timeout 600 python3 -m pytest tests/test_gpu_e2e.py tests/test_gpu_conv_wino.py tests/test_gpu_conv.py -q 2>&1 | tail -3
