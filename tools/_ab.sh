for i in 1 2; do timeout 300 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['value'], r['ms_per_step'], r['roofline'])"
done
timeout 900 python3 -m pytest tests/test_gpu_e2e.py tests/test_gpu_conv_wino.py tests/test_gpu_conv.py tests/test_gpu_conv_train.py tests/test_gpu_train_step_parity.py -q 2>&1 | tail -3
timeout 300 python3 tools/bench_train.py --batch 16 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-250
