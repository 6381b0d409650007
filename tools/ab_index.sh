for V in 0 1024 512; do
  export HVPR_INDEX_FUSED=$V
  echo "=== FUSED=$V"
  bash tools/prof_serial.sh ser_$V 2>&1 | tail -6
  grep -o '"value": [0-9.]*' gpurun_out/ser_$V/stats_serial.log | head -1
  timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 30 --probe-steps 10 > gpurun_out/bench_$V.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("gpurun_out/bench_$V.log") if x.startswith("{")][-1]
r=json.loads(l)
print("value", r["value"], "latency", r["single_graph_latency_mode"], "group live", r["roofline"]["isolated_warm_us_live"], "stage", r["stage_ms"])
PY
done
