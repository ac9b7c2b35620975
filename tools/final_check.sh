python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "conv2d_kernel or fused_epilogue" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r05_final6_bf16x3_default.json 2>/dev/null
python - <<PY
import json
b=json.loads(open("gpurun_out/bench_r05_final6_bf16x3_default.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["traffic"], b["fp32_exact"]["value"])
PY
