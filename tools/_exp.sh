python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_now.json; python -c "
import json; d=json.load(open('gpurun_out/bench_now.json')); print(d['ms_per_step'], d['kernels_ms'], d['value'], d['roofline']['frac'], d['regimes'])"
