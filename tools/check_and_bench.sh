timeout -k 10 200 python -u tools/diag_c5.py 0 > gpurun_out/diag_c5_m.txt 2>&1; grep -E " rep|differ:" gpurun_out/diag_c5_m.txt | cut -c1-60; timeout -k 10 800 python -u -m pytest tests/ -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log; timeout -k 10 200 python bench.py --no-c5 --no-other-configs --no-cpu-baseline > gpurun_out/bench_m.json 2>/dev/null; python -c "
import json
d=json.loads(open('gpurun_out/bench_m.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['kernel_ms'])"
