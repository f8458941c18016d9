python -m pytest tests/test_gpu_modules.py tests/test_gpu_entry_points.py -q 2>&1 | tail -3
python bench.py --no-sweep --no-dense --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print({k:r.get(k) for k in ('value','ms_per_step','api_ms_per_step','api_equals_engine','spread','api_spread')})"
