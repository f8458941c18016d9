python bench.py --gpus 2 --debug-single-device --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print({k:r.get(k) for k in ('value','ms_per_step','n_gpus','ranks_seen','ms_per_step_ranks','slowest_rank_per_block','spread')})"
python bench.py --config 4 --gpus 2 --debug-single-device --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print({k:r.get(k) for k in ('value','ms_per_step','n_gpus','ranks_seen','ms_per_step_ranks','slowest_rank_per_block','collective_ms','rows_per_block')})"
