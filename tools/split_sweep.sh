for L in 5 6 7 8; do
  SHK_SPLIT_LOG=$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stage_ms']; print('split_log $L', round(d['ms_per_step'],3), {k: round(s[k],3) for k in ('collapse_succ_split','collapse_walk','collapse_rank_device','collapse_emit','collapse_n_splitters_x1e-3')})"
done
