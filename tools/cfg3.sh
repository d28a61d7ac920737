set -e
mkdir -p gpurun_out/cfg3
timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --k 51 --err 0.01 > gpurun_out/cfg3/unmasked.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/cfg3/unmasked.json'))
print(d['ms_per_step'], d['config']['n_distinct_kmers'], d['config']['n_solid_kmers'], d['config']['ncontigs'])
for k,v in d['stage_ms'].items(): print("   %-45s %.3f"%(k,v))
PY
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --k 51 --err 0.01 --mask-errors > gpurun_out/cfg3/masked.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/cfg3/masked.json'))
print("masked:", d['ms_per_step'], d['config']['n_distinct_kmers'], d['config']['n_solid_kmers'], d['config']['ncontigs'], {k: round(v,2) for k,v in d['stage_ms'].items() if 'kernel' in k})
PY
