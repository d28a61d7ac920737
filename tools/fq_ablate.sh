set -e
for D in 0 1; do
  echo "SHK_DEBUG_FQ=$D: $(SHK_DEBUG_FQ=$D SKIP_HOST=1 ONLY_DEVICE=1 timeout -k 10 300 python tools/fastq_path.py 2>&1 | grep "run 1\|preprocess:" | cut -c1-200)"
done
