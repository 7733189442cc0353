timeout -k 5 120 python bench.py --no-legs --no-cpu-baseline --config msrvtt_care --batch 16384 --steps 5 2>&1 | tail -1 | cut -c1-200
echo "rc=$?"
