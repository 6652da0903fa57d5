python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile | cut -c1-140
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
