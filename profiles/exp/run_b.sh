python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --layers > gpurun_out/rb_ship.json 2> gpurun_out/rb_ship.txt
python -c "import json;d=json.load(open('gpurun_out/rb_ship.json'));print(d['value'], d['ms_per_step'], d['roofline'])"
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --dtype fp16 > gpurun_out/rb_ship_fp16.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/rb_ship_fp16.json'));print('fp16', d['value'])"
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --crop 240x320 > gpurun_out/rb_ship_240.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/rb_ship_240.json'));print('240x320', d['value'])"
