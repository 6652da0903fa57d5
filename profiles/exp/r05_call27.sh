#!/bin/bash
# round 5, final check of the committed tree: build() as the driver runs it, smoke(), the whole GPU suite, the driver's bench command
out=gpurun_out/r05ac; mkdir -p $out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -2 $out/pytest.txt | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; python -c "
import json; j=json.load(open('$out/bench.json')); print(j['value'], j['roofline']['traffic'], j['roofline']['frac'], j['cpu_baseline']['value'])"
