#!/bin/bash
# round 5, call 28: soak of the benched schedule on the final kernels -- the bit-identity / determinism tests 8 times over (the matrix-pipe fronts run beside the GEMMs
# and each other on two and three streams), then the packed-fp32 reproducer once more
out=gpurun_out/r05ae; mkdir -p $out
for i in 1 2 3 4 5 6 7 8; do
timeout 900 python -m pytest tests -m gpu -x -q -k "bit_identical or full_batch_properties or determin" > $out/soak_$i.txt 2>&1; echo "round $i rc $? $(tail -1 $out/soak_$i.txt | cut -c1-80)"
done | tee $out/soak.txt
