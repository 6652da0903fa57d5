cd profiles/exp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 valu_bench.hip -o /tmp/valu_bench && /tmp/valu_bench
