#!/bin/bash
# Tiled frames (the executables' default is the binary's 200-pixel tiles): one layer per launch on the planes (--fuse 0) against the
# pair kernel on the canvas of planes (--fuse 1), alternating bench runs on one box.  bash scripts/ab_tile_modes.sh > gpurun_out/ab_tile_modes.txt
cd "$(dirname "$0")/.."
for spec in "C2 200" "C2 100" "C2 400" "C3 200"; do
  set -- $spec
  for f in 0 1 0 1; do
    python3 bench.py --steps 150 --workload $1 --tile $2 --fuse $f --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 tile $2 fuse $f:', d['value'], 'frames/s; stages_ms', d['stages_ms'], '; body launch', d['roofline']['launch_us'], 'us x', d['roofline']['layers_per_launch'], 'layers')"
  done
done
