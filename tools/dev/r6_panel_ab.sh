#!/bin/bash
# round 6: the Cholesky panel step, before (panel_dev0.bin = HEAD's kernel) / after, phase stamps + the C3 fit
cd /root/repo
for i in 1 2; do
  echo "== baseline (panel_dev0) run $i"; tools/dev/panel_dev0.bin | tail -13
  echo "== new (panel_dev) run $i"; tools/dev/panel_dev.bin | tail -13
done
python tools/linalg_bench.py 2>&1 | tail -12
python tools/fit_only.py c3 z 2>&1 | tail -4
