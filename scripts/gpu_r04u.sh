#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04u
mkdir -p $OUT
for w in 0 1 3; do
  ME_AMD_HOST=python WGRAD_WPC=$w timeout 300 python scripts/unet_layers.py > $OUT/layers_wpc$w.log 2>&1
  echo "wpc=$w $(grep '^step' $OUT/layers_wpc$w.log)"
done
