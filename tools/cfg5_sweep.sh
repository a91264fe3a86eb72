#!/bin/bash
# sweep of the sort-stage knobs of the cell-order pipeline on cfg5 (kernel-trace per setting)
for cfg in "1024 512 2048" "512 512 2048" "256 1024 2048" "128 1024 2048" "256 1024 256" "256 1024 512" "256 1024 1024" "1024 512 256"; do
  set -- $cfg
  export BSK_BIN_CHUNKS=$1 BSK_BIN_BLOCK=$2 BSK_UNP_GRID=$3
  echo "== chunks $1 block $2 unpermute grid $3"
  bash tools/cfg5_prof.sh sweep | grep "bin_\|eval_cell"
done
