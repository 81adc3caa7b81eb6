#!/bin/bash
# usage: scripts/video_Nx.sh SOURCE_DIR OUTPUT_DIR DS_FACTOR N [NGPUS]   (reference scripts/video_Nx.sh:1-13)
SOURCE_PATH=$1
OUTPUT_PATH=$2
DS_FACTOR=$3
N=$4
NGPUS=${5:-1}
HERE="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$NGPUS" -gt 1 ]; then
  LAUNCH="python -m torch.distributed.run --nnodes=1 --nproc-per-node $NGPUS --master-addr 127.0.0.1 --master-port 29511"
else
  LAUNCH="python"
fi
$LAUNCH "$HERE/src/video_Nx.py" \
    --source-path "$SOURCE_PATH" \
    --output-path "$OUTPUT_PATH" \
    --ds-factor "$DS_FACTOR" \
    --N "$N" \
    -m="$HERE/configs/gimmvfi/gimmvfi_r_arb.yaml" \
    -l='pretrained_ckpt/gimmvfi_r_arb_lpips.pt' \
    --eval
