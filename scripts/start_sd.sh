#!/usr/bin/env bash
# Same four-step recipe as the reference's scripts/start_sd.sh, on the MI355X-native implementation.
export PYTHONPATH=$(pwd)
set -e
python src/sd/run_content_inversion_sd.py --content_path examples/contents/mallard-fly --output_path results/contents-inv --is_opt
python src/sd/run_style_inversion_sd.py --style_path examples/styles/00033.png --output_path results/styles-inv
python src/mask_propagation.py --feature_path results/contents-inv/sd/mallard-fly/features/inversion_feature_map_2_block_301_step.pt \
       --backbone sd --mask_path examples/masks/mallard-fly.png --output_path results/masks
# NGPU=8 scripts/start_sd.sh shards the clip's frames over the GPUs of the node (one process per GPU; rank 0 writes the PNGs);
# SMOOTHER=pixel|latent adds the sliding-window flow smoothing (BASELINE config 3)
NGPU=${NGPU:-1}
RUN="python"
if [ "$NGPU" -gt 1 ]; then RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $NGPU --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29533}"; fi
$RUN src/sd/run_video_style_transfer_sd.py --content_inv_path results/contents-inv/sd/mallard-fly/inversion \
       --style_inv_path results/styles-inv/sd/00033/inversion --mask_path results/masks/sd/mallard-fly --output_path results/stylizations \
       --smoother ${SMOOTHER:-none} --content_path examples/contents/mallard-fly
