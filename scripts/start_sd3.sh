#!/usr/bin/env bash
# Same four-step recipe as the reference's scripts/start_sd3.sh, on the MI355X-native implementation (native MM-DiT, processors,
# rectified-flow inversions and transfer loop; the VAE and the CLIP / T5 text encoders stay stock and must be available locally).
export PYTHONPATH=$(pwd)
set -e
python src/sd3/run_content_inversion_sd3.py --content_path examples/contents/mallard-fly --output_path results/contents-inv --is_rf_solver
python src/sd3/run_style_inversion_sd3.py --style_path examples/styles/00033.png --output_path results/styles-inv --is_rf_solver
python src/mask_propagation.py --feature_path results/contents-inv/sd3/mallard-fly/features/inversion_feature_map_20_block_5_step.pt \
       --backbone sd3 --mask_path examples/masks/mallard-fly.png --output_path results/masks
NGPU=${NGPU:-1}        # NGPU=8 scripts/start_sd3.sh: one process per GPU, frames sharded, rank 0 writes the PNGs
RUN="python"
if [ "$NGPU" -gt 1 ]; then RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $NGPU --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29534}"; fi
$RUN src/sd3/run_video_style_transfer_sd3.py --content_inv_path results/contents-inv/sd3/mallard-fly/inversion \
       --style_inv_path results/styles-inv/sd3/00033/inversion --mask_path results/masks/sd3/mallard-fly --output_path results/stylizations
