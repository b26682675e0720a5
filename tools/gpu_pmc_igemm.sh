#!/bin/bash
# PMC passes (each counter set in its own run) of the ResNet trunk's kernels on single shapes: bash tools/gpu_pmc_igemm.sh <tag>
tag=${1:-pmc_igemm}
for l in layer2 layer4; do bash tools/pmc_kernel.sh $tag/$l "conv_igemm_h2_kernel<128, 128, 2, true>" python tools/pmc_igemm_target.py $l > gpurun_out/$tag.$l.log 2>&1; tail -30 gpurun_out/$tag.$l.log | grep -E "mfma_busy|l2_hit|fabric_bytes\"|kernel_ms|effective_clock|lds_bank"; done
bash tools/pmc_kernel.sh $tag/layer1 "conv_igemm_h2_kernel<256, 64, 2>" python tools/pmc_igemm_target.py layer1 > gpurun_out/$tag.layer1.log 2>&1; grep -E "mfma_busy|l2_hit|fabric_bytes\"|kernel_ms|effective_clock|lds_bank" gpurun_out/$tag.layer1.log
bash tools/pmc_kernel.sh $tag/stem "conv_stem_pool_patch_kernel" python tools/pmc_igemm_target.py stem > gpurun_out/$tag.stem.log 2>&1; grep -E "mfma_busy|l2_hit|fabric_bytes\"|kernel_ms|effective_clock|lds_bank" gpurun_out/$tag.stem.log
