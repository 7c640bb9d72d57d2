#!/bin/bash
# Every alternate library tools/next_gpu_session.sh runs (tools/_var/, git-ignored, travels with the gpurun snapshot).  The experimental
# ones are probe builds as well (-DSTATTN_PROBES: their A/Bs use tool switches such as STATTN_SHARED_MIN).
set -e
cd "$(dirname "$0")/.."
tools/build_variant.sh dpp "-DSTATTN_DPP_REDUCE=1"
tools/build_variant.sh exp "-DSTATTN_EXPERIMENTAL=1 -DSTATTN_PROBES" attn.hip bwd.hip
tools/build_variant.sh expdpp "-DSTATTN_EXPERIMENTAL=1 -DSTATTN_PROBES -DSTATTN_DPP_REDUCE=1"
tools/build_variant.sh wps2 "-DSTATTN_BWD_BF16_WPS16=2" bwd.hip
tools/build_variant.sh fwdsched "-DSTATTN_BF16_FWD_SCHED=1" attn.hip
tools/build_variant.sh pnv2 "-DSTATTN_PN_V2=1" panel.hip panelw.hip
tools/build_tools_lib.sh
mkdir -p tools/bin
for v in 0 1; do hipcc -O3 --offload-arch=gfx950 -DSTATTN_DPP_REDUCE=$v -I video-description-with-spatial-temporal-attention_amd/csrc tools/probes/dpp_check.hip -o tools/bin/dpp_check$v; done
ls -la tools/_var/*.so tools/bin/dpp_check*
