#!/bin/bash
# Development aid: variant libraries of the window attention kernel's load/store schedule for same-box A/B runs.
# usage: tools/win_variants.sh   (writes inklayer_amd/lib/libinklayer_hip_w<N>.so; run tools/attn_time.py with INKLAYER_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
python -m inklayer_amd.build > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
i=0
for defs in "-DINK_WIN_KV_STEP=5" "-DINK_WIN_KV_STEP=6" "-DINK_WIN_ST_STEP=2" "-DINK_WIN_ST_STEP=4 -DINK_WIN_KV_STEP=5" "-DINK_WIN_QA_EARLY=1" "-DINK_WIN_QA_EARLY=1 -DINK_WIN_KV_STEP=5"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc $F $defs -c inklayer_amd/csrc/attention_win.hip -o /tmp/attention_win_v$i.o
  objs=$(ls inklayer_amd/lib/obj/*.o | grep -v attention_win.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o inklayer_amd/lib/libinklayer_hip_w$i.so $objs /tmp/attention_win_v$i.o
  echo "w$i: $defs"
done
