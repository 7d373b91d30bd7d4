# Timing variants of the fused FFN kernel (csrc/ffn_fused.hip), built HERE (hipcc cross-compiles) into tools/micro/*.so
# and timed on the GPU box by tools/ffn_time.py.  The NODMA / NOREAD variants compute garbage on purpose.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -w"
for v in "base:" "nodma:-DINK_FFN_NODMA" "noread:-DINK_FFN_NOREAD" "nodma_noread:-DINK_FFN_NODMA -DINK_FFN_NOREAD" "depth3:-DINK_FFN_DEPTH=3" "depth10:-DINK_FFN_DEPTH=10" $EXTRA; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $F $flags inklayer_amd/csrc/ffn_fused.hip -o tools/micro/ffn_$name.so
  echo "built ffn_$name.so ($flags)"
done
