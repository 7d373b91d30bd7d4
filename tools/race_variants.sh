# Experimental builds of the whole library for tools/race_bisect.py (selected with INKLAYER_HIP_LIB): which property of the
# window-attention kernel disturbs kernels of another stream?  Built HERE into tools/micro/.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -w"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  d=/tmp/ink_var_$name; mkdir -p $d
  for f in inklayer_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    if [ "$b" = attention ] || [ "$b" = attention_win ] || [ ! -f inklayer_amd/lib/obj/$b.o ]; then
      /opt/rocm/bin/hipcc $F $flags -c $f -o $d/$b.o &
    else
      cp inklayer_amd/lib/obj/$b.o $d/$b.o
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libink_$name.so $d/*.o
  echo "built tools/micro/libink_$name.so ($flags)"
done
