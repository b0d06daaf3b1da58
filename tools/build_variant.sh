#!/bin/bash
# A/B build of ONE source file with extra -D flags into a second library (same ABI): tools/build_variant.sh <name> <file.hip> <flags...>
# -> image2text_amd/csrc/libi2t_<name>.so ; run with I2T_LIB=image2text_amd/csrc/libi2t_<name>.so
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../image2text_amd/csrc"
EXTRA=""
case $SRC in attention.hip|attention_g.hip) EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $EXTRA "$@" -c $SRC -o /tmp/variant_${NAME}.o
OBJS=""
for f in abi comm gemm norm attention elementwise conv conv_mfma decode sample attention_g family grouped llama lora vit fp8; do
  if [ "$f.hip" = "$SRC" ] || [ "$f.cpp" = "$SRC" ]; then OBJS="$OBJS /tmp/variant_${NAME}.o"; else OBJS="$OBJS $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libi2t_${NAME}.so $OBJS -ldl
echo libi2t_${NAME}.so
