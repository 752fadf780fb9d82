#!/bin/bash
# Development: another build of ONE translation unit linked with the objects of the regular build
# into safe_learning_amd/libslhip_<name>.so (selected at run time with SL_LIB_PATH).
#   tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]
# <source.hip> is a path (e.g. an older revision extracted from git) or a file name in csrc/.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
[ -f "$src" ] || src=safe_learning_amd/csrc/$src
base=$(basename "$src")
mkdir -p safe_learning_amd/build/variants
obj=safe_learning_amd/build/variants/${name}_${base}.o
extra=""
[ "$base" = sl_gp4.hip ] && extra="-mllvm -amdgpu-spill-vgpr-to-agpr=0"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden \
    -Iinclude -Isafe_learning_amd/csrc $extra "$@" -c "$src" -o "$obj"
objs=""
for o in safe_learning_amd/build/*.hip.o; do
    [ "$(basename "$o")" = "${base}.o" ] && continue
    objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o safe_learning_amd/libslhip_${name}.so $objs "$obj" -ldl
echo safe_learning_amd/libslhip_${name}.so
