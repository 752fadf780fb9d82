#!/bin/bash
# Development (-DSL_DIAG: the unit reads the work-skipping switches SL_GP4_SKIP / SL_BM_FLAGS /
# SL_B4P_FLAGS, which the shipped library does not have): another build of sl_gp4.hip (or any one unit) linked with the objects of the regular
# build into safe_learning_amd/libslhip_<name>.so (selected at run time with SL_LIB_PATH).
#   tools/build_variant.sh <name> <unit> <source.hip> [extra hipcc flags...]
# <unit>: the object stem it replaces (sl_gp4_d4, sl_bellman4, ...) or "sl_gp4_all" for a source that
# defines every dimension itself (an older revision extracted from git: replaces sl_gp4_d1..4).
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; src=$3; shift 3
[ -f "$src" ] || src=safe_learning_amd/csrc/$src
mkdir -p safe_learning_amd/build/variants
obj=safe_learning_amd/build/variants/${name}_${unit}.o
extra=""
case $unit in sl_gp4_*) extra="-mllvm -amdgpu-spill-vgpr-to-agpr=0";; esac
case $unit in sl_gp4_d?) extra="$extra -DSL_GP4_DIM=${unit#sl_gp4_d}";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC ${SL_VARIANT_VIS--fvisibility=hidden} \
    -DSL_DIAG -Iinclude -Isafe_learning_amd/csrc $extra "$@" -c "$src" -o "$obj"
objs=""
for o in safe_learning_amd/build/*/*.o; do
    stem=$(basename "$o" .o)
    case $o in */sl_no_*/*|*/variants/*|*-hip-amdgcn*|*-host-*) continue;; esac
    [ "$stem" = "$unit" ] && continue
    [ "$unit" = sl_gp4_all ] && case $stem in sl_gp4_d?) continue;; esac
    objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o safe_learning_amd/libslhip_${name}.so $objs "$obj" -ldl
echo safe_learning_amd/libslhip_${name}.so
