#!/bin/bash
# bash tools/build_variant.sh <name> <file.hip> [-Dflags...]: a copy of the library with ONE source rebuilt under extra flags,
# permuto_sdf_amd/lib/variants/libpsdf_<name>.so (git-ignored; selected with PSDF_LIB_PATH).  CPU side only.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; src=$2; shift 2
V=$R/permuto_sdf_amd/lib/variants; mkdir -p $V/obj_$name
extra=$(python - "$src" <<PY
import sys
sys.path.insert(0, "$R")
from permuto_sdf_amd import build
print(" ".join(build.FLAGS + build.EXTRA.get(sys.argv[1], [])))
PY
)
/opt/rocm/bin/hipcc $extra "$@" -I $R/permuto_sdf_amd/csrc -c $R/permuto_sdf_amd/csrc/$src -o $V/obj_$name/${src%.hip}.o
objs=""
for o in $R/permuto_sdf_amd/lib/obj/*.o; do
  b=$(basename $o); if [ "$b" = "${src%.hip}.o" ]; then objs="$objs $V/obj_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o $V/libpsdf_$name.so
echo $V/libpsdf_$name.so
