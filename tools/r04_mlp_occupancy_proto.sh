#!/bin/bash
# Round 4: upper-bound measurement for the workgroup-cooperative MLP backward (DESIGN.md "Next").  Builds mlp_bwd_split_f16.hip
# with -DPSDF_F16_PROTO_OCC (eight waves per workgroup, four dW accumulators per layer and wave: WRONG sums, the same MFMA / VALU
# instruction mix, none of the LDS exchange the real design needs) and times the bench step's MLP backward with it.
# A measurement build: the library it produces is never shipped (lib/variants/ is git-ignored).
# The -DPSDF_F16_PROTO_OCC switch lived in csrc/mlp_bwd_split_f16.hip up to commit 2b8b417 (it left with the rewrite of the
# kernel's instruction stream in the second half of round 4): check that commit out to repeat the measurement.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R/permuto_sdf_amd
F="-O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1"
mkdir -p lib/variants
[ -f lib/variants/libpsdf_f16occ.so ] || { /opt/rocm/bin/hipcc $F -DPSDF_F16_PROTO_OCC -I csrc -c csrc/mlp_bwd_split_f16.hip -o lib/variants/mlp_bwd_split_f16_occ.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls lib/obj/*.o | grep -v "/mlp_bwd_split_f16.o") lib/variants/mlp_bwd_split_f16_occ.o -o lib/variants/libpsdf_f16occ.so; }
cd $R
for v in "" "PSDF_LIB_PATH=$R/permuto_sdf_amd/lib/variants/libpsdf_f16occ.so"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('${v:-default}', 'ms/step %.4f' % d['ms_per_step'], d['kernel_ms'])"
done
