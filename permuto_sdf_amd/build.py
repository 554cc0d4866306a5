"""Build recipe: compile every HIP source under csrc/ for gfx950 into ONE C-ABI shared library,
``permuto_sdf_amd/lib/libpsdf_hip.so`` (in-tree so it travels with the repo snapshot).

hipcc cross-compiles without a GPU.  Flags:
  -ffp-contract=off      results do not depend on FMA fusion choices (matches the CPU oracle)
  -munsafe-fp-atomics    fp32 atomicAdd lowers to global_atomic_add_f32 instead of a CAS loop
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIBNAME = "libpsdf_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wno-unused-result",
         "--offload-arch=" + ARCH]


# per-file extra flags.  mlp_bwd_split.hip: let MFMAs write plain VGPRs, so that only the persistent accumulators live in
# AGPRs (otherwise every chain / transpose result is copied out with v_accvgpr_read and the kernel spills; the flag crashes
# this compiler on mlp_bwd.hip, hence per file)
EXTRA = {"mlp_bwd_split.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
         # -fno-slp-vectorize: the operand split relies on v_fma_mixlo/hi_f16 being selected for fp16(fma(x, 1, -high)); the SLP
         # vectoriser turns the two fmas of a pair into v_pk_fma_f32 + conversions (5 instead of 3 instructions per pair)
         "mlp_bwd_split_f16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"],
         # the BASELINE net's instantiation alone: ILP-first scheduling (the others spill under it, see the file)
         "mlp_bwd_split_double.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
# .hip files that include another .hip file (rebuilt when that one changes)
INCLUDES = {"mlp_bwd_split_double.hip": ["mlp_bwd_split.hip"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    srcs = sources()
    objs = []
    jobs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        deps = headers + [os.path.join(CSRC, d) for d in INCLUDES.get(os.path.basename(s), [])]
        stale = force or _newer(s, o) or any(_newer(h, o) for h in deps) or _newer(__file__, o)
        if stale:
            jobs.append([hipcc] + FLAGS + EXTRA.get(os.path.basename(s), []) + ["-I", CSRC, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[psdf build]", " ".join(cmd[-3:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    lib = os.path.join(LIBDIR, LIBNAME)
    if jobs or not os.path.exists(lib):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH] + objs + ["-o", lib])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
