"""Register / LDS / scratch use of every kernel in the library, read from the gfx950 code objects (no GPU needed).
python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import build  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj, tmp):
    fat, co = os.path.join(tmp, "x.fatbin"), os.path.join(tmp, "x.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.devnull])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out = []
    for block in re.split(r"\n  - \.", notes):
        name = re.search(r"\.name:\s+(\S+)", block)
        if name:
            f = {k: int(v) for k, v in re.findall(r"\.?(\w+):\s+(\d+)\s*$", block, flags=re.M)}
            out.append((name.group(1), f))
    return out


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return p.stdout.splitlines() if p.returncode == 0 else names
    except FileNotFoundError:
        return names


def waves(v):      # 512 registers per SIMD lane, allocation granule 8, at most 8 waves
    return min(8, 512 // max(8, (v + 7) // 8 * 8))


def main():
    build.build(verbose=False)
    print("%-24s %-92s %5s %5s %5s %6s %7s %8s %6s" % ("file", "kernel", "regs", "agpr", "sgpr", "spill", "scratch", "LDS(st)", "w/SIMD"))
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(os.listdir(build.OBJDIR)):
            if not o.endswith(".o"):
                continue
            ks = kernels(os.path.join(build.OBJDIR, o), tmp)
            for (_, f), d in zip(ks, demangle([k for k, _ in ks])):
                d = re.sub(r"\(anonymous namespace\)::", "", d)
                d = re.sub(r"^void ", "", d)
                d = re.sub(r"\(.*$", "", d)
                print("%-24s %-92s %5d %5d %5d %6d %7d %8d %6d" % (o[:-2], d[:92], f["vgpr_count"], f.get("agpr_count", 0), f["sgpr_count"],
                                                                   f["vgpr_spill_count"], f["private_segment_fixed_size"],
                                                                   f["group_segment_fixed_size"], waves(f["vgpr_count"])))
    print("\nregs = unified VGPR + AGPR count per lane; w/SIMD = waves per SIMD the register count allows (LDS and launch bounds may lower it);"
          "\nLDS(st) = static LDS only (most kernels here size their LDS at launch)")


if __name__ == "__main__":
    main()
