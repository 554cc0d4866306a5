"""Instruction mix of one kernel (by mangled-name substring) from the gfx950 code object of an object file (no GPU needed):
python tools/kernel_isa_mix.py permuto_sdf_amd/lib/obj/mlp_bwd_split_f16.o mlp_bwd_split_f16_kernelILi3ELb1E"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj, tmp):
    fat, co = os.path.join(tmp, "x.fatbin"), os.path.join(tmp, "x.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.devnull])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)


def mix(text, needle):
    out = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1) if needle in m.group(1) else None
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        d = out.setdefault(cur, {})
        cls = ("mfma" if op.startswith("v_mfma") else "accvgpr" if "accvgpr" in op else "v_cvt" if op.startswith("v_cvt") else
               "v_perm" if op.startswith("v_perm") else "v_pk" if op.startswith("v_pk_") else "valu_other" if op.startswith("v_") else
               "lds" if op.startswith("ds_") else "waitcnt" if op == "s_waitcnt" else "s_nop" if op == "s_nop" else
               "salu" if op.startswith("s_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
        d[cls] = d.get(cls, 0) + 1
    return out


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        for k, d in mix(disassemble(sys.argv[1], tmp), sys.argv[2]).items():
            valu = sum(v for c, v in d.items() if c in ("v_cvt", "v_perm", "v_pk", "valu_other"))
            print(k[:80], "| VALU", valu, "|", " ".join("%s %d" % kv for kv in sorted(d.items())))
