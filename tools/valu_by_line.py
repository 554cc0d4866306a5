#!/usr/bin/env python3
"""Static vector-instruction count of one kernel by source line (CPU only; how round 5 found the waste in the encode kernels).

  python tools/valu_by_line.py permuto_sdf_amd/csrc/encode.hip encode_bwd_kernelILi3ELi2ELb1ELb0ELb1ELb0E [top]

compiles the file for gfx950 with -g -S (same flags as permuto_sdf_amd/build.py), takes the first kernel whose mangled name
contains the given substring, and prints per (file, line) of the .loc directives: VALU instructions, SALU instructions -- plus the
totals, the register counts and the opcode histogram.  Static counts: paths behind wave-uniform branches that a workload never
takes (general modulo, partial k-steps) are counted too; scheduling moves instructions across .loc boundaries, so a line's number
is where the compiler PUT the instruction.  Quarter-rate opcodes (transcendentals, 32-bit integer multiplies, v_mad_u64_u32) are
listed separately: each costs four issue slots."""
import collections
import os
import re
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
QUARTER = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_mul_lo_u32", "v_mul_hi_u32",
           "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_rcp_iflag_f32")


def main():
    src, key = os.path.abspath(sys.argv[1]), sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    from permuto_sdf_amd import build
    flags = build.FLAGS + build.EXTRA.get(os.path.basename(src), [])
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-g", "-I", os.path.join(R, "permuto_sdf_amd/csrc"), "--cuda-device-only",
                                                                 "-S", src, "-o", out], cwd=td)
        s = open(out).read()
    m = re.search(r"^(_Z\S*%s\S*):" % re.escape(key), s, re.M)
    if not m:
        sys.exit("no kernel matching %r" % key)
    k = s[m.start():]
    k = k[:k.index(".Lfunc_end")]
    files = {int(a): (c or b) for a, b, c in re.findall(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
    cur, valu, salu, ops = None, collections.Counter(), collections.Counter(), collections.Counter()
    for l in k.split("\n"):
        l = l.strip()
        loc = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
        if loc:
            cur = (os.path.basename(files.get(int(loc.group(1)), "?")), int(loc.group(2)))
            continue
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        op = l.split()[0]
        ops[op] += 1
        if op.startswith("v_"):
            valu[cur] += 1
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
            salu[cur] += 1
    print(m.group(1))
    tail = s[m.start():]
    for name in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy"):
        r = re.search(r"; %s: (\d+)" % name, tail)
        print("%s %s" % (name, r.group(1) if r else "?"), end="   ")
    print("\nVALU %d (quarter rate: %d)  SALU %d  s_nop %d  LDS %d  VMEM %d" % (
        sum(valu.values()), sum(n for o, n in ops.items() if o.startswith(QUARTER)), sum(salu.values()), ops["s_nop"],
        sum(n for o, n in ops.items() if o.startswith("ds_")), sum(n for o, n in ops.items() if o.startswith(("global_", "buffer_", "flat_")))))
    print("%-28s %6s %6s" % ("file:line", "VALU", "SALU"))
    for (f, ln), c in sorted(valu.items(), key=lambda x: -x[1])[:top]:
        print("%-28s %6d %6d" % ("%s:%d" % (f, ln), c, salu[(f, ln)]))
    print("opcodes:", ", ".join("%s %d" % kv for kv in ops.most_common(24)))


if __name__ == "__main__":
    main()
