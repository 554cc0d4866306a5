"""Main loop of a kernel, read from the gfx950 code object of an object file (no GPU needed): instruction mix, the order of the
instruction classes as one string (M = MFMA, v = VALU, t = transcendental, a = v_accvgpr move, l = LDS, g = global / scratch,
s = SALU, n = s_nop, w = s_waitcnt), runs of back-to-back MFMAs, and a crude single-wave timing model:

    in-order issue; 4 cycles per VALU instruction (8 for v_exp / v_rcp / ...), 2 per SALU, 8 per 128-bit LDS access;
    an MFMA 16x16x32 occupies the matrix pipe for 16 cycles (the next one waits for it) and its result is ready 2 cycles later;
    an instruction that reads a register an MFMA is still producing waits for it; s_waitcnt lgkmcnt waits for LDS data
    (LAT cycles after issue); s_nop N costs N + 1.

The model ignores everything else (other waves, DMA, bank conflicts, the clock).  Round 4 used it to rank build variants of
mlp_bwd_split_f16_kernel before they went to the GPU: it reproduced the measured 17 600 cycles per tile of the round-3 kernel
as 16 000 = the plain SUM of VALU, MFMA and wait times (one wave overlaps nothing), over-estimated what requesting weights
earlier would buy (-9 % predicted, -1.6 % measured) and was right about the direction of every instruction-count change.

    python tools/isa_loop_model.py permuto_sdf_amd/lib/obj/mlp_bwd_split_f16.o mlp_bwd_split_f16_kernelILi3E [-p] [-r]
      -p  print the class string      -r  model cycles per 60-instruction region
The loop = the longest backward branch of the kernel.  Branches are taken as fall-through (not-taken paths are counted)."""
import collections
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_isa_mix as K  # noqa: E402


def regs(tok):
    out = []
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        else:
            out.append((m.group(4), int(m.group(5))))
    return out


def load(obj, needle):
    """[(opcode, operand text, address)] of the main loop of the kernel whose mangled name contains `needle`"""
    with tempfile.TemporaryDirectory() as t:
        txt = K.disassemble(obj, t)
    ins, cur = [], False
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = needle in m.group(1)
            continue
        if not cur:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):\s*((?:[0-9A-F]{8}\s*)+)", line)
        if m:
            ins.append((m.group(1), m.group(2), int(m.group(3), 16)))
    if not ins:
        raise SystemExit("no kernel matches %r" % needle)
    a2i = {a: i for i, (_, _, a) in enumerate(ins)}
    best = None
    for i, (op, args, a) in enumerate(ins):
        if op.startswith(("s_branch", "s_cbranch")):
            off = int(args.split()[0])
            off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + off * 4
            if off < 0 and tgt in a2i and (best is None or i - a2i[tgt] > best[1] - best[0]):
                best = (a2i[tgt], i)
    if best is None:
        raise SystemExit("kernel has no loop")
    return ins[best[0]:best[1] + 1]


def cls(op):
    return ("M" if op.startswith("v_mfma") else "a" if "accvgpr" in op else
            "t" if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt")) else "v" if op.startswith("v_") else
            "l" if op.startswith("ds_") else "w" if op == "s_waitcnt" else "n" if op == "s_nop" else
            "s" if op.startswith("s_") else "g")


def sim(loop, lat=96):
    t, mfree, ready, lds_out = 0, 0, {}, []
    stall = collections.Counter()
    for op, args, _ in loop:
        toks = [x.strip() for x in args.split(",")]
        c = cls(op)
        if c == "M":
            dep = max([ready.get(r, 0) for tk in toks[1:4] for r in regs(tk)] + [0])
            st = max(t, mfree, dep)
            stall["mfma_pipe"] += max(0, min(mfree, st) - t)
            stall["mfma_dep"] += max(0, st - max(t, mfree))
            t, mfree = st + 4, st + 16
            for r in regs(toks[0]):
                ready[r] = st + 18
        elif c == "n":
            t += int(toks[0]) + 1
        elif c == "w":
            if "lgkmcnt" in args and lds_out:
                n = int(re.search(r"lgkmcnt\((\d+)\)", args).group(1))
                must = lds_out[:len(lds_out) - n] if n < len(lds_out) else []
                if must:
                    stall["lds"] += max(0, max(must) - t)
                    t = max(t, max(must))
                lds_out = lds_out[len(lds_out) - n:] if n else []
            t += 21 if "vmcnt" in args else 1
        elif c == "l":
            t += 8 if "b128" in op else 4
            lds_out.append(t + lat)
        elif c == "s":
            t += 2
        elif c == "g":
            t += 32
        else:
            dep = max([ready.get(r, 0) for tk in toks[1:] for r in regs(tk)] + [0])
            if dep > t:
                stall["valu_dep"] += dep - t
                t = dep
            t += 8 if c == "t" else 4
    return t, dict(stall)


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-")]
    pos = [a for a in sys.argv[1:] if not a.startswith("-")]
    loop = load(pos[0], pos[1])
    s = "".join(cls(op) for op, _, _ in loop)
    print("loop instructions", len(s), dict(collections.Counter(s)))
    print("MFMA runs (length: count)", sorted(collections.Counter(len(m.group(0)) for m in re.finditer(r"M+", s)).items()))
    print("model cycles per iteration %d, stalls %s" % sim(loop))
    if "-p" in flags:
        for i in range(0, len(s), 140):
            print(s[i:i + 140])
    if "-r" in flags:
        prev = 0
        for i in range(60, len(loop) + 60, 60):
            c, _ = sim(loop[:min(i, len(loop))])
            print("%5d %5d  %s" % (i - 60, c - prev, s[i - 60:i]))
            prev = c
