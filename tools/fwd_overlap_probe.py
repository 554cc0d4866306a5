"""Round 5 probe: do the encode forward (L2-line bound, 28 registers, no LDS) and the MLP forward (matrix pipe, 133 registers,
LDS image) overlap when they run on two streams?  Both at the bench size, independent buffers; serial time, concurrent time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from permuto_sdf_amd.encoding import encode_forward_raw  # noqa: E402
from permuto_sdf_amd.hotpath import SdfHotPath  # noqa: E402
from permuto_sdf_amd.mlp import mlp_forward_raw, pack_params, f16_forward_supported  # noqa: E402

dev = torch.device("cuda:0")
hp = SdfHotPath(nr_levels=16, hidden=64, out_channels=1, device=dev, seed=0)
rs, rgb, aux = bench.make_batch(dev, 7)
pos = rs.samples_pos
N = pos.shape[0]
enc, mlp = hp.enc, hp.mlp
win = torch.ones(16, device=dev)
ws, bs = [l.weight for l in mlp.layers], [l.bias for l in mlp.layers]
f16 = f16_forward_supported(mlp.dims)
packed = pack_params(mlp.dims, ws, bs, f16=f16)
featA = encode_forward_raw(enc.cfg, pos, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win)
featB = featA.clone()
outA = torch.empty_like(featA)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def enc_():
    encode_forward_raw(enc.cfg, pos, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win, out=outA)


def mlp_():
    mlp_forward_raw(mlp.dims, featB, packed, f16=f16)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def both():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    sa.wait_event(ev)
    sb.wait_event(ev)
    with torch.cuda.stream(sa):
        enc_()
        ea = torch.cuda.Event()
        ea.record(sa)
    with torch.cuda.stream(sb):
        mlp_()
        eb = torch.cuda.Event()
        eb.record(sb)
    cur.wait_event(ea)
    cur.wait_event(eb)


te, tm = timed(enc_), timed(mlp_)
tb = timed(both)
print("encode fwd %.3f ms, mlp fwd %.3f ms, sum %.3f; on two streams %.3f ms" % (te, tm, te + tm, tb))
# chunked pipeline emulation: 4 chunks, encode of chunk k+1 beside the MLP of chunk k (independent buffers of chunk size)
C = 4
Nc = N // C
posc = [pos[i * Nc:(i + 1) * Nc].contiguous() for i in range(C)]
featc = [encode_forward_raw(enc.cfg, p, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win) for p in posc]


def serial_chunks():
    for i in range(C):
        encode_forward_raw(enc.cfg, posc[i], enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win, out=featc[i])
        mlp_forward_raw(mlp.dims, featc[i], packed, f16=f16)


def piped_chunks():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    sa.wait_event(ev)
    sb.wait_event(ev)
    evs = []
    for i in range(C):
        with torch.cuda.stream(sa):
            encode_forward_raw(enc.cfg, posc[i], enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win, out=featc[i])
            e = torch.cuda.Event()
            e.record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(e)
            mlp_forward_raw(mlp.dims, featc[i], packed, f16=f16)
    eb = torch.cuda.Event()
    eb.record(sb)
    cur.wait_event(eb)


print("4 chunks: serial %.3f ms, pipelined on two streams %.3f ms" % (timed(serial_chunks), timed(piped_chunks)))
