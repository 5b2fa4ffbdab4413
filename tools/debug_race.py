#!/usr/bin/env python3
"""Repeat a fused mode at a batch size and list the conformers whose forces differ from the row path (race hunting).
usage: python tools/debug_race.py [B] [mode] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else '6'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.eval()
z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
os.environ['NNHIP_MOL_FUSED'] = '0'
for _ in range(2):
    ref = model(z, pos, cell, batch)
f0, e0 = ref.gradient_force.clone(), ref.energy.clone()
os.environ['NNHIP_MOL_FUSED'] = mode
prev = None
first = None
for k in range(reps):
    out = model(z, pos, cell, batch)
    f = out.gradient_force.clone()
    fw = (out.energy.clone(), out.atom_node.clone(), out.force_node.clone())
    if first is None:
        first = fw
    else:
        names = ('energy', 'atom_node', 'force_node')
        diff = [nm for nm, a_, b_ in zip(names, fw, first) if not torch.equal(a_, b_)]
        if diff:
            rows = torch.nonzero((fw[2] - first[2]).abs().reshape(B, -1).amax(dim=1) > 0).flatten().tolist()
            print(f'rep {k}: FORWARD outputs differ from rep 0: {diff}; force_node differs in conformers {rows[:8]}', flush=True)
    d = (f - f0).abs().view(B, -1).amax(dim=1)
    bad = torch.nonzero(d > 2e-6).flatten().tolist()
    same = None if prev is None else bool(torch.equal(f, prev))
    print(f'rep {k}: max |dF| {d.max():.2e}, conformers beyond 2e-6: {bad[:12]}{"..." if len(bad) > 12 else ""} ({len(bad)}); bitwise equal to previous rep: {same}', flush=True)
    prev = f
    if os.environ.get('M2_DBG_CHECK') == '1':
        import ctypes as C
        from newtonnet_amd import hip
        buf = (C.c_int * (1 + 64 * 8))()
        L = hip.lib()
        L.nnhip_m2_dbg_read.argtypes = [C.POINTER(C.c_int)]
        torch.cuda.synchronize()
        if L.nnhip_m2_dbg_read(buf) == 0 and buf[0] > 0:
            import struct
            print(f'rep {k}: LDS CHECK: {buf[0]} mismatches', flush=True)
            for q in range(min(buf[0], 12)):
                e = buf[1 + 8 * q: 9 + 8 * q]
                gf_, wf_ = struct.unpack('f', struct.pack('i', e[6]))[0], struct.unpack('f', struct.pack('i', e[7]))[0]
                print(f'     code {e[0]} (1 = X tile, 2 = phi tile, 3 = msg tile of pass 1) mol {e[1]} mlp {e[2]} tile {e[3]} row {e[4]} col {e[5]} got {gf_:.6e} want {wf_:.6e}', flush=True)
