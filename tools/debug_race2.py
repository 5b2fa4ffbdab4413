#!/usr/bin/env python3
"""Race hunting, second stage: repeat one step of a fused mode through hip.energy_forces (explicit graph, known workspace layout)
and, when a repeat's forces differ from the first repeat's, list which of the workspace's arrays differ from their copies of
the first repeat, and in which rows / columns.
usage: python tools/debug_race2.py [B] [mode] [reps] [max_reports]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd import hip
from newtonnet_amd.models import NewtonNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else '6'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
max_reports = int(sys.argv[4]) if len(sys.argv) > 4 else 4
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.eval()
z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
os.environ['NNHIP_MOL_FUSED'] = mode
m = model._hip_model(0)
g = hip.build_graph(pos, cell, batch, 5.0, model.embedding_layers.edge_embedding.embedding.frequencies)
N, E, L = g.n_atoms, g.n_edges, m.n_layers
n = N // B
P = E // 2
lay = hip.workspace_layout(N, E, B, L)
pair_ptr = g.pair_ptr.cpu() if getattr(g, 'pair_ptr', None) is not None else None
print(f'N {N} E {E} pairs {P}; pair_ptr {"yes" if pair_ptr is not None else "no"}', flush=True)


def arrays(ws):
    out = {}
    def view(off, rows, cols):
        return ws[off: off + rows * cols * 4].view(torch.float32).view(rows, cols)
    for l in range(L):
        for nm, rows, cols in (('m', N, 128), ('msg', P, 128), ('h12', P, 256), ('phi1', P, 128), ('phi2', P, 128), ('a_mid', N, 128),
                               ('a_out', N, 128), ('f_out', N, 384), ('q', N, 128)):
            out[f'{nm}[{l}]'] = view(getattr(lay, nm)[l], rows, cols)
    for nm, rows, cols in (('g_x', P, 1), ('g_u', P, 3), ('g_a', N, 128), ('g_f', N, 384)):
        out[nm] = view(getattr(lay, nm), rows, cols)
    return out


first = None
reports = 0
res = None
for k in range(reps):
    res = hip.energy_forces(m, z, pos, cell, g, want_nodes=False, workspace=None if res is None else res['workspace'], out=res)
    f = res['forces']
    torch.cuda.synchronize()
    if first is None:
        first = (f.clone(), {nm: v.clone() for nm, v in arrays(res['workspace']).items()})
        continue
    if torch.equal(f, first[0]):
        continue
    d = (f - first[0]).abs().view(B, -1).amax(dim=1)
    bad = torch.nonzero(d > 0).flatten().tolist()
    print(f'rep {k}: forces differ in conformers {bad[:8]} (max {d.max():.2e})', flush=True)
    for nm, v in arrays(res['workspace']).items():
        ref = first[1][nm]
        if torch.equal(v, ref):
            continue
        dd = (v - ref).abs()
        rows = torch.nonzero(dd.amax(dim=1) > 0).flatten()
        r0 = int(rows[0])
        cols = torch.nonzero(dd[r0] > 0).flatten().tolist()
        if v.shape[0] == N:
            where = f'atoms {rows[:8].tolist()} of conformers {sorted(set((rows // n).tolist()))[:6]}'
        else:
            p0 = int(pair_ptr[bad[0] * n]) if pair_ptr is not None and bad else 0
            where = f'pair rows {rows[:10].tolist()} (molecule-local {[int(r) - p0 for r in rows[:10]]})'
        print(f'    {nm}: {rows.numel()} rows differ ({where}); first row {r0}: {len(cols)} cols {cols[:8]}..{cols[-1]}; '
              f'max |d| {dd.max():.3e} (|ref| max in row {ref[r0].abs().max():.3e})', flush=True)
        if nm.startswith('msg') and rows.numel() == 1 and len(cols) <= 32:
            c0 = cols[0] & ~3
            print('        got ', [f'{float(x):+.5e}' for x in v[r0, c0:c0 + 8]], flush=True)
            print('        want', [f'{float(x):+.5e}' for x in ref[r0, c0:c0 + 8]], flush=True)
            for dr in (-32, -8, 8, 32):
                if 0 <= r0 + dr < ref.shape[0]:
                    print(f'        row {dr:+d}', [f'{float(x):+.5e}' for x in ref[r0 + dr, c0:c0 + 8]], flush=True)
            ratio = (v[r0, cols] / ref[r0, cols]).tolist()
            print('        got / want over the differing cols:', [f'{x:.5f}' for x in ratio], flush=True)
    reports += 1
    if reports >= max_reports:
        break
print('done', flush=True)
