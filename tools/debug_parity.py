#!/usr/bin/env python3
"""Run one golden case through the HIP path on the GPU and print, tensor by tensor, the max error of every
workspace intermediate against the fp64 CPU trace (tests/trace.py).  Localises a parity failure to a kernel.

usage: python tools/debug_parity.py [case ...]     (default: aspirin8_rand mixed_rand pbc216_rand)
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from newtonnet_amd import hip  # noqa: E402
from newtonnet_amd.models import NewtonNet  # noqa: E402
from tests import trace, util  # noqa: E402


def view(ws, off, shape):
    n = int(np.prod(shape))
    return ws[off:off + 4 * n].view(torch.float32).reshape(shape).cpu().double()


def report(name, got, want):
    got, want = got.double(), want.double()
    if got.shape != want.shape:
        print(f'  {name:14s} SHAPE {tuple(got.shape)} vs {tuple(want.shape)}')
        return
    if want.numel() == 0:
        print(f'  {name:14s} (empty)')
        return
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    flag = '' if err <= 1e-4 * max(scale, 1e-3) else '   <<<<<<'
    print(f'  {name:14s} max|err| {err:10.3e}   max|ref| {scale:10.3e}   rel {err / max(scale, 1e-30):9.2e}{flag}')


def run(case):
    which = case.split('_')[-1]
    z, pos, cell, batch, c = util.case_inputs(case, torch.float32)
    sd32 = util.load_state(which, torch.float32)
    sd64 = {k: v.double() for k, v in sd32.items()}
    T = trace.trace(sd64, z, pos.double(), cell.double(), batch)
    model = NewtonNet(output_properties=['energy', 'gradient_force'])
    model.load_state_dict(sd32)
    model = model.to('cuda')
    model.eval()
    dev = 'cuda'
    zc, pc, cc, bc = z.to(dev), pos.to(dev), cell.to(dev), batch.to(dev)
    m = model._hip_model(0)
    g = hip.build_graph(pc, cc, bc, 5.0, model.embedding_layers.edge_embedding.embedding.frequencies, want_rbf=True)
    print(f'== {case}: N={g.n_atoms} E={g.n_edges} (ref {T["edge_index"].shape[1]}) B={g.n_mol}')
    ei_ok = np.array_equal(g.edge_index.cpu().numpy(), T['edge_index'].numpy())
    print('  edge_index bit-exact:', ei_ok)
    if not ei_ok:
        return
    rev = g.rev.cpu().numpy()
    ei = g.edge_index.cpu().numpy()
    print('  rev consistent:', bool(np.all(ei[0][rev] == ei[1]) and np.all(ei[1][rev] == ei[0])))
    report('disp', g.disp.cpu(), T['disp'])
    report('dir', g.geo[:, :3].cpu(), T['u'])
    report('r', g.geo[:, 3].cpu(), T['r'])
    report('rbf', g.rbf.cpu(), T['rbf'])
    res = hip.energy_forces(m, zc, pc, cc, g, want_forces=True, want_virial=True, want_nodes=False)
    torch.cuda.synchronize()
    ws = res['workspace']
    N, E, B, L = g.n_atoms, g.n_edges, g.n_mol, m.n_layers
    lay = hip.workspace_layout(N, E, B, L)
    Pn = E // 2
    pid = g.pid.cpu().long()          # edge tensors live once per undirected pair: compare row pid[e] with edge e
    report('a0', view(ws, lay.a0, (N, 128)), T['a0'])
    for l in range(L):
        print(f'  -- layer {l}')
        report('hn', view(ws, lay.hn[l], (N, 128)), T[f'hn_{l}'])
        report('m', view(ws, lay.m[l], (N, 128)), T[f'm_{l}'])
        report('msg', view(ws, lay.msg[l], (Pn, 128))[pid], T[f'msg_{l}'])
        report('a_mid', view(ws, lay.a_mid[l], (N, 128)), T[f'a_mid_{l}'])
        h12 = view(ws, lay.h12[l], (Pn, 256))[pid]
        report('h1', h12[:, :128], T[f'h1_{l}'])
        report('phi1', view(ws, lay.phi1[l], (Pn, 128))[pid], T[f'phi1_{l}'])
        if l > 0:
            report('h2', h12[:, 128:], T[f'h2_{l}'])
            report('phi2', view(ws, lay.phi2[l], (Pn, 128))[pid], T[f'phi2_{l}'])
        report('f_out', view(ws, lay.f_out[l], (N, 3, 128)), T[f'f_out_{l}'])
        report('q', view(ws, lay.q[l], (N, 3, 128)), T[f'q_{l}'])
        report('a_out', view(ws, lay.a_out[l], (N, 128)), T[f'a_out_{l}'])
        gx = view(ws, lay.g_x + 4 * l * E, (E,))
        rev = g.rev.cpu().long()
        report('g_x pair', gx + gx[rev], T[f'g_x_{l}'] + T[f'g_x_{l}'][rev])   # only the pair sum is defined
        report('g_u', view(ws, lay.g_u + 16 * l * E, (E, 4))[:, :3], T[f'g_u_{l}'])
    report('atom_energy', res['atom_energy'].cpu(), T['atom_energy'])
    report('energy', res['energy'].cpu(), T['energy'])
    report('forces', res['forces'].cpu(), T['forces'])
    f = res['forces'].cpu().double()
    print(f'  force MAE {((f - T["forces"]).abs().mean()).item():.3e}  max {((f - T["forces"]).abs().max()).item():.3e}'
          f'   (tol {util.FORCE_MAE_TOL:.0e} / {util.FORCE_MAX_TOL:.0e})')


if __name__ == '__main__':
    cases = sys.argv[1:] or ['aspirin8_rand', 'mixed_rand', 'pbc216_rand']
    print('lib version', hip.lib().nnhip_version(), 'device', torch.cuda.get_device_name(0))
    for cs in cases:
        run(cs)
