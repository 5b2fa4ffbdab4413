"""Staged fp64 CPU model of the TRAINING algorithm of the HIP path (TEST INFRASTRUCTURE, like tests/trace.py).

The HIP training path (csrc/train.hip) does not run torch autograd through the force graph.  It evaluates
    dL/dtheta  for any loss L(E, F), F = -dE/dpos,
as "tangent over reverse":  with  c_b = dL/dE_b  and  d = dL/dF,
    dL/dtheta = sum_b c_b dE_b/dtheta  -  D_d [ grad_theta E_tot ]        (D_d: directional derivative along d in pos space)
i.e. the reverse sweep that produces grad_theta E is differentiated once more in FORWARD (tangent) mode along
v = -d, with the seed of the reverse sweep carried as the dual number 1 + eps c_b.  Four sweeps:
    1 forward (values)   2 reverse (values: forces)   3 tangent forward   4 tangent reverse + weight-gradient products
This file writes all four out stage by stage in torch fp64 -- the same stages, in the same order and with the same
intermediates as the kernels -- so that every HIP stage can be checked against it, and is itself checked against the
oracle's autograd double backward (oracle/newtonnet_ref.py:training_loss_grads; tests/test_oracle.py).

Reference semantics restated: newtonnet/models/newtonnet.py:207-231, output.py:66-73,98-100, scalers.py:55-58,
train/trainer.py:299-313, train/loss.py:48,72,96.
"""
from __future__ import annotations

import math

import torch

from oracle import newtonnet_ref as ref


# activation, first and second derivative (activations.py:5-30)
def _sig(x):
    return torch.sigmoid(x)


ACTS = {
    'swish': (lambda x: x * _sig(x),
              lambda x: _sig(x) * (1 + x * (1 - _sig(x))),
              lambda x: _sig(x) * (1 - _sig(x)) * (2 + x * (1 - 2 * _sig(x)))),
    'tanh': (torch.tanh, lambda x: 1 - torch.tanh(x) ** 2, lambda x: -2 * torch.tanh(x) * (1 - torch.tanh(x) ** 2)),
    'sigmoid': (_sig, lambda x: _sig(x) * (1 - _sig(x)), lambda x: _sig(x) * (1 - _sig(x)) * (1 - 2 * _sig(x))),
    'softplus': (torch.nn.functional.softplus, _sig, lambda x: _sig(x) * (1 - _sig(x))),
    'ssp': (lambda x: torch.nn.functional.softplus(x) - math.log(2.0), _sig, lambda x: _sig(x) * (1 - _sig(x))),
    'relu': (torch.relu, lambda x: (x > 0).to(x.dtype), lambda x: torch.zeros_like(x)),
    'leaky_relu': (lambda x: torch.where(x > 0, x, 0.01 * x), lambda x: torch.where(x > 0, 1.0, 0.01).to(x.dtype),
                   lambda x: torch.zeros_like(x)),
    'elu': (torch.nn.functional.elu, lambda x: torch.where(x > 0, torch.ones_like(x), torch.exp(x)),
            lambda x: torch.where(x > 0, torch.zeros_like(x), torch.exp(x))),
    'gelu': (torch.nn.functional.gelu,
             lambda x: 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi),
             lambda x: (2 - x * x) * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)),
}
ACTS['silu'] = ACTS['swish']


def _scat(src, index, n):
    return src.new_zeros((n,) + tuple(src.shape[1:])).index_add_(0, index, src)


def train_grads(sd, z, pos, cell, batch, g_energy, g_forces, cutoff=5.0, activation='swish', keep=False):
    """dL/dtheta for dL/dE = g_energy [B], dL/dF = g_forces [N,3].  Returns (energy, forces, {name: grad}[, stages])."""
    act, dact, d2act = ACTS[activation]
    dt = pos.dtype
    N, B = pos.shape[0], cell.shape[0]
    L = ref.n_layers(sd)
    S = {}

    # ---------------------------------------------------------------- geometry (values and d/dx, d2/dx2 of the radial basis)
    ei, disp = ref.radius_graph(pos, cell, batch, cutoff)
    i, j = ei[0], ei[1]
    E = ei.shape[1]
    r = disp.norm(dim=1, keepdim=True)
    u = disp / r
    x = r / cutoff
    freq = sd['embedding_layers.edge_embedding.embedding.frequencies'].to(dt)
    env = 1 - 55 * x ** 9 + 99 * x ** 10 - 45 * x ** 11
    denv = -495 * x ** 8 + 990 * x ** 9 - 495 * x ** 10
    bes = torch.sin(freq * x) / x
    dbes = (freq * torch.cos(freq * x) - bes) / x
    rbf = env * bes                                  # [E, nb]
    drbf = denv * bes + env * dbes                   # d rbf / dx

    W = lambda k: sd[k].to(dt)  # noqa: E731
    P = lambda l, k: W(f'interaction_layers.{l}.{k}')  # noqa: E731

    # ---------------------------------------------------------------- sweep 1: forward (values)
    a = W('embedding_layers.node_embedding.weight')[z]
    f = torch.zeros(N, 3, a.shape[1], dtype=dt)
    Ls = []
    for l in range(L):
        st = dict(a_in=a, f_in=f)
        st['hn'] = a @ P(l, 'message_nodepart.0.weight').T + P(l, 'message_nodepart.0.bias')
        st['m'] = act(st['hn']) @ P(l, 'message_nodepart.2.weight').T + P(l, 'message_nodepart.2.bias')
        st['eps'] = rbf @ P(l, 'message_edgepart.weight').T
        st['deps'] = drbf @ P(l, 'message_edgepart.weight').T
        m = st['m']
        st['msg'] = st['eps'] * m[i] * m[j]
        st['a_mid'] = a + _scat(st['msg'], i, N)
        for k in (1, 2):
            st[f'h{k}'] = st['msg'] @ P(l, f'equiv_message{k}.0.weight').T
            st[f'phi{k}'] = act(st[f'h{k}']) @ P(l, f'equiv_message{k}.2.weight').T
        eq = st['phi1'].unsqueeze(1) * u.unsqueeze(2) + st['phi2'].unsqueeze(1) * f[j]
        st['f_out'] = f + _scat(eq, i, N)
        st['q'] = st['f_out'] @ P(l, 'equiv_update.weight').T
        st['a_out'] = st['a_mid'] + (st['f_out'] * st['q']).sum(1)
        a, f = st['a_out'], st['f_out']
        Ls.append(st)
    H0, c0 = W('output_layers.0.layers.0.weight'), W('output_layers.0.layers.0.bias')
    H2, c2 = W('output_layers.0.layers.2.weight'), W('output_layers.0.layers.2.bias')
    w4, b4 = W('output_layers.0.layers.4.weight').reshape(-1), W('output_layers.0.layers.4.bias')
    sc, sh = W('scalers.0.scale.weight').reshape(-1)[z], W('scalers.0.shift.weight').reshape(-1)[z]
    e1 = a @ H0.T + c0
    e2 = act(e1) @ H2.T + c2
    eps_atom = act(e2) @ w4 + b4
    e_atom = eps_atom * sc + sh
    energy = _scat(e_atom, batch, B)

    # ---------------------------------------------------------------- sweep 2: reverse (values, seed 1) -> forces
    g_e2 = (sc.unsqueeze(1) * w4) * dact(e2)
    t_e1 = g_e2 @ H2
    g_e1 = t_e1 * dact(e1)
    GA = g_e1 @ H0
    Gf = torch.zeros_like(f)
    g_d = torch.zeros(E, 3, dtype=dt)
    for l in range(L - 1, -1, -1):
        st = Ls[l]
        Wu = P(l, 'equiv_update.weight')
        st['GA'] = GA
        st['gf'] = Gf + GA.unsqueeze(1) * st['q'] + (GA.unsqueeze(1) * st['f_out']) @ Wu
        gf = st['gf']
        st['g_phi1'] = (gf[i] * u.unsqueeze(2)).sum(1)
        st['g_phi2'] = (gf[i] * st['f_in'][j]).sum(1)
        g_u = (gf[i] * st['phi1'].unsqueeze(1)).sum(2)                     # [E,3]
        Gf = gf + _scat(st['phi2'].unsqueeze(1) * gf[i], j, N)
        g_msg = 0
        for k in (1, 2):
            st[f't{k}'] = st[f'g_phi{k}'] @ P(l, f'equiv_message{k}.2.weight')
            st[f'g_h{k}'] = st[f't{k}'] * dact(st[f'h{k}'])
            g_msg = g_msg + st[f'g_h{k}'] @ P(l, f'equiv_message{k}.0.weight')
        st['G'] = g_msg + GA[i]
        G, m = st['G'], st['m']
        st['g_m'] = _scat(G * st['eps'] * m[j], i, N) + _scat(G * st['eps'] * m[i], j, N)
        st['g_eps'] = G * m[i] * m[j]
        g_x = (st['g_eps'] * st['deps']).sum(1, keepdim=True)
        st['t_n'] = st['g_m'] @ P(l, 'message_nodepart.2.weight')
        st['g_hn'] = st['t_n'] * dact(st['hn'])
        GA = GA + st['g_hn'] @ P(l, 'message_nodepart.0.weight')
        g_d = g_d + (g_x / cutoff) * u + (g_u - (g_u * u).sum(1, keepdim=True) * u) / r
    GA0 = GA
    forces = -(_scat(g_d, i, N) - _scat(g_d, j, N))

    # ---------------------------------------------------------------- sweep 3: tangent forward along v = -dL/dF
    v = -g_forces.to(dt)
    dd = v[i] - v[j]
    dr = (u * dd).sum(1, keepdim=True)
    du = (dd - u * dr) / r
    dx = dr / cutoff
    da = torch.zeros_like(Ls[0]['a_in'])
    df = torch.zeros_like(f)
    for l in range(L):
        st = Ls[l]
        m = st['m']
        st['da_in'], st['df_in'] = da, df
        st['dhn'] = da @ P(l, 'message_nodepart.0.weight').T
        st['dm'] = (st['dhn'] * dact(st['hn'])) @ P(l, 'message_nodepart.2.weight').T
        dm = st['dm']
        st['dmsg'] = st['deps'] * dx * m[i] * m[j] + st['eps'] * (dm[i] * m[j] + m[i] * dm[j])
        da_mid = da + _scat(st['dmsg'], i, N)
        for k in (1, 2):
            st[f'dh{k}'] = st['dmsg'] @ P(l, f'equiv_message{k}.0.weight').T
            st[f'dphi{k}'] = (st[f'dh{k}'] * dact(st[f'h{k}'])) @ P(l, f'equiv_message{k}.2.weight').T
        deq = (st['dphi1'].unsqueeze(1) * u.unsqueeze(2) + st['phi1'].unsqueeze(1) * du.unsqueeze(2)
               + st['dphi2'].unsqueeze(1) * st['f_in'][j] + st['phi2'].unsqueeze(1) * df[j])
        st['df_out'] = df + _scat(deq, i, N)
        st['dq'] = st['df_out'] @ P(l, 'equiv_update.weight').T
        st['da_out'] = da_mid + (st['df_out'] * st['q'] + st['f_out'] * st['dq']).sum(1)
        da, df = st['da_out'], st['df_out']
    de1 = da @ H0.T
    de2 = (de1 * dact(e1)) @ H2.T
    deps_atom = (dact(e2) * de2) @ w4

    # ---------------------------------------------------------------- sweep 4: tangent reverse (seed tangent c_b) + weight grads
    c = g_energy.to(dt)[batch]                                            # per atom
    grads = {}
    dg_e2 = (sc.unsqueeze(1) * w4) * (c.unsqueeze(1) * dact(e2) + d2act(e2) * de2)
    dt_e1 = dg_e2 @ H2
    dg_e1 = dt_e1 * dact(e1) + t_e1 * d2act(e1) * de1
    grads['output_layers.0.layers.4.weight'] = ((sc * c).unsqueeze(1) * act(e2) + sc.unsqueeze(1) * dact(e2) * de2).sum(0, keepdim=True)
    grads['output_layers.0.layers.4.bias'] = (sc * c).sum().reshape(1)
    grads['scalers.0.scale.weight'] = _scat(c * eps_atom + deps_atom, z, 119).unsqueeze(1)
    grads['scalers.0.shift.weight'] = _scat(c, z, 119).unsqueeze(1)
    grads['output_layers.0.layers.2.weight'] = dg_e2.T @ act(e1) + g_e2.T @ (dact(e1) * de1)
    grads['output_layers.0.layers.2.bias'] = dg_e2.sum(0)
    grads['output_layers.0.layers.0.weight'] = dg_e1.T @ Ls[-1]['a_out'] + g_e1.T @ Ls[-1]['da_out']
    grads['output_layers.0.layers.0.bias'] = dg_e1.sum(0)
    dGA = dg_e1 @ H0
    dGf = torch.zeros_like(f)
    for l in range(L - 1, -1, -1):
        st = Ls[l]
        p = f'interaction_layers.{l}.'
        Wu = P(l, 'equiv_update.weight')
        GA, gf, m, dm = st['GA'], st['gf'], st['m'], st['dm']
        gq, dgq = GA.unsqueeze(1) * st['f_out'], dGA.unsqueeze(1) * st['f_out'] + GA.unsqueeze(1) * st['df_out']
        st['dgf'] = dGf + dGA.unsqueeze(1) * st['q'] + GA.unsqueeze(1) * st['dq'] + dgq @ Wu
        dgf = st['dgf']
        F_ = gq.shape[-1]
        grads[p + 'equiv_update.weight'] = dgq.reshape(-1, F_).T @ st['f_out'].reshape(-1, F_) + \
            gq.reshape(-1, F_).T @ st['df_out'].reshape(-1, F_)
        st['dg_phi1'] = (dgf[i] * u.unsqueeze(2) + gf[i] * du.unsqueeze(2)).sum(1)
        st['dg_phi2'] = (dgf[i] * st['f_in'][j] + gf[i] * st['df_in'][j]).sum(1)
        dGf = dgf + _scat(st['dphi2'].unsqueeze(1) * gf[i] + st['phi2'].unsqueeze(1) * dgf[i], j, N)
        dg_msg = 0
        for k in (1, 2):
            V0, V2 = P(l, f'equiv_message{k}.0.weight'), P(l, f'equiv_message{k}.2.weight')
            h, dh = st[f'h{k}'], st[f'dh{k}']
            dtk = st[f'dg_phi{k}'] @ V2
            st[f'dg_h{k}'] = dtk * dact(h) + st[f't{k}'] * d2act(h) * dh
            dg_msg = dg_msg + st[f'dg_h{k}'] @ V0
            grads[p + f'equiv_message{k}.2.weight'] = st[f'dg_phi{k}'].T @ act(h) + st[f'g_phi{k}'].T @ (dact(h) * dh)
            grads[p + f'equiv_message{k}.0.weight'] = st[f'dg_h{k}'].T @ st['msg'] + st[f'g_h{k}'].T @ st['dmsg']
        dG, G = dg_msg + dGA[i], st['G']
        eps, deps_x = st['eps'], st['deps'] * dx
        st['dg_m'] = (_scat(dG * eps * m[j] + G * deps_x * m[j] + G * eps * dm[j], i, N)
                      + _scat(dG * eps * m[i] + G * deps_x * m[i] + G * eps * dm[i], j, N))
        st['dg_eps'] = dG * m[i] * m[j] + G * (dm[i] * m[j] + m[i] * dm[j])
        grads[p + 'message_edgepart.weight'] = st['dg_eps'].T @ rbf + st['g_eps'].T @ (drbf * dx)
        W0n, W2n = P(l, 'message_nodepart.0.weight'), P(l, 'message_nodepart.2.weight')
        dt_n = st['dg_m'] @ W2n
        st['dg_hn'] = dt_n * dact(st['hn']) + st['t_n'] * d2act(st['hn']) * st['dhn']
        grads[p + 'message_nodepart.2.weight'] = st['dg_m'].T @ act(st['hn']) + st['g_m'].T @ (dact(st['hn']) * st['dhn'])
        grads[p + 'message_nodepart.2.bias'] = st['dg_m'].sum(0)
        grads[p + 'message_nodepart.0.weight'] = st['dg_hn'].T @ st['a_in'] + st['g_hn'].T @ st['da_in']
        grads[p + 'message_nodepart.0.bias'] = st['dg_hn'].sum(0)
        dGA = dGA + st['dg_hn'] @ W0n
    grads['embedding_layers.node_embedding.weight'] = _scat(dGA, z, 119)
    out = (energy, forces, grads)
    if keep:
        S.update(layers=Ls, edge_index=ei, u=u, du=du, dx=dx, rbf=rbf, drbf=drbf, e1=e1, e2=e2, de1=de1, de2=de2,
                 g_e2=g_e2, dg_e2=dg_e2, g_e1=g_e1, dg_e1=dg_e1, GA0=GA0, dGA0=dGA, v=v)
        out = out + (S,)
    return out
