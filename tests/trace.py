"""Per-kernel reference values: the oracle's forward re-run with every intermediate tensor kept and the
gradients the reverse sweep must produce (torch autograd on CPU, fp64).  Used by the GPU parity tests and
by tools/debug_parity.py to localise a mismatch to one kernel."""
import torch
import torch.nn.functional as Fn

from oracle import newtonnet_ref as ref


def trace(sd, z, pos, cell, batch, cutoff=5.0):
    """sd: fp64 state; inputs fp64.  Returns dict of intermediates (detached) + per-layer geometry gradients."""
    L = ref.n_layers(sd)
    pos = pos.clone().requires_grad_(True)
    edge_index, disp = ref.radius_graph(pos, cell, batch, cutoff)
    freq = sd['embedding_layers.edge_embedding.embedding.frequencies']
    r = disp.norm(dim=-1, keepdim=True)
    u = disp / r
    x = (r / cutoff).squeeze(-1)
    x_l = [x.clone() for _ in range(L)]          # per-layer leaves -> per-layer dE/dx
    u_l = [u.clone() for _ in range(L)]
    for t in x_l + u_l:
        t.retain_grad()
    i, j = edge_index
    n = z.shape[0]
    T = dict(edge_index=edge_index, disp=disp.detach(), r=r.detach().squeeze(-1), u=u.detach(), x=x.detach())
    xx = x.detach().unsqueeze(-1)
    T['rbf'] = (ref.poly_envelope(xx) * torch.sin(freq * xx) / xx)
    a = sd['embedding_layers.node_embedding.weight'][z]
    T['a0'] = a.detach()
    f = torch.zeros(n, 3, a.shape[1], dtype=a.dtype)
    keep = []
    for l in range(L):
        p = f'interaction_layers.{l}.'
        xl = x_l[l].unsqueeze(-1)
        rbf = ref.poly_envelope(xl) * torch.sin(freq * xl) / xl
        hn = Fn.linear(a, sd[p + 'message_nodepart.0.weight'], sd[p + 'message_nodepart.0.bias'])
        m = Fn.linear(Fn.silu(hn), sd[p + 'message_nodepart.2.weight'], sd[p + 'message_nodepart.2.bias'])
        msg = Fn.linear(rbf, sd[p + 'message_edgepart.weight']) * m[i] * m[j]
        a_mid = a + torch.zeros_like(a).index_add_(0, i, msg)
        h1 = Fn.linear(msg, sd[p + 'equiv_message1.0.weight'])
        h2 = Fn.linear(msg, sd[p + 'equiv_message2.0.weight'])
        phi1 = Fn.linear(Fn.silu(h1), sd[p + 'equiv_message1.2.weight'])
        phi2 = Fn.linear(Fn.silu(h2), sd[p + 'equiv_message2.2.weight'])
        eq = phi1.unsqueeze(1) * u_l[l].unsqueeze(2) + phi2.unsqueeze(1) * f[j]
        f_out = f + torch.zeros_like(f).index_add_(0, i, eq)
        q = Fn.linear(f_out, sd[p + 'equiv_update.weight'])
        a_out = a_mid + (f_out * q).sum(1)
        layer = dict(hn=hn, m=m, msg=msg, a_mid=a_mid, h1=h1, h2=h2, phi1=phi1, phi2=phi2, f_out=f_out, q=q, a_out=a_out)
        for t in layer.values():
            if t.requires_grad:
                t.retain_grad()
        keep.append(layer)
        a, f = a_out, f_out
    energy, e_atom = ref.energy_head(sd, 0, a, z, batch, cell.shape[0])
    energy.sum().backward()
    T['energy'] = energy.detach()
    T['atom_energy'] = e_atom.detach().reshape(-1)
    T['forces'] = -pos.grad
    for l, layer in enumerate(keep):
        for k, t in layer.items():
            T[f'{k}_{l}'] = t.detach()
            T[f'g_{k}_{l}'] = t.grad if t.requires_grad else None
        T[f'g_x_{l}'] = x_l[l].grad
        T[f'g_u_{l}'] = u_l[l].grad
    return T
