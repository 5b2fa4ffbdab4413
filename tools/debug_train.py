#!/usr/bin/env python3
"""Stage-by-stage check of the fused training path (newtonnet_amd/train_fused.py, csrc/train.hip) against the fp64 model of the
same algorithm (tests/tangent_ref.py) and the oracle's autograd double backward.  GPU only.  usage: tools/debug_train.py [case]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from newtonnet_amd import hip, train_fused  # noqa: E402
from newtonnet_amd.models import NewtonNet  # noqa: E402
from oracle import newtonnet_ref as ref  # noqa: E402
from tests import tangent_ref as tr, util  # noqa: E402


def main(case='ethanol4_rand', activation='swish'):
    z, pos, cell, batch, c = util.case_inputs(case, torch.float32)
    torch.manual_seed(0)
    model = NewtonNet(activation=activation, output_properties=['energy', 'gradient_force'])
    if activation == 'swish':
        model.load_state_dict(util.load_state('rand', torch.float32))
    sd = {k: v.detach().clone().double() for k, v in model.state_dict().items()}
    model = model.cuda()
    model.train()
    g = torch.Generator().manual_seed(3)
    B, N = cell.shape[0], pos.shape[0]
    e_lab, f_lab = torch.randn(B, generator=g), torch.randn(N, 3, generator=g)
    p = pos.cuda().requires_grad_(True)
    out = model(z.cuda(), p, cell.cuda(), batch.cuda())
    loss = torch.nn.functional.mse_loss(out.energy, e_lab.cuda()) + 50.0 * torch.nn.functional.mse_loss(out.gradient_force, f_lab.cuda())
    gE, gF = torch.autograd.grad(loss, (out.energy, out.gradient_force), retain_graph=True)
    loss.backward()
    torch.cuda.synchronize()
    ws = model._train_ws[-1]
    E_, F_, grads, S = tr.train_grads(sd, z, pos.double(), cell.double(), batch, gE.cpu().double(), gF.cpu().double(),
                                       activation=activation, keep=True)
    ei = out.edge_index.cpu()
    assert torch.equal(ei, S['edge_index'])
    gr = hip.build_graph(pos.cuda(), cell.cuda(), batch.cuda(), 5.0, model.embedding_layers.edge_embedding.embedding.frequencies)
    pid, i_, j_ = gr.pid.cpu().long(), ei[0], ei[1]
    rev = gr.rev.cpu().long()

    def rel(name, got, want):
        got, want = got.detach().cpu().double(), want.double()
        err = (got - want).abs().max().item()
        scale = max(want.abs().max().item(), 1e-30)
        flag = '' if err <= 2e-4 * scale else '   <<<<<<'
        print(f'{name:14s} max err {err:.3e}  scale {scale:.3e}  rel {err / scale:.2e}{flag}')

    rel('energy', out.energy, E_)
    rel('forces', out.gradient_force, F_)
    Ls = S['layers']
    for l, st in enumerate(Ls):
        pairsum = lambda x: x + x[rev]  # noqa: E731  (directed-edge adjoints -> both directions of the pair)
        rel(f'msg{l}', ws.msg[l][pid], st['msg'])
        rel(f'h1_{l}', ws.h1[l][pid], st['h1'])
        rel(f'phi1_{l}', ws.phi1[l][pid], st['phi1'])
        rel(f'GA{l}', ws.GA[l], st['GA'])
        rel(f'gf{l}', ws.gf[l], st['gf'])
        rel(f'g_phi1_{l}', ws.g_h12[l][:, :128][pid], pairsum(st['g_phi1']))
        rel(f't1_{l}', ws.t1[l][pid], pairsum(st['t1']))
        rel(f'g_msg{l}', ws.g_msg[l][pid], pairsum(st['G'] - st['GA'][i_]))
        rel(f'dmsg{l}', ws.dmsg[l][pid], st['dmsg'])
        rel(f'dh1_{l}', ws.dh1[l][pid], st['dh1'])
        rel(f'dphi1_{l}', ws.dphi1[l][pid], st['dphi1'])
        rel(f'df_out{l}', ws.df_out[l], st['df_out'])
        rel(f'dq{l}', ws.dq[l], st['dq'])
        rel(f'da_out{l}', ws.da_out[l], st['da_out'])
        rel(f'dg_phi1_{l}', ws.dg_h12[l][:, :128][pid], pairsum(st['dg_phi1']))
        rel(f'dg_h1_{l}', ws.dg_h1[l][pid], pairsum(st['dg_h1']))
        rel(f'dg_m{l}', ws.dg_m[l], st['dg_m'])
        rel(f'dg_hn{l}', ws.dg_hn[l], st['dg_hn'])
        rel(f'g_eps{l}', ws.g_eps[l][pid], pairsum(st['g_eps']))
        rel(f'dg_eps{l}', ws.dg_eps[l][pid], pairsum(st['dg_eps']))
        if l > 0:
            rel(f'phi2_{l}', ws.phi2[l][pid], st['phi2'])
            rel(f'dphi2_{l}', ws.dphi2[l][pid], st['dphi2'])
            rel(f'g_phi2_{l}', ws.g_h12[l][:, 128:][pid], pairsum(st['g_phi2']))
            rel(f'dg_phi2_{l}', ws.dg_h12[l][:, 128:][pid], pairsum(st['dg_phi2']))
            rel(f'dg_h2_{l}', ws.dg_h2[l][pid], pairsum(st['dg_h2']))
            rel(f'dm{l}', ws.dm[l], st['dm'])
            rel(f'g_m{l}', ws.g_m[l], st['g_m'])
    rel('de2', ws.de2, S['de2'])
    rel('dg_e2', ws.dg_e2, S['dg_e2'])
    rel('dg_e1', ws.dg_e1, S['dg_e1'])
    rel('dGA(final)', ws.dGA, S['dGA0'])
    want_loss, want = ref.training_loss_grads(sd, z, pos.double(), cell.double(), batch, e_lab.double(), f_lab.double())
    print('loss', loss.item(), want_loss.item())
    tot_e = tot_n = 0.0
    for name, prm in model.named_parameters():
        if not prm.requires_grad:
            continue
        got = prm.grad.detach().cpu().double()
        w = want[name]
        err, nrm = (got - w).norm().item(), w.norm().item()
        tot_e += err ** 2
        tot_n += nrm ** 2
        print(f'{name:55s} err {err:.3e} norm {nrm:.3e} rel {err / max(nrm, 1e-30):.2e}' + ('   <<<<' if err > 1e-4 * max(nrm, 1e-9) else ''))
    print('TOTAL rel grad-norm error', np.sqrt(tot_e / tot_n))


if __name__ == '__main__':
    main(*sys.argv[1:])
