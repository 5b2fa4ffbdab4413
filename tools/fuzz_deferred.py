"""Tooling: random batches through the deferred step (NewtonNet.forward in eval mode) against the synchronous path of a fresh module,
bit for bit, and the neighbor list against the CPU oracle.  Molecule counts 1..1200, molecule sizes 1..30 with now and then a 40- or
1100-atom one, some molecules in their own periodic box, three calls per batch (synchronous, deferred, deferred) and a second batch
of the SAME shape with other sizes (the guess about molecule sizes then goes wrong in both directions).
usage (through gpurun): python tools/fuzz_deferred.py [rounds] [seed]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import util
from tests.test_hip_parity import make_model
from newtonnet_amd.models import NewtonNet
from oracle import newtonnet_ref as ref


def fresh(model):
    m = NewtonNet(output_properties=['energy', 'gradient_force'])
    m.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    m = m.cuda()
    m.eval()
    return m


def make_sizes(gen, B, kind):
    if kind == 0:
        s = torch.randint(1, 25, (B,), generator=gen)                      # all small
    elif kind == 1:
        s = torch.randint(1, 31, (B,), generator=gen)                      # some above the staging limit
    else:
        s = torch.randint(1, 25, (B,), generator=gen)
        s[int(torch.randint(0, B, (1,), generator=gen))] = 1100 if B < 200 else 40
    return s


def make_batch(gen, sizes, pbc_frac):
    B = len(sizes)
    zs, ps, bs = [], [], []
    cell = torch.zeros(B, 3, 3)
    for b, n in enumerate(sizes.tolist()):
        m = int(round(n ** (1.0 / 3.0))) + 1
        grid = torch.stack(torch.meshgrid(*[torch.arange(m)] * 3, indexing='ij'), dim=-1).reshape(-1, 3)[:n].float()
        a = 1.6 + 0.8 * float(torch.rand(1, generator=gen))
        p = a * grid + 0.2 * torch.randn(n, 3, generator=gen)
        if float(torch.rand(1, generator=gen)) < pbc_frac:
            L = max(a * m, 5.5)
            cell[b] = L * torch.eye(3)
            if b % 2:
                cell[b, 1, 0] = 0.2 * L
        ps.append(p + 60.0 * torch.rand(3, generator=gen))
        zs.append(torch.tensor([1, 6, 7, 8])[torch.randint(0, 4, (n,), generator=gen)])
        bs.append(torch.full((n,), b))
    return torch.cat(zs), torch.cat(ps), cell, torch.cat(bs)


def same(a, b):
    return (torch.equal(a.energy, b[0]) and torch.equal(a.gradient_force, b[1]) and torch.equal(a.edge_index, b[2])
            and torch.equal(a.atom_node, b[3]))


def main(rounds=40, seed=0):
    gen = torch.Generator().manual_seed(seed)
    model, sd = make_model('rand')
    t0, n_calls, n_def, n_rep = time.time(), 0, 0, 0
    for rd in range(rounds):
        u = float(torch.rand(1, generator=gen))
        B = int(torch.randint(1, 12, (1,), generator=gen)) if u < 0.3 else (int(torch.randint(12, 200, (1,), generator=gen)) if u < 0.7
                                                                           else int(torch.randint(640, 1200, (1,), generator=gen)))
        pbc = 1.0 if float(torch.rand(1, generator=gen)) < 0.35 else 0.0       # (the reference takes all cells or none)
        k1, k2 = int(torch.randint(0, 3, (1,), generator=gen)), int(torch.randint(0, 3, (1,), generator=gen))
        s1 = make_sizes(gen, B, k1)
        s2 = make_sizes(gen, B, k2)
        # the second batch keeps the atom count of the first (so the module stays on the deferred path): move atoms between molecules
        d = int(s1.sum() - s2.sum())
        i = 0
        while d != 0 and i < 100000:
            b = i % B
            if d > 0 and s2[b] < 24:
                s2[b] += 1; d -= 1
            elif d < 0 and s2[b] > 1:
                s2[b] -= 1; d += 1
            i += 1
        if d != 0:
            s2 = s1.clone()
        for sizes in (s1, s2):
            z, pos, cell, batch = make_batch(gen, sizes, pbc)
            args = (z.cuda(), pos.cuda(), cell.cuda(), batch.cuda())
            o = fresh(model)(*args)
            want = (o.energy.clone(), o.gradient_force.clone(), o.edge_index.clone(), o.atom_node.clone())
            assert torch.isfinite(want[1]).all()
            for rep in range(3):
                out = model(*args)
                rec = model.__dict__.get('_last_deferred')
                queued = rec is not None and rec.state == rec.QUEUED
                n_def += queued
                ok = same(out, want)
                n_rep += bool(queued and rec.count > rec.cap)
                n_calls += 1
                if not ok:
                    print('MISMATCH round', rd, 'B', B, 'N', int(sizes.sum()), 'kinds', k1, k2, 'rep', rep, 'pbc', pbc,
                          'dE', float((out.energy - want[0]).abs().max()), 'dF', float((out.gradient_force - want[1]).abs().max()),
                          'edges equal', torch.equal(out.edge_index, want[2]), 'small', None if rec is None else rec.small_molecules)
                    raise SystemExit(1)
            if int(sizes.sum()) <= 6000:      # the list against the CPU oracle (all pairs inside molecules: affordable below this)
                ei_ref, _ = ref.radius_graph(pos, cell if pbc else None, batch, 5.0)
                if not torch.equal(want[2].cpu(), ei_ref):
                    print('LIST MISMATCH round', rd, 'B', B, 'N', int(sizes.sum()))
                    raise SystemExit(1)
    print(f'fuzz ok: {rounds} rounds, {n_calls} calls ({n_def} deferred, {n_rep} overflow repeats) in {time.time() - t0:.0f} s')
    return n_calls, n_def, n_rep


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
