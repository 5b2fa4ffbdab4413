"""CPU-side checks: the C-ABI library loads and exports what include/newtonnet_hip.h declares, the module
mirrors the reference's state_dict, and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from newtonnet_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    header = open(os.path.join(ROOT, 'include', 'newtonnet_hip.h')).read()
    declared = set(re.findall(r'\b(nnhip_[a-z_0-9]+)\s*\(', header))
    assert declared, 'no declarations found'
    for sym in declared:
        assert hasattr(lib, sym), f'{sym} declared in the header but not exported'
    assert set(hip.EXPORTED_SYMBOLS) == declared
    assert lib.nnhip_version() >= 100


def test_train_ws_mirror_matches_header():
    """The ctypes mirror of nnhip_train_ws lists the header's members in the header's order, and has the library's size."""
    import ctypes as C
    from newtonnet_amd import hip
    text = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'newtonnet_hip.h')).read()
    body = text[:text.index('} nnhip_train_ws;')]
    body = re.sub(r'/\*.*?\*/', '', body[body.rindex('typedef struct {'):], flags=re.S)
    names = []
    for stmt in body.split(';'):
        stmt = stmt.replace('typedef struct {', '')
        if not stmt.strip():
            continue
        # "const float* a, b_" / "float* hn[NNHIP_MAX_LAYERS]" / "int32_t n_atoms, n_edges"
        decl = re.sub(r'^\s*(const\s+)?\w+\s*\*?\s*', '', stmt.strip())
        names += [re.match(r'\*?\s*(\w+)', part.strip()).group(1) for part in decl.split(',')]
    assert names == [n for n, _ in hip.TrainWs._fields_]
    assert C.sizeof(hip.TrainWs) == hip.lib().nnhip_train_ws_bytes()


def _header_struct_members(name):
    text = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'newtonnet_hip.h')).read()
    body = text[:text.index('} %s;' % name)]
    body = re.sub(r'/\*.*?\*/', '', body[body.rindex('typedef struct {'):], flags=re.S)
    names = []
    for stmt in body.split(';'):
        stmt = stmt.replace('typedef struct {', '')
        if not stmt.strip():
            continue
        decl = re.sub(r'^\s*(const\s+)?\w+\s*\*?\s*', '', stmt.strip())
        names += [re.match(r'\*?\s*(\w+)', part.strip()).group(1) for part in decl.split(',')]
    return names


def test_deferred_step_structs_mirror_the_header():
    """nnhip_step_layout / nnhip_step_dev (the one-call deferred step, nnhip_forward_dev): the ctypes mirrors list the header's
    members in the header's order, and the layout the library computes is self-consistent (no overlaps, everything inside)."""
    import ctypes as C
    from newtonnet_amd import hip
    assert _header_struct_members('nnhip_step_layout') == [n for n, _ in hip.StepLayout._fields_]
    assert _header_struct_members('nnhip_step_dev') == [n for n, _ in hip.StepDev._fields_]
    N, B, cap = 21504, 1024, 332826
    lay = hip.step_layout(N, B, cap)
    n_scan = (N + 1023) // 1024 + 1
    ints = [(lay.mol_ptr, B + 1), (lay.row_ptr, N + 1 + 1 + n_scan), (lay.pair_ptr, N + 1), (lay.pair_scan, n_scan), (lay.tail, 2),
            (lay.mol_scratch, 2 * B + B // 1024 + 4),
            (lay.xg, 2 * cap), (lay.col, cap), (lay.rev, cap), (lay.pid, cap)]
    flts = [(lay.geo, 4 * cap), (lay.disp, 3 * cap), (lay.energy, B), (lay.forces, 3 * N), (lay.virial, 9 * B), (lay.atom_energy, N)]
    assert lay.status == lay.row_ptr + N + 1                     # (count, status) adjacent: one 8-byte copy
    for spans, total in ((ints, lay.i32_count), (flts, lay.f32_count)):
        spans = sorted(spans)
        assert all(a % 4 == 0 for a, _ in spans)                 # 16-byte granules
        assert all(a + n <= b for (a, n), (b, _) in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] <= total
    assert 0 <= hip.lib().nnhip_graph_small_max_atoms() <= 1024


def test_workspace_layout_is_consistent():
    from newtonnet_amd import hip
    lay = hip.workspace_layout(21504, 310406, 1024, 3)
    assert lay.total == hip.lib().nnhip_workspace_bytes(21504, 310406, 1024, 3)
    offs = sorted([lay.a0, lay.e1, lay.e2, lay.g_x, lay.g_u, lay.g_a, lay.g_f] +
                  [getattr(lay, n)[l] for n in ('m', 'hn', 'msg', 'h12', 'phi1', 'phi2', 'a_mid', 'a_out', 'f_out', 'q')
                   for l in range(3)])
    assert all(o % 256 == 0 for o in offs) and len(set(offs)) == len(offs) and offs[-1] < lay.total


def test_state_dict_matches_reference_layout():
    from newtonnet_amd.models import NewtonNet
    model = NewtonNet(output_properties=['energy', 'gradient_force'])
    ref_sd = util.load_state('ckpt')
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref_sd.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), k
    model.load_state_dict(ref_sd)
    assert sum(p.numel() for p in model.parameters()) == 401155
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 401135     # frequencies frozen


def test_same_seed_same_init_as_reference():
    """torch.manual_seed(0) + default construction reproduces the reference's initial weights
    (tests/golden/rand_state_seed0.npz was produced by constructing the reference model)."""
    from newtonnet_amd.models import NewtonNet
    torch.manual_seed(0)
    model = NewtonNet(output_properties=['energy', 'gradient_force'])
    want = util.load_state('rand', torch.float32)
    for k, v in model.state_dict().items():
        assert torch.equal(v, want[k]), k


def test_train_eval_quirk_and_create_graph():
    from newtonnet_amd.models import NewtonNet
    from newtonnet_amd.models.output import DerivativeProperty
    model = NewtonNet(output_properties=['energy', 'gradient_force'])
    assert model.embedding_layers.requires_dr is True
    assert model.eval() is None                     # reference quirk: train() returns None (newtonnet.py:106-113)
    assert all(not ol.create_graph for ol in model.output_layers if isinstance(ol, DerivativeProperty))
    model.train()
    assert all(ol.create_graph for ol in model.output_layers if isinstance(ol, DerivativeProperty))


def test_no_cpu_fallback():
    from newtonnet_amd.models import NewtonNet
    model = NewtonNet(output_properties=['energy', 'gradient_force'])
    model.eval()
    with pytest.raises(RuntimeError, match='no CPU path'):
        model(torch.tensor([1]), torch.zeros(1, 3), torch.zeros(1, 3, 3), torch.zeros(1, dtype=torch.long))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'newtonnet_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith('.py'):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f'{fn} imports the oracle'


def test_unknown_keys_raise_like_reference():
    from newtonnet_amd.layers import get_activation_by_string, get_precision_by_string
    from newtonnet_amd.models import get_output_by_string
    with pytest.raises(NotImplementedError):
        get_activation_by_string('nope')
    with pytest.raises(ValueError):
        get_precision_by_string('int8')
    with pytest.raises(NotImplementedError):
        get_output_by_string('nope')


def test_whole_module_pickle_roundtrip(tmp_path):
    """trainer.py:219 / ase_interface.py:87: models are saved and loaded as whole-module pickles; kernel handles must not
    live in picklable attributes."""
    from newtonnet_amd.models import NewtonNet
    model = NewtonNet(output_properties=['energy', 'gradient_force'])
    path = tmp_path / 'best_model.pt'
    torch.save(model, path)
    loaded = torch.load(path, map_location='cpu', weights_only=False)
    assert isinstance(loaded, NewtonNet) and loaded.output_properties == ['energy', 'gradient_force']
    for (k0, v0), (k1, v1) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert k0 == k1 and torch.equal(v0, v1)


def test_model_surgery_like_ase_interface():
    """ase_interface.py:97-121 appends / pops heads on the live module lists."""
    from newtonnet_amd.layers import get_scaler_by_string
    from newtonnet_amd.models import NewtonNet, get_aggregator_by_string, get_output_by_string
    model = NewtonNet(output_properties=['energy'])
    assert model.embedding_layers.requires_dr is False
    model.output_properties.append('gradient_force')
    model.output_layers.append(get_output_by_string('gradient_force'))
    model.scalers.append(get_scaler_by_string('gradient_force'))
    model.aggregators.append(get_aggregator_by_string('gradient_force'))
    assert len(model.output_layers) == 2 and model.scalers[1].scale is None
    with pytest.raises(NotImplementedError):
        get_output_by_string('hessian')


def test_bench_self_launch_command(monkeypatch):
    """bench.py --gpus N without a launcher starts `python -m torch.distributed.run --nproc-per-node N bench.py <same argv>` as a
    CHILD process (rendezvous on 127.0.0.1) and returns its exit code -- it never execs, and it does so before any GPU call."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert not torch.cuda.is_initialized()


def test_no_packed_fp32_instruction_in_any_kernel():
    """build.sh compiles every kernel without packed-fp32 instructions: dependent v_pk_*_f32 chains with op_sel mis-execute on MI355X
    for some code alignments (tools/probes/pk_chain_probe.hip; profiles/r05_mol_fused2_soak.txt), and which kernel is exposed changes
    with every recompile.  Checked on the objects of the last build (skipped when there are none, e.g. before the first build)."""
    import glob
    import shutil
    import subprocess
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    objs = sorted(glob.glob(os.path.join(ROOT, 'newtonnet_amd', 'csrc', 'build', 'obj', '*.o')))
    if not objs or not (os.path.exists(objdump) or shutil.which('llvm-objdump')):
        pytest.skip('no build objects / no llvm-objdump here')
    objdump = objdump if os.path.exists(objdump) else shutil.which('llvm-objdump')
    import tempfile
    seen_mfma = False
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            # (`--offloading` unbundles the gfx950 code object next to the file it is given; the disassembly of that is the device code)
            local = shutil.copy(o, tmp)
            subprocess.run([objdump, '-d', '--offloading', local], capture_output=True, text=True, cwd=tmp)
            dev = [f for f in glob.glob(local + '.*') if 'amdgcn' in f]
            if not dev:          # (a file of launchers only: nothing for the device)
                continue
            dis = subprocess.run([objdump, '-d', dev[0]], capture_output=True, text=True).stdout
            # ANY packed VOP3P instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, but also v_pk_mov_b32, v_pk_*_f16 with op_sel ...):
            # the claim of profiles/r05_mol_fused2_soak.txt section 7 is "no v_pk_* of any type", so that is what is checked
            pk = sorted(set(re.findall(r'\bv_pk_\w+', dis)))
            assert not pk, (o, pk)
            seen_mfma |= 'v_mfma_' in dis
    assert seen_mfma      # (the disassembly really is the device code)


def test_tooling_builds_are_marked_and_refused(tmp_path):
    """A library compiled with any extra flag (the ablation switches of tools/ablate_*.sh produce WRONG results) carries
    nnhip_build_flags() bit 0 and the package refuses to load it; the shipped library is unmarked; defining an ablation switch
    without the marker does not compile at all."""
    import subprocess
    import sys
    from newtonnet_amd import hip
    assert hip.lib().nnhip_build_flags() == 0
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-std=c++17', '-fsyntax-only', '-DEDGE_ABL_TABLE',
                        os.path.join(ROOT, 'newtonnet_amd', 'csrc', 'edge.hip')], capture_output=True, text=True)
    assert r.returncode != 0 and 'ablation switches produce wrong results' in r.stderr
    # a marked library: one small translation unit is enough to show the load-time refusal
    src = tmp_path / 'marked.cpp'
    src.write_text('extern "C" int nnhip_version(void) { return 101; }\nextern "C" int nnhip_build_flags(void) { return 1; }\n')
    so = os.path.join(ROOT, 'newtonnet_amd', 'lib', 'libmarked_test.so')
    try:
        subprocess.run(['g++', '-shared', '-fPIC', str(src), '-o', so], check=True)
        code = ("import os, sys; sys.path.insert(0, %r); os.environ['NNHIP_LIB_NAME'] = 'libmarked_test.so'\n"
                "from newtonnet_amd import hip\n"
                "try:\n    hip.lib()\nexcept hip.HipLibraryError as e:\n    print('REFUSED' if 'TOOLING build' in str(e) else e)\n" % ROOT)
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
        assert 'REFUSED' in out.stdout, out.stdout + out.stderr
    finally:
        if os.path.exists(so):
            os.remove(so)


def test_fused_step_refuses_a_model_with_both_force_heads():
    """the fused training step carries ONE force term (loss.py:30-47 sums a term per configured head); with both
    'gradient_force' and 'direct_force' it must refuse instead of training the direct-force head with a zero gradient."""
    from newtonnet_amd.distributed import _force_key
    from newtonnet_amd.models import NewtonNet
    assert _force_key(NewtonNet(output_properties=['energy', 'gradient_force'])) == 'gradient_force'
    assert _force_key(NewtonNet(output_properties=['energy', 'direct_force'])) == 'direct_force'
    assert _force_key(NewtonNet(output_properties=['energy'])) is None
    with pytest.raises(NotImplementedError):
        _force_key(NewtonNet(output_properties=['energy', 'gradient_force', 'direct_force']))


def test_inference_lanes_share_parameters_and_nothing_else(tmp_path):
    """model.inference_lanes(n) (host logic, no GPU): lane 0 is the module; the other lanes are shallow views with the SAME Parameter
    and submodule objects and NONE of the run-time state; they are cached, handed out by any lane, eval-only without touching the
    owner's flags, and never travel in a pickle (trainer.py:219 saves whole modules)."""
    import pickle
    from newtonnet_amd.models import NewtonNet
    m = NewtonNet(output_properties=['energy', 'gradient_force'])
    m.train()
    m.__dict__['_edge_hint'] = (21, 400)            # (run-time state a call would have left)
    lanes = m.inference_lanes(3)
    assert lanes[0] is m and len({id(l) for l in lanes}) == 3
    for l in lanes[1:]:
        assert l._parameters is m._parameters and l._modules is m._modules and l.interaction_layers is m.interaction_layers
        assert '_edge_hint' not in l.__dict__ and '_lanes' not in l.__dict__ and l.__dict__['_lane_of'] is m
        assert l.training is False
        with pytest.raises(RuntimeError):
            l.train()
        assert l.eval() is None
    assert m.training and m.interaction_layers.training and all(ol.create_graph for ol in m.output_layers if hasattr(ol, 'create_graph'))
    assert m.inference_lanes(2)[1] is lanes[1] and lanes[2].inference_lanes(3)[1] is lanes[1]
    # a parameter update on the owner is the lanes' update (same objects)
    with torch.no_grad():
        m.interaction_layers[0].equiv_update.weight.add_(1.0)
    assert torch.equal(lanes[1].interaction_layers[0].equiv_update.weight, m.interaction_layers[0].equiv_update.weight)
    back = pickle.loads(pickle.dumps(m))
    assert '_lanes' not in back.__dict__ and '_edge_hint' not in back.__dict__
    lane_back = pickle.loads(pickle.dumps(lanes[1]))
    assert '_lane_of' not in lane_back.__dict__             # (a pickled lane comes back as an ordinary module)


def test_library_config_survives_hostile_environment_values():
    """nnhip_config writes the NNHIP_* switches that are set into a JSON string; a value with a quote or a backslash must not
    cost the caller its bench line (bench.py calls hip.config() on rank 0).  No GPU needed: the call only formats text."""
    import json
    import subprocess
    import sys
    code = ("import sys, json; sys.path.insert(0, %r)\n"
            "from newtonnet_amd import hip\n"
            "print(json.dumps(hip.config()))\n" % ROOT)
    env = dict(os.environ, NNHIP_EDGE_LDS='12"3\\4', NNHIP_MOL_KERNELS_MIN='700')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    cfg = json.loads(r.stdout.strip().splitlines()[-1])
    assert cfg['env']['NNHIP_EDGE_LDS'] == '12?3?4' and cfg['env']['NNHIP_MOL_KERNELS_MIN'] == '700'
    assert cfg['molecule_forms']['edge_kernels_from_molecules'] == 700 and cfg['version'] >= 108
