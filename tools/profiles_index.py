#!/usr/bin/env python3
"""Regenerate profiles/INDEX.md: every committed measurement file -> what it measured -> the tree it was taken on.
The description comes from the file-name pattern (the names follow r<round>_[v<pass>_]<what>), the tree from the `# tree <sha>`
header tools/profile_round.sh writes (or the bench line's own fields).   usage: python tools/profiles_index.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')

PATTERNS = [
    (r'INDEX\.md$', 'this index (tools/profiles_index.py)'),
    (r'HISTORY_.*\.md$', 'experiment log moved out of DESIGN.md (what was tried, measured and not kept)'),
    (r'README\.md$', 'how the summaries in this directory are produced'),
    (r'\.csrc_sha$', 'sha256 prefix of the kernel sources the LATEST config-2 trace / PMC passes ran (bench.py compares it with the tree)'),
    (r'fresh_lease_.*\.json$', 'bench.py line taken as the FIRST command of a fresh GPU lease (headline reproducibility)'),
    (r'box100k.*bench.*\.json$|bench_box100k\.json$', 'bench.py --workload box100k line (BASELINE configs[4], 100k-atom periodic box)'),
    (r'bench_mode_train.*\.json$', 'bench.py --mode train line (data-parallel training step, configs[3] per-rank batch)'),
    (r'train_large\.json$', 'large-batch (1024 aspirin conformers) training step: ms/step, weight-gradient launch'),
    (r'bench.*\.json$', 'bench.py line, default command (config 2: 1024 aspirin conformers, 1 GPU)'),
    (r'bench_train\.txt$', 'tools/bench_train.py: training step times across batch shapes'),
    (r'box100k.*kernel_stats\.txt$', 'rocprofv3 --kernel-trace --stats summary of the 100k-atom box step'),
    (r'box100k.*pmc_(fetch|write)_size\.txt$', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel, 100k-atom box step'),
    (r'train.*kernel_stats.*\.txt$', 'rocprofv3 --kernel-trace --stats summary of a training step'),
    (r'train.*pmc_(fetch|write)_size\.txt$', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel, training step'),
    (r'md_kernel_stats\.txt$', 'rocprofv3 kernel trace of the one-molecule MD step (calculator path)'),
    (r'kernel_stats_2_in_flight\.txt$', 'the same trace with the bench default of two steps in flight on two lanes / streams (kernels overlap: durations include what they share the chip with)'),
    (r'two_stream.*\.txt$', 'steps in flight (model.inference_lanes): one batch split over streams vs whole independent steps over 2 / 3 / 4 lanes, us per step by batch size'),
    (r'kernel_stats\.txt$', 'rocprofv3 --kernel-trace --stats summary (kernel x grid: calls, total ms, avg us) of the config-2 bench command'),
    (r'pmc_fetch_size\.txt$', 'rocprofv3 --pmc FETCH_SIZE per kernel x grid (KiB; x2 on gfx950 for wide reads), config-2 step'),
    (r'pmc_write_size\.txt$', 'rocprofv3 --pmc WRITE_SIZE per kernel x grid (KiB), config-2 step'),
    (r'pmc_edge_p\d\.txt$', 'rocprofv3 counter pass over the edge kernels (pass N of tools/pmc_edge*.sh: L2 / TA / SQ wait counters)'),
    (r'pmc_issue_p\d\.txt$', 'rocprofv3 SQ issue / wait counters (tools/pmc_issue.sh)'),
    (r'pmc_mfma\.txt$|pmc_sq_.*\.txt$', 'rocprofv3 matrix-pipe / SQ busy counters of the dense kernels'),
    (r'pmc_mol_vs_row.*\.txt$|split_rows_pmc.*\.txt$', 'rocprofv3 counters of two forms of an edge kernel side by side'),
    (r'phase_clock.*\.txt$', 'wall-clock stamps behind every workgroup barrier of a kernel (tooling build)'),
    (r'timeline.*\.txt$', 'dispatch timeline of one step (start / end / gap of every kernel): idle time between launches'),
    (r'md_latency.*\.txt$', 'one-molecule latency path: model() / calculator step times (tools/bench_latency.py)'),
    (r'host_profile.*\.txt$', 'cProfile of the host side of a one-molecule call'),
    (r'sweep_.*\.txt$', 'throughput sweep over batch sizes / molecule shapes'),
    (r'small_thresholds\.txt$|crossover\.txt$|by_molecule_size\.txt$', 'where one kernel form overtakes another (threshold choice)'),
    (r'ubench_.*\.txt$', 'micro-benchmark of the pool (tools/ubench): streaming bandwidth / first touch / grid barrier'),
    (r'lookback\.txt$', 'single-launch per-molecule neighbor list with a decoupled look-back: timing (not kept)'),
    (r'deferred_fuzz\.txt$', 'tools/fuzz_deferred.py: deferred step vs synchronous path vs oracle over random batches'),
    (r'gpu_tests.*\.txt$', 'pytest -m gpu output on the GPU box'),
    (r'_ab.*\.txt$|_ab_.*\.txt$|experiment.*\.txt$|variants.*\.txt$', 'same-box A/B of a code variant (kept or not: see the file head and HISTORY / DESIGN)'),
    (r'train_step\.txt$', 'training step timing'),
]


def tree_of(path):
    try:
        with open(path, errors='replace') as f:
            head = f.read(4096)
    except OSError:
        return ''
    m = re.search(r'#\s*tree\s+([0-9a-f]{7,40}|unknown)', head)
    if m:
        return m.group(1)[:9]
    if path.endswith('.json'):
        try:
            d = json.loads(head if head.rstrip().endswith('}') else open(path).read().splitlines()[-1])
            for k in ('head', 'tree', 'git_head'):
                if isinstance(d, dict) and k in d:
                    return str(d[k])[:9]
        except Exception:  # noqa: BLE001
            pass
    return ''


_ADDED = {}


def added_in(name):
    """the commit that added the file (for files without a `# tree` header: the measured tree is that commit's)"""
    if not _ADDED:
        import subprocess
        try:
            out = subprocess.run(['git', 'log', '--diff-filter=A', '--name-only', '--format=@%h', '--', 'profiles'], cwd=ROOT,
                                 capture_output=True, text=True, timeout=60).stdout
        except Exception:  # noqa: BLE001
            out = ''
        sha = ''
        for line in out.splitlines():
            if line.startswith('@'):
                sha = line[1:]
            elif line.startswith('profiles/'):
                _ADDED.setdefault(os.path.basename(line), sha)
        _ADDED.setdefault('', '')
    return _ADDED.get(name, '')


def describe(name):
    for pat, text in PATTERNS:
        if re.search(pat, name):
            return text
    return ''


def main():
    names = sorted(os.listdir(PROF), key=lambda n: (re.sub(r'_v(\d)_', r'_v0\1_', n)))
    rows = ['# profiles/ index', '',
            'file -> what it measured -> tree.  Names: `r<round>_[v<pass>_]<what>`; a later pass of a round supersedes an earlier one of the',
            'same `<what>`; `*_ab*` files are same-box A/Bs of one change.  bench.py quotes the LATEST `r*_v*_kernel_stats.txt` /',
            '`*_pmc_{fetch,write}_size.txt` of the config-2 command and reports whether `.csrc_sha` still matches the kernel sources.',
            'Regenerate with `python tools/profiles_index.py`.', '', '| file | what | tree |', '|---|---|---|']
    for n in names:
        if n.startswith('.') and n != '.csrc_sha':
            continue
        t = tree_of(os.path.join(PROF, n))
        if not t or t == 'unknown':
            a = added_in(n)
            t = f'(added in {a})' if a else ''
        rows.append(f'| `{n}` | {describe(n)} | {t} |')
    with open(os.path.join(PROF, 'INDEX.md'), 'w') as f:
        f.write('\n'.join(rows) + '\n')
    missing = [n for n in names if not describe(n) and not n.startswith('.')]
    print(f'{len(names)} files indexed; {len(missing)} without a description: {missing[:10]}')


if __name__ == '__main__':
    main()
