#!/usr/bin/env python3
"""List the places in the gfx950 ISA of a .hip file where a wave resumes on `s_waitcnt vmcnt(N)`, N > 0, while a STORE that is
younger than a load the wait is meant to cover is still outstanding.

Why it exists: the compiler counts vector-memory loads and stores on one in-order counter (gfx9 family: no separate store counter), so
`vmcnt(N)` is taken to mean "everything but the youngest N operations has completed".  While hunting round 5's intermittent wrong
molecule ONE HYPOTHESIS was that on MI355X a store's acknowledgement can overtake an OLDER load's data, letting such a wait through
early.  That hypothesis was TESTED AND NOT CONFIRMED: tools/probes/vmcnt_order_probe.hip checked 1.5e10 lane-values under memory
pressure with 0 wrong (profiles/r05_vmcnt_order_probe.txt), and the error was traced to packed-fp32 `op_sel` chains instead
(profiles/r05_mol_fused2_soak.txt section 7; build.sh now compiles without them).  The compiler's vmcnt accounting is NOT known to be
unsafe on this part.  The tool stays as an over-approximating audit helper: it lists where a kernel RELIES on load / store return
order ([... load L ... store S ...] pending and a wait with N >= the operations issued after L), nothing more.

The scan simulates the pending queue over the linear text of each kernel and replays every loop body once more from its
back-edge (loop-carried prefetches).  It over-approximates (every path is taken as fall-through), so a clean report is the useful
outcome; a listed site needs a look at the source.

usage: python tools/scan_vmcnt.py newtonnet_amd/csrc/edge.hip [-D...]     (or a .s file)"""
import os, re, subprocess, sys, tempfile

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def isa_of(path, extra):
    if path.endswith('.s'):
        return open(path).read()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-I', os.path.join(ROOT, 'include'),
                        path, '-o', out] + extra, check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


LOAD = re.compile(r'^\s*(global_load|buffer_load|flat_load|scratch_load|global_atomic\S*\s.*\bsc0\b|buffer_inv)')
STORE = re.compile(r'^\s*(global_store|buffer_store|flat_store|scratch_store|global_atomic)')
WAIT = re.compile(r'^\s*s_waitcnt\b(.*)$')
VM = re.compile(r'vmcnt\((\d+)\)')
LABEL = re.compile(r'^(\.LBB\d+_\d+):')
BRANCH = re.compile(r'^\s*s_c?branch\S*\s+(\.LBB\d+_\d+)')


def scan_function(name, lines):
    labels = {}
    for i, ln in enumerate(lines):
        m = LABEL.match(ln)
        if m:
            labels[m.group(1)] = i
    sites = {}

    def run(lo, hi, pending, replay):
        i = lo
        while i < hi:
            ln = lines[i]
            if LOAD.match(ln):
                pending.append(('L', i))
            elif STORE.match(ln):
                pending.append(('S', i))
            else:
                w = WAIT.match(ln)
                if w:
                    v = VM.search(w.group(1))
                    if v is not None or 'vmcnt' not in w.group(1) and w.group(1).strip() in ('0', ''):
                        n = int(v.group(1)) if v else 0
                        if n == 0:
                            pending.clear()
                        elif len(pending) > n:
                            retire, keep = pending[:-n], pending[-n:]
                            if any(k == 'S' for k, _ in keep) and any(k == 'L' for k, _ in retire):
                                sites.setdefault(i, (n, [j for k, j in retire if k == 'L'][-1], [j for k, j in keep if k == 'S'][0]))
                            del pending[:-n]
                elif not replay:
                    b = BRANCH.match(ln)
                    if b and b.group(1) in labels and labels[b.group(1)] <= i:
                        run(labels[b.group(1)], i, list(pending), True)       # the loop body once more, entered from its back-edge
            i += 1

    run(0, len(lines), [], False)
    return sites


def main():
    path = sys.argv[1]
    text = isa_of(path, sys.argv[2:])
    lines = text.split('\n')
    starts = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.match(r'^(_Z\w+):', ln)] if m]
    total = 0
    for k, (i, name) in enumerate(starts):
        end = next((j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end')), len(lines))
        sites = scan_function(name, lines[i:end])
        try:
            pretty = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
        except OSError:
            pretty = name
        print(f'{pretty}: {len(sites)} exposed wait(s)')
        for at, (n, ld, st) in sorted(sites.items())[:int(os.environ.get('SCAN_SHOW', '6'))]:
            print(f'    line +{at}: s_waitcnt vmcnt({n})  covers the load at +{ld} [{lines[i + ld].strip()[:60]}]  with the store at +{st} '
                  f'[{lines[i + st].strip()[:50]}] still counted')
        total += len(sites)
    print(f'total: {total}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
