"""In-tree build of the HIP library (hipcc, gfx950 only).  `python -m diffalign_amd.build`."""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libdiffalign_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result']
# the diffusion kernels keep the CPU oracle's arithmetic order (no fma contraction): probabilities match to 2e-6
EXTRA = {'k_diffusion.hip': ['-ffp-contract=off'], 'k_elbo.hip': ['-ffp-contract=off']}


def _stale(src, obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    jobs = []
    for f in srcs:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f[:-4] + '.o')
        if force or _stale(src, obj, hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ['-c', src, '-o', obj], capture_output=True, text=True)
        return src, r.returncode, r.stderr

    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for src, rc, err in ex.map(cc, jobs):
            if verbose:
                print(f'[build] {os.path.basename(src)} rc={rc}')
            if rc != 0:
                raise RuntimeError(f'hipcc failed for {src}:\n{err}')
    objs = [os.path.join(OBJ, f[:-4] + '.o') for f in srcs]
    manifest = os.path.join(OBJ, '.sources')                 # a removed or added source file changes the link line
    listed = open(manifest).read().split() if os.path.exists(manifest) else None
    if force or jobs or not os.path.exists(LIB) or listed != srcs or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr)
        with open(manifest, 'w') as f:
            f.write('\n'.join(srcs))
        if verbose:
            print(f'[build] linked {LIB}')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
