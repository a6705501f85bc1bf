"""Builds libexorl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / 'csrc'
LIB = HERE / 'libexorl_hip.so'
SOURCES = ['api.cpp', 'comm.cpp', 'gemm.hip', 'rowops.hip', 'fused.hip', 'loss.hip', 'cql.hip', 'optim.hip', 'replay.hip', 'agent.hip', 'knn.hip', 'intr.hip', 'pixels.hip', 'pixel_agent.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
if os.environ.get('EXORL_GEMM_EXPERIMENTS'):       # the measured-and-not-adopted GEMM schedules (tools/micro/ws_bench.py, ms16_bench.py)
    FLAGS.append('-DEXORL_GEMM_EXPERIMENTS')


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob('*')) + [HERE.parent / 'include' / 'exorl_hip.h']
    return any(p.stat().st_mtime > t for p in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = HERE / 'build'
    objdir.mkdir(exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = objdir / (src.rsplit('.', 1)[0] + '.o')
        cmd = [hipcc, *FLAGS, '-x', 'hip', '-c', str(CSRC / src), '-o', str(obj)]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        if verbose and out.strip():
            print(out, file=sys.stderr)
        objs.append(str(obj))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(LIB), *objs, '-L/opt/rocm/lib', '-lrccl']      # RCCL: exorl_comm_*
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
