"""Build libi2t_hip.so (gfx950) in-tree with hipcc.  ``python -m image2text_amd.build`` or ``build_library()``."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libi2t_hip.so')
SOURCES = ['abi.cpp', 'comm.cpp', 'gemm.hip', 'norm.hip', 'attention.hip', 'elementwise.hip', 'conv.hip', 'conv_mfma.hip', 'decode.hip', 'sample.hip',
           'attention_g.hip', 'family.hip', 'grouped.hip', 'llama.hip', 'lora.hip', 'vit.hip', 'fp8.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast']
# per-file additions.  attention: keep MFMA accumulators in VGPRs -- the softmax rescales them every key tile, and the
# AGPR form cost ~110 v_accvgpr_read/write per tile in kernels that are VALU-bound
EXTRA_FLAGS = {'attention.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'attention_g.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'attention_common.h'),
                                                      os.path.join(CSRC, '..', '..', 'include', 'i2t.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    # one builder at a time (ranks of one job, pytest-xdist workers): the others wait, then find the library fresh
    import fcntl
    with open(os.path.join(CSRC, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return LIB
        return _build(verbose, force_all=force)


def _build(verbose: bool, force_all: bool = False) -> str:
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    headers = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'attention_common.h'), os.path.join(CSRC, '..', '..', 'include', 'i2t.h'),
               os.path.abspath(__file__)]
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + '.o')
        objs.append(obj)
        if not force_all and os.path.exists(obj) and all(os.path.getmtime(d) < os.path.getmtime(obj) for d in [src] + headers):
            continue                                      # this object is newer than its source and the shared headers
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + (['-x', 'hip'] if s.endswith('.cpp') else []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n{out.decode()}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-o', LIB] + objs + ['-ldl']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout.decode()}')
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
