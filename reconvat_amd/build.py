"""Build libreconvat_hip.so (gfx950 only) in-tree with hipcc.  No JIT cache: the .so lives next to the
sources so it travels with the repo snapshot to the GPU box."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libreconvat_hip.so')
SOURCES = ['conv.hip', 'conv_wino2.hip', 'bn.hip', 'gemm.hip', 'gemm_bf16x6.hip', 'attn.hip', 'elementwise.hip', 'mel.hip', 'data.hip', 'lstm.hip', 'api.cpp']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def source_digest():
    """sha256 (first 16 hex digits) over every kernel / ABI source in csrc/ (names and contents, sorted): baked into the library as
    rv_source_digest() and compared by reconvat_amd._lib.load() with the sources that travelled next to the .so."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(('.hip', '.h', '.cpp')):
            h.update(name.encode() + b'\0')
            with open(os.path.join(CSRC, name), 'rb') as fh:
                h.update(fh.read())
            h.update(b'\0')
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
    path = os.path.join(CSRC, src)
    deps = [path, os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_shared.h')]
    extra, stale = [], _stale(obj, deps)
    if src == 'api.cpp':                       # carries the digest of ALL sources: rebuilt whenever any of them changed
        dig = source_digest()
        extra = [f'-DRV_SOURCE_DIGEST="{dig}"']
        side = obj + '.digest'
        stale = stale or not os.path.exists(side) or open(side).read().strip() != dig
    if stale:
        cmd = [HIPCC] + FLAGS + extra + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', path, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stderr}')
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        if src == 'api.cpp':
            with open(obj + '.digest', 'w') as fh:
                fh.write(extra[0].split('"')[1])
    return obj


def build(force=False, verbose=False):
    if force:
        for s in SOURCES:
            o = os.path.join(CSRC, os.path.splitext(s)[0] + '.o')
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stderr}')
    if verbose:
        print('built', LIB)
    return LIB


def build_ablation():
    """Timing-experiment build (wrong results by design): conv.hip with -DRV_ABLATION so that RV_ABLATE=bits switches parts of
    the conv / wgrad kernels off.  Lives next to the real library as libreconvat_hip_abl.so; load it with
    RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_abl.so (tools only -- never the product default)."""
    build()
    obj = os.path.join(CSRC, 'conv_abl.o')
    r = subprocess.run([HIPCC] + FLAGS + ['-DRV_ABLATION', '-c', os.path.join(CSRC, 'conv.hip'), '-o', obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    objs = [os.path.join(CSRC, os.path.splitext(s)[0] + '.o') for s in SOURCES if s != 'conv.hip'] + [obj]
    lib = os.path.join(HERE, 'libreconvat_hip_abl.so')
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', lib], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    print('built', lib)
    return lib


if __name__ == '__main__':
    if '--ablation' in sys.argv:
        build_ablation()
    else:
        build(force='--force' in sys.argv, verbose=True)
