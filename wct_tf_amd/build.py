"""Build libwct_hip.so (gfx950) in-tree with hipcc.  `python -m wct_tf_amd.build`."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libwct_hip.so')
SOURCES = ['api.hip', 'conv.hip', 'conv_wino.hip', 'conv_tail.hip', 'wct.hip', 'coral.hip', 'train.hip']


STAMP = LIB + '.src.sha256'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result']
# WCT_BUILD_TUNING=1 (experiments on the GPU box, tools/): compile the A-B / tuning switches in (-DWCT_TUNING: they are read
# from the environment); the product build has them as constants (csrc/common.h)
if os.environ.get('WCT_BUILD_TUNING'):
    FLAGS = FLAGS + ['-DWCT_TUNING'] + os.environ.get('WCT_BUILD_DEFS', '').split()      # (extra -D switches of an experiment)
# per-file additions.  wct.hip: the SLP vectoriser packs the eigensolver's rotation arithmetic into v_pk_fma_f32 pairs at the
# price of ~20 v_mov per rotation set for operand assembly (measured: Jacobi 25.2 -> 23.8 ms per 32-pair step without it)
FILE_FLAGS = {'wct.hip': ['-fno-slp-vectorize']}


def source_digest():
    """sha256 over every source the library is built from (csrc/*.hip, *.h, the public header) and the flags."""
    h = hashlib.sha256((' '.join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'wct_hip.h'))
    for d in deps:
        h.update(os.path.basename(d).encode())
        h.update(open(d, 'rb').read())
    return h.hexdigest()


def needs_build():
    """The .so is current iff the digest stored beside it equals the digest of the sources as they are now
    (modification times do not survive a snapshot to the GPU box; content does)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    return open(STAMP).read().strip() != source_digest()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(source_digest() + '\n')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
