"""The round-4 pair-problem kernel source (wct_tf_amd/csrc/jacobi_dev.h, namespace r4: the 64 x 64 pair problem resident
in registers, the 256-thread tile update, the look-ahead assembly) compiled for the HOST and executed lane by lane
(tests/emul/hip_emul.h: 256 threads per workgroup, pthread barriers, emulated DPP / shuffles / MFMA) against plain
sequential arithmetic in double precision.  There is no GPU in the build container: this is how the kernel's data routing
is checked on every CPU test run; the -m gpu tests then check the same code on the hardware against LAPACK."""
import os
import subprocess

import pytest

from conftest import ROOT

CLANG = '/opt/rocm/lib/llvm/bin/clang++'


@pytest.mark.skipif(not os.path.exists(CLANG), reason='ROCm clang++ not found')
@pytest.mark.parametrize('variant,u_f16', [(0, 0), (1, 0), (0, 1)], ids=['8-strips-4-waves', '16-strips-8-waves', '8-strips-4-waves-split-fp16-tile-update'])
def test_r4_pair_problem_source_emulated_on_the_cpu(tmp_path, variant, u_f16):
    exe = str(tmp_path / 'jacobi_r4_emul')
    src = os.path.join(ROOT, 'tests', 'emul', 'jacobi_r4_emul.cpp')
    subprocess.check_call([CLANG, '-std=c++17', '-O1', '-pthread', '-Wno-unused-value', '-DEMUL_VAR=%d' % variant] + (['-DEMUL_U_F16'] if u_f16 else [])
                          + ['-o', exe, src])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert 'all checks passed' in out.stdout and 'FAIL' not in out.stdout


def test_patch_routing_prototype():
    """tools/jacobi_patch_proto.py: the routing the kernel was written from (P = 1, 2, 4 cells per lane edge)"""
    import runpy
    runpy.run_path(os.path.join(ROOT, 'tools', 'jacobi_patch_proto.py'), run_name='__main__')
