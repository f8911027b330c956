"""Resource table of every kernel the library ships: hipcc -Rpass-analysis=kernel-resource-usage over csrc/*.hip with the product's flags
(VGPRs / AGPRs / scratch bytes per lane / spills / occupancy / static LDS).  usage: python tools/kernel_resources.py > profiles/<tag>_kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wct_tf_amd import build
src = [os.path.join(ROOT, 'wct_tf_amd', 'csrc', s) for s in build.SOURCES]
print('# hipcc %s -Rpass-analysis=kernel-resource-usage  (the flags of wct_tf_amd/build.py)' % ' '.join(build.FLAGS))
print('%-8s %5s %5s %8s %7s %7s %5s %8s  %s' % ('file', 'VGPR', 'AGPR', 'scratch', 'vspill', 'sspill', 'occ', 'LDS', 'kernel'))
worst = 0
for f in src:
    out = subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + build.FLAGS + build.FILE_FLAGS.get(os.path.basename(f), []) +
                         ['-I', os.path.join(ROOT, 'include'), '-c', f, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'],
                         capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r'remark:\s+Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r'remark:\s+([A-Za-z /\[\]]+?):\s+(\S+)', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    for r in rows:
        if 'VGPRs' not in r:
            continue
        name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(.*', '', name.replace('(anonymous namespace)::', '')).replace('void ', '')
        sc = int(r.get('ScratchSize [bytes/lane]', '0'))
        worst = max(worst, sc)
        print('%-8s %5s %5s %8d %7s %7s %5s %8s  %s' % (os.path.basename(f)[:8], r['VGPRs'], r.get('AGPRs', '0'), sc, r.get('VGPRs Spill', '0'), r.get('SGPRs Spill', '0'),
                                                     r.get('Occupancy [waves/SIMD]', '?'), r.get('LDS Size [bytes/block]', '0'), name))
print('# largest scratch size of any kernel: %d bytes per lane' % worst)
