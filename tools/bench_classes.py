"""bench.py at a few batch sizes, every class of the per-step breakdown printed (for A-B runs of a switch or a build)."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for b in (sys.argv[1:] or ['32', '8', '1']):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', b, '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-latency'],
                         capture_output=True, text=True).stdout
    l = json.loads([x for x in out.strip().splitlines() if x.startswith('{')][-1])
    br = l['breakdown_ms_per_step']
    print('batch %3s: %7.1f frames/s %6.2f ms/step (no_prof %.2f) | ' % (b, l['value'], l['ms_per_step'], l['no_prof']['ms_per_step'])
          + ' '.join('%s %.2f' % (k, v) for k, v in br.items() if v > 0.005) + ' | frames ' + (l.get('frames_sha256') or '')[:12], flush=True)
