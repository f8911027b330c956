"""Sustained clock / power of the GPU while the conv stack runs alone (AdaIN path: no eigensolver) and while the full
WCT step runs: `rocm-smi` sampled from a side thread.  The fp16 dense peak the roofline is priced against (2.5 PFLOP/s)
assumes the 2.4 GHz boost clock; this records what the chip sustains under these kernels.
usage: python tools/r03_clocks.py [seconds per leg]"""
import sys, os, time, threading, subprocess, re, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd import weights as W

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
TARGETS = ['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1']
ctx = Context(0)
ctx.set_weights(W.synthetic_weights(0, TARGETS))
B, S = 32, 512
rng = np.random.default_rng(0)
c = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
s = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
dc, ds, do = ctx.dev_alloc(c.nbytes), ctx.dev_alloc(s.nbytes), ctx.dev_alloc(c.nbytes)
ctx.h2d(dc, c); ctx.h2d(ds, s)


def sample(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(t)
            card = d[sorted(d)[0]]
            rec = {}
            for k, v in card.items():
                m = re.search(r'\((\d+)Mhz\)', str(v))
                if 'sclk' in k and m:
                    rec['sclk'] = int(m.group(1))
                if 'mclk' in k and m:
                    rec['mclk'] = int(m.group(1))
                if 'ower' in k and 'W' in k:
                    try:
                        rec['power'] = float(v)
                    except ValueError:
                        pass
            out.append(rec)
        except Exception as e:                      # noqa: BLE001
            out.append({'err': str(e)[:80]})
        time.sleep(0.05)


def leg(name, adain):
    for _ in range(2):
        ctx.stylize_batch_dev(dc, S, S, ds, S, S, B, TARGETS, 0.8, do, adain=adain)
    ctx.sync()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(4):
            ctx.stylize_batch_dev(dc, S, S, ds, S, S, B, TARGETS, 0.8, do, adain=adain)
        ctx.sync(); n += 4
    dt = time.time() - t0
    stop.set(); th.join()
    good = [r for r in out if 'sclk' in r]
    sc = sorted(r['sclk'] for r in good)
    pw = sorted(r['power'] for r in good if 'power' in r)
    print('%-28s %6.1f frames/s  %.2f ms/step | samples %d  sclk MHz min/median/max %s  power W median %s max %s  mclk %s' % (
        name, n * B / dt, dt / n * 1e3, len(good),
        (sc[0], sc[len(sc) // 2], sc[-1]) if sc else None, pw[len(pw) // 2] if pw else None, pw[-1] if pw else None,
        sorted(set(r.get('mclk') for r in good))), flush=True)
    if not good and out:
        print('  sampler:', out[:2])


idle = []
ev = threading.Event(); th = threading.Thread(target=sample, args=(ev, idle)); th.start(); time.sleep(0.6); ev.set(); th.join()
print('idle:', idle[-1] if idle else None)
leg('conv stack alone (AdaIN)', True)
leg('full WCT step', False)
