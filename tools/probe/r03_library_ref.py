"""What the vendor libraries reach on this box for the same arithmetic (reference points for the conv roofline fraction;
torch is used here as a front-end to hipBLASLt / MIOpen only, nothing in the product path calls them):
  * fp16 GEMMs of the conv layers' implicit-GEMM shapes (M = pixels of 32 frames, N = Cout, K = 9 Cin) and 8192^3,
  * MIOpen's fp16 3x3 convolution on the layer shapes (NHWC / channels_last),
with the sustained clock / power sampled from rocm-smi beside them."""
import sys, os, time, threading, subprocess, re, json
import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')


def sampler(stop, out):
    while not stop.is_set():
        try:
            d = json.loads(subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout)
            card = d[sorted(d)[0]]
            rec = {}
            for k, v in card.items():
                m = re.search(r'\((\d+)Mhz\)', str(v))
                if 'sclk' in k and m:
                    rec['sclk'] = int(m.group(1))
                if 'ower' in k and 'W' in k:
                    try:
                        rec['power'] = float(v)
                    except ValueError:
                        pass
            out.append(rec)
        except Exception:                            # noqa: BLE001
            pass
        time.sleep(0.05)


def timed(name, fn, flops, secs=2.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    one = e0.elapsed_time(e1)
    reps = max(5, int(secs * 1e3 / max(one, 1e-3)))
    stop, out = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, out)); th.start()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    ms = e0.elapsed_time(e1) / reps
    sc = sorted(r['sclk'] for r in out if 'sclk' in r); pw = sorted(r['power'] for r in out if 'power' in r)
    print('%-44s %8.3f ms %7.0f TFLOP/s  sclk median %s MHz  power median %s W' % (
        name, ms, flops / ms / 1e9, sc[len(sc) // 2] if sc else None, pw[len(pw) // 2] if pw else None), flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else 'gemm,conv'
if 'gemm' in which:
    for m, n, k in [(8192, 8192, 8192), (32 * 64 * 64, 512, 4608), (32 * 128 * 128, 256, 2304), (32 * 256 * 256, 128, 1152), (32 * 512 * 512, 64, 576)]:
        a = torch.randn(m, k, device=dev, dtype=torch.float16); b = torch.randn(n, k, device=dev, dtype=torch.float16)
        timed('hipBLASLt fp16 GEMM %d x %d x %d (NT)' % (m, n, k), lambda: torch.matmul(a, b.t()), 2.0 * m * n * k)
        del a, b
if 'conv' in which:
    torch.backends.cudnn.benchmark = True
    for cin, cout, h in [(512, 512, 64), (256, 256, 128), (128, 128, 256), (64, 64, 512)]:
        x = torch.randn(32, cin, h, h, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.float16) * 0.05).contiguous(memory_format=torch.channels_last)
        try:
            timed('MIOpen fp16 conv3x3 %d->%d @%d x32 (NHWC, zero pad)' % (cin, cout, h), lambda: F.conv2d(x, w, None, padding=1),
                  2.0 * 32 * h * h * 9 * cin * cout)
        except Exception as e:                       # noqa: BLE001
            print('MIOpen conv %d->%d @%d failed: %s' % (cin, cout, h, str(e)[:200]))
        del x, w
