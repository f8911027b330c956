"""Decoder training step (wct_train_step) against a torch-autograd CPU reference of
model.py:123-223 (oracle/train_oracle.py): losses, weight and bias gradients of every decoder conv, the Adam
update, and that training actually reduces the loss.  Tolerances: the GPU forward runs the inference kernels
(fp16 operands, fp32 accumulate) and the reference emulates exactly that rounding, so losses agree to ~1e-3;
the backward pass is fp32 on both sides but the GPU differentiates through the fp32 master weights while the
emulation differentiates through their fp16 roundings: gradients agree to ~1e-2 in relative L2."""
import numpy as np
import pytest

from conftest import rel_err
from oracle.train_oracle import train_losses_and_grads
from wct_tf_amd.weights import synthetic_weights, synthetic_image

pytestmark = pytest.mark.gpu


def fp16_exact(weights):
    """Weights that are exactly representable in fp16: the GPU's fp32 master copies (used by the backward pass)
    and the fp16 copies the forward kernels read are then the same numbers, as in the reference emulation."""
    r = lambda a: np.float32(np.float16(a))   # noqa: E731
    enc = {k: ((w, b) if k in ('preprocess', 'conv1_1') else (r(w), b)) for k, (w, b) in weights['encoder'].items()}
    dec = {k: [(r(w), b) for w, b in v] for k, v in weights['decoder'].items()}
    return {'encoder': enc, 'decoder': dec}


def batch(seed, b, h, w):
    return np.stack([synthetic_image(seed + i, h, w) / 255. for i in range(b)]).astype(np.float32)


def tame(weights, s):
    """Scale every conv after conv1_1 by s < 1: a contractive net.  The He-normal synthetic nets amplify a one-ulp
    fp16 difference in an activation by orders of magnitude on the way to a deep gradient (tests/test_oracle.py
    shows the same for the forward pass), which would hide a wrong adjoint behind 'noise'; the contractive nets do
    not, so their gradients must agree tightly layer by layer."""
    enc = {k: ((w, b) if k in ('preprocess', 'conv1_1') else (w * s, b)) for k, (w, b) in weights['encoder'].items()}
    dec = {k: [(w * s, b) for w, b in v] for k, v in weights['decoder'].items()}
    return {'encoder': enc, 'decoder': dec}


def run_both(relu, h, w, weights, fw=1.0, pw=0.7, tvw=1e-4):
    from wct_tf_amd.context import Context
    ctx = Context(0)
    ctx.set_weights(weights)
    x = batch(11, 2, h, w)
    got = ctx.train_step(relu, x, step=1, learning_rate=0.0, feature_weight=fw, pixel_weight=pw, tv_weight=tvw)
    mine = ctx.get_decoder(relu, grads=True)
    unchanged = ctx.get_decoder(relu)
    ctx.close()
    want, grads = train_losses_and_grads(x, weights, relu, fw, pw, tvw)
    for k in ('feature_loss', 'pixel_loss', 'tv_loss', 'total_loss'):
        assert abs(got[k] - want[k]) <= 2e-3 * abs(want[k]) + 1e-7, (k, got, want)
    for (w0, b0), (w1, b1) in zip(weights['decoder'][relu], unchanged):      # lr = 0 left the weights alone
        assert np.array_equal(np.float32(w0), w1) and np.array_equal(np.float32(b0), b1)
    errs = [(rel_err(gw, ww), rel_err(gb, wb)) for (gw, gb), (ww, wb) in zip(mine, grads)]
    print(relu, (h, w), 'loss %.6f / %.6f' % (got['total_loss'], want['total_loss']), ['%.1e' % e for e, _ in errs])
    return x, errs


@pytest.mark.parametrize('relu,h,w,tol', [('relu1_1', 12, 10, 1e-3), ('relu2_1', 16, 12, 1e-2), ('relu3_1', 16, 24, 1e-2),
                                          ('relu5_1', 32, 32, 2e-3)])
def test_train_step_gradients_match_autograd_contractive_nets(relu, h, w, tol):
    """Every adjoint of the chain (conv dgrad/wgrad, reflect-pad fold, ReLU masks, ceil-mode max-pool, x2 upsample,
    conv1_1 with the folded preprocess, the three losses) in the deepest configuration: relu5_1 = 13 decoder
    convs, 4 upsamples, 13 encoder convs, 4 pools."""
    weights = fp16_exact(tame(synthetic_weights(42, relu_targets=[relu]), 1.0 if relu == 'relu1_1' else 0.35))
    _, errs = run_both(relu, h, w, weights)
    for i, (ew, eb) in enumerate(errs):
        assert ew < tol and eb < tol, (relu, i, ew, eb)


@pytest.mark.parametrize('relu,h,w', [('relu2_1', 16, 12), ('relu3_1', 16, 24)])
def test_train_step_gradients_on_the_synthetic_nets(relu, h, w):
    """On the He-normal nets a one-ulp fp16 difference grows on its way to a deep gradient: the fp16-emulating
    reference and the exact-arithmetic version of the SAME net already differ by 1e-2 (relu2_1) .. 7e-2 (relu3_1).
    The GPU must be closer to the fp16 emulation than exact arithmetic is."""
    weights = fp16_exact(synthetic_weights(42, relu_targets=[relu]))
    x, errs = run_both(relu, h, w, weights, tvw=0.0)
    _, g16 = train_losses_and_grads(x, weights, relu, 1.0, 0.7, 0.0, emulate_fp16=True)
    _, g32 = train_losses_and_grads(x, weights, relu, 1.0, 0.7, 0.0, emulate_fp16=False)
    for i, ((ew, _), (a, _), (b, _)) in enumerate(zip(errs, g16, g32)):
        spread = rel_err(b, a)
        print('  conv %d: gpu vs fp16 emulation %.2e, exact arithmetic vs fp16 emulation %.2e' % (i, ew, spread))
        assert ew < spread and ew < 6e-2, (relu, i)


def test_training_reduces_the_loss_and_updates_the_forward_weights():
    from wct_tf_amd.context import Context
    relu = 'relu2_1'
    weights = synthetic_weights(42, relu_targets=[relu])
    ctx = Context(0)
    ctx.set_weights(weights)
    x = batch(21, 4, 32, 32)
    feat = ctx.encode(x[0], relu)
    before = ctx.decode(feat, relu)
    losses = [ctx.train_step(relu, x, step=t, learning_rate=1e-3)['total_loss'] for t in range(1, 41)]
    print(['%.4f' % v for v in losses[::5]])
    assert losses[-1] < 0.7 * losses[0]
    after = ctx.decode(feat, relu)                      # the inference path sees the trained (re-packed fp16) weights
    assert not np.array_equal(before, after)
    trained = ctx.get_decoder(relu)
    assert any(not np.array_equal(np.float32(w0), w1) for (w0, _), (w1, _) in zip(weights['decoder'][relu], trained))
    # Adam's first step moves every weight with a non-zero gradient by ~lr
    ctx2 = Context(0)
    ctx2.set_weights(weights)
    ctx2.train_step(relu, x, step=1, learning_rate=1e-3)
    w_first = ctx2.get_decoder(relu)[0][0]
    step = np.abs(w_first - np.float32(weights['decoder'][relu][0][0]))
    assert 0.5e-3 < np.median(step[step > 0]) < 1.5e-3
    # determinism: the same step from the same state gives the same bits
    ctx3 = Context(0)
    ctx3.set_weights(weights)
    ctx3.train_step(relu, x, step=1, learning_rate=1e-3)
    assert np.array_equal(ctx3.get_decoder(relu)[0][0], w_first)
    for c in (ctx, ctx2, ctx3):
        c.close()


def test_train_cli_end_to_end(tmp_path):
    """python -m wct_tf_amd.train on synthetic data: the loss goes down, checkpoints rotate, a second run resumes
    from the saved step, and the WCT facade stylizes with the trained decoder directory."""
    import json
    import os
    from wct_tf_amd.train import train
    from wct_tf_amd.wct import WCT
    from wct_tf_amd.weights import save_weights
    ck = str(tmp_path / 'ckpt')
    common = ['--checkpoint', ck, '--relu-target', 'relu2_1', '--synthetic-weights', '42', '--synthetic-data', '16',
              '--batch-size', '4', '--crop', '32', '--learning-rate', '1e-3', '--lr-decay', '1e-3', '--save-iter', '5',
              '--summary-iter', '4', '--max-to-keep', '2']
    train(common + ['--max-iter', '12'])
    log = [json.loads(l) for l in open(os.path.join(ck, 'log', 'train_log.jsonl'))]
    assert [r['step'] for r in log] == list(range(1, 13))
    assert log[-1]['total_loss'] < log[0]['total_loss']
    assert abs(log[5]['lr'] - 1e-3 / (1 + 5 * 1e-3)) < 1e-12 and 'val_total_loss' in log[4]
    state = json.load(open(os.path.join(ck, 'train_state.json')))
    assert state['step'] == 12 and len(state['snapshots']) == 2
    assert sorted(f for f in os.listdir(ck) if f.endswith('.npz')) == sorted(state['snapshots'] + ['decoder_relu2_1.npz'])
    train(common + ['--max-iter', '3'])                           # resumes: steps 13..15
    log = [json.loads(l) for l in open(os.path.join(ck, 'log', 'train_log.jsonl'))]
    assert [r['step'] for r in log][-3:] == [13, 14, 15]
    # the trained decoder loads through the reference-shaped facade
    weights = synthetic_weights(42, relu_targets=['relu2_1'])
    vgg = str(tmp_path / 'vgg.npz')
    save_weights(vgg, {'encoder': weights['encoder'], 'decoder': {}})
    out = WCT(checkpoints=[ck], relu_targets=['relu2_1'], vgg_path=vgg).predict(synthetic_image(1, 32, 32), synthetic_image(2, 32, 32), 0.8)
    assert out.shape == (32, 32, 3) and out.dtype == np.uint8


def test_split_step_equals_fused_step_and_rccl_allreduce_on_the_gradient_buffer():
    """The data-parallel building blocks on one GPU: train_step(lr=0) + train_apply(lr) equals train_step(lr) bit
    for bit; the library's contiguous gradient buffer can be wrapped by torch without a copy and all-reduced over
    RCCL (world size 1 here: the N>1 launch is `torchrun -m wct_tf_amd.train`)."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from wct_tf_amd.context import Context
    from wct_tf_amd.train import _DevArray
    relu = 'relu2_1'
    weights = synthetic_weights(42, relu_targets=[relu])
    x = batch(31, 4, 32, 32)
    a, b = Context(0), Context(0)
    for c in (a, b):
        c.set_weights(weights)
    for t in range(1, 4):
        ra = a.train_step(relu, x, step=t, learning_rate=1e-3)
        rb = b.train_step(relu, x, step=t, learning_rate=0.0)
        b.train_apply(relu, t, 1e-3)
        assert ra == rb
    for (wa, ba), (wb, bb) in zip(a.get_decoder(relu), b.get_decoder(relu)):
        assert np.array_equal(wa, wb) and np.array_equal(ba, bb)
    # the gradient buffer as a torch view + one RCCL all-reduce
    ptr, count = b.train_grad_buffer(relu)
    g = torch.as_tensor(_DevArray(ptr, count), device='cuda:0')
    assert g.numel() == count and g.data_ptr() == ptr
    before = g.clone()
    sock = socket.socket(); sock.bind(('127.0.0.1', 0)); port = sock.getsockname()[1]; sock.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert torch.equal(g, before)
    gw0 = b.get_decoder(relu, grads=True)[0][0]
    assert np.array_equal(gw0.ravel(), before[:gw0.size].cpu().numpy())      # layer 0's kernel gradient opens the buffer
    a.close(); b.close()
