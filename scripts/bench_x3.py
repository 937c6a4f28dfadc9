"""k3 s1 conv at the bench batch: fp32 Winograd kernels vs the split-bf16 ("f32x3") direct kernel, hipGraph timed, with
the error of each against an fp64 reference.   usage: python scripts/bench_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
ROWS = int(os.environ.get('ROWS', 1280))


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for ci, co, L in ((64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7)):
    torch.manual_seed(0)
    x = torch.randn(ROWS, L, ci, device='cuda'); w = torch.randn(co, ci, 3, device='cuda') * (2.0 / (3 * co)) ** 0.5
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), padding=1).transpose(1, 2)
    scale = ref.abs().max().item()
    u = H.wino_weights(w, points=6 if ci >= 512 else 4)
    yw = torch.empty(ROWS, L, co, device='cuda'); H.conv3_winograd(x, u, out=yw)
    wf, wd = H.pack_conv3_x3(w)
    yx = H.conv3_x3(x, wf)
    wb, _ = H.pack_conv3_bf16(w)
    yb = H.conv3_bf16(x, wb)
    wfd, _ = H.repack_weight(w, True, True)
    yd = H.conv_fwd(x, wfd, 1, 1)
    ew, ex, eb, ed = [((y.double() - ref).abs().max().item() / scale) for y in (yw, yx, yb, yd)]
    rw, rx = [((y.double() - ref).pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item()) for y in (yw, yx)]
    tw = graph_time(lambda: H.conv3_winograd(x, u, out=yw))
    tx = graph_time(lambda: H.conv3_x3(x, wf, out=yx))
    tb = graph_time(lambda: H.conv3_bf16(x, wb, out=yb))
    fl = 2.0 * ROWS * L * ci * co * 3
    print('%4d->%4d L %2d | fp32 winograd %6.1f us %6.1f TF(alg) maxerr %.1e rms %.1e | f32x3 %6.1f us %6.1f TF(alg) maxerr %.1e rms %.1e '
          'x%.2f | bf16 %6.1f us maxerr %.1e | fp32 direct maxerr %.1e' %
          (ci, co, L, tw, fl / tw / 1e6, ew, rw, tx, fl / tx / 1e6, ex, rx, tw / tx, tb, eb, ed), flush=True)
    # data gradient pack: conv of dy with the reversed, transposed taps
    dy = torch.randn(ROWS, L, co, device='cuda')
    dref = torch.nn.functional.conv_transpose1d(dy.double().transpose(1, 2), w.double(), padding=1).transpose(1, 2)
    dx = H.conv3_x3(dy, wd)
    print('      dgrad maxerr %.1e' % ((dx.double() - dref).abs().max().item() / dref.abs().max().item()), flush=True)
