"""k3 s1 conv at the bench batch: Winograd fp32 (F(2,3) / F(4,3)) vs the bf16-MFMA kernel; errors vs torch fp64.
usage: python scripts/bench_bf16.py"""
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
    x = torch.randn(ROWS, L, ci, device='cuda'); w = torch.randn(co, ci, 3, device='cuda') * (2.0 / (3 * co)) ** 0.5
    ref = torch.nn.functional.conv1d(x.permute(0, 2, 1).double(), w.double(), padding=1).permute(0, 2, 1)
    u = H.wino_weights(w, points=6 if ci >= 512 else 4)
    wf, wd = H.pack_conv3_bf16(w)
    yw = H.conv3_winograd(x, u); yb = H.conv3_bf16(x, wf)
    sc = ref.abs().max().item()
    ew, eb = (yw.double() - ref).abs().max().item() / sc, (yb.double() - ref).abs().max().item() / sc
    # what exact bf16 rounding of the operands gives in fp64
    refb = torch.nn.functional.conv1d(x.bfloat16().double().permute(0, 2, 1), w.bfloat16().double(), padding=1).permute(0, 2, 1)
    ebb = (yb.double() - refb).abs().max().item() / sc
    tw = graph_time(lambda: H.conv3_winograd(x, u, out=yw)); tb = graph_time(lambda: H.conv3_bf16(x, wf, out=yb))
    fl = 2.0 * ROWS * L * ci * co * 3
    print('%4d->%4d L %2d  winograd fp32 %6.1f us (err %.1e) | bf16 %6.1f us %6.1f TF  x%.2f  err %.1e (vs bf16-rounded operands %.1e)'
          % (ci, co, L, tw, ew, tb, fl / tb / 1e6, tw / tb, eb, ebb))
