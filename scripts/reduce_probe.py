"""Slab reduction (da_wgrad_reduce_multi) alone, at the resnet18 B = 64 step's shapes: us per launch (hipGraph replay of 20
launches, warm) against a plain copy of the same bytes.  usage: python scripts/reduce_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H

# (co, ci, k, splits) of the step: Winograd-form slabs of 640 pairs / 320 quads, the direct kernels' plans approximated
SHAPES = [(64, 64, 3, 56)] * 4 + [(128, 128, 3, 28)] * 3 + [(256, 256, 3, 14)] * 3 + [(512, 512, 3, 8)] * 3 + \
    [(128, 64, 3, 16), (128, 64, 1, 16), (256, 128, 3, 12), (256, 128, 1, 12), (512, 256, 3, 8), (512, 256, 1, 8)]


def timed(fn, n=20, rounds=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        best = 1e9
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    dev = torch.device('cuda:0')
    items, nbytes = [], 0
    for co, ci, k, sp in SHAPES:
        slab = torch.randn(sp * k * co * ci, device=dev)
        dw = torch.zeros(co, ci, k, device=dev)
        items.append(((slab, sp, k, co, ci), dw))
        nbytes += slab.numel() * 4
    for acc in (False, True):
        t = timed(lambda: H.wgrad_reduce_multi(items, accumulate=acc))
        print('reduce accumulate=%d: %.1f us, %.0f MB of slabs -> %.2f TB/s' % (acc, t, nbytes / 1e6, nbytes / t / 1e6))
    big = [it for it in items if it[0][3] == 512 and it[0][2] == 3][:3]
    bb = sum(it[0][0].numel() * 4 for it in big)
    t = timed(lambda: H.wgrad_reduce_multi(big, accumulate=True))
    print('three 512x512x3 x 8 splits: %.1f us, %.0f MB -> %.2f TB/s' % (t, bb / 1e6, bb / t / 1e6))
    src = torch.randn(nbytes // 8, device=dev)
    dst = torch.empty_like(src)
    t = timed(lambda: dst.copy_(src))
    print('copy of %.0f MB (read) + same written: %.1f us' % (nbytes / 2e6, t))
    # reference: a pure read-reduce of the same bytes (sum over dim 0 of a [8][n] view)
    v = torch.randn(8, nbytes // 32, device=dev)
    o = torch.empty(nbytes // 32, device=dev)
    t = timed(lambda: torch.sum(v, 0, out=o))
    print('torch.sum over 8 slabs of the same bytes: %.1f us' % t)


def cold():
    """the same launch with the slabs cold (a 1 GB buffer written in between): what the step sees behind 700 us of weight
    gradients"""
    dev = torch.device('cuda:0')
    items, nbytes = [], 0
    for co, ci, k, sp in SHAPES:
        slab = torch.randn(sp * k * co * ci, device=dev)
        dw = torch.zeros(co, ci, k, device=dev)
        items.append(((slab, sp, k, co, ci), dw))
        nbytes += slab.numel() * 4
    junk = torch.empty(256 * 1024 * 1024, device=dev)
    v = torch.randn(8, nbytes // 32, device=dev)
    o = torch.empty(nbytes // 32, device=dev)
    for name, fn in (('reduce', lambda: H.wgrad_reduce_multi(items, accumulate=True)), ('torch.sum 8 slabs', lambda: torch.sum(v, 0, out=o))):
        ts = []
        for _ in range(6):
            junk.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print('cold %s: %s us (single launches, event-bracketed: +~8 us of bracket)' % (name, ' '.join('%.1f' % t for t in ts)))


if __name__ == '__main__':
    main()
    cold()
