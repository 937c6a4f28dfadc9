#!/usr/bin/env python3
"""bench.py -- breath-sequences/sec of one cnn_linear TRAIN step (BASELINE.json metric).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = zero-grad + forward + BCE-with-logits + backward + (+-0.01 clamp, weight decay, SGD-Nesterov
update) over one synthetic batch of (B, 20, 1, 224) fp32 windows already resident in HBM -- the
reference's run_train_epoch body (train_ards_detector.py:139-173).  N > 1: one process per GPU,
each rank owns B windows (weak scaling), one RCCL all-reduce of the flat gradient bucket per step.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel family (implicit-GEMM conv on the fp32 MFMA pipe): algorithmic
                  FLOPs of its launches / their HIP-event durations, measured live on the launch stream
  cpu_baseline -- the reference's CPU path (oracle/torch_ref.py, stock ATen ops, all host threads)
                  timed on a bounded sample of the same workload (N=1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per breath-sequence: FLOPs train, activation bytes train fp32, #params
WORK = {
    'resnet18': dict(flops=228.665e6, act_bytes=1421.1e3, act_bytes_bf16=710.5e3, params=3864386),
    'densenet18': dict(flops=33.404e6, act_bytes=1052.8e3, act_bytes_bf16=526.4e3, params=214850),
    # BASELINE configs[4]'s tile shape (NB 40, L 512; head 5120*40 -> 2): SURVEY 8d row 3
    ('resnet18', 40, 512): dict(flops=522.711e6, act_bytes=3248.1e3, act_bytes_bf16=1624.1e3, params=3864386 - 20482 + 2 * 5120 * 40 + 2),
}
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), headline sparsity figure NOT used
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--backbone', default='resnet18', choices=['resnet18', 'densenet18'])
    ap.add_argument('--nb', type=int, default=20, help='sub-batch rows per window (BASELINE configs[4]: 40)')
    ap.add_argument('--seq-len', type=int, default=224, help='samples per row (BASELINE configs[4]: 512)')
    ap.add_argument('--batch', type=int, default=64, help='windows per GPU (BASELINE configs[1]: B=64)')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='STRONG scaling: this many windows per step over ALL ranks (G / world per rank, as nn.DataParallel '
                         'scatters a batch; BASELINE configs[3]: 16 or 64 over 4 GPUs) instead of --batch per rank')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--min-seconds', type=float, default=1.0, help='repeat the K-step timed bracket until this much timed work')
    ap.add_argument('--max-rounds', type=int, default=200)
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary densenet18 measurement')
    ap.add_argument('--storage', default=None, choices=['f32', 'bf16'],
                    help='activation storage under --dtype bf16: bf16 (default: BASELINE configs[2] / [4], bf16 storage with '
                         'fp32 statistics and accumulators) or f32 (round 1: bf16 operands only)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'f32x3p'],
                    help="arithmetic of the k3 s1 convs' forward / data gradient: f32 (the headline, BASELINE configs[1]) or "
                         "bf16 operands with fp32 sums (BASELINE configs[2])")
    return ap.parse_args()


class KernelTimer(object):
    """Wraps C-ABI entry points with HIP events recorded on the launch stream (eager mode only)."""

    REPEAT = 8                              # launches per bracket in the repeated measurement
    REPEATED = ('da_conv3_winograd', 'da_conv3_winograd4', 'da_conv3_bf16')     # x,u,y,rows,L,ldx,C,ldy,N,accumulate,stream
    REPEATED_X3P = ('da_conv3_x3p',)                                                            # x,wpk,y,rows,L,C,ldy,N,accumulate,stream

    def __init__(self, lib, torch, act_bytes=4.0):
        self.lib, self.torch, self.act_bytes = lib, torch, act_bytes
        self.records = {}
        self.rep_records = {}
        self.orig = {}
        self.scratch = None

    def _repeat(self, name, fn, a):
        """The same launch REPEAT times back to back inside ONE event bracket, writing a scratch output (accumulate
        off): an event bracket costs several microseconds of its own (marker packets, cache write-back at the
        timestamp), which a single-launch bracket adds to a 20-50 us kernel and a bracket of 8 does not."""
        import ctypes
        x3p = name in self.REPEATED_X3P
        need = int(a[3]) * int(a[4]) * int(a[6 if x3p else 7])
        if self.scratch is None or self.scratch.numel() < need:
            self.scratch = self.torch.empty(need, device='cuda', dtype=self.torch.float32)
        b = list(a)
        b[2] = ctypes.c_void_p(self.scratch.data_ptr())
        b[8 if x3p else 9] = 0
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(self.REPEAT):
            fn(*b)
        e1.record()
        self.rep_records.setdefault(name, []).append((e0, e1) + self.work_of(name, a))

    def flops_of(self, name, a):
        return self.work_of(name, a)[0]

    def work_of(self, name, a):
        """(algorithmic FLOPs, algorithmic HBM bytes) of one launch of C-ABI entry point `name` with ctypes args `a`.
        GEMM entries: the direct convolution's FLOPs and input + output + weights; bandwidth entries: every tensor
        they read or write, once."""
        f = self.act_bytes                                                    # bytes per activation element (storage)
        if name == 'da_conv_gemm':       # x,w,y,rows,Lm,Lsrc,ldx,C,Ldst,ldy,N,...,ntaps at index 14
            return (2.0 * a[3] * a[4] * a[7] * a[10] * a[14],
                    f * (a[3] * a[5] * a[7] + a[3] * a[4] * a[10] + a[14] * a[7] * a[10]))
        if name == 'da_conv_wgrad':      # dy,x,dw,ws,rows,Lm,Ldy,lddy,N,Lx,ldx,C,...,ntaps at index 15
            return (2.0 * a[4] * a[5] * a[8] * a[11] * a[15],
                    f * (a[4] * a[6] * a[8] + a[4] * a[9] * a[11] + a[15] * a[8] * a[11]))
        if name in ('da_conv3_winograd', 'da_conv3_winograd4', 'da_conv3_bf16'):  # x,u,y,rows,L,ldx,C,ldy,N,accumulate
            pts = {'da_conv3_winograd': 4, 'da_conv3_winograd4': 6, 'da_conv3_bf16': 1.5}[name]
            return (2.0 * a[3] * a[4] * a[6] * a[8] * 3,
                    f * (a[3] * a[4] * a[6] + a[3] * a[4] * a[8] * (2 if a[9] else 1) + pts * a[6] * a[8]))
        if name == 'da_conv3_x3p':          # x,wpk,y,rows,L,C,ldy,N,accumulate: x3 input (6 B / element), fp32 output, 18 B / weight
            return (2.0 * a[3] * a[4] * a[5] * a[7] * 3,
                    6.0 * a[3] * a[4] * a[5] + 4.0 * a[3] * a[4] * a[7] * (2 if a[8] else 1) + 18.0 * a[5] * a[7])
        if name == 'da_conv_x3p_s2_fwd':    # x3,w1pk,wdpk,y1,yd,rows,Lin,C,N: 4 taps (k3 + 1x1) over rows * Lin / 2 outputs
            return (2.0 * a[5] * (a[6] // 2) * a[7] * a[8] * 4,
                    6.0 * a[5] * a[6] * a[7] + 2 * 4.0 * a[5] * (a[6] // 2) * a[8] + 4 * 6.0 * a[7] * a[8])
        if name == 'da_conv_x3p_s2_dgrad':  # dy1_3,w1pk,dyd_3,wdpk,dx,rows,Lout,N,C
            return (2.0 * a[5] * a[6] * a[7] * a[8] * 4,
                    2 * 6.0 * a[5] * a[6] * a[7] + 4.0 * a[5] * 2 * a[6] * a[8] + 4 * 6.0 * a[7] * a[8])
        if name in ('da_conv_wgrad_multi', 'da_conv_wgrad_multi_reduce'):   # jobs (host array of da_wgrad_job), n; x3 operands (code 49) are 6 bytes / element
            # (the chained form: all kinds of one call + the slab reductions they carry inside one event bracket)
            j = a[0]
            return (sum(2.0 * j[i].rows * j[i].Lm * j[i].N * j[i].C * j[i].ntaps for i in range(a[1])),
                    sum((6.0 if j[i].winograd == 49 else f) * (j[i].rows * j[i].Ldy * j[i].N + j[i].rows * j[i].Lx * j[i].C)
                        for i in range(a[1])))
        if name in ('da_conv_gemm_multi', 'da_conv_bf16_multi'):   # jobs (host array of da_conv_job), n
            j = a[0]
            return (sum(2.0 * j[i].rows * j[i].Lm * j[i].N * j[i].C * j[i].ntaps for i in range(a[1])),
                    sum(f * (j[i].rows * j[i].Lsrc * j[i].C + j[i].rows * j[i].Lm * j[i].N + j[i].ntaps * j[i].C * j[i].N)
                        for i in range(a[1])))
        if name in ('da_bn_fwd', 'da_bn_fwd_mask'):       # x,ldx,res,ldr,out,ldo,W,Wn,C,...
            return 0.0, f * a[6] * a[7] * a[8] * (3 if a[2] else 2)
        if name in ('da_bn_bwd', 'da_bn_bwd_add'):        # dout,ldd,x,ldx,out,ldo,dx,lddx,g,ldg,W,Wn,C,mean,invstd,gamma,beta,mode
            t = 3 + (1 if a[17] == 2 else 0) + (1 if a[8] else 0) + (1 if name == 'da_bn_bwd_add' else 0)
            return 0.0, f * a[10] * a[11] * a[12] * t
        if name == 'da_bn_bwd_mask':                      # dout,ldd,x,ldx,dx,lddx,g,ldg,W,Wn,C,...
            return 0.0, f * a[8] * a[9] * a[10] * (3 + (1 if a[6] else 0))
        if name == 'da_bn_bwd_mask2':                     # dout,ldd,dout2,ldd2,x,ldx,dx,lddx,g,ldg,W,Wn,C,...
            return 0.0, f * a[10] * a[11] * a[12] * (4 + (1 if a[8] else 0))
        if name == 'da_bn_bwd_pair2':                     # dout,ldd,dout2,ldd2,descs,W,Wn,C,...: two BatchNorms: (dout, dout2) + 2 x (x, dx)
            return 0.0, f * a[5] * a[6] * a[7] * 6
        if name == 'da_bn_bwd_pair':                      # dout,ldd,descs,W,Wn,C,...
            return 0.0, f * a[3] * a[4] * a[5] * 5
        if name == 'da_bn_fwd_pool':                      # x,ldx,res,ldr,flat,W,Wn,C,L,...: the map is never stored
            return 0.0, f * a[5] * a[6] * a[7] * (2 if a[2] else 1)
        if name == 'da_bn_bwd_pool':                      # dflat,ldd,x,ldx,dx,lddx,g,ldg,W,Wn,C,L,...
            return 0.0, f * a[8] * a[9] * a[10] * (2 + (1 if a[6] else 0))
        if name == 'da_bn_fwd_x':                         # x,ldx,res,ldr,out,ldo,W,Wn,C,...,mask(15),res_x3(16),out_x3(17)
            return 0.0, a[6] * a[7] * a[8] * (4.0 + ((6.0 if a[16] else 4.0) if a[2] else 0.0) + (6.0 if a[17] else 4.0))
        if name == 'da_bn_bwd_x':                         # dout,ldd,x,ldx,dx,lddx,g,ldg,W,Wn,C,...,dx_x3(18)
            return 0.0, a[8] * a[9] * a[10] * (8.0 + (6.0 if a[18] else 4.0) + (4.0 if a[6] else 0.0))
        if name == 'da_pool_bwd':                         # dout,ldd,y,ldy,dz,lddz,rows,R,lin,C,...
            return 0.0, f * a[6] * a[9] * (2 * a[8] + (a[8] - 1) // 2 + 1)
        if name == 'da_bn_relu_pool_fwd':                 # y,ldy,out,ldo,rows,R,lin,C,...
            return 0.0, f * a[4] * a[7] * (a[6] + (a[6] - 1) // 2 + 1)
        if name == 'da_stem_conv_fwd':                    # x,w,y,rows,lin,c0,ldy
            return 2.0 * 7 * a[3] * (a[4] // 2) * a[5], f * a[3] * (a[4] + (a[4] // 2) * a[5])
        if name == 'da_stem_conv_wgrad':                  # dy,lddy,x,dw,ws,rows,lin,c0,...
            return 2.0 * 7 * a[5] * (a[6] // 2) * a[7], f * a[5] * (a[6] + (a[6] // 2) * a[7])
        if name in ('da_clamp_sgd_nesterov',):            # p,g,buf,n
            return 0.0, f * 5 * a[3]
        # ---- the dense block as one design (fp32) ----
        if name == 'da_conv1x1_bn':       # x,ldx,w,y,ldy,rows,R,Lin,C,N,pool,...: reads x once, writes y (half the positions when pooled)
            lo = a[7] // 2 if a[10] else a[7]
            return 2.0 * a[5] * lo * a[8] * a[9], 4.0 * (a[5] * a[7] * a[8] + a[5] * lo * a[9] + a[8] * a[9])
        if name == 'da_conv3_winograd_drop':   # x,u,y,rows,L,ldx,C,ldy,N,...
            return 2.0 * a[3] * a[4] * a[6] * a[8] * 3, 4.0 * (a[3] * a[4] * a[6] + a[3] * a[4] * a[8] + 4 * a[6] * a[8])
        if name == 'da_conv3_winograd_bn':     # x,u,y,rows,L,C,ldy,N,R,...
            return 2.0 * a[3] * a[4] * a[5] * a[7] * 3, 4.0 * (a[3] * a[4] * a[5] + a[3] * a[4] * a[7] + 4 * a[5] * a[7])
        if name == 'da_bn_stats_fused':        # x,ldx,W,Wn,C,...
            return 0.0, 4.0 * a[2] * a[3] * a[4]
        if name == 'da_bn_bwd_ss':             # dout,ldd,x,ldx,out,ldo,dx,lddx,add,ldadd,W,Wn,C,...,relu(18),half(19)
            t = 2.0 + (0.5 if a[19] else 1.0) + (1.0 if a[8] else 0.0) + (1.0 if a[18] == 2 else 0.0)
            return 0.0, 4.0 * a[10] * a[11] * a[12] * t
        return 0.0, 0.0

    def install(self, names):
        for n in names:
            fn = getattr(self.lib, n)
            self.orig[n] = fn

            def wrapped(*a, _fn=fn, _n=n):
                if _n == 'da_conv_wgrad_multi':
                    # one call launches one kernel per job KIND (Winograd / split-bf16 / bf16 forms, direct fp32 form): the
                    # instrumented steps issue the kinds one after the other, each inside its own event bracket
                    jobs, cnt = a[0], a[1]
                    kinds = {}
                    for i in range(cnt):
                        kinds.setdefault('direct' if jobs[i].winograd == 0 else 'code%d' % jobs[i].winograd, []).append(i)
                    rc = 0
                    for kind, idx in kinds.items():
                        sub = (type(jobs[0]) * len(idx))(*[jobs[i] for i in idx])
                        e0 = self.torch.cuda.Event(enable_timing=True)
                        e1 = self.torch.cuda.Event(enable_timing=True)
                        e0.record()
                        rc = rc or _fn(sub, len(idx), *a[2:])
                        e1.record()
                        self.records.setdefault('%s[%s]' % (_n, kind), []).append((e0, e1) + self.work_of(_n, (sub, len(idx))))
                    return rc
                e0 = self.torch.cuda.Event(enable_timing=True)
                e1 = self.torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*a)
                e1.record()
                self.records.setdefault(_n, []).append((e0, e1) + self.work_of(_n, a))
                if _n in self.REPEATED or _n in self.REPEATED_X3P:
                    self._repeat(_n, _fn, a)
                return rc
            setattr(self.lib, n, wrapped)

    def remove(self):
        for n, fn in self.orig.items():
            setattr(self.lib, n, fn)

    def summary(self):
        self.torch.cuda.synchronize()
        out = {}
        for n, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            out[n] = dict(calls=len(recs), total_ms=ms, avg_us=1e3 * ms / len(recs), flops=sum(r[2] for r in recs),
                          bytes=sum(r[3] for r in recs))
        for n, recs in self.rep_records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs) / self.REPEAT
            out[n].update(rep_total_ms=ms, rep_avg_us=1e3 * ms / len(recs))
        return out


def host_cores():
    """CPUs this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    cores = min(cores, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return cores


def cpu_baseline(backbone, batch, seconds):
    """Reference CPU path (stock ATen ops through oracle/torch_ref.py) on the host cores."""
    import torch
    from oracle import torch_ref
    from oracle.weights import seeded_params
    cores = host_cores()
    torch.set_num_threads(cores)
    p = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, 0).items()}
    tr = torch_ref.CpuReferenceTrainer(p, backbone, drop_rate=0.2 if backbone == 'densenet18' else 0.0)
    g = torch.Generator().manual_seed(0)
    b = batch                                            # the same batch as the GPU line (same B, same step)
    x = torch.randn(b, 20, 1, 224, generator=g)
    t = torch.zeros(b, 2)
    t[torch.arange(b), torch.randint(0, 2, (b,), generator=g)] = 1
    say('cpu baseline: %d threads' % cores)
    tr.step(x, t)                                        # warm-up
    say('cpu baseline: warm-up step done')
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(x, t)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= 2:
            break
    return dict(value=round(n * b * 20 / dt, 2), unit='breath-sequences/s', cores=cores, kind='port',
                sample='%d train steps of B=%d windows (20x1x224) in %.1f s, %s, torch %s CPU, %d threads' %
                       (n, b, dt, backbone, torch.__version__, cores))


def csrc_sha16():
    """Identity of the kernel sources (the snapshot on the GPU box has no .git): sha256 over deepards_amd/csrc/*."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'deepards_amd', 'csrc')
    for n in sorted(os.listdir(d)):
        if n.endswith(('.hip', '.h')):
            h.update(n.encode())
            h.update(open(os.path.join(d, n), 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(entry, mode=''):
    """HBM bytes per launch of the dominant kernel (C-ABI entry point `entry`) from the newest committed rocprofv3
    PMC passes (profiles/rNN_traffic.json), or None.  bench.py cannot run the profiler on itself
    (scripts/profile_round.sh re-collects them); the file carries the sha of the kernel sources it was profiled at,
    and the figure is dropped (null) when the sources have changed since or it is about another kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic%s.json' % ('_' + mode if mode else ''))), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        ent = d.get('kernels', {}).get(entry) if 'kernels' in d else (d if d.get('entry') == entry else None)
        if ent is None:
            return None, '%s holds no figure for %s' % (os.path.basename(path), entry)
        if d.get('csrc_sha16') != csrc_sha16():
            return None, '%s was profiled at kernel sources %s, now %s: stale, dropped' % (
                os.path.basename(path), d.get('csrc_sha16'), csrc_sha16())
        return ent['hbm_bytes_per_launch'], '%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, sources %s)' % (
            os.path.basename(path), d.get('csrc_sha16'))
    return None, 'no profiles/r*_traffic%s.json' % ('_' + mode if mode else '')


def rocprof_in_graph_us(entry, dtype, batch):
    """Average duration of the dominant kernel INSIDE the replayed graph from the newest committed rocprofv3 --stats run of
    this same command (profiles/rNN_dominant_kernel*.json, written by scripts/profile_round.sh), or None when that file is
    about another kernel / arithmetic / batch or was taken at other kernel sources (its csrc_sha16 is checked)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_dominant_kernel*.json')), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get('entry') != entry or d.get('csrc_sha16') != csrc_sha16() or d.get('dtype') != dtype or \
                d.get('batch_per_gpu') != batch:
            continue
        return float(d['rocprof_avg_us']), os.path.basename(path)
    return None, None


def np_isfinite(v):
    return v == v and abs(v) != float('inf')


def say(*a):
    print('[bench]', *a, file=sys.stderr, flush=True)


def main():
    args = parse()
    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`' % (args.gpus, args.gpus))
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the product path')
    # DA_BENCH_REHEARSE=gloo: rehearsal of the N > 1 launch on a box with fewer GPUs than ranks (ranks share the
    # cards, gradients travel through gloo's host staging).  Exercises rendezvous, sharding, the barrier-bracketed
    # timing and the max over ranks; its throughput means nothing and the JSON line says so.
    rehearse = os.environ.get('DA_BENCH_REHEARSE', '') == 'gloo'
    if rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from deepards_amd import _lib
    lib = _lib.lib()                                     # fail loudly if the HIP library is missing
    import deepards_amd.models as M
    from deepards_amd.train import HotPathTrainer

    from deepards_amd import functional as F_
    F_.set_conv_dtype(args.dtype)
    fp32like = args.dtype in ('f32', 'f32x3p')            # f32x3p: fp32-equivalent products on the bf16 pipe (opt-in)
    storage = args.storage or ('f32' if fp32like else args.dtype)
    if fp32like and storage != 'f32':
        raise SystemExit('--storage bf16 needs --dtype bf16')
    if args.backbone != 'resnet18':
        storage = 'f32'                                  # no bf16-storage DenseNet (96-channel convs)
    F_.set_storage_dtype(storage)
    torch.manual_seed(0)                                 # same init on every rank (replicas start identical)
    bb = M.resnet18() if args.backbone == 'resnet18' else M.densenet18()
    NB, SL = args.nb, args.seq_len
    c5_shape = (NB, SL) != (20, 224)
    if c5_shape:
        # BASELINE configs[4]'s tile shape.  The reference cannot run it (CNNLinearNetwork refuses seq_len != 224 and
        # never concatenates the 9 metadata inputs, SURVEY finding 8), so the model around the tile is STATED here, not
        # mirrored: the breath block on (B*NB, 1, L) rows with per-window BatchNorm, its features (AvgPool1d(7,1) leaves
        # L/32 - 6 positions per channel) flattened per window like view(-1), one Linear(F*NB, 2) head, no metadata.
        if (args.backbone, NB, SL) not in WORK:
            raise SystemExit('no algorithmic-work row for %s NB=%d L=%d (SURVEY 8d has resnet18 40 x 512)' % (args.backbone, NB, SL))

        model = M.BreathBlockLinear(bb, NB, SL).to(dev)    # deepards_amd/models/torch_cnn_linear_network.py
    else:
        model = M.CNNLinearNetwork(bb, 20, 0).to(dev)
    B = args.batch
    if args.global_batch:                                # strong scaling: the global batch is fixed, a rank takes G / world
        from deepards_amd.train import legal_batch_len
        if args.global_batch % world or legal_batch_len(args.global_batch, args.global_batch, world) != args.global_batch:
            raise SystemExit('--global-batch %d does not split into even shards over %d ranks' % (args.global_batch, world))
        B = args.global_batch // world
    g = torch.Generator().manual_seed(1000 + rank)       # each rank its own shard of the global batch
    x = torch.randn(B, NB, 1, SL, generator=g).to(dev)
    t = torch.zeros(B, 2)
    t[torch.arange(B), torch.randint(0, 2, (B,), generator=g)] = 1
    t = t.to(dev)
    tr = HotPathTrainer(model, optimizer='sgd', world_size=world, rank=rank, use_graph=not args.no_graph)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    say('model built; setup')
    for i in range(2):                                   # setup (not warm-up): eager first step, then the graph capture
        tr.train_step(x, t)
        torch.cuda.synchronize()
    if tr.static_batch() is not None:
        # the batch lives in the buffers the captured step reads (a DeviceTileStore fills them in place in training):
        # no per-step device copy of the inputs
        xs, ts = tr.static_batch()
        xs.copy_(x)
        ts.copy_(t)
        x, t = xs, ts
    say('captured; warmup')
    for i in range(args.warmup):                         # W untimed steps of exactly what is timed (graph replays)
        tr.train_step(x, t)
    def timed_round():
        """EXACTLY args.steps steps between two barrier + synchronize brackets; max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.train_step(x, t)
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt)
        return d

    # The K-step bracket is repeated until the timed work adds up to >= --min-seconds (default 1 s; K = 20 steps are
    # 60 ms, too short for clocks and samplers to settle) and the MEDIAN round is reported.  The round count comes
    # from the first round's max-over-ranks time, so every rank runs the same number.
    rounds = [timed_round()]
    n_rounds = max(1, min(args.max_rounds, int(args.min_seconds / rounds[0] + 0.999)))
    while len(rounds) < n_rounds:
        rounds.append(timed_round())
    srt = sorted(rounds)
    dt = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    loss = float(tr.last_loss)
    say('timed region done: %d rounds of %d steps, median %.3f ms/step (min %.3f, max %.3f)' %
        (len(rounds), args.steps, 1e3 * dt / args.steps, 1e3 * srt[0] / args.steps, 1e3 * srt[-1] / args.steps))

    seqs = world * B * NB * args.steps
    value = seqs / dt
    w = WORK[(args.backbone, NB, SL)] if c5_shape else WORK[args.backbone]
    out = {
        'metric': ('breath-sequences/sec (train step) cnn_linear nb20 seq224' if not c5_shape else
                   'breath-sequences/sec (train step) %s breath block + linear head nb%d seq%d' % (args.backbone, NB, SL)),
        'value': round(value, 1), 'unit': 'breath-sequences/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 4), 'higher_is_better': True,
        'scaling': 'strong' if args.global_batch else 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': (('%s breath block + Linear(F*NB, 2) head (stated, not mirrored: the reference cannot run this shape), '
                                 'synthetic (B=%d per GPU, %d, 1, %d) train step, %s (tile shape of BASELINE configs[4])' %
                                 (args.backbone, B, NB, SL, 'fp32' if args.dtype == 'f32' else 'fp32 via three-term bf16 splits (opt-in)' if args.dtype == 'f32x3p' else 'bf16 MFMA operands / fp32 sums in the residual-block convs, %s activation storage, fp32 statistics' % storage))
                                if c5_shape else
                                ('cnn_linear+%s, synthetic (B=%d per GPU, 20, 1, 224) fp32 train step (BASELINE configs[1])'
                                 if args.dtype == 'f32' else
                                 'cnn_linear+%s, synthetic (B=%d per GPU, 20, 1, 224) fp32 train step (BASELINE configs[1]); arithmetic: fp32 storage '
                                 'of parameters / statistics / conv outputs / gradients, the k3 s1 conv products (forward, data and weight '
                                 'gradient) as exact three-term bf16 splits on the bf16 matrix cores (six MFMA products per multiply, fp32 sums; '
                                 'their operands stored pre-split by the BatchNorm / pool kernels), stride-2 / 1x1 convs on the fp32 matrix cores'
                                 if args.dtype == 'f32x3p' else
                                 'cnn_linear+%s, synthetic (B=%d per GPU, 20, 1, 224) train step, bf16 MFMA operands / fp32 sums in the '
                                 'residual-block convs (forward, data and weight gradient), ' + storage + ' activation storage, fp32 '
                                 'statistics / optimizer (BASELINE configs[2])') % (args.backbone, B)),
                   'backbone': args.backbone, 'batch_per_gpu': B, 'global_batch': B * world,
                   'batch_rule': ('--global-batch %d: fixed global batch, %d windows per rank (strong scaling)' % (args.global_batch, B)
                                  if args.global_batch else '--batch %d per rank (weak scaling: the global batch grows with the ranks)' % B),
                   'n_sub_batches': NB,
                   'seq_len': SL, 'optimizer': 'sgd-nesterov+clamp', 'parallelism': 'dp%d' % world,
                   'hipgraph': not args.no_graph},
        'final_loss': round(loss, 6),
        'timing': {'rounds': len(rounds), 'steps_per_round': args.steps, 'reported': 'median round',
                   'ms_per_step_min': round(1e3 * srt[0] / args.steps, 4), 'ms_per_step_max': round(1e3 * srt[-1] / args.steps, 4),
                   'timed_seconds': round(sum(rounds), 3)},
    }
    if not np_isfinite(loss):
        raise SystemExit('bench: the loss went non-finite (%r) -- the step is broken, no number is reported' % loss)
    if world > 1:
        # the exchange step alone: 20 eager all-reduces of the flat gradient bucket, back to back
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            tr.bucket.allreduce(tr.group)
        torch.cuda.synchronize()
        ar = (time.perf_counter() - t0) / 20
        tt = torch.tensor([ar], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out['allreduce_ms'] = round(1e3 * float(tt), 4)
        out['allreduce_bytes'] = int(tr.bucket.numel * 4)
        # round 3: under RCCL the all-reduce is a node of the captured step graph (no host round trip, no extra launches); it still
        # runs AFTER the backward it reduces (not overlapped).  The two-graph form (eager all-reduce between two replays) is the fallback.
        out['allreduce_in_graph'] = bool(getattr(tr, 'allreduce_in_graph', False))
        out['allreduce_exposed'] = True
    if rehearse:
        out['rehearsal'] = 'gloo: %d ranks sharing %d GPU(s); throughput is NOT a measurement' % (world, torch.cuda.device_count())
    step_flops = w['flops'] * B * NB
    step_bytes = (w['act_bytes'] * B * NB) + 8 * 4 * w['params']
    per_gpu_dt = dt / args.steps
    if fp32like:
        out['step_roofline'] = {
            'alg_tflops': round(step_flops / per_gpu_dt / 1e12, 2), 'peak_tflops_fp32_mfma': PEAK_FP32_MFMA_TFLOPS,
            'frac_compute': round(step_flops / per_gpu_dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            'alg_gbs': round(step_bytes / per_gpu_dt / 1e9, 1), 'peak_gbs': PEAK_HBM_GBS,
            'frac_hbm': round(step_bytes / per_gpu_dt / 1e9 / PEAK_HBM_GBS, 4), 'binding': 'compute(fp32 mfma)'}
    else:
        # bf16 arithmetic makes the step memory-bound: price it against HBM with the ALGORITHMIC bytes of
        # a bf16-storage step (what the path should move), next to the bytes the current storage really implies
        bf16_bytes = (w['act_bytes_bf16'] * B * NB) + 8 * 4 * w['params']
        out['step_roofline'] = {
            'bound': 'hbm', 'alg_bytes_bf16_storage': int(bf16_bytes),
            'alg_gbs': round(bf16_bytes / per_gpu_dt / 1e9, 1), 'peak_gbs': PEAK_HBM_GBS,
            'frac_hbm': round(bf16_bytes / per_gpu_dt / 1e9 / PEAK_HBM_GBS, 4),
            'storage_now': F_.storage_dtype(),
            'alg_bytes_at_current_storage': int(step_bytes if F_.storage_dtype() == 'f32' else bf16_bytes),
            'alg_tflops': round(step_flops / per_gpu_dt / 1e12, 2), 'peak_tflops_bf16_mfma': PEAK_BF16_MFMA_TFLOPS,
            'frac_compute': round(step_flops / per_gpu_dt / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4), 'binding': 'hbm'}

    if rank == 0 and not args.no_roofline:
        # instrumented EAGER steps: HIP events around every C-ABI launch, on the launch stream
        kt = KernelTimer(lib, torch, act_bytes=2.0 if F_.storage_dtype() == 'bf16' else 4.0)
        names = [n for n in _lib.SIGNATURES if n not in ('da_version', 'da_conv_wgrad_workspace', 'da_stem_wgrad_workspace', 'da_bn_workspace', 'da_debug_set', 'da_bn_chunks', 'da_conv_wgrad_splits', 'da_conv_wgrad_plan', 'da_bn_debug_two_stage', 'da_bn_debug_target_blocks', 'da_abi_sizes', 'da_wino_debug_tail', 'da_wino_debug_pchunk', 'da_wino_debug_tapmod', 'da_stat_records_floats', 'da_head_groups', 'da_stem_bwd_partials', 'da_stem_bwd_workspace', 'da_wino_weights', 'da_wino4_weights',
                                                          'da_hip_runtime_symbol', 'da_stem_wgrad_workspace_g', 'da_set_act_dtype', 'da_get_act_dtype',
                                                          'da_sizeof_wgrad_reduce_desc', 'da_sizeof_bn_running_desc', 'da_sizeof_bn_pgrad_desc', 'da_bn_mask_words',
                                                          'da_bn_pool_ok', 'da_bn_two_ok')]      # (host-only queries)
        tr_e = HotPathTrainer(model, optimizer='sgd', use_graph=False)
        tr_e.bucket, tr_e.state = tr.bucket, tr.state
        try:
            tr_e._eager_step(x, t)                       # untimed
            kt.install(names)
            nprof = 3
            for _ in range(nprof):
                tr_e._eager_step(x, t)
            summ = kt.summary()
        finally:
            kt.remove()
        say('roofline pass done')
        # Which kernel each single-kernel entry point launches (rocprofv3 names).  da_conv_wgrad_multi launches TWO
        # kernels per call (wino_wgrad_multi_kernel + conv_wgrad_multi_kernel<>), so HIP events around it time a pair:
        # it is listed in kernel_time_share but cannot be "the dominant kernel" (by rocprof each half is smaller than the
        # forward / data-gradient kernels, profiles/*kernel_stats.csv).
        KERNEL_OF = {
            'da_conv3_winograd': 'conv3_wino_kernel (k3 s1 conv forward + data gradient, Winograd F(2,3) on v_mfma_f32_16x16x4_f32; '
                                 'algorithmic = direct-conv FLOPs, 2/3 of them executed)',
            'da_conv3_winograd4': 'conv3_wino4k_kernel (k3 s1 conv forward + data gradient of the 512-channel stage, Winograd '
                                  'F(4,3); algorithmic = direct-conv FLOPs, 1/2 of them executed)',
            'da_conv_gemm': 'conv_gemm_tailed_kernel<*> / conv_gemm_kernel<*> (conv fwd + dgrad implicit GEMM, v_mfma_f32_32x32x2_f32)',
            'da_conv_gemm_multi': 'conv_gemm_multi_kernel (stride-2 block heads + 1x1 downsamples, fwd + dgrad, v_mfma_f32_32x32x2_f32)',
            'da_conv3_bf16': 'conv3_bf16_kernel (k3 s1 conv forward + data gradient, v_mfma_f32_32x32x16_bf16)',
            'da_conv3_x3p': 'conv3_x3p_dma_kernel (k3 s1 conv forward + data gradient on pre-split (x3) operands: fp32 products as six '
                            'v_mfma_f32_32x32x16_bf16 of exact three-term splits, operands by LDS-DMA, no VALU in the K loop; peak = the bf16 MFMA peak / 6)',
            'da_conv_x3p_s2_fwd': 'conv_x3p_s2_kernel<false> (stride-2 block entry: k3 s2 conv + 1x1 s2 downsample in one launch, x3 operands)',
            'da_conv_x3p_s2_dgrad': 'conv_x3p_s2_kernel<true> (their summed data gradient, x3 operands)',
            'da_conv_bf16_multi': 'conv_bf16_gen_kernel<*> (stride-2 / 1x1 convs, bf16 operands)',
            'da_bn_fwd': 'bn_fwd_fused_kernel<*> (per-window BatchNorm (+ReLU)(+residual) forward, single pass)',
            'da_bn_fwd_mask': 'bn_fwd_fused_kernel<*> (block-output BatchNorm + residual + ReLU forward, ReLU bit mask)',
            'da_bn_bwd': 'bn_bwd_fused_kernel<*> (per-window BatchNorm backward, single pass)',
            'da_bn_bwd_mask': 'bn_bwd_fused_kernel<*> (block-output BatchNorm backward from the ReLU bit mask)',
            'da_bn_bwd_add': 'bn_bwd_fused_kernel<*> (BatchNorm backward + concat pass-through)',
            'da_pool_bwd': 'pool_bwd_kernel (stem max/avg pool + ReLU backward)',
            'da_conv_wgrad_multi[code1]': 'wino_wgrad_multi_kernel (k3 s1 weight gradients, Winograd F(2,3) form on v_mfma_f32_32x32x2_f32)',
            'da_conv_wgrad_multi[code6]': 'wino4_wgrad_multi_kernel (512-channel k3 s1 weight gradients, Winograd F(4,3) form: 6 contractions '
                                          'over output quads, 1/2 of the direct FLOPs executed)',
            'da_conv_wgrad_multi[code49]': 'wgrad_x3p_multi_kernel (k3 s1 weight gradients on pre-split (x3) operands, six v_mfma_f32_32x32x16_bf16 '
                                           'products per multiply)',
            'da_conv_wgrad_multi[code16]': 'wgrad_bf16_multi_kernel<*, 1> (weight gradients, bf16 operands)',
            'da_conv_wgrad_multi[direct]': 'conv_wgrad_multi_kernel<*> (stride-2 / 1x1 / small-channel weight gradients, v_mfma_f32_32x32x2_f32)',
            'da_bn_fwd_x': 'bn_fwd_fused_kernel<float, *, x3> (BatchNorm forward storing / reading the x3 format)',
            'da_bn_bwd_x': 'bn_bwd_fused_kernel<float, *, x3> (BatchNorm backward storing dx in the x3 format)',
        }
        PEAK = {'da_conv3_bf16': PEAK_BF16_MFMA_TFLOPS, 'da_conv_bf16_multi': PEAK_BF16_MFMA_TFLOPS,
                'da_conv3_x3p': round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1),
                'da_conv_x3p_s2_fwd': round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1), 'da_conv_x3p_s2_dgrad': round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1),
                'da_conv_wgrad_multi[code49]': round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1),
                'da_conv_wgrad_multi[code16]': PEAK_BF16_MFMA_TFLOPS}
        cands = [k for k in summ if k in KERNEL_OF and (summ[k]['flops'] or summ[k]['bytes'])]
        dname = max(cands, key=lambda k: summ[k].get('rep_total_ms', summ[k]['total_ms']))   # argmax over all of them
        dom = summ[dname]
        single_us = dom['avg_us']
        if 'rep_total_ms' in dom:                 # per-launch time from the 8-launch brackets (see KernelTimer._repeat)
            dom = dict(dom, total_ms=dom['rep_total_ms'], avg_us=dom['rep_avg_us'])
        if dom['flops'] and fp32like:
            ach = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
            peak, bound, unit = PEAK.get(dname, PEAK_FP32_MFMA_TFLOPS), 'mfma', 'TFLOP/s'
        else:                                     # bf16 arithmetic: every kernel of the step is priced against HBM
            ach = dom['bytes'] / (dom['total_ms'] * 1e-3) / 1e9
            peak, bound, unit = PEAK_HBM_GBS, 'hbm', 'GB/s'
        if c5_shape or B != 64:
            traffic, traffic_note = None, 'the PMC passes under profiles/ ran the default workload (B=64, nb20, seq224) only'
        else:
            traffic, traffic_note = pmc_traffic(dname, ('bf16' if F_.storage_dtype() == 'bf16' else 'bf16_f32storage') if args.dtype == 'bf16' else
                                                (args.dtype if args.dtype == 'f32x3p' else ''))
        out['roofline'] = {'bound': bound, 'kernel': KERNEL_OF[dname], 'entry': dname,
                           'achieved': round(ach, 2), 'peak': peak, 'unit': unit,
                           'frac': round(ach / peak, 4), 'traffic': traffic, 'traffic_source': traffic_note,
                           'launches_per_step': dom['calls'] // nprof, 'avg_launch_us': round(dom['avg_us'], 2),
                           'avg_launch_us_single_bracket': round(single_us, 2),
                           'alg_flops_per_launch': round(dom['flops'] / dom['calls'], 1),
                           'alg_bytes_per_launch': round(dom['bytes'] / dom['calls'], 1)}
        # provenance: the event brackets above time eager, warm-cache re-launches; the timed region replays a graph.  When a
        # rocprofv3 --stats run of this very command at these very kernel sources is committed, its in-graph average is
        # printed beside and the LOWER fraction is the one reported as `frac`.
        rp_us, rp_src = rocprof_in_graph_us(dname, args.dtype, B) if not c5_shape else (None, None)
        out['roofline']['frac_hip_events'] = out['roofline']['frac']
        if rp_us:
            rp_ach = (dom['flops'] / dom['calls'] / (rp_us * 1e-6) / 1e12) if unit == 'TFLOP/s' else \
                (dom['bytes'] / dom['calls'] / (rp_us * 1e-6) / 1e9)
            out['roofline']['avg_launch_us_rocprof_in_graph'] = rp_us
            out['roofline']['frac_rocprof_in_graph'] = round(rp_ach / peak, 4)
            out['roofline']['rocprof_source'] = rp_src
            if rp_ach / peak < out['roofline']['frac']:
                out['roofline']['frac'] = round(rp_ach / peak, 4)
                out['roofline']['achieved'] = round(rp_ach, 2)
        else:
            out['roofline']['frac_rocprof_in_graph'] = None
        executed = {'da_conv3_winograd': 2.0 / 3.0, 'da_conv3_winograd4': 0.5}.get(dname)
        if executed and bound == 'mfma':          # Winograd: `achieved` counts the direct convolution's FLOPs (the contract's
            # ALGORITHMIC work), the matrix pipe executes fewer -- the fraction of the pipe's peak it really runs at:
            out['roofline']['executed_flops_per_alg_flop'] = round(executed, 4)
            out['roofline']['frac_of_peak_executed'] = round(ach * executed / peak, 4)
        out['kernel_roofline'] = {}
        for k in cands:                           # every single-kernel entry against ITS roofline (eager, single brackets)
            v = summ[k]
            ms = v.get('rep_total_ms', v['total_ms'])
            if v['flops'] and fp32like:
                out['kernel_roofline'][k] = {'tflops': round(v['flops'] / ms / 1e9, 1),
                                             'frac': round(v['flops'] / ms / 1e9 / PEAK.get(k, PEAK_FP32_MFMA_TFLOPS), 3)}
            else:
                out['kernel_roofline'][k] = {'gbs': round(v['bytes'] / ms / 1e6, 1), 'frac': round(v['bytes'] / ms / 1e6 / PEAK_HBM_GBS, 3)}
        tot = sum(v['total_ms'] for v in summ.values())
        out['kernel_time_share'] = {k: round(v['total_ms'] / tot, 4) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['total_ms'])}
        wgk = [v for k, v in summ.items() if k.startswith('da_conv_wgrad_multi[')]
        wg = (dict(flops=sum(v['flops'] for v in wgk), total_ms=sum(v['total_ms'] for v in wgk)) if wgk else None) or \
            summ.get('da_conv_wgrad_multi_reduce') or summ.get('da_conv_wgrad')
        if wg and wg['flops']:
            out['wgrad_tflops'] = round(wg['flops'] / (wg['total_ms'] * 1e-3) / 1e12, 2)
        out['eager_kernel_ms_per_step'] = round(tot / nprof, 3)

    if world == 1 and not args.no_extra and not c5_shape:
        # forward-only (run_test_epoch step: no_grad train-mode forward + loss + argmax), same batch, graph replayed
        for _ in range(3):
            tr.test_step(x, t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tr.test_step(x, t)
        torch.cuda.synchronize()
        d1 = (time.perf_counter() - t1) / args.steps
        out.setdefault('extra', {})['inference'] = {'value': round(B * NB / d1, 1), 'unit': 'breath-sequences/s',
                                                     'ms_per_step': round(1e3 * d1, 4),
                                                     'note': 'forward-only test step of %s, B=%d' % (args.backbone, B)}
        say('inference extra done')

    if world == 1 and not args.no_extra and args.backbone == 'resnet18' and args.dtype == 'f32' and not c5_shape:
        # (fp32 only: densenet's 64+32k channel counts have no bf16 kernels)
        # secondary figure: the reference's DEFAULT backbone (defaults.yml:18), same step definition, dropout active
        torch.manual_seed(0)
        m2 = M.CNNLinearNetwork(M.densenet18(), 20, 0).to(dev)
        tr2 = HotPathTrainer(m2, optimizer='sgd', use_graph=not args.no_graph)
        for _ in range(3):
            tr2.train_step(x, t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tr2.train_step(x, t)
        torch.cuda.synchronize()
        d2 = (time.perf_counter() - t1) / args.steps
        w2 = WORK['densenet18']
        out.setdefault('extra', {})['densenet18'] = {
            'value': round(B * 20 / d2, 1), 'ms_per_step': round(1e3 * d2, 4),
            'alg_tflops': round(w2['flops'] * B * 20 / d2 / 1e12, 2),
            'alg_gbs': round((w2['act_bytes'] * B * 20 + 32 * w2['params']) / d2 / 1e9, 1),
            'note': 'cnn_linear+densenet18 (reference default backbone), drop_rate 0.2 active'}
        if rank == 0 and not args.no_roofline:
            # where the DenseNet step goes: the same instrumented eager pass as for the headline (HIP events around every
            # C-ABI launch), time share per entry point and each against its own roofline
            kt2 = KernelTimer(lib, torch)
            tr2e = HotPathTrainer(m2, optimizer='sgd', use_graph=False)
            tr2e.bucket, tr2e.state = tr2.bucket, tr2.state
            try:
                tr2e._eager_step(x, t)
                kt2.install(names)       # every kernel-launching entry point (round 3 counted by name prefixes and missed the
                                         # recomputing stem's three calls and the two global pools: its 109 is 114 on this count)
                for _ in range(3):
                    tr2e._eager_step(x, t)
                s2 = kt2.summary()
            finally:
                kt2.remove()
            tot2 = sum(v['total_ms'] for v in s2.values())
            dn = out['extra']['densenet18']
            dn['launches_per_step'] = sum(v['calls'] for v in s2.values()) // 3
            dn['kernel_time_share'] = {k: round(v['total_ms'] / tot2, 4) for k, v in sorted(s2.items(), key=lambda kv: -kv[1]['total_ms'])[:12]}
            dn['kernel_roofline'] = {}
            for k, v in s2.items():
                if v['flops']:
                    dn['kernel_roofline'][k] = {'tflops': round(v['flops'] / v['total_ms'] / 1e9, 1),
                                                'frac_mfma': round(v['flops'] / v['total_ms'] / 1e9 / PEAK_FP32_MFMA_TFLOPS, 3),
                                                'frac_hbm': round(v['bytes'] / v['total_ms'] / 1e6 / PEAK_HBM_GBS, 3)}
                elif v['bytes']:
                    dn['kernel_roofline'][k] = {'gbs': round(v['bytes'] / v['total_ms'] / 1e6, 1),
                                                'frac_hbm': round(v['bytes'] / v['total_ms'] / 1e6 / PEAK_HBM_GBS, 3)}
            dn['eager_kernel_ms_per_step'] = round(tot2 / 3, 3)
        say('densenet18 extra done')

    if world == 1 and not args.no_extra and args.backbone == 'resnet18' and args.dtype == 'f32' and not c5_shape:
        # BASELINE configs[2] ("resnet18-1D ... bf16"): same step with bf16 operands in the k3 s1 conv forward / data gradient
        F_.set_conv_dtype('bf16')
        F_.set_storage_dtype('bf16')
        try:
            torch.manual_seed(0)
            m3 = M.CNNLinearNetwork(M.resnet18(), 20, 0).to(dev)
            tr3 = HotPathTrainer(m3, optimizer='sgd', use_graph=not args.no_graph)
            for _ in range(3):
                tr3.train_step(x, t)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                l3 = tr3.train_step(x, t)
            torch.cuda.synchronize()
            d3 = (time.perf_counter() - t1) / args.steps
            out.setdefault('extra', {})['resnet18_bf16'] = {
                'value': round(B * 20 / d3, 1), 'ms_per_step': round(1e3 * d3, 4), 'dtype': 'bf16',
                'final_loss': round(float(l3), 6),
                'note': 'cnn_linear+resnet18 (BASELINE configs[2]): every residual-block conv (forward, data and weight gradient) on '
                        'v_mfma_f32_32x32x16_bf16 with fp32 sums, activations and activation gradients STORED in bf16, '
                        'statistics / parameters / optimizer fp32'}
        finally:
            F_.set_conv_dtype('f32')
        say('bf16 extra done')

    if world == 1 and not args.no_extra and args.backbone == 'resnet18' and args.dtype == 'f32' and not c5_shape:
        # the same fp32 step with the residual-block convs on the bf16 matrix cores through exact three-term splits
        # (--dtype f32x3p / --conv-dtype f32x3p: fp32 results, opt-in): the headline stays on the native fp32 kernels
        F_.set_conv_dtype('f32x3p')
        try:
          try:
              torch.manual_seed(0)
              m4 = M.CNNLinearNetwork(M.resnet18(), 20, 0).to(dev)
              tr4 = HotPathTrainer(m4, optimizer='sgd', use_graph=not args.no_graph)
              for _ in range(10):
                  tr4.train_step(x, t)
              torch.cuda.synchronize()
              t1 = time.perf_counter()
              for _ in range(max(args.steps, 50)):
                  l4 = tr4.train_step(x, t)
              torch.cuda.synchronize()
              d4 = (time.perf_counter() - t1) / max(args.steps, 50)
              out.setdefault('extra', {})['resnet18_f32x3p'] = {
                  'value': round(B * 20 / d4, 1), 'ms_per_step': round(1e3 * d4, 4), 'dtype': 'f32 (six bf16 MFMA products of exact three-term splits per fp32 product)',
                  'final_loss': round(float(l4), 6),
                  'note': 'cnn_linear+resnet18, same workload as the headline under conv arithmetic f32x3p: activations stored pre-split '
                          '[h|m|l] by the BatchNorm / pool kernels, every residual-block conv (k3 s1, the stride-2 block entries, their '
                          'data and weight gradients) on v_mfma_f32_32x32x16_bf16; opt-in (DESIGN_APPENDIX.md 7c: the 2.7 ms bar for making it '
                          'the default was not met)'}
          except Exception as e:                     # an opt-in extra never takes the headline line down with it
            out.setdefault('extra', {})['resnet18_f32x3p'] = {'error': '%s: %s' % (type(e).__name__, e)}
        finally:
            F_.set_conv_dtype('f32')
        say('f32x3p extra done')

    if world == 1 and not args.no_extra and args.backbone == 'resnet18' and args.dtype == 'f32' and not c5_shape and not args.no_graph:
        # two k-fold replicas side by side on this GPU (train_ards_detector --folds-in-flight 2): each its own model,
        # optimizer, captured step and stream, replayed round-robin; aggregate rate of the pair
        from deepards_amd.train import concurrent_streams
        reps = []
        fold_streams = concurrent_streams(2)        # two streams on different hardware queues (measured with a spin kernel)
        for r in range(2):
            torch.manual_seed(100 + r)
            mr = M.CNNLinearNetwork(M.resnet18(), 20, 0).to(dev)
            trr = HotPathTrainer(mr, optimizer='sgd', use_graph=True)
            sr = fold_streams[r]
            with torch.cuda.stream(sr):
                for _ in range(3):
                    trr.train_step(x, t)
            torch.cuda.synchronize()
            reps.append((trr, sr, trr.static_batch()))
        from deepards_amd.train import place_replicas_on_streams
        placed = place_replicas_on_streams([r_[0] for r_ in reps])     # measured stream placement (state restored afterwards)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            for (trr, _, st), sr in zip(reps, placed):
                with torch.cuda.stream(sr):
                    trr.train_step(st[0], st[1])
        torch.cuda.synchronize()
        d4 = (time.perf_counter() - t1) / args.steps
        out.setdefault('extra', {})['two_folds_in_flight'] = {
            'value': round(2 * B * 20 / d4, 1), 'ms_per_round': round(1e3 * d4, 4), 'vs_one_after_the_other': round(2 * B * 20 / d4 / value, 3),
            'note': 'two independent cnn_linear+resnet18 training replicas (two k-folds) of B=%d each on one GPU, own streams; '
                    'aggregate breath-sequences/s of the pair -- NOT the headline workload (one model)' % B}
        for trr, _, _ in reps:
            trr.release_graphs()
        say('two-fold extra done')
        # the reference's DEFAULT protocol (defaults.yml: densenet18, batch_size 16, 5 folds): five fold replicas of B=16
        # one after the other vs in flight together
        x16, t16 = x[:16].contiguous(), t[:16].contiguous()
        reps = []
        for r in range(5):
            torch.manual_seed(200 + r)
            mr = M.CNNLinearNetwork(M.densenet18(), 20, 0).to(dev)
            trr = HotPathTrainer(mr, optimizer='sgd', use_graph=True)
            for _ in range(3):
                trr.train_step(x16, t16)
            torch.cuda.synchronize()
            reps.append((trr, trr.static_batch()))

        def five(streams, n):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                for (trr, st), sr in zip(reps, streams):
                    with torch.cuda.stream(sr):
                        trr.train_step(st[0], st[1])
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n
        one = torch.cuda.Stream()
        seq5 = five([one] * 5, args.steps)
        par5 = five(place_replicas_on_streams([r_[0] for r_ in reps]), args.steps)
        out['extra']['five_folds_in_flight_reference_defaults'] = {
            'value': round(5 * 16 * 20 / par5, 1), 'one_after_the_other': round(5 * 16 * 20 / seq5, 1),
            'ms_per_round': round(1e3 * par5, 4), 'vs_one_after_the_other': round(seq5 / par5, 3),
            'note': 'five independent cnn_linear+densenet18 replicas (the 5 folds of defaults.yml) of batch_size 16 each on one GPU: '
                    'aggregate breath-sequences/s with the folds in flight together (own streams) and one after the other -- NOT '
                    'the headline workload'}
        for trr, _ in reps:
            trr.release_graphs()
        say('five-fold extra done')

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not c5_shape:      # the reference's CPU path cannot run the C5 shape
        say('cpu baseline ...')
        out['cpu_baseline'] = cpu_baseline(args.backbone, B, args.cpu_seconds)
        out['speedup_vs_cpu_baseline'] = round(value / out['cpu_baseline']['value'], 1)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
