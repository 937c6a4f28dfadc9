"""Debug: per-block gradient comparison HIP vs torch CPU fp64 (resnet18, golden b2 input)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from oracle.weights import seeded_params, seeded_batch
from oracle import torch_ref as T
import deepards_amd.models as M
from deepards_amd.functional import bce_with_logits

x, t = seeded_batch(2, 20, 0)
p64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in seeded_params('resnet18', 0).items()}
# reference with intermediates (all windows batched is NOT the same BN; loop per window and stack)
def ref_blocks(xw):
    hs = []
    h = T._stem(p64, xw, 'breath_block.conv1.weight', 'breath_block.bn1'); h.retain_grad(); hs.append(h)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512)):
        for bi in range(2):
            s = 2 if (li > 0 and bi == 0) else 1
            bp = 'breath_block.layer%d.%d.' % (li + 1, bi)
            o = F.relu(T._bn(p64, bp + 'bn1', F.conv1d(h, p64[bp + 'conv1.weight'], None, s, 1)))
            o = T._bn(p64, bp + 'bn2', F.conv1d(o, p64[bp + 'conv2.weight'], None, 1, 1))
            r = h
            if s != 1 or inpl != planes:
                r = T._bn(p64, bp + 'downsample.1', F.conv1d(h, p64[bp + 'downsample.0.weight'], None, s, 0))
            h = F.relu(o + r); h.retain_grad(); hs.append(h)
            inpl = planes
    return F.avg_pool1d(h, 7, 1).flatten(1), hs
xt = torch.from_numpy(x).double()
outs, allhs = [], []
for i in range(2):
    f, hs = ref_blocks(xt[i]); allhs.append(hs)
    outs.append(F.linear(f.reshape(-1), p64['linear_final.weight'], p64['linear_final.bias']).unsqueeze(0))
loss = torch.nn.BCEWithLogitsLoss()(torch.cat(outs), torch.from_numpy(t).double())
loss.backward()

model = M.CNNLinearNetwork(M.resnet18(), 20, 0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params('resnet18', 0).items()}, strict=False)
model = model.cuda().train()
bb = model.breath_block
caps = []
import deepards_amd.functional as F_
xc = torch.from_numpy(x).cuda().reshape(40, 1, 224)
h = F_.StemFunction.apply(xc.view(40, 224), bb.conv1.weight, bb.bn1.weight, bb.bn1.bias, 20, 0, F_.BNState(bb.bn1))
h.retain_grad(); caps.append(h)
for layer in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
    for blk in layer:
        h = blk.forward_rlc(h, 20); h.retain_grad(); caps.append(h)
feat = F_.GlobalAvgPoolFunction.apply(h)
logits = F_.Linear2Function.apply(feat.view(2, -1), model.linear_final.weight, model.linear_final.bias)
fns = [c.grad_fn for c in caps]
REC = []
_orig_bwd = F_.BasicBlockFunction.backward
from deepards_amd import hip_ops as H
TRACE = {'on': False, 'log': []}
def _wrap(name):
    orig = getattr(H, name)
    def w(*a, **k):
        if TRACE['on']:
            ins = [round(float(t.double().abs().sum()), 9) if torch.is_tensor(t) else t for t in a]
            kin = {kk: (round(float(v.double().abs().sum()), 9) if torch.is_tensor(v) else v) for kk, v in k.items()}
        r = orig(*a, **k)
        if TRACE['on']:
            outs = [round(float(t.double().abs().sum()), 9) for t in (r if isinstance(r, tuple) else (r,)) if torch.is_tensor(t)]
            TRACE['log'].append((name, ins, kin, outs))
        return r
    setattr(H, name, w)
for nm_ in ('bn_bwd', 'conv_wgrad', 'conv_dgrad', 'repack_weight'):
    _wrap(nm_)
NCALL = [0]
def _rec_bwd(ctx, dout):
    NCALL[0] += 1
    TRACE['on'] = NCALL[0] == 4
    res = _orig_bwd(ctx, dout)
    TRACE['on'] = False
    REC.append((dout.clone(), res[0].clone(), dout.is_contiguous(), tuple(dout.stride()), tuple(dout.shape)))
    return res
F_.BasicBlockFunction.backward = staticmethod(_rec_bwd)
from deepards_amd import hip_ops as H
blks = [None] + [b for l in (bb.layer1, bb.layer2, bb.layer3, bb.layer4) for b in l]
for n_, f, blk in zip(['stem'] + ['layer%d.%d' % (l, b) for l in (1, 2, 3, 4) for b in (0, 1)], fns, blks):
    if blk is None: continue
    sv = f.saved_tensors
    x_, w1_, g1_, b1_, w2_, g2_, b2_, y1_, m1_, i1_, h1_, y2_, m2_, i2_, out_ = sv[:15]
    st_ = blk.stride
    def chk(tag, a_, b_):
        if not torch.equal(a_, b_):
            dd = (a_ != b_).nonzero()
            print('MISMATCH', n_, tag, tuple(a_.shape), 'n', len(dd), 'first', dd[:2].tolist(), 'last', dd[-2:].tolist(), 'maxdiff', float((a_-b_).abs().max()))
    chk('y1', y1_, H.conv_fwd(x_, H.repack_weight(w1_)[0], st_, 1))
    mm, ii = H.bn_stats(y1_, 20); chk('m1', m1_, mm); chk('i1', i1_, ii)
    chk('h1', h1_, H.bn_apply(y1_, 20, m1_, i1_, g1_, b1_, relu=True))
    chk('y2', y2_, H.conv_fwd(h1_, H.repack_weight(w2_)[0], 1, 1))
    mm, ii = H.bn_stats(y2_, 20); chk('m2', m2_, mm); chk('i2', i2_, ii)
    if len(sv) > 15:
        wd_, gd_, bd_, yd_, md_, id_ = sv[15:]
        chk('yd', yd_, H.conv_fwd(x_, H.repack_weight(wd_)[0], st_, 0))
        mm, ii = H.bn_stats(yd_, 20); chk('md', md_, mm); chk('id', id_, ii)
        res_ = H.bn_apply(yd_, 20, md_, id_, gd_, bd_, relu=False)
    else:
        res_ = x_
    chk('out', out_, H.bn_apply(y2_, 20, m2_, i2_, g2_, b2_, relu=True, res=res_))
snap = [[None if s_ is None else s_.clone() for s_ in f.saved_tensors] for f in fns]
bce_with_logits(logits, torch.from_numpy(t).cuda()).backward(retain_graph=True)
torch.cuda.synchronize()
for n_, f, sn in zip(['stem'] + ['layer%d.%d' % (l, b) for l in (1, 2, 3, 4) for b in (0, 1)], fns, snap):
    for i_, (a_, b_) in enumerate(zip(f.saved_tensors, sn)):
        if a_ is not None and not torch.equal(a_, b_):
            dd = (a_ != b_).nonzero()
            print('CORRUPT', n_, 'saved', i_, tuple(a_.shape), 'n', len(dd), 'first', dd[:3].tolist(), 'last', dd[-3:].tolist())
names = ['stem'] + ['layer%d.%d' % (l, b) for l in (1, 2, 3, 4) for b in (0, 1)]
for k, n in enumerate(names):
    ref_act = torch.cat([allhs[i][k] for i in range(2)]).detach().numpy()        # (40,C,L)
    ref_grad = torch.cat([allhs[i][k].grad for i in range(2)]).numpy()
    act = caps[k].detach().cpu().numpy().transpose(0, 2, 1)
    grad = caps[k].grad.cpu().numpy().transpose(0, 2, 1)
    ea, eg = np.abs(act - ref_act).max(), np.abs(grad - ref_grad).max()
    bad = np.argwhere(np.abs(grad - ref_grad) > 1e-3 * np.abs(ref_grad).max())
    print('%-10s act err %.2e (max %.2e)  grad-out err %.2e (max %.2e) nbad %d %s' % (n, ea, np.abs(ref_act).max(), eg, np.abs(ref_grad).max(), len(bad), bad[:6].tolist()))
k = names.index('layer2.1')
ref_grad = torch.cat([allhs[i][k].grad for i in range(2)]).numpy()
grad = caps[k].grad.cpu().numpy().transpose(0, 2, 1)
d = np.abs(grad - ref_grad)
bad = d > 1e-3 * np.abs(ref_grad).max()
print('bad rows', np.unique(np.argwhere(bad)[:, 0]).tolist())
print('bad chans', np.unique(np.argwhere(bad)[:, 1]).tolist()[:64], '...')
print('bad pos', np.unique(np.argwhere(bad)[:, 2]).tolist())
for r in np.unique(np.argwhere(bad)[:, 0])[:3]:
    bb_ = bad[r]
    print('row', r, 'nbad', bb_.sum(), 'per-pos', bb_.sum(0).tolist())
    print('  sample got/ref', grad[r, 0, :8], ref_grad[r, 0, :8])
# second run: deterministic?
print('n recorded', len(REC))
order = ['layer4.1', 'layer4.0', 'layer3.1', 'layer3.0', 'layer2.1', 'layer2.0', 'layer1.1', 'layer1.0']
for (dr, dxr, cont, strd, shp), nm in zip(REC, order):
    k_ = names.index(nm)
    ref_do = torch.cat([allhs[i][k_].grad for i in range(2)]).numpy()
    ref_dx = torch.cat([allhs[i][k_ - 1].grad for i in range(2)]).numpy()
    e1 = np.abs(dr.cpu().numpy().transpose(0, 2, 1) - ref_do).max()
    e2 = np.abs(dxr.cpu().numpy().transpose(0, 2, 1) - ref_dx).max()
    e3 = np.abs(caps[k_ - 1].grad.cpu().numpy() - dxr.cpu().numpy()).max()
    print('%-9s dout-received err %.2e  dx-returned err %.2e  |retained(in)-dx| %.2e contig %s stride %s shape %s' % (nm, e1, e2, e3, cont, strd, shp))
print('---- isolate layer3.0 with in-context data')
k = names.index('layer3.0')
blk = bb.layer3[0]
sv = fns[k].saved_tensors
x_ = sv[0]
dout_ref = torch.cat([allhs[i][k].grad for i in range(2)])          # (40, 256, 14) fp64 exact
# CPU fp64 block
R = 20
def bnw(x, g, b):
    return torch.cat([F.batch_norm(x[i:i + R], None, None, g, b, True, 0.1, 1e-5) for i in range(0, x.shape[0], R)])
xc = x_.detach().cpu().double().permute(0, 2, 1).contiguous().requires_grad_(True)
P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in blk.named_parameters()}
y1 = F.conv1d(xc, P['conv1.weight'], None, 2, 1); y1.retain_grad()
h1 = F.relu(bnw(y1, P['bn1.weight'], P['bn1.bias'])); h1.retain_grad()
y2 = F.conv1d(h1, P['conv2.weight'], None, 1, 1); y2.retain_grad()
yd = F.conv1d(xc, P['downsample.0.weight'], None, 2, 0); yd.retain_grad()
out = F.relu(bnw(y2, P['bn2.weight'], P['bn2.bias']) + bnw(yd, P['downsample.1.weight'], P['downsample.1.bias']))
out.backward(dout_ref)
# GPU manual backward, step by step
x, w1, g1, b1, w2, g2, b2, y1g, m1, i1, h1g, y2g, m2, i2, outg, wd, gd, bd, ydg, md, idd = sv
doutg = dout_ref.permute(0, 2, 1).contiguous().float().cuda()
def cmpg(tag, a, b):
    a = a.detach().cpu().double().permute(0, 2, 1).numpy(); b = b.numpy()
    d = np.abs(a - b); badm = d > 1e-4 * np.abs(b).max()
    print('%-6s err %.2e max %.2e nbad %d rows %s pos %s' % (tag, d.max(), np.abs(b).max(), badm.sum(), np.unique(np.argwhere(badm)[:, 0]).tolist()[:8], np.unique(np.argwhere(badm)[:, 2]).tolist()[:12]))
dy2, dg2, db2, g = H.bn_bwd(doutg, y2g, R, m2, i2, g2, b2, 2, out=outg, want_g=True)
cmpg('dy2', dy2, y2.grad)
_, wdd2 = H.repack_weight(w2, need_fwd=False, need_dgrad=True)
dh1 = H.conv_dgrad(dy2, wdd2, 1, 1, h1g.shape[1])
cmpg('dh1', dh1, h1.grad)
dy1, dg1, db1, _ = H.bn_bwd(dh1, y1g, R, m1, i1, g1, b1, 1, dx=dh1)
cmpg('dy1', dy1, y1.grad)
dyd, dgd, dbd, _ = H.bn_bwd(g, ydg, R, md, idd, gd, bd, 0, dx=g)
cmpg('dyd', dyd, yd.grad)
_, wdd1 = H.repack_weight(w1, need_fwd=False, need_dgrad=True)
dx = H.conv_dgrad(dy1, wdd1, 2, 1, x.shape[1])
_, wddd = H.repack_weight(wd, need_fwd=False, need_dgrad=True)
H.conv_dgrad(dyd, wddd, 2, 0, x.shape[1], out=dx, accumulate=True)
cmpg('dx', dx, xc.grad)
print('dg2 err', float((dg2.cpu().double() - P['bn2.weight'].grad).abs().max()), 'max', float(P['bn2.weight'].grad.abs().max()))
print('---- with wgrad interleaved')
auto_log = TRACE['log']; TRACE['log'] = []; TRACE['on'] = True
dy2, dg2, db2, g = H.bn_bwd(doutg, y2g, R, m2, i2, g2, b2, 2, out=outg, want_g=True)
g0, dy20 = g.clone(), dy2.clone()
dw2 = H.conv_wgrad(dy2, h1g, 3, 1, 1)
print('after wgrad2: g same', torch.equal(g, g0), 'dy2 same', torch.equal(dy2, dy20))
_, wdd2 = H.repack_weight(w2, need_fwd=False, need_dgrad=True)
dh1 = H.conv_dgrad(dy2, wdd2, 1, 1, h1g.shape[1])
print('after dgrad2: g same', torch.equal(g, g0), 'dy2 same', torch.equal(dy2, dy20))
cmpg('dh1', dh1, h1.grad)
dy1, dg1, db1, _ = H.bn_bwd(dh1, y1g, R, m1, i1, g1, b1, 1, dx=dh1)
print('after bn1: g same', torch.equal(g, g0))
cmpg('dy1', dy1, y1.grad)
dy10 = dy1.clone()
dw1 = H.conv_wgrad(dy1, x, 3, 2, 1)
print('after wgrad1: g same', torch.equal(g, g0), 'dy1 same', torch.equal(dy1, dy10))
_, wdd1 = H.repack_weight(w1, need_fwd=False, need_dgrad=True)
dyd, dgd, dbd, _ = H.bn_bwd(g, ydg, R, md, idd, gd, bd, 0, dx=g)
cmpg('dyd', dyd, yd.grad)
dyd0 = dyd.clone()
dwd = H.conv_wgrad(dyd, x, 1, 2, 0)
print('after wgradd: dyd same', torch.equal(dyd, dyd0), 'dy1 same', torch.equal(dy1, dy10))
dx = H.conv_dgrad(dy1, wdd1, 2, 1, x.shape[1])
cmpg('dx part', dx, xc.grad)
_, wddd = H.repack_weight(wd, need_fwd=False, need_dgrad=True)
H.conv_dgrad(dyd, wddd, 2, 0, x.shape[1], out=dx, accumulate=True)
cmpg('dx', dx, xc.grad)

TRACE['on'] = False
man = TRACE['log']
print('autograd trace:')
for e in auto_log: print('  A', e)
print('manual trace:')
for e in man: print('  M', e)
