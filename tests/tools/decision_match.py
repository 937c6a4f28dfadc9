"""Test helper: gradient parity under matched activation decisions (see the docstring)."""
import numpy as np

from oracle import np_ref


def decision_matched_gradients(ref, ours, log_tag='', tols=(1e-5, 3e-5), log=print, max_eval=128, known=(), only=None):
    """The exact (fp64 oracle) gradients under the activation decisions THIS run took.

    An fp32 forward differs from the fp64 one by ~1e-6 relative, enough to take a ReLU / max-pool decision the other way
    on the few elements whose pre-activation is that close to zero; ONE such element changes every upstream gradient by
    up to ~3e-2 relative (it is one of ~10^4 elements of a late feature map).  The reference does it to itself: its own
    fp32 vs fp64 gradients differ by 3.1e-2 rel-l2 on densenet18_b4_flow.  So gradients are compared with the oracle's
    gradients under the SAME decisions.  The oracle lists its ambiguous decisions (|pre-activation| or max-pool gap
    < tol); for each the backward is re-run with that one decision flipped (forward untouched) -> its effect D_e.
    Matching pursuit over the candidates in order of decreasing effect: a flip is adopted when the current residual
    (ours - exact - adopted effects) projects onto D_e with coefficient > 1/2.  The adopted set is then applied at once
    in an exact re-run.  Effects are matched on a fixed coordinate subsample (<= 2048 per parameter), the verdict is
    taken by the caller on the full tensors.  At most ``max_eval`` candidates are evaluated per tolerance, those closest
    to their decision boundary first (every evaluation is one oracle backward: seconds for the deeper backbones).
    ``known``: flips [(name, index)] that are not searched for but KNOWN -- the decisions the run under test exported
    (``hip_relu_flips``) -- they are part of every backward here; ``only(name)``: restrict the search to those decisions
    (the ones the run cannot export: the stem's fused ReLU / max-pool).
    Returns (gradients, flips adopted (known ones first), number of candidates)."""
    known = list(known)
    base = ref['rebackward'](known) if known else ref['grads']
    names = [n for n in base if n in ours]
    # gradients that are analytically zero (a conv in front of a BatchNorm when every ReLU is active) get a unit
    # scale: their error is judged absolutely, not relative to a norm of ~1e-17
    # ... and one whose norm is small but not zero is weighted by the size of error the verdict tolerates on it (1e-4 per
    # element), or fp32 noise on ~1e-9 gradients would steer the matching
    scale = {n: max(float(np.linalg.norm(base[n])), 1e-4 * np.sqrt(base[n].size)) for n in names}
    known3 = [(n, i, 0.0) for n, i in known]
    if max(np.linalg.norm(ours[n] - base[n]) / scale[n] for n in names) < 2e-5:
        return base, known3, 0
    rng = np.random.RandomState(0)
    sub = {n: (np.arange(base[n].size) if base[n].size <= 2048 else np.sort(rng.choice(base[n].size, 2048, replace=False)))
           for n in names}
    pack = lambda g: np.concatenate([(g[n].ravel()[sub[n]] - base[n].ravel()[sub[n]]) *
                                     (np.sqrt(base[n].size / len(sub[n])) / scale[n]) for n in names])
    got, chosen, cands = base, [], []
    for tol in tols:
        cands = [c for c in np_ref.ambiguous_decisions(ref['tape'], tol) if (only is None or only(c[0])) and (c[0], c[1]) not in set(known)]
        cands = sorted(cands, key=lambda c: c[2])[:max_eval]
        effects = [pack(ref['rebackward'](known + [(name, i)])) for name, i, _ in cands]
        resid = pack(ours)
        chosen = []
        for k in np.argsort([-float(e @ e) for e in effects]):
            e = effects[k]
            if float(e @ e) < 1e-12:                          # cannot move any parameter by 1e-6 relative
                break
            if float(resid @ e) / float(e @ e) > 0.5:
                chosen.append(cands[k])
                resid = resid - e
        got = ref['rebackward'](known + [(n, i) for n, i, _ in chosen]) if chosen else base
        worst = max(np.linalg.norm(ours[n] - got[n]) / scale[n] for n in names)
        log('   %s decision matching: tol %.0e, %d candidates, %d flips adopted %s, worst rel-l2 after %.2e' %
            (log_tag, tol, len(cands), len(chosen), [(n.split('.', 1)[1], i, '%.1e' % m) for n, i, m in chosen], worst))
        if worst < 1e-4:
            break
    return got, known3 + chosen, len(cands)


STEM_DECISIONS = ('bn1.relu', 'bn1.maxpool', 'bn2.relu', 'bn2.maxpool', 'norm0.relu', 'norm0.maxpool')


def is_stem_decision(name):
    return name.endswith(STEM_DECISIONS)


def hip_relu_flips(tape, taps, log=print, tag='', near=3e-5):
    """The ReLU decisions the run under test TOOK, as flips of the oracle's: ``taps`` are the post-ReLU activations the
    block Functions recorded (deepards_amd.functional.DECISION_TAP: float (rows, L, C) or x3 tensors, in forward order),
    one per ReLU of the oracle's tape behind the stem (the stem's ReLU and max-pool are fused into one kernel and export
    nothing: ``decision_matched_gradients(only=is_stem_decision)`` searches those).  Every differing element must be one
    whose fp64 pre-activation is within ``near`` of zero -- anything else is a wrong value, not a decision.  ``near`` is
    the oracle's own ambiguity tolerance (``decision_matched_gradients(tols=...)``: 3e-5; the largest exported
    pre-activation ever measured is 7.5e-6): a looser bound would let a 1e-4-sized VALUE error pass as
    "a decision"."""
    names = [n for n in tape.order if tape.decisions[n]['kind'] == 'relu' and not is_stem_decision(n)]
    assert len(names) == len(taps), 'decision tap: %d activations for %d ReLUs' % (len(taps), len(names))
    flips = []
    for name, t in zip(names, taps):
        d = tape.decisions[name]
        if t.dim() == 5:                                    # x3 format: the sign of the leading term is the sign of the value
            t = t[:, :, :, 0, :].reshape(t.shape[0], t.shape[1], -1)
        mask = (t.detach().float() > 0).permute(0, 2, 1).cpu().numpy()
        assert mask.shape == d['mask'].shape, (name, mask.shape, d['mask'].shape)
        idx = np.flatnonzero(mask != d['mask'])
        if len(idx):
            far = np.abs(d['pre'].flat[idx]).max()
            assert far < near, '%s: a ReLU decision differs where the fp64 pre-activation is %.2e from zero' % (name, far)
            flips += [(name, int(i)) for i in idx]
    if flips:
        log('   %s exported decisions: %d ReLU elements on the other side of zero than the fp64 oracle (|pre| <= %.1e)' %
            (tag, len(flips), max(abs(tape.decisions[n]['pre'].flat[i]) for n, i in flips)))
    return flips


def rel_l2(a, b):
    nb = float(np.linalg.norm(b))
    return float(np.linalg.norm(a - b) / (nb if nb > 1e-9 else 1.0))      # ~zero references: absolute


def assert_gradients_match(ref, ours, tag='', strict=False, max_flips=12, log=print, allow=None, taps=None):
    """THE gradient yardstick of this suite (BASELINE north_star: 1e-4): every parameter of ``ours`` (name -> float64
    array) within 1e-4 (rel-l2, or max abs err <= 1e-4 * max(1, max|ref|)) of the oracle's exact gradients under the
    activation decisions this run took (``decision_matched_gradients``); at most ``max_flips`` adopted flips, no ReLU
    flip when ``strict`` (goldens whose every ReLU is active; a DenseNet stem's max-pool keeps its near-ties even there:
    two neighbouring conv outputs 1e-5 apart are not moved by a BatchNorm shift).  ``allow(name) -> bool`` exempts named parameters (callers document
    why).  ``taps``: the post-ReLU activations the run recorded (functional.DECISION_TAP) -- its decisions are then TAKEN from
    the run (``hip_relu_flips``) and only the stem's fused ReLU / max-pool decisions are searched.
    Returns (worst rel-l2, flips)."""
    if taps is not None:       # the run exported its ReLU decisions: nothing to search for behind the stem
        known = hip_relu_flips(ref['tape'], taps, log=log, tag=tag)
        matched, flips, ncand = decision_matched_gradients(ref, ours, tag, log=log, known=known, only=is_stem_decision)
    else:
        matched, flips, ncand = decision_matched_gradients(ref, ours, tag, log=log)
    if strict:
        assert not [f for f in flips if not f[0].endswith('.maxpool')], flips
    assert len(flips) <= max_flips, flips
    worst, bad = 0.0, []
    for n in ours:
        if n not in matched:
            continue
        abs_err = float(np.abs(ours[n] - matched[n]).max())
        rl2 = rel_l2(ours[n], matched[n])
        worst = max(worst, rl2)
        scale = max(1.0, float(np.abs(matched[n]).max()))
        if not (rl2 <= 1e-4 or abs_err <= 1e-4 * scale) and not (allow is not None and allow(n)):
            bad.append((n, abs_err, rl2, rel_l2(ours[n], ref['grads'][n])))
    log('   %s worst grad rel-l2 %.3e with %d flips of %d candidates' % (tag, worst, len(flips), ncand))
    assert not bad, bad
    return worst, flips


def feature_reference(params64, rows_per_window, x64, cotangent, backbone='resnet18', **tape_flags):
    """The oracle at breath-block level -- features of ``x64`` (rows, 1, L) and the parameter gradients of
    sum(features * cotangent) -- in the form ``decision_matched_gradients`` wants (grads, tape, rebackward).  Used where
    CNNLinearNetwork itself refuses the shape (seq_len != 224, torch_cnn_linear_network.py:106-107)."""
    t = np_ref._Tape(params64, rows_per_window)
    for k, v in tape_flags.items():
        setattr(t, k, v)
    fn = np_ref.resnet18_features if backbone.startswith('resnet') else np_ref.densenet18_features
    feat, bwd = fn(t, x64)
    extra = {}
    if callable(cotangent):                 # a head stated by the caller: feat -> (d loss / d feat, its own gradients, ...)
        cotangent, extra = cotangent(feat)

    def run():
        t.g = {}
        bwd(cotangent)
        g = dict(t.g)
        g.update(extra.get('grads', {}))
        return g

    def rebackward(flips):
        saved = {}
        for name, i in flips:
            d = t.decisions[name]
            key = 'mask' if d['kind'] == 'relu' else 'idx'
            saved.setdefault(name, (key, d[key]))
            np_ref._apply_flip(t, name, i)
        try:
            return run()
        finally:
            for name, (key, val) in saved.items():
                t.decisions[name][key] = val
    out = dict(feat=feat, grads=run(), tape=t, rebackward=rebackward)
    out.update({k: v for k, v in extra.items() if k != 'grads'})
    return out
