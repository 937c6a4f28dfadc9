"""Test helper: gradient parity under matched activation decisions (see the docstring)."""
import numpy as np

from oracle import np_ref


def decision_matched_gradients(ref, ours, log_tag='', tols=(1e-5, 3e-5), log=print):
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
    taken by the caller on the full tensors.  Returns (gradients, flips adopted, number of candidates)."""
    base = ref['grads']
    names = [n for n in base if n in ours]
    # gradients that are analytically zero (a conv in front of a BatchNorm when every ReLU is active) get a unit
    # scale: their error is judged absolutely, not relative to a norm of ~1e-17
    scale = {n: (float(np.linalg.norm(base[n])) if float(np.linalg.norm(base[n])) > 1e-9 else 1.0) for n in names}
    if max(np.linalg.norm(ours[n] - base[n]) / scale[n] for n in names) < 2e-5:
        return base, [], 0
    rng = np.random.RandomState(0)
    sub = {n: (np.arange(base[n].size) if base[n].size <= 2048 else np.sort(rng.choice(base[n].size, 2048, replace=False)))
           for n in names}
    pack = lambda g: np.concatenate([(g[n].ravel()[sub[n]] - base[n].ravel()[sub[n]]) *
                                     (np.sqrt(base[n].size / len(sub[n])) / scale[n]) for n in names])
    got, chosen, cands = base, [], []
    for tol in tols:
        cands = np_ref.ambiguous_decisions(ref['tape'], tol)
        effects = [pack(ref['rebackward']([(name, i)])) for name, i, _ in cands]
        resid = pack(ours)
        chosen = []
        for k in np.argsort([-float(e @ e) for e in effects]):
            e = effects[k]
            if float(e @ e) < 1e-12:                          # cannot move any parameter by 1e-6 relative
                break
            if float(resid @ e) / float(e @ e) > 0.5:
                chosen.append(cands[k])
                resid = resid - e
        got = ref['rebackward']([(n, i) for n, i, _ in chosen]) if chosen else base
        worst = max(np.linalg.norm(ours[n] - got[n]) / scale[n] for n in names)
        log('   %s decision matching: tol %.0e, %d candidates, %d flips adopted %s, worst rel-l2 after %.2e' %
            (log_tag, tol, len(cands), len(chosen), [(n.split('.', 1)[1], i, '%.1e' % m) for n, i, m in chosen], worst))
        if worst < 1e-4:
            break
    return got, chosen, len(cands)
