"""Re-run the test epoch over saved models, fold by fold -- the reference's ``deepards/evaluate.py:15-49`` (the entry
point of its Jetson evaluation image) on the MI355X hot path.

    python -m deepards_amd.evaluate -co evaluate_config/unpadded_centered_nb20_cnn_linear.yml [--saved-models-dir DIR]

The override file is an experiment file plus ``experiment_name`` and ``models: {fold: [file, ...]}`` (the reference's
``evaluate_config/*.yml``).  For every fold the test patients of that fold are pushed through every listed model with
``BaseTraining.run_test_epoch`` (no_grad forward with train-mode modules, window argmax, per-patient votes on the
device); each model counts as one "epoch" of the fold, like the reference's aggregation hack (:34-37).  Models are
loaded by ``deepards_amd.checkpoint.load_model_weights``: files saved by the reference are read without being
unpickled.  Printed / returned: per fold, patient-level accuracy and AUC of ``pred_frac`` (:39-46); the reference's
``aggregate_classification_results`` report (metrics.py, prettytable / pandas frames) is out of scope.
"""
import argparse
import os

import numpy as np
import torch

from .checkpoint import load_model_weights
from .config import Configuration
from .train_ards_detector import BUILD_DEFAULTS, build_parser, network_map


def patient_rows(res, store, fold, epoch):
    """One row per test patient of the fold: (fold, epoch, patient slot, true class, predicted class, ARDS vote share)
    -- the columns of DeepARDSResults.results the reference's table reads (patho, prediction, pred_frac)."""
    slot = np.asarray(store.patient_slot)
    labels = store.targets.argmax(dim=1).cpu().numpy()
    rows = []
    for p in np.nonzero(res['votes'].sum(axis=1))[0].tolist():
        truth = int(labels[np.nonzero(slot == p)[0][0]])
        rows.append((fold, epoch, p, truth, int(res['prediction'][p]), float(res['pred_frac'][p])))
    return rows


def fold_table(rows):
    """[(fold, patient accuracy, AUC of pred_frac)] -- accuracy_score / roc_auc_score of evaluate.py:42-45."""
    from sklearn.metrics import accuracy_score, roc_auc_score
    out = []
    for fold in sorted({r[0] for r in rows}):
        fr = [r for r in rows if r[0] == fold]
        truth, pred, frac = [r[3] for r in fr], [r[4] for r in fr], [r[5] for r in fr]
        auc = float('nan') if len(set(truth)) < 2 else round(float(roc_auc_score(truth, frac)), 4)
        out.append((fold, round(float(accuracy_score(truth, pred)), 4), auc))
    return out


def main(argv=None):
    parser = argparse.ArgumentParser(prog='deepards_amd.evaluate')
    parser.add_argument('-co', '--config-override', required=True, help='override file')
    parser.add_argument('--saved-models-dir', help='directory of the listed model files (default: '
                                                   'saved_models/<experiment_name> next to this package)')
    parser_args = parser.parse_args(argv)
    model_args = build_parser().parse_args([])
    model_args.config_override = parser_args.config_override
    args = Configuration(model_args, BUILD_DEFAULTS)
    cls = network_map[args.network](args)
    _, test_dataset = cls.get_base_datasets()
    saved = parser_args.saved_models_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'saved_models',
                                                         str(args.experiment_name))
    rows = []
    for fold in sorted(args.models):
        test_dataset.set_kfold_indexes_for_fold(fold)
        loader = (test_dataset, args.batch_size, True)
        for i, model_name in enumerate(args.models[fold]):
            model = load_model_weights(os.path.join(saved, model_name),
                                       lambda: cls.get_network(cls.get_base_network())).to(cls.device)
            res = cls.run_test_epoch(i, model, test_dataset, loader, fold)
            rows += patient_rows(res, test_dataset, fold, i)
    cls.restore_dtypes()
    table = fold_table(rows)
    print('\nMean Results')
    print('%-6s %-10s %-8s' % ('Fold', 'Accuracy', 'AUC'))
    for fold, acc, auc in table:
        print('%-6d %-10.4f %-8.4f' % (fold, acc, auc))
    return cls, rows, table


if __name__ == '__main__':
    main()
