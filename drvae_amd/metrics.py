"""Prediction metrics of ``eval_y_prediction`` (src/DGMMixin.py:158-190) computed where the
scores live (device tensors, float64): accuracy, ROC-AUC and average precision, binary and
macro-averaged.  The reference moves everything to numpy and calls scikit-learn; here the
sort/scan runs on the device and only the final scalars cross to the host.  Ties in the
scores are handled like scikit-learn does (one threshold per distinct score)."""
import math

import torch


def _groups_desc(score, positive):
    """distinct score values in DEscending order -> (cumulative positives, cumulative count)
    at the end of every tie group."""
    s, order = torch.sort(score.double(), descending=True)
    pos = positive[order].double()
    last = torch.ones_like(s, dtype=torch.bool)
    last[:-1] = s[1:] != s[:-1]
    ctp = torch.cumsum(pos, 0)[last]
    cnt = torch.arange(1, s.numel() + 1, device=s.device, dtype=torch.float64)[last]
    return ctp, cnt


def roc_auc(y_true, score):
    """area under the ROC curve (trapezoids over the distinct thresholds); nan when only one
    class is present (scikit-learn raises there and the reference turns that into nan,
    src/DGMMixin.py:168-171)."""
    positive = y_true.reshape(-1) > 0
    n = positive.numel()
    n_pos = int(positive.sum())
    if n == 0 or n_pos == 0 or n_pos == n:
        return float('nan')
    ctp, cnt = _groups_desc(score.reshape(-1), positive)
    cfp = cnt - ctp
    z = torch.zeros(1, dtype=torch.float64, device=ctp.device)
    tp, fp = torch.cat([z, ctp]), torch.cat([z, cfp])
    area = ((fp[1:] - fp[:-1]) * (tp[1:] + tp[:-1])).sum() * 0.5
    return float(area / (n_pos * (n - n_pos)))


def average_precision(y_true, score):
    """sum_k (R_k - R_{k-1}) P_k over the distinct thresholds (sklearn.metrics.average_precision_score)."""
    positive = y_true.reshape(-1) > 0
    n_pos = int(positive.sum())
    if positive.numel() == 0 or n_pos == 0:
        return 0.0
    ctp, cnt = _groups_desc(score.reshape(-1), positive)
    prev = torch.cat([torch.zeros(1, dtype=torch.float64, device=ctp.device), ctp[:-1]])
    return float((((ctp - prev) / n_pos) * (ctp / cnt)).sum())


def macro(metric, labels, proba):
    """unweighted mean of the one-vs-rest metric over the classes (``average='macro'`` on the one-hot
    labels: what src/DGMMixin.py:173-180 intends; that branch cannot run in the reference, see
    tests/golden/make_golden.py)."""
    vals = [metric(labels.reshape(-1) == j, proba[:, j]) for j in range(proba.shape[1])]
    return float('nan') if any(math.isnan(v) for v in vals) else sum(vals) / len(vals)


def eval_y_regression(pred, ylab):
    """dict(rmse, r2, pearr) for a continuous target (src/DGMMixin.py:181-188: numpy / sklearn r2_score /
    scipy pearsonr on the flattened vectors), in float64 where the tensors live"""
    y = ylab.reshape(-1).double()
    p = pred.reshape(-1).double()
    out = dict(rmse=float(torch.sqrt(((y - p) ** 2).mean())))
    ss_tot = ((y - y.mean()) ** 2).sum()
    out['r2'] = float(1.0 - ((y - p) ** 2).sum() / ss_tot) if float(ss_tot) > 0 else float('nan')
    yc, pc = y - y.mean(), p - p.mean()
    den = torch.sqrt((yc ** 2).sum() * (pc ** 2).sum())
    out['pearr'] = float((yc * pc).sum() / den) if float(den) > 0 else float('nan')
    return out


def eval_y_prediction(pred, proba, ylab, dim_y):
    """dict(acc, auroc, aupr) for a discrete target (src/DGMMixin.py:163-180)."""
    ylab = ylab.reshape(-1)
    out = dict()
    out['acc'] = float((pred.reshape(-1).int() == ylab.int()).float().sum() / max(ylab.numel(), 1)) \
        if ylab.numel() else float('nan')
    if dim_y == 2:
        out['auroc'] = roc_auc(ylab, proba[:, 1])
        out['aupr'] = average_precision(ylab, proba[:, 1])
    else:
        out['auroc'] = macro(roc_auc, ylab, proba)
        out['aupr'] = macro(average_precision, ylab, proba)
    return out


# ------------------------------------------------------------------------------------------------------------------
# The same metrics as 0-d float64 DEVICE tensors, fixed shapes and no host synchronisation anywhere (no ``int(...)``, no
# boolean-mask compaction): what the whole-set evaluation of ``fit`` captures into ONE hipGraph per dataset and reads
# back with a single copy (drvae_amd/fit.py).  Same float64 arithmetic per tie group as above; the degenerate cases
# (one class only, empty) come out as nan / 0 through ``torch.where`` instead of early returns.

def _tie_groups(score, positive):
    """sorted-by-descending-score views: (last-of-its-tie-group mask, cumulative positives, cumulative count, and the
    two cumulative values at the END OF THE PREVIOUS tie group -- 0 for the first)"""
    s, order = torch.sort(score.reshape(-1).double(), descending=True)
    pos = positive.reshape(-1)[order].double()
    last = torch.ones_like(s, dtype=torch.bool)
    last[:-1] = s[1:] != s[:-1]
    ctp = torch.cumsum(pos, 0)
    cnt = torch.arange(1, s.numel() + 1, device=s.device, dtype=torch.float64)
    zero = torch.zeros((), dtype=torch.float64, device=s.device)

    def prev_end(c):       # c is non-decreasing: the running max of its group-end values IS the latest group end's
        run = torch.cummax(torch.where(last, c, zero), 0).values
        return torch.cat([zero.reshape(1), run[:-1]])
    return last, ctp, cnt, prev_end(ctp), prev_end(cnt - ctp)


def roc_auc_dev(y_true, score):
    positive = y_true.reshape(-1) > 0
    n = positive.numel()
    nan = torch.full((), float('nan'), dtype=torch.float64, device=score.device)
    if n == 0:
        return nan
    n_pos = positive.sum().double()
    last, ctp, cnt, ptp, pfp = _tie_groups(score, positive)
    cfp = cnt - ctp
    area = torch.where(last, (cfp - pfp) * (ctp + ptp), torch.zeros_like(ctp)).sum() * 0.5
    return torch.where((n_pos == 0) | (n_pos == n), nan, area / (n_pos * (n - n_pos)))


def average_precision_dev(y_true, score):
    positive = y_true.reshape(-1) > 0
    zero = torch.zeros((), dtype=torch.float64, device=score.device)
    if positive.numel() == 0:
        return zero
    n_pos = positive.sum().double()
    last, ctp, cnt, ptp, _ = _tie_groups(score, positive)
    ap = torch.where(last, (ctp - ptp) * (ctp / cnt), torch.zeros_like(ctp)).sum() / torch.clamp(n_pos, min=1.0)
    return torch.where(n_pos == 0, zero, ap)


def eval_y_prediction_dev(pred, proba, ylab, dim_y):
    """``eval_y_prediction`` as a dict of 0-d float64 device tensors"""
    ylab = ylab.reshape(-1)
    dev = proba.device
    nan = torch.full((), float('nan'), dtype=torch.float64, device=dev)
    out = dict()
    out['acc'] = ((pred.reshape(-1).int() == ylab.int()).float().sum() / max(ylab.numel(), 1)).double() if ylab.numel() else nan
    if dim_y == 2:
        out['auroc'] = roc_auc_dev(ylab, proba[:, 1])
        out['aupr'] = average_precision_dev(ylab, proba[:, 1])
    else:       # macro average: nan as soon as one class's value is (the mean propagates it)
        out['auroc'] = torch.stack([roc_auc_dev(ylab == j, proba[:, j]) for j in range(proba.shape[1])]).mean()
        out['aupr'] = torch.stack([average_precision_dev(ylab == j, proba[:, j]) for j in range(proba.shape[1])]).mean()
    return out
