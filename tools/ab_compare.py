"""Compare two sets of solver outputs (r00..r02.g000.out.bin + .ctl, the format er3t/rtm/mca/mca_out.py:48-103 parses) of one
A/B case: tools/ab_compare.py <case> <dir_a> <dir_b>.  Per output variable: domain means, their difference in standard errors
of the difference (runs as batches), and per-pixel z-scores."""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from er3t_amd.rtm.mca.mca_out import mca_out_raw       # noqa: E402


def load(d):
    runs = []
    for f in sorted(glob.glob(os.path.join(d, 'r0?.g000.out.bin'))):
        runs.append([v['data'].astype(np.float64) for v in mca_out_raw(f).data])
    return runs


def main():
    case, da, db = sys.argv[1:4]
    a, b = load(da), load(db)
    if not a or not b or len(a) != len(b):
        print('%s: outputs missing (%d against %d runs)' % (case, len(a), len(b)))
        return 1
    n = len(a)
    for iv in range(len(a[0])):
        A = np.stack([r[iv] for r in a]); B = np.stack([r[iv] for r in b])
        ma, mb = A.mean(axis=tuple(range(1, A.ndim))), B.mean(axis=tuple(range(1, B.ndim)))
        se = np.sqrt(ma.var(ddof=1)/n + mb.var(ddof=1)/n)
        sep = np.sqrt(A.var(axis=0, ddof=1)/n + B.var(axis=0, ddof=1)/n)
        ok = sep > 0
        z = (A.mean(axis=0)-B.mean(axis=0))[ok]/sep[ok]
        print('%-10s var %d: mean %.6g against %.6g  (%+.2f sigma, %+.3f %%)   per-pixel z: mean %+.2f std %.2f  |z|>2 %.3f  |z|>3 %.4f'
              % (case, iv, ma.mean(), mb.mean(), (ma.mean()-mb.mean())/max(se, 1e-300), 100.0*(ma.mean()/mb.mean()-1.0) if mb.mean() else 0.0,
                 z.mean(), z.std(), np.mean(np.abs(z) > 2), np.mean(np.abs(z) > 3)))
    return 0


if __name__ == '__main__':
    sys.exit(main())
