"""Generates tests/golden/analyze_nll_vectors.json from the reference's own
`tracegnn/utils/analyze_nll.py` + `fscore_utils.py` (loaded by file path under a stand-in
package name, so that nothing else of the reference package is imported).  Run in the build
container only (the reference does not travel):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_analyze_vectors.py

The fixture holds inputs (scores, labels, threshold) and the reference's result dict.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference/tracegnn/utils'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'analyze_nll_vectors.json')


def load_reference():
    pkg = types.ModuleType('refutils')
    pkg.__path__ = [REF]
    sys.modules['refutils'] = pkg
    mods = {}
    for name in ('fscore_utils', 'analyze_nll'):
        spec = importlib.util.spec_from_file_location(f'refutils.{name}', os.path.join(REF, f'{name}.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f'refutils.{name}'] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods


def main():
    mods = load_reference()
    analyze = mods['analyze_nll'].analyze_anomaly_nll
    cases = []
    rng = np.random.RandomState(7)
    for n, frac_drop, frac_lat, shift in ((200, 0.1, 0.1, 3.0), (64, 0.25, 0.05, 1.0),
                                          (500, 0.02, 0.3, 0.5), (40, 0.5, 0.25, 6.0)):
        labels = rng.choice([0, 1, 2], size=n, p=[1 - frac_drop - frac_lat, frac_drop, frac_lat])
        labels[:3] = [0, 1, 2]          # every class present
        nll = rng.normal(10.0, 2.0, size=n) + shift * (labels == 1) + 0.7 * shift * (labels == 2)
        nll = np.round(nll, 3)          # ties happen in real score lists too
        thr = float(np.percentile(nll[labels == 0], 97.0))
        want = analyze(nll_list=nll, label_list=labels, threshold=thr)
        cases.append({'nll': nll.tolist(), 'labels': labels.tolist(), 'threshold': thr,
                      'result': {k: float(v) for k, v in want.items()}})
    with open(OUT, 'w') as f:
        json.dump({'source': 'tracegnn/utils/analyze_nll.py:18-110 + fscore_utils.py:13-41',
                   'cases': cases}, f)
    print('wrote', OUT, len(cases), 'cases;', sorted(cases[0]['result']))


if __name__ == '__main__':
    main()
