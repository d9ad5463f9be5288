"""Is the |d log2 P| of the HIP path against the reference's golden vectors rounding, or a consequence of the
rank-revealing early exit of the truncating QR passes (ops.RANK_TOL)?  Runs the G7 cases with the early exit on (2^-56,
the product default) and off (0 = the plain factorisation the reference performs) and prints both differences next to
the oracle's own sensitivity.  Output: one JSON document on stdout (committed as profiles/r02_ranktol_study.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np          # noqa: E402
import golden_inputs as gi  # noqa: E402
import tnac4o_amd           # noqa: E402
from tnac4o_amd import ops  # noqa: E402

CASES = [(128, 1, 0, 8, False), (128, 1, 3, 8, False), (128, 1, 0, 32, False), (128, 2, 1, 8, False), (128, 3, 2, 8, False),
         (128, 2, 0, 32, False), (128, 1, 0, 8, True), (128, 3, 2, 32, True), (512, 1, 0, 32, False)]


def run(L, ins, rot, chi, pre, tol):
    ops.RANK_TOL = tol
    n = {128: 4, 512: 8}[L]
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=gi.droplet_J(L, ins), beta=3.0)
    if rot:
        s.rotate_graph(rot)
    if pre:
        s.precondition(mode='balancing')
    s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=chi)
    return s


def main():
    with open(os.path.join(gi.GOLDEN_DIR, 'g7_search.json')) as f:
        g7 = json.load(f)
    out = []
    for (L, ins, rot, chi, pre) in CASES:
        want = g7['L%d_i%d_r%d_chi%d_pre%d' % (L, ins, rot, chi, int(pre))]
        row = {'case': 'L%d #%d rot%d chi%d pre%d' % (L, ins, rot, chi, int(pre)), 'golden_log2P': want['probability'],
               'golden_negative_probability': want['negative_probability']}
        for name, tol in (('rank_tol_2^-56', 2.0 ** -56), ('rank_tol_0', 0.0)):
            s = run(L, ins, rot, chi, pre, tol)
            row[name] = {'dlog2P': abs(float(s.probability[0]) - want['probability']),
                         'dE': abs(float(s.energy[0]) - want['energy']),
                         'same_state': [int(x) for x in s.states[0]] == want['state0'],
                         'negative_probability': float(s.negative_probability),
                         'max_bond_diff_vs_cap': int(max(max(m.D) for m in s.rhoT))}
        out.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    ops.RANK_TOL = 2.0 ** -56
    print(json.dumps({'cases': out}, indent=1))


if __name__ == '__main__':
    main()
