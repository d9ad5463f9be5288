"""Do two settings of the chain driver give the same boundary MPS?  Runs the bench instance's top-down sweep (L = 2048, chi = 64, seed
20260004) in child processes under the environment settings given on the command line (the switches are read once per process), keeps
the compressed boundary MPS of every row on the box, and prints per row: discarded weight of every variant and 1 - fidelity of every
variant against the first.  Usage: ab_state.py "TN_GAUGE_SVD=1" "TN_VAR1_SKIP=0" "" ...   ("" = the defaults)"""
import os, pickle, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))

CHILD = r'''
import os, sys, pickle
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
s = tnac4o_amd.tnac4o(mode='Ising', Nx=16, Ny=16, Nc=8, J=synthetic_chimera(16, 16, 20260004), beta=3.0)
s._setup_rhoT(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
torch.cuda.synchronize()
out = dict(disc=[float(x) for x in s.rhoT_discarded], ovl=[float(x) for x in s.rhoT_overlap],
           A=[[a.detach().cpu().numpy() for a in m.A] for m in s.rhoT if m is not None])
pickle.dump(out, open(sys.argv[1], 'wb'))
'''

def main():
    settings = sys.argv[1:] or ['', 'TN_GAUGE_SVD=1']
    outs = []
    for i, st in enumerate(settings):
        env = dict(os.environ)
        for kv in st.split():
            k, v = kv.split('=', 1)
            env[k] = v
        path = '/tmp/ab_state_%d.pkl' % i
        subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT), path], env=env, check=True)
        outs.append(pickle.load(open(path, 'rb')))
    import numpy as np
    from oracle import mps_ref as mr

    def chain(As):
        o = mr.RefMPS(d=[a.shape[1] for a in As], L=len(As), Dmax=1, canonise=None)
        o.A = As
        return o
    print('settings:', ['(defaults)' if not s else s for s in settings])
    nrow = len(outs[0]['A'])
    for r in range(nrow):
        base = chain(outs[0]['A'][r])
        nb = mr.mps_dot(base, base)
        line = 'row %2d  disc' % r
        for o in outs:
            line += ' %.6e' % o['disc'][r]
        line += '   1-F'
        for o in outs[1:]:
            c = chain(o['A'][r])
            f = abs(mr.mps_dot(base, c)) / np.sqrt(nb * mr.mps_dot(c, c))
            line += ' %.2e' % (1.0 - f)
        line += '   bonds ' + ' | '.join(','.join(str(a.shape[2]) for a in o['A'][r][:-1]) for o in outs)
        print(line, flush=True)

if __name__ == '__main__':
    main()
