import os, sys, cProfile, pstats, io, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, tnac4o_amd, golden_inputs as gi
s = tnac4o_amd.tnac4o(mode='Ising', Nx=16, Ny=16, Nc=8, J=gi.droplet_J(2048, 1), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=32, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw); torch.cuda.synchronize()
orig = s._setup_rhoT
s._setup_rhoT = lambda **k: None          # reuse the sweep; profile the beam part only
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
s.search_ground_state(M=1024, relative_P_cutoff=1e-8, Dmax=32); torch.cuda.synchronize()
dt = time.perf_counter() - t0; pr.disable()
print('beam part: %.2f s, E=%.6f' % (dt, s.energy[0]))
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(18); print(st.getvalue()[:4500])
