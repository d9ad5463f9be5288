"""Test adapter: the CPU oracle's branch-and-bound with the product's droplet bookkeeping (tnac4o_amd.droplets) plugged
into its merge hook -- exercises the host-side spectrum logic without a GPU, against vectors captured
from the reference (tests/golden/g10_spectrum.npz, g11_spectrum_adjacency.npz)."""
import numpy as np

from oracle import solver_ref as sr
from tnac4o_amd import droplets


class SpectrumRef(sr.RefSolver):
    def add_noise(self, amplitude=1e-7):
        """tnac4o.py:917-941 on the oracle's dense upper-triangular couplings."""
        if self.mode == 'Ising':
            rows, cols = self.J.nonzero()
            self.J[rows, cols] += (np.random.rand(len(rows)) * 2 - 1) * amplitude
        else:
            fun = {}
            for key, val in self.J['fun'].items():
                fun[key] = np.array(val, dtype=float, copy=True)
                if fun[key].ndim == 1:
                    fun[key] += (np.random.rand(fun[key].shape[0]) * 2 - 1) * amplitude
            self.J['fun'] = fun
        self._divide_couplings()

    def search_low_energy_spectrum(self, excitations_encoding=1, max_dEng=0., lim_hd=0, **kw):
        self.excitations_encoding = enc = excitations_encoding
        ising = self.mode == 'Ising'
        if not hasattr(self, 'J0_'):
            self.J0_ = None
        if enc == 1:
            rec = droplets.ExcitationRecorder(max_dEng, lim_hd, self.mode)
        else:
            conn = droplets.Connectivity(self.mode, self.Nx, J=self.J if ising else None, ind=self.ind if ising else None)
            rec = (droplets.AdjacencyRecorder if enc == 2 else droplets.FlatRecorder)(max_dEng, lim_hd, self.mode, conn)
        kw2 = dict(kw)
        E = self.search_ground_state(merge_hook=rec.merge_step, row_hook=getattr(rec, 'end_row', None), **kw2)
        self.el, self.d = rec.finish(self.order_i)
        if enc > 1:
            # unrotated adjacency: rotate the couplings back through the cell order
            self._conn = droplets.Connectivity(self.mode, self.Nx_model, J=self.J_unrotated() if ising else None,
                                               ind=self.ind0 if ising else None)
        return E

    def J_unrotated(self):
        """Couplings in the original spin order: cell k of the rotated lattice is cell order_i... of the original."""
        Nc = self.Nc
        perm = (np.asarray(self.order)[:, None] * Nc + np.arange(Nc)[None, :]).reshape(-1)   # original spin -> rotated spin
        Jp = self.J[np.ix_(perm, perm)]
        return np.triu(Jp) + np.tril(Jp, -1).T

    def decode_low_energy_states(self, max_dEng=0., max_states=1024):
        if self.excitations_encoding == 1:
            E, st = droplets.decode_states(self.states[0], self.el, self.d, self.Nx * self.Ny, max_dEng, max_states,
                                           self.indtype)
        else:
            E, st = droplets.decode_states_adjacent(self.states[0], self.el, self.d, self._conn, max_dEng, max_states,
                                                    self.indtype, one_layer=(self.excitations_encoding == 3))
        self.energy = E + self.energy[0]
        self.states = st
        return E[0]
