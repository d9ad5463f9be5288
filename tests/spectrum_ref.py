"""Test adapter: the CPU oracle's branch-and-bound with the product's droplet bookkeeping (tnac4o_amd.droplets) plugged
into its merge hook -- exercises the host-side spectrum logic without a GPU, against vectors captured
from the reference (tests/golden/g10_spectrum.npz)."""
from oracle import solver_ref as sr
from tnac4o_amd import droplets


class SpectrumRef(sr.RefSolver):
    def search_low_energy_spectrum(self, excitations_encoding=1, max_dEng=0., lim_hd=0, **kw):
        assert excitations_encoding == 1
        rec = droplets.ExcitationRecorder(max_dEng, lim_hd, self.mode)
        E = self.search_ground_state(merge_hook=rec.merge_step, **kw)
        self.el, self.d = rec.finish(self.order_i)
        return E

    def decode_low_energy_states(self, max_dEng=0., max_states=1024):
        E, st = droplets.decode_states(self.states[0], self.el, self.d, self.Nx * self.Ny, max_dEng, max_states, self.indtype)
        self.energy = E + self.energy[0]
        self.states = st
        return E[0]
