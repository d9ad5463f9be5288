"""CPU oracle for the boundary-MPS PEPS contraction path of marekrams/tnac4o.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy/scipy restatement of the
reference algorithm (every function cites the reference file:line it follows).  It is
the *checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``tnac4o_amd/`` does,
and the product path raises if the HIP library is missing instead of falling back here.

Parity pin: the oracle is checked (tests/test_oracle_golden.py) against
  * the reference's own golden data copied as fixtures (ground-state energies and bit
    strings of the droplet instances, J124 energy/degeneracy), and
  * vectors captured by importing the reference in the authoring container
    (tools/make_golden.py -> tests/golden/*.npz).
"""
