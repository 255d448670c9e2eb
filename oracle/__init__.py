"""CPU oracle for the DeepSolid VMC inner loop -- TEST INFRASTRUCTURE ONLY.

This package is a torch-CPU float64 restatement of the reference algorithm
(bytedance/DeepSolid, files cited function by function).  It exists to check
the HIP path; it is never the thing measured or shipped.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``deepsolid_amd/`` imports
it, and the product path raises if the HIP library is missing -- there is no
CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * forward log-psi / orbitals / Ewald / PBC wrap: pinned against vectors
    produced by executing the reference's own ``network.py`` / ``ewaldsum.py`` /
    ``distance.py`` / ``supercell.py`` (``tests/golden/*.npz``, generator
    ``tools/make_golden.py``; jax.numpy replaced by a numpy stand-in because
    JAX is not installable here).
  * kinetic energy and energy gradient: pinned DIRECTLY -- the reference's own
    ``hamiltonian.py`` (``local_energy_seperate``, all four modes) and
    ``train.py`` (``jax.value_and_grad(make_loss)``) are executed over its own
    ``network.py`` under a torch-backed ``jax`` stand-in
    (``tools/jax_torch_standin.py``); every fixture holds ``ke_ref`` / ``ew_ref``
    (and ``grad_ref_*``), the oracle and the HIP chain are asserted against them
    to 1e-9 Ha.  What is not pinned is XLA's own numerics (slogdet, erfc and the
    autodiff underneath are numpy / scipy / torch here).  Finite differences of
    the reference-executed forward and the mode equivalences stay as side checks.
  * Metropolis step: pinned by explicit-noise vectors (JAX threefry stream is
    not reproducible here; decisions are compared for supplied noise).
"""
