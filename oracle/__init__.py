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
  * kinetic energy: the reference has no golden numbers and its autodiff
    runtime (jax.grad/jvp) is absent, so the `for`-mode restatement is pinned
    indirectly: finite differences of the reference-executed forward, mode
    equivalence for == hessian == partition == dim_batch, and the three
    wavefunction properties the reference's test/test_network.py checks.
  * Metropolis step: pinned by explicit-noise vectors (JAX threefry stream is
    not reproducible here; decisions are compared for supplied noise).
"""
