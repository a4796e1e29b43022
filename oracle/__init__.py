"""oracle/ — TEST INFRASTRUCTURE ONLY.

A float64 numpy restatement of the sparse-recovery solver path of vlaxose/jstsp19
(proposed_algorithm / proposed_algorithm_angles / svt / OMP / sparse_admm / mc_svt /
mc_admm / vamp) and of the system-model functions that build their inputs.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from this package, and there only as the checker / the timed CPU
baseline — never as the thing measured or shipped.  The product (``jstsp19_amd``)
must not import it and must fail loudly when its HIP library is missing.

PARITY PINNING.  The reference is MATLAB-only, ships no tests, no golden vectors and
never seeds its RNG; MATLAB/Octave are absent from the build
container, so no output of the reference itself can be produced here.  The oracle is
therefore pinned by (a) a *literal* restatement that follows each ``.m`` file line by
line (dense ``kron``/``lu``/full ``svd`` exactly in the reference's operation order)
agreeing with (b) the *structured* restatement to <=1e-9, (c) closed-form known-answer
tests, and (d) the order-of-magnitude NMSE band of ``results/errorVSsnr_angles.fig``
(one unseeded trial per point).  Against the reference's own numerical outputs parity
is **unpinned** — DESIGN.md says the same.
"""
