"""CPU oracle for the depth+VO training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  The product path (``depth-vo-feat_amd/``) never imports this
package and raises if the HIP library is missing.

What it is: a plain-torch-CPU restatement (explicit arithmetic, frozen
``align_corners`` semantics, no reference imports) of the reference's
``pytorch_version`` hot path.  Every function cites the reference file:line it
follows.  It is pinned against golden vectors produced by importing the
reference itself in the build container (``tests/golden/gen_golden.py`` ->
``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).

Third-party arithmetic: convolution / transposed convolution / area and
bilinear interpolation are ``torch`` (un-vendored, unpinned by the reference;
2.10.0 CPU here).  The oracle calls the same aten CPU ops for those and
restates everything the reference writes itself (geometry, sampling
coordinates, masks, losses, the training-step body and Adam).
"""
