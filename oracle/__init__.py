"""CPU oracle for the key-estimation hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / stock torch CPU ops) of the
reference algorithm for the path  CQT front end -> PitchClassNet forward
(-> loss / MIREX score).  It is the *checker*:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * the product package (``audio-key-estimation_amd``) never imports it and
    has no CPU fallback: without the HIP library it raises.

Pinning status (see DESIGN.md section "Oracle"):

  * ``pcnet_oracle``  -- PINNED: equals the reference ``models.py`` (imported in
    the build container with inert stubs, ``oracle/make_golden.py``) to <=1e-12
    in float64 on the committed fixtures under ``tests/golden/``.
  * ``mirex_oracle``  -- PINNED against the reference ``mirex_score`` and the
    ``mel_shifting_*`` helpers the same way.
  * ``loss_oracle``   -- PINNED against the reference ``general_step`` loss.
  * ``cqt_oracle``    -- PARITY UNPINNED.  The reference's CQT is one call into
    third-party ``librosa`` (pinned 0.9.2 in requirements.txt:250, not vendored,
    not installable here) and the reference holds no fixture at that boundary.
    ``cqt_oracle`` restates librosa's *published definition* (direct-form
    constant-Q transform) in float64; it is the
    build's own specification of that stage.
"""
