"""CPU oracle for the one-to-many GAN G+D training step.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 restatement of the
reference algorithm (struan-robertson/one-to-many-gan: src/model/layers.py,
src/model/blocks.py, src/model/builder.py, src/model/loss.py, src/core/training.py).
It exists so that the HIP path can be checked against it; it is never the thing that is
measured or shipped.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``one_to_many_gan_amd``) must never import from here.

Parity status: PINNED.  The reference holds no golden vectors of its own (it has no tests),
so the oracle is pinned against outputs of the reference itself, produced in the build
container by ``tools/make_golden.py`` (which imports /root/reference) and committed as
small fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every one
of them.  The single exception is the third-party ``ada`` augmentation (pytorch-ada @
99754cb4, not vendored, not importable offline): it is held at p = 0 (identity) on both
sides and is "parity unpinned".
"""
