"""CPU oracle for the gan-control StyleGAN2 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the reported CPU
baseline.  The product (``gan-control_amd/``) never imports this package and
fails loudly when its HIP library is missing.

The oracle is a plain-PyTorch (CPU, fp32 or fp64) restatement of the
reference's FUSED=False path (``src/gan_control/models/gan_model.py:24-50``,
``src/gan_control/models/pytorch_upfirdn2d.py:9-51``) and of the step maths in
``src/gan_control/trainers/generator_trainer.py``.  It is pinned by
``oracle/make_golden.py``: run in the build container, that script imports the
reference from ``/root/reference/src``, asserts that every oracle function
agrees with it, and writes the golden vectors under ``tests/golden/``.  The
reference holds no tests or golden vectors of its own for this path
(SURVEY.md section 4), so those generated fixtures are the pins.
"""
