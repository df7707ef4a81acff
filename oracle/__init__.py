"""CPU oracle for the SEQUOIA hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 / numpy) of the reference
algorithm for the path BASELINE.json names:

    uint8 patch -> ImageNet normalise -> ResNet-50 forward_extract
    -> per-slide KMeans(100, random_state=0) -> cluster means
    -> ViS (SummaryMixing aggregator) -> 20 820-gene head
    (+ MSE / AdamW training step, MAE / Pearson / SMAPE metrics)

It exists so the HIP path in ``sequoia-pub_amd/`` can be checked for parity and
so ``bench.py`` can time a CPU baseline on the GPU box's host cores.  ONLY
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  The product package never imports it and has no CPU fallback:
without the HIP extension and a GPU it raises.

Pinning (see DESIGN.md "Oracle"): every function here was checked in the build
container against the importable reference modules (``/root/reference/src/
tformer_lin.py``, ``src/resnet.py``, ``src/vit.py`` + ``src/he2rna.py`` with
stubbed imports) and against scikit-learn 1.7.2 ``KMeans``; the outputs of those
runs are committed as golden vectors under ``tests/golden/`` together with the
script that generated them (``tests/golden/make_golden.py``).  The reference has
no tests or golden vectors of its own.
"""
