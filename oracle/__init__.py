"""oracle/ -- TEST INFRASTRUCTURE ONLY (not product code).

A CPU restatement (stock ATen fp32 ops on the host) of the reference's
DeepLabV3+ training hot path: dilated ResNet backbone -> ASPP -> V3+ decoder ->
weighted CE / focal loss -> backward -> optimizer step.  Every function cites the
reference file:line it restates (paths relative to the reference checkout).

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker / the CPU
baseline, never as the thing measured or shipped.  ``iswm_amd`` (the product)
never imports it and has no CPU fallback.

Parity pin: ``oracle/make_golden.py`` imports the reference's own modules in the
build container and writes small input/output vectors under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against them.  The
reference publishes no tests or golden vectors of its own,
so those generated vectors *are* the pin.
"""
