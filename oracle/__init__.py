"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the reference hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the reported CPU baseline.  The product path
(``myriad_amd``) never imports this package and fails loudly when its HIP
library is missing.

Parity status: PINNED for rows a-1..a-13 of SURVEY.md section 8 against golden
vectors generated in the build container by importing the reference's own
modules (``tools/make_golden.py`` -> ``tests/golden/*.npz``).  Row a-14 (PEFT
LoRA arithmetic) is "parity unpinned": ``peft`` is an un-vendored, un-pinned
third-party dependency that is absent from the container; its published
formula is restated (see ``myriad_ref.llama_forward``).
"""
