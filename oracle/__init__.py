"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the RVC hot path.

Nothing in the product package may import this directory.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
there only as the checker (see DESIGN.md, section "Oracle").

Contents
--------
nsf_oracle.py   torch-CPU fp32 restatement of the NSF-HiFi-GAN generator
                (reference: rvc/layers/{nsf,generators,residuals}.py).  Pinned:
                oracle/make_golden.py runs it against the reference's own modules
                imported from /root/reference and writes tests/golden/*.npz.
ivf_oracle.py   numpy fp64 restatement of faiss IndexIVFFlat search (nprobe
                lists, squared L2, top-k) + the reference's inverse-square blend
                (infer/modules/vc/pipeline.py:126-138).  PARITY UNPINNED: faiss is
                an un-vendored, un-pinned wheel (requirements/cpu.txt:8) that is
                not installable offline and the reference has no test or golden
                vector at this boundary.
synth.py        seeded synthetic weights / inputs / indices (no checkpoints or
                .index files exist offline).
ivf_scan.c      plain-C restatement of the IVF search used for the multi-core
                CPU baseline timing and as a second, independent checker.
"""
