"""CPU oracle for the SceneSplat PTv3 hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / pure-PyTorch (fp32, CPU) restatement of the
reference algorithm for the hot path named in BASELINE.json.  It exists so
that the HIP path can be checked against something that is itself pinned to
the reference.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; ``scenesplat_amd`` (the
product) never does and fails loudly when its HIP library is missing.

Parity pinning (see DESIGN.md "Oracle"):
  * serialization, window padding, windowed attention (non-flash math),
    pooling / unpooling, Block wiring incl. the stale-CPE quirk, the three
    distillation losses and the end-to-end tiny PTv3: PINNED against outputs of
    the reference itself, imported in the build container by
    ``tests/golden/make_golden.py`` and stored under ``tests/golden/*.npz``.
  * ``spconv.SubMConv3d`` and ``torch_scatter.segment_csr`` live in
    un-vendored third-party CUDA wheels (env.yaml:42-53) that are absent from
    /root/reference: "parity unpinned" for their arithmetic; restated from
    their published definition and cross-checked against a dense
    ``torch.nn.functional.conv3d`` (tests/test_oracle.py).
  * libs/pointops, pointops2, pointgroup_ops: CUDA sources cannot be built
    here (no nvcc) and the reference holds no fixtures for them: "parity
    unpinned"; restated from the .cu files cited in each docstring.
"""
