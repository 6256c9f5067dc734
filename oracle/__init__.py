"""CPU oracle for the contrastive data-path hot path (TEST INFRASTRUCTURE, not product).

Plain PyTorch-CPU / numpy / scipy restatement of the reference algorithm
(barry-mir/mixing-style-transfer: src/mixing_utils.py, src/model.py, src/loss.py,
src/data.py) used ONLY by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg as the checker / reported baseline.  The product package
(`mixing-style-transfer_amd/`) never imports it.

Pinning (see DESIGN.md "Oracle"):
  * features / encoder / InfoNCE / augmentation: pinned against the reference's own
    Python code executed in the build container (tests/golden/make_golden.py; the
    outputs are committed under tests/golden/, the reference never travels).
  * mel front end: the reference delegates to `torchaudio.transforms.MelSpectrogram`
    (torchaudio>=2.0.0, unpinned, not installed here); restated from its documented
    algorithm -> "pinned to algorithm", equality with a real torchaudio unverified.
  * SCNet separation and src/data.py: not importable (un-vendored submodule)
    -> dataset crop/collate logic restated from source, parity unpinned.
  * retrieval metric / style-transfer sampling order (oracle/retrieval.py): restated
    from src/validation_utils.py:217-282 and src/data.py:467-519; the reference holds
    no fixtures for them and validation_utils.py needs librosa -> parity unpinned.
"""
