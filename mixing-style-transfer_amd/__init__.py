"""MI355X-native contrastive data path (mel + mixing features + band-split encoder forward).

Mirror of the reference's Python call contract (src/mixing_utils.py, src/model.py,
src/data.py, src/loss.py) on top of hand-written gfx950 HIP kernels behind the C ABI
declared in include/mst.h.  Import as `mst_amd` (alias package at the repo root) or put
this directory on sys.path in place of the reference's `src/`.
"""
__version__ = "0.1.0"
