"""Host-side data contract of the hot path.  The reference's dataset / augmentation code
(datasets/, 875 LoC) is out of scope; what the losses consume — the target tensors
produced by `to_heatmap` + `collate_fn_ctnet` — is restated in transforms/functional.py and
used by the synthetic VisDrone-shaped generator in synthetic.py."""
from .synthetic import SyntheticDronesDET, make_dataloader  # noqa: F401
