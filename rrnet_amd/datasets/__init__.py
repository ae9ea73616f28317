"""Data contract of the hot path.  The reference's dataset / augmentation code (datasets/, 875 LoC) is out of
scope; what the losses consume — the target tensors of the CenterNet collate contract — is produced on
the device by rr_ctnet_targets and used by the synthetic VisDrone-shaped generator in synthetic.py."""
from .synthetic import SyntheticDronesDET, make_dataloader  # noqa: F401
