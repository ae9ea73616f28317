"""utils/model_tools.py:9-33 of the reference, restricted to what the RRNet path selects
(configs/rrnet_config.py:80 -> 'hourglass'); 'hourglass_tiny' is the builder-defined backbone of
BASELINE.json config 1.  The other names belong to out-of-scope models."""
from rrnet_amd.backbones.hourglass import hourglass_net, hourglass_tiny


def get_backbone(backbone, pretrained=False, num_stacks=2):
    if backbone == 'hourglass':
        return hourglass_net(num_stacks=num_stacks)
    if backbone == 'hourglass_tiny':
        return hourglass_tiny(num_stacks=num_stacks)
    raise NotImplementedError(
        "backbone %r is outside the accelerated RRNet path (only 'hourglass' / 'hourglass_tiny')" % (backbone,))
