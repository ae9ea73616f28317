"""The two test-time helpers of the reference's datasets/transforms/functional.py that the operators call
(flip evaluation, operators/centernet_operator.py:259-262).  Target generation (gaussian splat + regression
targets, functional.py:177-262 of the reference) is the device kernel rr_ctnet_targets, reached through
rrnet_amd.datasets.synthetic.collate_ctnet_device / the ToHeatmap transform."""


def flip_img(data):
    """datasets/transforms/functional.py:13-19: horizontal flip of a [C,H,W] image tensor."""
    return data.flip(dims=(2,))


def flip_annos(data, w):
    """datasets/transforms/functional.py:22-29: x -> w - x - width for xywh rows (in place, like the reference)."""
    data[:, 0] = w - data[:, 0] - data[:, 2]
    return data
