"""oracle/thirdparty.py restates two un-vendored dependencies from their published structure.
For torchvision there ARE independent published facts to hold the restatement to -- the model
table of the torchvision documentation (parameter counts, multiply-accumulates of one 224x224
frame) and the state_dict layout every published ResNet checkpoint has -- so the key counts in
test_oracle_golden.py are no longer the only (circular) check of that layer.  habitat-lab's
GroupNorm ResNet has no such table; it stays "parity unpinned" (oracle/thirdparty.py header)."""
import pytest
import torch
from torch.utils.flop_counter import FlopCounterMode

from oracle import thirdparty as tp

# torchvision.models documentation, "Table of all available classification weights":
#   ResNet18  11,689,512 parameters, 1.81 GFLOPS;  ResNet50  25,557,032 parameters, 4.09 GFLOPS
# (their "GFLOPS" column counts multiply-accumulates of one 3x224x224 frame)
PUBLISHED = {
    "tv_resnet18": dict(params=11_689_512, gmacs=1.81, entries=122, fc_in=512),
    "tv_resnet50": dict(params=25_557_032, gmacs=4.09, entries=320, fc_in=2048),
}


@pytest.mark.parametrize("name", sorted(PUBLISHED))
def test_restated_torchvision_resnet_has_the_published_size_and_work(name):
    want = PUBLISHED[name]
    net = getattr(tp, name)().eval()
    assert sum(p.numel() for p in net.parameters()) == want["params"]
    sd = net.state_dict()
    assert len(sd) == want["entries"]
    assert tuple(sd["fc.weight"].shape) == (1000, want["fc_in"])
    assert tuple(sd["conv1.weight"].shape) == (64, 3, 7, 7)
    # checkpoint layout: stem, then layer{1..4}.{block}.{conv,bn}{k} (+ downsample.{0,1})
    keys = list(sd)
    assert keys[:6] == ["conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean",
                        "bn1.running_var", "bn1.num_batches_tracked"]
    assert "layer2.0.downsample.0.weight" in sd and "layer2.0.downsample.1.running_var" in sd
    assert "layer1.1.downsample.0.weight" not in sd
    with FlopCounterMode(display=False) as counter, torch.no_grad():
        out = net(torch.zeros(1, 3, 224, 224))
    assert tuple(out.shape) == (1, 1000)
    assert round(counter.get_total_flops() / 2 / 1e9, 2) == want["gmacs"]


def test_bottleneck_stride_sits_on_the_3x3():
    """torchvision's ResNet-50 ("v1.5") downsamples in the 3x3 convolution of a stage's first
    block, not in its first 1x1 (the original paper's placement): same parameter count, different
    function -- the count test above cannot see it."""
    blk = tp.tv_resnet50().layer2[0]
    assert blk.conv1.stride == (1, 1) and blk.conv2.stride == (2, 2) and blk.conv3.stride == (1, 1)
    assert blk.downsample[0].stride == (2, 2)
