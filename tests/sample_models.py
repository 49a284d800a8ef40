"""The three sample architectures of the reference (quant/Weight_Only/models.py) rebuilt for
the tests: same layer names, shapes and construction order, so `torch.manual_seed(0)` followed
by construction yields the weights whose digests tests/golden/config1_models.npz records."""
from collections import OrderedDict

import torch
import torch.nn as nn


class LinearNet(nn.Module):
    def __init__(self, input_size=3072, num_classes=10):
        super().__init__()
        widths = [input_size, 2048, 1024, 64, num_classes]
        for i in range(4):
            setattr(self, f"linear{i + 1}", nn.Linear(widths[i], widths[i + 1]))
        self.relu = nn.ReLU()

    def forward(self, x):
        x = x.view(-1, 3 * 32 * 32)
        for i in range(1, 4):
            x = self.relu(getattr(self, f"linear{i}")(x))
        return self.linear4(x)


class VGG(nn.Module):
    PLAN = [64, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"]

    def __init__(self):
        super().__init__()
        layers, n, cin = [], {"conv": 0, "bn": 0, "relu": 0, "pool": 0}, 3

        def add(kind, mod):
            layers.append((f"{kind}{n[kind]}", mod))
            n[kind] += 1
        for width in self.PLAN:
            if width == "M":
                add("pool", nn.MaxPool2d(2))
            else:
                add("conv", nn.Conv2d(cin, width, 3, padding=1, bias=False))
                add("bn", nn.BatchNorm2d(width))
                add("relu", nn.ReLU(True))
                cin = width
        self.backbone = nn.Sequential(OrderedDict(layers))
        self.classifier = nn.Linear(512, 10)

    def forward(self, x):
        return self.classifier(self.backbone(x).mean([2, 3]))


class SimpleCNN(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 32, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=3, padding=1)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.fc1 = nn.Linear(64 * 8 * 8, 512)
        self.fc2 = nn.Linear(512, 10)

    def forward(self, x):
        x = self.pool(torch.relu(self.conv1(x)))
        x = self.pool(torch.relu(self.conv2(x)))
        x = torch.relu(self.fc1(x.view(-1, 64 * 8 * 8)))
        return self.fc2(x)


def build(name):
    torch.manual_seed(0)
    return {"SimpleCNN": SimpleCNN, "LinearNet": lambda: LinearNet(3072, 10), "VGG": VGG}[name]().eval()


def loader(batches, bs, seed):
    """The synthetic calibration loader oracle/gen_golden.py used."""
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(bs, 3, 32, 32, generator=g), torch.randint(0, 10, (bs,), generator=g))
            for _ in range(batches)]
