"""VGGSound ResNet-18 spectrogram encoder, HIP-backed (reference: backbones/resnet.py:17-154).

Same module tree / state-dict keys as the reference (`conv1, bn1, layer{1..4}.{0,1}.{conv1,bn1,
conv2,bn2,downsample.{0,1}}`); returns the layer4 map.  All convs run as implicit GEMMs with the
BatchNorm folded in and the residual add + ReLU in the epilogue."""
import torch
import torch.nn as nn

from .. import engine as E
from ..module import HipModule


class BasicBlock(HipModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def _pack(self):
        s = self.stride
        pk = {"c1": E.pack_conv(self.conv1.weight, None, self.bn1, (1, s, s), (0, 1, 1), E.ACT_RELU),
              "c2": E.pack_conv(self.conv2.weight, None, self.bn2, (1, 1, 1), (0, 1, 1), E.ACT_RELU)}
        if self.downsample is not None:
            pk["ds"] = E.pack_conv(self.downsample[0].weight, None, self.downsample[1], (1, s, s), (0, 0, 0), E.ACT_NONE)
        return pk

    def run(self, x):
        pk = self.pk
        idt = E.conv(x, pk["ds"]) if "ds" in pk else x
        return E.conv(E.conv(x, pk["c1"]), pk["c2"], res=idt)


class ResNet(HipModule):
    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(1, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        for m in self.modules():  # backbones/resnet.py:92-97
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.normal_(m.weight, mean=1, std=0.02)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [block(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _pack(self):
        return E.pack_conv(self.conv1.weight, None, self.bn1, (1, 2, 2), (0, 3, 3), E.ACT_RELU)

    @torch.no_grad()
    def forward_cl(self, x):
        """x: [B,1,F,T'] fp32 spectrograms on the GPU -> CL [B,1,F/32,T'/32,512]."""
        self._check_eval()
        y = E.maxpool(E.conv(x, self.pk), (1, 3, 3), (1, 2, 2), (0, 1, 1))
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                y = blk.run(y)
        return y

    def forward(self, x):
        return self.forward_cl(x).as_ncdhw().squeeze(2)


def get_resnet18(block=BasicBlock, layers=(2, 2, 2, 2), pretrained=True, path=None, **kwargs):
    """backbones/resnet.py:149-154.  With pretrained=True the file must exist (as upstream)."""
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        model.load_state_dict(torch.load(path, map_location="cpu"))
    return model
