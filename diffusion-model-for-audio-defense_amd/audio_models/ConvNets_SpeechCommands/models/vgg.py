"""VGG19_bn for 1x32x32 mel spectrograms (reference models/vgg.py:31-52,69-89,190-201) as a
HIP-backed module.  The parameter containers (`features`, `classifier`) keep the reference's layout so
that state dicts and pickled checkpoints (`models.vgg.VGG` inside a DataParallel) load unchanged; the
forward pass runs in libdmad_hip.so: 3x3 convs as implicit GEMM on the fp32 matrix cores over NHWC
activations with eval-mode BatchNorm folded into a per-channel scale/shift, 2x2 max-pools, and the
three Linear layers."""
import math

import torch
import torch.nn as nn

_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def _features(in_channels):
    mods, c = [], in_channels
    for v in _CFG_E:
        if v == 'M':
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            mods += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            c = v
    return nn.Sequential(*mods)


class VGG(nn.Module):

    def __init__(self, features, num_classes=1000, init_weights=True):
        super().__init__()
        self.features = features
        self.classifier = nn.Sequential(nn.Linear(512, 4096), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(4096, num_classes))
        if init_weights:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
                    m.bias.data.zero_()
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.data.fill_(1); m.bias.data.zero_()
                elif isinstance(m, nn.Linear):
                    m.weight.data.normal_(0, 0.01); m.bias.data.zero_()

    # -- HIP engine binding ---------------------------------------------------------------------
    def bind_engine(self, engine=None):
        """Fold BatchNorm (eval statistics) and upload the weights into the engine (once).  An explicit `engine` that
        already holds ANOTHER classifier is refused (DmadError); the shared engine in that case is left alone and this
        module gets an engine of its own."""
        from dmad_hip import engine as _eng
        eng = _eng.bind_classifier(self.state_dict(), 'load_vgg19_bn', engine)
        self.__dict__['engine'] = eng
        return self

    def forward(self, x):
        if self.training:
            raise NotImplementedError('the HIP VGG19_bn is inference-only: call .eval() first')
        if torch.is_grad_enabled() and x.requires_grad:
            # callers that differentiate through the system: the module's own layers are the torch restatement of
            # reference models/vgg.py:48-52 (eval mode: BatchNorm on its running statistics, Dropout inactive); CUDA tensors only
            if not x.is_cuda:
                raise NotImplementedError('the VGG19_bn mirror has no CPU path (gradient branch included)')
            return self.classifier(self.features(x).view(x.size(0), -1))
        if 'engine' not in self.__dict__:
            self.bind_engine()
        return self.__dict__['engine'].classify(x)


def vgg19_bn(pretrained=False, in_channels=3, **kwargs):
    if pretrained:
        raise NotImplementedError('no network access: ImageNet weights cannot be fetched')
    return VGG(_features(in_channels), **kwargs)
