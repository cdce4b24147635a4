"""ResNeXt29 8x64d for 1x32x32 mel spectrograms (reference models/resnext.py:23-142: `CifarResNeXt`, the default
classifier of certified_robustness_eval.py:57) as a HIP-backed module.  The parameter containers keep the reference's
names (`conv_1_3x3`, `bn_1`, `stage_<s>.stage_<s>_bottleneck_<k>.{conv_reduce,bn_reduce,conv_conv,bn,conv_expand,
bn_expand,shortcut.shortcut_conv,shortcut.shortcut_bn}`, `classifier`) so state dicts and pickled checkpoints
(`models.resnext.CifarResNeXt` inside a DataParallel) load unchanged; the forward pass runs in libdmad_hip.so:
1x1 convs as plain GEMMs, the cardinality-8 3x3 conv as 8 gather-GEMMs in one launch (one group per grid.z), the
bottleneck sum + ReLU in the expand GEMM's epilogue, all on the fp32 matrix cores over NHWC maps with eval-mode
BatchNorm folded into per-channel scale/shift."""
import torch
import torch.nn as nn

__all__ = ['CifarResNeXt', 'ResNeXtBottleneck']


class ResNeXtBottleneck(nn.Module):
    """Parameter container of one type-C bottleneck (reference l.23-62)."""

    def __init__(self, in_channels, out_channels, stride, cardinality, base_width, widen_factor):
        super().__init__()
        width_ratio = out_channels / (widen_factor * 64.)
        D = cardinality * int(base_width * width_ratio)
        self.conv_reduce = nn.Conv2d(in_channels, D, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn_reduce = nn.BatchNorm2d(D)
        self.conv_conv = nn.Conv2d(D, D, kernel_size=3, stride=stride, padding=1, groups=cardinality, bias=False)
        self.bn = nn.BatchNorm2d(D)
        self.conv_expand = nn.Conv2d(D, out_channels, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn_expand = nn.BatchNorm2d(out_channels)
        self.shortcut = nn.Sequential()
        if in_channels != out_channels:
            self.shortcut.add_module('shortcut_conv', nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride,
                                                                padding=0, bias=False))
            self.shortcut.add_module('shortcut_bn', nn.BatchNorm2d(out_channels))

    def forward(self, x):
        """reference l.47-62 — used on the gradient branch of CifarResNeXt.forward only (inference runs in the HIP engine)."""
        b = torch.relu(self.bn_reduce(self.conv_reduce(x)))
        b = torch.relu(self.bn(self.conv_conv(b)))
        b = self.bn_expand(self.conv_expand(b))
        return torch.relu(self.shortcut(x) + b)


class CifarResNeXt(nn.Module):

    def __init__(self, nlabels, cardinality=8, depth=29, base_width=64, widen_factor=4, in_channels=3):
        super().__init__()
        if (cardinality, depth, base_width, widen_factor) != (8, 29, 64, 4):
            raise NotImplementedError('the HIP engine builds ResNeXt29 8x64d (cardinality 8, depth 29, base width 64, widen 4)')
        self.cardinality, self.depth, self.base_width, self.widen_factor = cardinality, depth, base_width, widen_factor
        self.block_depth = (depth - 2) // 9
        self.nlabels = nlabels
        self.output_size = 64
        self.stages = [64, 64 * widen_factor, 128 * widen_factor, 256 * widen_factor]
        self.conv_1_3x3 = nn.Conv2d(in_channels, 64, 3, 1, 1, bias=False)
        self.bn_1 = nn.BatchNorm2d(64)
        self.stage_1 = self.block('stage_1', self.stages[0], self.stages[1], 1)
        self.stage_2 = self.block('stage_2', self.stages[1], self.stages[2], 2)
        self.stage_3 = self.block('stage_3', self.stages[2], self.stages[3], 2)
        self.classifier = nn.Linear(self.stages[3], nlabels)
        nn.init.kaiming_normal_(self.classifier.weight)
        for key, v in self.state_dict().items():          # reference l.105-112
            if key.split('.')[-1] == 'weight':
                if 'conv' in key:
                    nn.init.kaiming_normal_(v, mode='fan_out')
                if 'bn' in key:
                    v[...] = 1
            elif key.split('.')[-1] == 'bias':
                v[...] = 0

    def block(self, name, in_channels, out_channels, pool_stride=2):
        block = nn.Sequential()
        for bottleneck in range(self.block_depth):
            name_ = '%s_bottleneck_%d' % (name, bottleneck)
            if bottleneck == 0:
                block.add_module(name_, ResNeXtBottleneck(in_channels, out_channels, pool_stride, self.cardinality,
                                                          self.base_width, self.widen_factor))
            else:
                block.add_module(name_, ResNeXtBottleneck(out_channels, out_channels, 1, self.cardinality, self.base_width,
                                                          self.widen_factor))
        return block

    # -- HIP engine binding ---------------------------------------------------------------------
    def bind_engine(self, engine=None):
        """Fold BatchNorm (eval statistics) and upload the weights into the engine (once).  An explicit `engine` that
        already holds ANOTHER classifier is refused (DmadError); the shared engine in that case is left alone and this
        module gets an engine of its own."""
        from dmad_hip import engine as _eng
        if self.conv_1_3x3.in_channels != 1:
            raise NotImplementedError('the HIP ResNeXt29 takes the 1x32x32 mel spectrogram (in_channels = 1)')
        eng = _eng.bind_classifier(self.state_dict(), 'load_resnext29', engine)
        self.__dict__['engine'] = eng
        return self

    def forward(self, x):
        if self.training:
            raise NotImplementedError('the HIP ResNeXt29 is inference-only: call .eval() first')
        if torch.is_grad_enabled() and x.requires_grad:
            # callers that differentiate through the system: the module's own layers, reference l.133-142; CUDA only
            if not x.is_cuda:
                raise NotImplementedError('the ResNeXt29 mirror has no CPU path (gradient branch included)')
            h = torch.relu(self.bn_1(self.conv_1_3x3(x)))
            h = self.stage_3(self.stage_2(self.stage_1(h)))
            h = torch.nn.functional.avg_pool2d(h, 8, 1).view(-1, self.stages[3])
            return self.classifier(h)
        if 'engine' not in self.__dict__:
            self.bind_engine()
        return self.__dict__['engine'].classify(x)
