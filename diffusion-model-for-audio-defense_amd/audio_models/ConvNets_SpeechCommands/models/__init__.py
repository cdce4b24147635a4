"""Model zoo entry kept under the reference's module path (models/__init__.py:17-46).  Built natively: the
classifier of the benchmarked path (vgg19_bn, the `else` default of the reference's table) and the certification
script's default (resnext29_8_64); the other architectures of the reference zoo are out of scope."""
from .resnext import CifarResNeXt  # noqa: F401
from .vgg import VGG, vgg19_bn  # noqa: F401

available_models = ['vgg19_bn', 'resnext29_8_64']


def create_model(model_name, num_classes, in_channels):
    if model_name == 'resnext29_8_64':
        return CifarResNeXt(nlabels=num_classes, in_channels=in_channels)
    if model_name != 'vgg19_bn':
        raise NotImplementedError('%s is not built natively for MI355X yet (vgg19_bn, resnext29_8_64)' % model_name)
    return vgg19_bn(num_classes=num_classes, in_channels=in_channels)
