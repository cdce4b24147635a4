"""Model zoo entry kept under the reference's module path (models/__init__.py:17-46).  Only the
classifier of the benchmarked path (vgg19_bn, the `else` default of the reference's table) is built
natively; the other architectures of the reference zoo are out of scope."""
from .vgg import VGG, vgg19_bn  # noqa: F401

available_models = ['vgg19_bn']


def create_model(model_name, num_classes, in_channels):
    if model_name != 'vgg19_bn':
        raise NotImplementedError('%s is not built natively for MI355X yet (only vgg19_bn)' % model_name)
    return vgg19_bn(num_classes=num_classes, in_channels=in_channels)
