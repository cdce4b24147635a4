__all__ = ['Compose']


class Compose:
    """torchvision.transforms.Compose: apply the callables in order."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x

    def __repr__(self):
        return 'Compose(%s)' % ', '.join(repr(t) for t in self.transforms)
