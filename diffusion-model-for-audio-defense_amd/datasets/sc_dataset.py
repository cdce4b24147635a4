"""SC09 evaluation index used by the certification driver (contract of the reference's datasets/sc_dataset.py:90-134,
consumer certified_robustness_eval.py:99-106): `SC09Dataset(folder, transform, classes, num_per_class)` lists, for every
digit sub-folder of `folder`, its first `num_per_class` directory entries in `os.listdir` order; item i is
`transform({'path': ..., 'target': class index})`.  Every class but the last two must be present.
`make_weights_for_balanced_classes()` (reference :136-149; its training scripts feed it to a WeightedRandomSampler,
audio_models/M5/train.py:45) returns one float64 weight per item, N / (items of the item's class)."""
import os

import numpy as np
from torch.utils.data import Dataset

__all__ = ['CLASSES', 'SC09_CLASSES', 'SC09Dataset']

CLASSES = 'unknown, silence, yes, no, up, down, left, right, on, off, stop, go'.split(', ')
SC09_CLASSES = 'zero, one, two, three, four, five, six, seven, eight, nine'.split(', ')


class SC09Dataset(Dataset):

    def __init__(self, folder, transform=None, classes=SC09_CLASSES, num_per_class=100):
        present = [c for c in classes if not c.startswith('_') and os.path.isdir(os.path.join(folder, c))]
        missing = [c for c in classes[:-2] if c not in present]
        assert not missing, 'class folders missing under %s: %s' % (folder, missing)
        self.classes = classes
        self.transform = transform
        self.data = [(os.path.join(folder, c, name), classes.index(c))
                     for c in present for name in os.listdir(os.path.join(folder, c))[:num_per_class]]

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        path, target = self.data[index]
        item = {'path': path, 'target': target}
        return item if self.transform is None else self.transform(item)

    def make_weights_for_balanced_classes(self):
        targets = np.fromiter((t for _, t in self.data), dtype=np.int64, count=len(self.data))
        per_class = np.bincount(targets, minlength=len(self.classes)).astype(np.float64)
        with np.errstate(divide='ignore'):          # an absent class gets an infinite weight nobody is assigned (as in the reference)
            return (float(len(self.data)) / per_class)[targets]
