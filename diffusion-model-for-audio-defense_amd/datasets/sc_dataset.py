"""SC09 test-set index of the certification driver (reference datasets/sc_dataset.py:90-134): one sub-folder per
digit, the first `num_per_class` directory entries of each (in `os.listdir` order, as the reference takes them),
items `{'path', 'target'}` passed through `transform`."""
import os

import numpy as np
from torch.utils.data import Dataset

__all__ = ['CLASSES', 'SC09_CLASSES', 'SC09Dataset']

CLASSES = 'unknown, silence, yes, no, up, down, left, right, on, off, stop, go'.split(', ')
SC09_CLASSES = 'zero, one, two, three, four, five, six, seven, eight, nine'.split(', ')


class SC09Dataset(Dataset):

    def __init__(self, folder, transform=None, classes=SC09_CLASSES, num_per_class=100):
        all_classes = [d for d in classes if os.path.isdir(os.path.join(folder, d)) and not d.startswith('_')]
        for c in classes[:-2]:
            assert c in all_classes
        class_to_idx = {classes[i]: i for i in range(len(classes))}
        for c in all_classes:
            if c not in class_to_idx:
                class_to_idx[c] = len(classes) - 1
        data = []
        for c in all_classes:
            d = os.path.join(folder, c)
            target = class_to_idx[c]
            entries = os.listdir(d)
            for f in entries[:min(num_per_class, len(entries))]:
                data.append((os.path.join(d, f), target))
        self.classes = classes
        self.data = data
        self.transform = transform

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        path, target = self.data[index]
        data = {'path': path, 'target': target}
        if self.transform is not None:
            data = self.transform(data)
        return data

    def make_weights_for_balanced_classes(self):
        nclasses = len(self.classes)
        count = np.zeros(nclasses)
        for item in self.data:
            count[item[1]] += 1
        N = float(sum(count))
        weight_per_class = N / count
        weight = np.zeros(len(self))
        for idx, item in enumerate(self.data):
            weight[idx] = weight_per_class[item[1]]
        return weight
