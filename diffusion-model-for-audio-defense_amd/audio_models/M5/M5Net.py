"""M5 raw-waveform classifier (reference audio_models/M5/M5Net.py:4-38), needed so that the bundled
whole-module M5 pickles (`M5Net.M5`) can be unpickled by create_model().  27.8 k parameters: it is
not on the MFMA-bound part of the path and runs as ordinary torch ops on whatever device it is on
(SURVEY section 2, row 5); only the purification in front of it and the vote count are HIP."""
import torch.nn as nn
import torch.nn.functional as F


class M5(nn.Module):
    # attribute names (conv1..4, bn1..4, pool1..4, fc1) are fixed by the pickled checkpoints
    def __init__(self, n_input=1, first_kernel_size=80, n_output=35, stride=16, n_channel=32):
        super().__init__()
        widths = [(n_input, n_channel, first_kernel_size, stride), (n_channel, n_channel, 3, 1),
                  (n_channel, 2 * n_channel, 3, 1), (2 * n_channel, 2 * n_channel, 3, 1)]
        for i, (cin, cout, k, s) in enumerate(widths, start=1):
            setattr(self, 'conv%d' % i, nn.Conv1d(cin, cout, kernel_size=k, stride=s))
            setattr(self, 'bn%d' % i, nn.BatchNorm1d(cout))
            setattr(self, 'pool%d' % i, nn.MaxPool1d(4))
        self.fc1 = nn.Linear(2 * n_channel, n_output)

    def forward(self, x):
        for i in (1, 2, 3, 4):
            x = getattr(self, 'conv%d' % i)(x)
            x = getattr(self, 'pool%d' % i)(F.relu(getattr(self, 'bn%d' % i)(x)))
        x = F.avg_pool1d(x, x.shape[-1]).flatten(1)
        return F.log_softmax(self.fc1(x), dim=1)
