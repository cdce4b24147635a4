"""Host mirror of the reference's acoustic_system.AcousticSystem (acoustic_system.py:3-51):
defender (waveform purifier) -> transform (waveform -> spectrogram) -> classifier, with the same
constructor, attributes and error behaviour.  The stages themselves are the HIP-backed modules of this
package (DiffWave, MelSpectrogramDB, VGG) or any callables the caller passes."""
import torch


class AcousticSystem(torch.nn.Module):

    def __init__(self, classifier: torch.nn.Module, transform, defender: torch.nn.Module = None, defense_type: str = 'wave'):
        super().__init__()
        self.classifier = classifier
        self.transform = transform
        self.defender = defender
        self.defense_type = defense_type
        if self.defense_type not in ['wave', 'spec']:
            raise NotImplementedError('argument defense_type should be \'wave\' or \'spec\'!')

    def forward(self, x, defend=True):
        # int16-range input is rescaled to [-1, 1] (acoustic_system.py:29-30)
        if 0.9 * x.max() > 1 and 0.9 * x.min() < -1:
            x = x / (2 ** 15)
        use_defender = defend == True and self.defender is not None   # noqa: E712 (reference semantics)
        output = self.defender(x) if (use_defender and self.defense_type == 'wave') else x
        if self.transform is not None:
            output = self.transform(output)
        if use_defender and self.defense_type == 'spec':
            output = self.defender(output)
        return self.classifier(output)
