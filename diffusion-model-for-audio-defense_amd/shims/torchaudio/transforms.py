import torch

from dmad_hip import engine as _eng


class MelSpectrogram(torch.nn.Module):
    """torchaudio.transforms.MelSpectrogram for the one configuration of the reference
    (n_fft=2048, hop_length=512, n_mels=32, norm='slaney', pad_mode='constant', mel_scale='slaney',
    sample_rate=16000, power=2): [B,1,16000] -> power mel spectrogram [B,1,32,32]."""
    _dmad_stage = 'mel_power'

    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None, pad=0,
                 n_mels=128, window_fn=torch.hann_window, power=2.0, normalized=False, wkwargs=None, center=True,
                 pad_mode='reflect', onesided=True, norm=None, mel_scale='htk'):
        super().__init__()
        got = dict(sample_rate=sample_rate, n_fft=n_fft, win_length=win_length or n_fft, hop_length=hop_length, f_min=f_min,
                   f_max=f_max or sample_rate / 2, pad=pad, n_mels=n_mels, power=power, normalized=normalized, center=center,
                   pad_mode=pad_mode, norm=norm, mel_scale=mel_scale)
        want = dict(sample_rate=16000, n_fft=2048, win_length=2048, hop_length=512, f_min=0.0, f_max=8000.0, pad=0, n_mels=32,
                    power=2.0, normalized=False, center=True, pad_mode='constant', norm='slaney', mel_scale='slaney')
        if got != want or window_fn is not torch.hann_window:
            diff = {k: (got[k], want[k]) for k in want if got[k] != want[k]}
            raise NotImplementedError('the HIP mel front-end is built for the reference configuration only; differs in %s' % diff)
        self.n_mels = n_mels

    @torch.no_grad()
    def forward(self, waveform):
        return _eng.get_engine().mel_power(waveform)


class AmplitudeToDB(torch.nn.Module):
    """torchaudio.transforms.AmplitudeToDB(stype='power', top_db=None): 10 * log10(clamp(x, 1e-10))."""
    _dmad_stage = 'power_to_db'

    def __init__(self, stype='power', top_db=None):
        super().__init__()
        if stype != 'power' or top_db is not None:
            raise NotImplementedError("only AmplitudeToDB(stype='power', top_db=None) is provided")

    @torch.no_grad()
    def forward(self, x):
        return _eng.get_engine().power_to_db(x)
