"""LoadAudio / FixAudioLength with the semantics of the reference's transforms/transforms_wav.py:16-48,
without librosa: a RIFF/WAVE reader on the standard library (`wave`) + numpy.

`librosa.load(path, sr=16000)` returns mono float32 in [-1, 1): PCM16 samples / 32768 (soundfile's scaling),
channels averaged, resampled if the file's rate differs.  SC09 is 16 kHz mono PCM16, so the loader accepts
PCM 8/16/24/32-bit files of any channel count at the requested rate and refuses other rates instead of silently
using a different resampler than the reference's (soxr_hq)."""
import wave

import numpy as np

__all__ = ['LoadAudio', 'FixAudioLength', 'read_wav']


def read_wav(path):
    """-> (float32 mono samples in [-1, 1), sample_rate)"""
    with wave.open(path, 'rb') as w:
        nch, width, rate, nframes = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(nframes)
    if width == 2:
        x = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
    elif width == 1:                                    # unsigned 8-bit
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = v.astype(np.float32) / 8388608.0
    elif width == 4:
        x = (np.frombuffer(raw, dtype='<i4').astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        raise ValueError('unsupported PCM sample width %d in %s' % (width, path))
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1).astype(np.float32)
    return x, rate


class LoadAudio(object):
    """Loads an audio into a numpy array (data['path'] -> data['samples'], data['sample_rate'])."""

    def __init__(self, sample_rate=16000):
        self.sample_rate = sample_rate

    def __call__(self, data):
        path = data['path']
        if path:
            samples, sample_rate = read_wav(path)
            if sample_rate != self.sample_rate:
                raise ValueError('%s is sampled at %d Hz, expected %d Hz (resampling is not provided)'
                                 % (path, sample_rate, self.sample_rate))
        else:
            # silence
            sample_rate = self.sample_rate
            samples = np.zeros(sample_rate, dtype=np.float32)
        data['samples'] = samples
        data['sample_rate'] = sample_rate
        return data


class FixAudioLength(object):
    """Either pads or truncates an audio into a fixed length."""

    def __init__(self, time=1):
        self.time = time

    def __call__(self, data):
        samples = data['samples']
        sample_rate = data['sample_rate']
        length = int(self.time * sample_rate)
        if length < len(samples):
            data['samples'] = samples[:length]
        elif length > len(samples):
            data['samples'] = np.pad(samples, (0, length - len(samples)), "constant")
        return data
