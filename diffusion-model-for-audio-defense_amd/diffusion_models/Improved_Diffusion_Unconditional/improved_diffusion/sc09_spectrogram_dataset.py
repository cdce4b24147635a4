"""The mel-dB <-> [-1, 1] maps of the reference's sc09_spectrogram_dataset.py:58-81 (bounds of the SC09 mel data)."""
MEL_UPPER_BOUND = 38.22
MEL_LOWER_BOUND = -100.0


def melspec_standardize(x):
    return 2 * (x - MEL_LOWER_BOUND) / (MEL_UPPER_BOUND - MEL_LOWER_BOUND) - 1


def melspec_inv_standardize(x):
    return (x + 1) * (MEL_UPPER_BOUND - MEL_LOWER_BOUND) / 2 + MEL_LOWER_BOUND
