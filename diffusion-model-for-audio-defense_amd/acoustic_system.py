"""Host mirror of the reference's acoustic_system.AcousticSystem (acoustic_system.py:3-51):
defender (waveform purifier) -> transform (waveform -> spectrogram) -> classifier, with the same
constructor, attributes and error behaviour.  The stages themselves are the HIP-backed modules of this
package (DiffWave, MelSpectrogramDB, VGG) or any callables the caller passes.

`query(x, repeats)` is the batched entry of the gradient-free attack drivers (EOT / NES): every clip evaluated
`repeats` times with fresh purification noise.  When the three stages are this package's HIP stages on one engine it is
ONE C-ABI call — dmad_query_logits for defense_type 'wave' (repeat -> DDPM purify -> mel dB -> classifier -> arg-max),
dmad_spec_query_logits for 'spec' (repeat -> mel dB -> spec-domain purifier -> classifier -> arg-max); otherwise it loops over
forward()."""
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

    @staticmethod
    def _rescale(x):
        # int16-range input is rescaled to [-1, 1] (acoustic_system.py:29-30)
        if 0.9 * x.max() > 1 and 0.9 * x.min() < -1:
            x = x / (2 ** 15)
        return x

    def forward(self, x, defend=True):
        x = self._rescale(x)
        use_defender = defend == True and self.defender is not None   # noqa: E712 (reference semantics)
        output = self.defender(x) if (use_defender and self.defense_type == 'wave') else x
        if self.transform is not None:
            output = self.transform(output)
        if use_defender and self.defense_type == 'spec':
            output = self.defender(output)
        return self.classifier(output)

    # ------------------------------------------------------------------------------------------------------------
    def _engine_chain(self, defend):
        """The engine that can run this system as one dmad_query_logits call, with the sampler id, or (None, 0)."""
        from diffusion_models.diffwave_ddpm import DiffWave
        from dmad_hip.transforms import MelSpectrogramDB
        cls, tr, den = self.classifier, self.transform, self.defender
        eng = getattr(cls, 'engine', None) if 'engine' in getattr(cls, '__dict__', {}) else None
        if eng is None or not eng.has_classifier:
            return None, 0
        mel = isinstance(tr, MelSpectrogramDB) and tr.engine is eng
        if not mel:
            mel = [getattr(t, '_dmad_stage', None) for t in getattr(tr, 'transforms', [])] == ['mel_power', 'power_to_db']
        if not mel:
            return None, 0
        if not (defend == True and den is not None):                  # noqa: E712
            return eng, 0
        if self.defense_type == 'spec':                               # sampler 3: the spec-domain chain (dmad_spec_query_logits)
            from diffusion_models.improved_diffusion_ddpm import SpecPurifier
            return (eng, 3) if (type(den) is SpecPurifier and den.engine is eng) else (None, 0)
        if type(den) is DiffWave and den.noise_source == 'device' and den.engine is eng and eng.has_wavenet:
            return eng, 1
        return None, 0

    @torch.no_grad()
    def query(self, x, repeats: int, defend=True, per_call: int = None):
        """x [B,1,L] -> (logits [repeats, B, C], decisions int64 [repeats, B]); repeat r of clip b is row (r, b), the
        order of `model(x.repeat(repeats, 1, 1))` (robustness_eval/_EOT.py:36-40).  Without a one-engine chain (other
        stages, or a purifier on the reference's CPU noise stream) the system is called the way the reference's EOT calls
        it: `repeats / per_call` forward passes over x.repeat(per_call, 1, 1), which also consumes the CPU generator in
        the reference's order."""
        x = self._rescale(x)
        B = x.shape[0]
        eng, sampler = self._engine_chain(defend)
        if eng is not None and x.is_cuda:
            if sampler == 1:
                den = self.defender
                ts, c_a, c_b, c_eps, c_div, c_sig = den.purify_coefficients()
                logits, dec = eng.query_logits(x, repeats, 1, ts, c_a, c_b, c_eps, c_div, c_sig, seed=den.seed, sample0=den._draws)
                den._draws += repeats * B
            elif sampler == 3:
                from diffusion_models.Improved_Diffusion_Unconditional.improved_diffusion.sc09_spectrogram_dataset import MEL_LOWER_BOUND, MEL_UPPER_BOUND
                den = self.defender
                logits, dec = eng.spec_query_logits(x, repeats, *den.purifier.purify_coefficients(), MEL_LOWER_BOUND, MEL_UPPER_BOUND,
                                                    seed=den.seed, sample0=den._draws)
                den._draws += repeats * B
            else:
                logits, dec = eng.query_logits(x, repeats, 0)
            return logits.view(repeats, B, -1), dec.view(repeats, B).long()
        per_call = per_call or repeats
        assert repeats % per_call == 0
        logits = torch.cat([self.forward(x.repeat(per_call, 1, 1), defend).view(per_call, B, -1) for _ in range(repeats // per_call)])
        return logits, logits.argmax(-1)
