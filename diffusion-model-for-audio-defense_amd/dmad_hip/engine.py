"""Host-side handle on one dmad_engine (one per process per GPU).

PyTorch is plumbing here: device memory (torch tensors), the current HIP stream and
torch.distributed.  All arithmetic of the hot path runs inside libdmad_hip.so."""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import warnings
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import DmadConfig, DmadError, check

BF16, FP32, EXACT = 0, 1, 2                       # enum dmad_precision
MODE_FAST, MODE_EXACT_VOTES, MODE_FP32 = 0, 1, 2  # enum dmad_mode (EXACT engines)
HALF_BF16, HALF_F16 = 0, 1                        # enum dmad_half_type: operand format of the 16-bit MFMA path
WAVE_16BIT, WAVE_FP32, WAVE_SPLIT = 0, 1, 2       # dmad_set_waveform_tier: WaveNet tier of the waveform-returning surfaces (EXACT engines)
# Recheck bound of the exact-vote mode: a Monte Carlo sample whose 16-bit-path top-2 logit margin is below it is
# re-evaluated on the higher tiers.  Let i be the exact path's arg-max and e = (16-bit logits) - (exact logits).  If the 16-bit
# margin is >= tau and the 16-bit leader were some j != i, then l~_j - l~_i >= tau with l_j - l_i <= 0, i.e. e_j - e_i >= tau:
# so the vote is unchanged whenever tau exceeds E = max_j |e_j - e_i|, the largest error of a logit DIFFERENCE AGAINST THE
# EXACT LEADER.  Measured on 9 x 4096 samples (3 clips x sigma 0.25 / 0.5 / 1.0, tools/gpu_flip_study.py,
# profiles/r02_flip_study.md): f16 operands E = 0.0244 (0.0287 over all pairs i, j; 35 flips, the largest at margin 0.011),
# bf16 operands 0.207 (0.221; 261 flips) -> bounds with ~1.4x headroom.  Overridable: DMAD_RECHECK_MARGIN / recheck_margin=.
DEFAULT_RECHECK_MARGIN = {1: 0.034, 0: 0.30}          # by dmad_half_type: HALF_F16, HALF_BF16
# The error a given eps error turns into is a property of the classifier.  With the calibrated synthetic ResNeXt29 (first pass of the
# exact-vote mode = f16 WaveNet + the classifier's split-f16 tier; tools/gpu_flip_study.py with CLASSIFIER=resnext29 FIRSTPASS=1,
# profiles/r05h_flip_study_resnext29_first_pass.json, 36 864 samples): E = 0.0303, Gaussian scale 0.0065 -> 1.5 x E = 0.045 = 7 scales,
# P(E >= 0.045) = 4.5e-11 per sample.  load_resnext29 widens the bound to this floor.
# (the bf16 entry is the f16 one scaled by the VGG table's ratio 0.30 / 0.034, NOT measured: calibrate before using bf16 operands with it)
DEFAULT_RECHECK_MARGIN_RESNEXT29 = {1: 0.045, 0: 0.40}
# The queued samples first go through the split-f16 tier (fp32 pipeline, three f16 MFMAs per product, ~22 significant bits);
# only those whose margin is inside ITS error bound reach the exact-fp32 path.
DEFAULT_RECHECK_MARGIN2 = 1e-3
# Spec-domain vote loop (BASELINE C5): the UNet's 16-bit tier runs the whole 26-evaluation chain on f16 operands; a sample whose
# top-2 logit margin is below this bound re-runs its chain on a higher tier.  Measured with tools/gpu_c5_flip_study.py at the bench's
# engine batch 2048 on the CALIBRATED synthetic VGG19_bn (profiles/r05c_c5_flip_study.json, 6 144 samples, sigma 0.5, t* 25): the same
# leader-difference statistic as DEFAULT_RECHECK_MARGIN, 0.084 max, Gaussian scale 0.0198 -> 0.13 = 1.5 x max = 6.6 scales:
# P(E >= 0.13) = 4.9e-10 per sample by the Gaussian tail (the pessimistic one here: the generalised-Pareto fit of the top 61 has a
# negative shape and ends below 0.10; profiles/r05c_c5_recheck_tail_fit.txt), 1.5e-12 with the margin condition.  Like every bound of
# this file it is a property of the WEIGHTS: calibrate (calibrate_spec_recheck / RobustCertificate(calibrate=...)) before certifying
# with real checkpoints — round 4's 0.5 belonged to an uncalibrated stand-in whose logits were 50 x larger.
DEFAULT_SPEC_RECHECK_MARGIN = 0.13
# ... and the samples it queues first re-run their chain on the UNet's split-f16 tier (fp32 pipeline, three f16 MFMAs per product); only
# those whose margin is inside THAT tier's error bound reach the exact-fp32 UNet (dmad_set_spec_recheck_margin2; < 0: no middle tier).
DEFAULT_SPEC_RECHECK_MARGIN2 = 5e-4          # 2.2 x the largest leader-difference error (2.3e-4) of the split-f16 chain on 6 144 samples (profiles/r05c_c5_flip_study.json)
# Tail rule shared by the committed default and calibrate_recheck (tools/fit_recheck_tail.py): with s the
# Gaussian scale of the per-sample leader-difference error, a bound of TAIL_Z * s keeps the modelled miss probability per
# sample (error beyond the bound AND an exact margin small enough to be overturned) at or below 1e-9.
TAIL_Z = 5.4
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _as_np(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().float().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def state_fingerprint(sd: Dict[str, object]) -> str:
    """Content hash of a state dict (names, shapes, bytes; BatchNorm's num_batches_tracked counters excepted: they do not
    enter inference): identifies WHICH weights an engine holds."""
    h = hashlib.sha1()
    for k in sorted(sd):
        if k.endswith('num_batches_tracked'):
            continue
        v = sd[k]
        a = v.detach().cpu().contiguous().numpy() if isinstance(v, torch.Tensor) else np.ascontiguousarray(v)
        h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode())
        h.update(memoryview(a.reshape(-1)).cast('B'))
    return h.hexdigest()


def fold_wavenet_state_dict(sd: Dict[str, object], num_res_layers: int) -> Dict[str, np.ndarray]:
    """Reference checkpoint layout -> folded fp32 arrays named as in dmad.h.
    Weight norm is folded with torch._weight_norm, the very op nn.utils.weight_norm evaluates on each
    forward of the reference (WaveNet.py:27-28,66-72)."""
    def T(k):
        v = sd[k]
        return v.detach().cpu().float() if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v, dtype=np.float32))

    def fold(prefix):
        return torch._weight_norm(T(prefix + '.weight_v'), T(prefix + '.weight_g'), 0)

    out = {'init.w': fold('init_conv.0.conv').reshape(256), 'init.b': T('init_conv.0.conv.bias'),
           'fc_t1.w': T('residual_layer.fc_t1.weight'), 'fc_t1.b': T('residual_layer.fc_t1.bias'),
           'fc_t2.w': T('residual_layer.fc_t2.weight'), 'fc_t2.b': T('residual_layer.fc_t2.bias')}
    for n in range(num_res_layers):
        p = 'residual_layer.residual_blocks.%d' % n
        out['fc_t.%d.w' % n] = T(p + '.fc_t.weight'); out['fc_t.%d.b' % n] = T(p + '.fc_t.bias')
        out['dil.%d.w' % n] = fold(p + '.dilated_conv_layer.conv'); out['dil.%d.b' % n] = T(p + '.dilated_conv_layer.conv.bias')
        out['res.%d.w' % n] = fold(p + '.res_conv').reshape(256, 256); out['res.%d.b' % n] = T(p + '.res_conv.bias')
        out['skip.%d.w' % n] = fold(p + '.skip_conv').reshape(256, 256); out['skip.%d.b' % n] = T(p + '.skip_conv.bias')
    out['f0.w'] = fold('final_conv.0.conv').reshape(256, 256); out['f0.b'] = T('final_conv.0.conv.bias')
    out['f2.w'] = T('final_conv.2.conv.weight').reshape(256); out['f2.b'] = T('final_conv.2.conv.bias').reshape(1)
    return {k: _as_np(v) for k, v in out.items()}


def fold_vgg19_bn_state_dict(sd: Dict[str, object], eps: float = 1e-5) -> Dict[str, np.ndarray]:
    """models/vgg.py vgg19_bn state dict -> conv weights + eval-mode BatchNorm folded to scale/shift
    (float64 on the host, rounded once):  y = scale * conv(x) + shift,
    scale = gamma / sqrt(var + eps), shift = (bias - mean) * scale + beta."""
    def A(k):
        v = sd[k]
        return (v.detach().cpu().double().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float64))
    out, idx, li = {}, 0, 0
    for v in VGG19_CFG:
        if v == 'M':
            idx += 1
            continue
        b = idx + 1
        scale = A('features.%d.weight' % b) / np.sqrt(A('features.%d.running_var' % b) + eps)
        shift = (A('features.%d.bias' % idx) - A('features.%d.running_mean' % b)) * scale + A('features.%d.bias' % b)
        out['vgg.conv%d.w' % li] = A('features.%d.weight' % idx)
        out['vgg.conv%d.scale' % li] = scale
        out['vgg.conv%d.shift' % li] = shift
        idx += 3
        li += 1
    for j, i in enumerate((0, 3, 6)):
        out['vgg.fc%d.w' % j] = A('classifier.%d.weight' % i)
        out['vgg.fc%d.b' % j] = A('classifier.%d.bias' % i)
    return {k: _as_np(v) for k, v in out.items()}


def fold_resnext29_state_dict(sd: Dict[str, object], eps: float = 1e-5) -> Dict[str, np.ndarray]:
    """models/resnext.py CifarResNeXt (8x64d, depth 29) state dict -> conv weights + eval-mode BatchNorm folded to
    scale/shift per conv (float64 on the host, rounded once); bottleneck i = 3 * (stage - 1) + k."""
    def A(k):
        v = sd[k]
        return (v.detach().cpu().double().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float64))

    def bn(prefix):
        scale = A(prefix + '.weight') / np.sqrt(A(prefix + '.running_var') + eps)
        return scale, A(prefix + '.bias') - A(prefix + '.running_mean') * scale

    out = {'rx.conv1.w': A('conv_1_3x3.weight')}
    out['rx.conv1.scale'], out['rx.conv1.shift'] = bn('bn_1')
    for st in (1, 2, 3):
        for k in range(3):
            src = 'stage_%d.stage_%d_bottleneck_%d.' % (st, st, k)
            dst = 'rx.b%d.' % (3 * (st - 1) + k)
            for conv, norm, name in (('conv_reduce', 'bn_reduce', 'reduce'), ('conv_conv', 'bn', 'conv'), ('conv_expand', 'bn_expand', 'expand')):
                w = A(src + conv + '.weight')
                out[dst + name + '.w'] = w.reshape(w.shape[0], w.shape[1]) if name != 'conv' else w
                out[dst + name + '.scale'], out[dst + name + '.shift'] = bn(src + norm)
            if src + 'shortcut.shortcut_conv.weight' in sd:
                w = A(src + 'shortcut.shortcut_conv.weight')
                out[dst + 'short.w'] = w.reshape(w.shape[0], w.shape[1])
                out[dst + 'short.scale'], out[dst + 'short.shift'] = bn(src + 'shortcut.shortcut_bn')
    out['rx.fc.w'] = A('classifier.weight')
    out['rx.fc.b'] = A('classifier.bias')
    return {k: _as_np(v) for k, v in out.items()}


class Engine:
    """One libdmad_hip engine bound to the current CUDA(HIP) device."""

    @staticmethod
    def geometry(wavenet_config: Optional[dict]) -> tuple:
        wc = dict(res_channels=256, skip_channels=256, num_res_layers=36, dilation_cycle=12,
                  diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512, diffusion_step_embed_dim_out=512)
        wc.update({k: v for k, v in (wavenet_config or {}).items() if k in wc})
        return tuple(sorted(wc.items()))

    def __init__(self, wavenet_config: Optional[dict] = None, clip_len: int = 16000, max_batch: int = 64,
                 num_classes: int = 10, precision: int = BF16, with_classifier: bool = True, recheck_batch: int = 0,
                 recheck_margin: Optional[float] = None, half_type: Optional[int] = None, with_wavenet: bool = True):
        if not torch.cuda.is_available():
            raise DmadError('no MI355X/HIP device visible: the dmad engine has no CPU path')
        self.lib = _lib.load()
        wc = dict(res_channels=256, skip_channels=256, num_res_layers=36, dilation_cycle=12,
                  diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512, diffusion_step_embed_dim_out=512)
        wc.update(wavenet_config or {})
        if wc.get('in_channels', 1) != 1 or wc.get('out_channels', 1) != 1:
            raise DmadError('only in_channels = out_channels = 1 is supported')
        if half_type is None:    # f16 operands for the exact-vote engine (smaller recheck band), bf16 for the plain 16-bit engine
            half_type = {'bf16': HALF_BF16, 'f16': HALF_F16}[os.environ.get('DMAD_HALF_TYPE', 'f16' if precision == EXACT else 'bf16').lower()]
        self.cfg = DmadConfig(wc['res_channels'], wc['skip_channels'], wc['num_res_layers'], wc['dilation_cycle'],
                              wc['diffusion_step_embed_dim_in'], wc['diffusion_step_embed_dim_mid'],
                              wc['diffusion_step_embed_dim_out'], clip_len, max_batch, num_classes, precision,
                              1 if with_classifier else 0, recheck_batch, half_type, 1 if with_wavenet else 0)
        self.half_type = half_type
        self.with_wavenet = bool(with_wavenet)
        self.L, self.max_batch, self.num_classes, self.precision = clip_len, max_batch, num_classes, precision
        self.num_res_layers = wc['num_res_layers']
        self.wavenet_geometry = Engine.geometry(wc)
        self.device = torch.device('cuda', torch.cuda.current_device())
        h = C.c_void_p()
        check(self.lib.dmad_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        self.has_wavenet = False
        self.has_classifier = False
        self.has_unet = False
        # which weights are resident (state_fingerprint): a module that binds to an engine holding OTHER weights must
        # not silently run them
        self.wavenet_owner = self.classifier_owner = self.unet_owner = None
        self.classifier_kind = None
        self.mode = {BF16: MODE_FAST, FP32: MODE_FP32, EXACT: MODE_EXACT_VOTES}[precision]
        self.calibration = None                # what the last calibrate_recheck() observed
        self.waveform_tier = WAVE_SPLIT if precision == EXACT else WAVE_16BIT
        # what calibrations may never go below: the committed defaults, or a wider bound the caller chose (constructor / environment /
        # a direct set_recheck_margin call)
        self.floor1, self.floor2, self.floor_spec = DEFAULT_RECHECK_MARGIN[half_type], DEFAULT_RECHECK_MARGIN2, DEFAULT_SPEC_RECHECK_MARGIN
        self.floor_spec2 = DEFAULT_SPEC_RECHECK_MARGIN2
        self.spec_calibration = None
        if precision == EXACT:
            wt = os.environ.get('DMAD_WAVEFORM_TIER')
            if wt:
                self.set_waveform_tier({'16bit': WAVE_16BIT, 'fp32': WAVE_FP32, 'split': WAVE_SPLIT}[wt.lower()])
            if recheck_margin is None:
                recheck_margin = float(os.environ.get('DMAD_RECHECK_MARGIN', DEFAULT_RECHECK_MARGIN[half_type]))
            self.set_recheck_margin(recheck_margin)
            self.set_recheck_margin2(float(os.environ.get('DMAD_RECHECK_MARGIN2', DEFAULT_RECHECK_MARGIN2)))
            self.set_spec_recheck_margin(float(os.environ.get('DMAD_SPEC_RECHECK_MARGIN', DEFAULT_SPEC_RECHECK_MARGIN)))
            self.set_spec_recheck_margin2(float(os.environ.get('DMAD_SPEC_RECHECK_MARGIN2', DEFAULT_SPEC_RECHECK_MARGIN2)))

    def close(self):
        if getattr(self, '_h', None):
            torch.cuda.synchronize()
            self.lib.dmad_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def _load(self, arrays: Dict[str, np.ndarray]):
        for name, a in arrays.items():
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            check(self.lib.dmad_load_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        check(self.lib.dmad_finalize_weights(self._h))
        note = self.lib.dmad_last_warning()
        if note:
            warnings.warn('dmad engine: ' + note.decode(), RuntimeWarning, stacklevel=3)

    def load_wavenet(self, state_dict):
        if self.has_wavenet:
            raise DmadError('WaveNet weights are already loaded into this engine')
        self._load(fold_wavenet_state_dict(state_dict, self.num_res_layers))
        self.has_wavenet = True
        self.wavenet_owner = state_fingerprint(state_dict)

    def load_vgg19_bn(self, state_dict):
        if self.has_classifier:
            raise DmadError('classifier weights are already loaded into this engine')
        self._load(fold_vgg19_bn_state_dict(state_dict))
        self.has_classifier = True
        self.classifier_owner, self.classifier_kind = state_fingerprint(state_dict), 'vgg19_bn'

    def bind(self, part: str, state_dict, loader) -> None:
        """Make `state_dict` the resident weights of `part` ('wavenet' / 'classifier' / 'unet'): upload them if the
        part is empty, accept them if they ARE the resident ones, refuse anything else (an engine holds one weight set per
        part for its lifetime; use get_engine(fresh=True) / Engine(...) for a second model)."""
        owner = getattr(self, part + '_owner')
        if not getattr(self, 'has_' + part):
            loader(state_dict)
            return
        fp = state_fingerprint(state_dict)
        if owner != fp:
            raise DmadError('this engine already holds different %s weights (resident %s..., offered %s...): bind the module to '
                            'its own engine (dmad_hip.engine.get_engine(fresh=True))' % (part, str(owner)[:10], fp[:10]))

    # ------------------------------------------------------------------ exact-vote mode (EXACT engines)
    def set_mode(self, mode: int):
        check(self.lib.dmad_set_mode(self._h, int(mode)))
        self.mode = int(mode)

    def set_waveform_tier(self, tier: int):
        """dmad_set_waveform_tier: WaveNet tier of the waveform-returning surfaces (wavenet_eps, one_shot, ddpm_step / purify, the purifier
        inside query_logits) of an exact-vote engine: WAVE_SPLIT (default: fp32-grade, 8e-5), WAVE_FP32, WAVE_16BIT (4e-3)."""
        check(self.lib.dmad_set_waveform_tier(self._h, int(tier)))
        self.waveform_tier = int(tier)

    def set_recheck_margin(self, tau: float, calibrated: bool = False):
        """bound of the 16-bit tier.  A value the CALLER sets (calibrated = False) also becomes the floor of later calibrations when it
        is wider than the committed default: a calibration may only widen what is in force."""
        check(self.lib.dmad_set_recheck_margin(self._h, float(tau)))
        self.recheck_margin = float(tau)
        if not calibrated:
            self.floor1 = max(self._committed_margin1(), float(tau))

    def _committed_margin1(self) -> float:
        """the committed tier-1 bound for this engine's operand format and resident classifier kind"""
        table = DEFAULT_RECHECK_MARGIN_RESNEXT29 if getattr(self, 'classifier_kind', None) == 'resnext29' else DEFAULT_RECHECK_MARGIN
        return table[self.half_type]

    def set_recheck_margin2(self, tau2: float, calibrated: bool = False):
        """bound of the split-f16 middle tier (< 0: tier off, queued samples go straight to the fp32 path)."""
        check(self.lib.dmad_set_recheck_margin2(self._h, float(tau2)))
        self.recheck_margin2 = float(tau2)
        if not calibrated:
            self.floor2 = max(DEFAULT_RECHECK_MARGIN2, float(tau2))

    def set_spec_recheck_margin(self, tau: float, calibrated: bool = False):
        """bound of the spec-domain vote loop's 16-bit UNet tier (dmad_set_spec_recheck_margin)."""
        check(self.lib.dmad_set_spec_recheck_margin(self._h, float(tau)))
        self.spec_recheck_margin = float(tau)
        if not calibrated:
            self.floor_spec = max(DEFAULT_SPEC_RECHECK_MARGIN, float(tau))

    def set_spec_recheck_margin2(self, tau2: float, calibrated: bool = False):
        """bound of the spec-domain loop's split-f16 UNet tier (dmad_set_spec_recheck_margin2); < 0: queued samples go straight to fp32."""
        check(self.lib.dmad_set_spec_recheck_margin2(self._h, float(tau2)))
        self.spec_recheck_margin2 = float(tau2)
        if not calibrated:
            self.floor_spec2 = max(DEFAULT_SPEC_RECHECK_MARGIN2, float(tau2))

    def spec_recheck_stats(self, reset: bool = False, detail: bool = False):
        """-> (samples voted by spec_smooth_votes, samples whose chain left the 16-bit tier) since the last reset; detail: + the samples
        that reached the exact-fp32 UNet."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.dmad_spec_recheck_stats2(self._h, C.byref(a), C.byref(b), C.byref(c), 1 if reset else 0))
        return (int(a.value), int(b.value), int(c.value)) if detail else (int(a.value), int(b.value))

    def recheck_stats(self, reset: bool = False, detail: bool = False):
        """-> (samples voted, samples that left the 16-bit pass) since the last reset; detail: + samples that reached fp32."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.dmad_recheck_stats(self._h, C.byref(a), C.byref(b), C.byref(c), 1 if reset else 0))
        return (int(a.value), int(b.value), int(c.value)) if detail else (int(a.value), int(b.value))

    def calibrate_recheck(self, clip, sigma: float, sqrt_abar_star: float, t: int, c_a: float, c_b: float,
                          n: int = 1024, n_fp32: int = 512, headroom: float = 1.5, seed: int = 0xCA11B):
        """Measure, for THE RESIDENT WEIGHTS, what the defaults were measured for on the synthetic VGG19_bn (see
        DEFAULT_RECHECK_MARGIN): the largest error the 16-bit pass makes on a logit difference against the leader (vs the split-f16
        tier, n Philox samples per clip at this sigma) and the largest error of the split-f16 tier (against the exact-fp32 path,
        n_fp32 samples per clip).  `clip`: one clip or a list of clips (the maxima are taken over all of them).  The bounds become
            tau2 = max(committed default, headroom * e2),
            tau1 = max(committed default, headroom * e1 + tau2, TAIL_Z * s1 + tau2),
        s1 = the Gaussian scale of the per-sample error statistic read off its upper quantiles (q90, q99 of max_j |e_j - e_i|
        ~ the maximum of 9 normal differences): TAIL_Z * s1 is where the tail model of tools/fit_recheck_tail.py puts the
        per-sample miss probability at 1e-9 (on the committed study: 5.4 x 0.0056 = 0.030 < 0.034).
        A calibration can only WIDEN a bound — a maximum over a few hundred samples underestimates the tail the committed
        defaults were derived from (36 864 samples), so it is never allowed to go below them.  The logit sensitivity of a
        classifier — hence the error a given eps error turns into — is a property of its weights: call this once per
        (WaveNet, classifier, sigma) before certifying with checkpoints other than the ones the defaults were measured on.
        Returns (tau1, tau2, e1, e2); the observed errors are also kept in self.calibration."""
        if self.precision != EXACT:
            raise DmadError('calibrate_recheck needs an EXACT engine')
        clips = list(clip) if isinstance(clip, (list, tuple)) else [clip]
        mode, tau1, tau2 = self.mode, self.recheck_margin, self.recheck_margin2
        n_fp32 = min(n_fp32, n)

        def lead_err(a, b):                  # per-sample error of a logit difference against the reference's leader (see DEFAULT_RECHECK_MARGIN)
            e = a - b
            return (e - e.gather(1, b.argmax(1, keepdim=True))).abs().max(1).values

        def pair_err(a, b):
            return float(lead_err(a, b).max())

        def gauss_scale(le):                 # P(max of 9 |normal differences| > x) ~= 18 Q(x / s): s from the q90 and q99 points
            zs = {0.9: 2.5392, 0.99: 3.2608}            # 18 Q(z) = 1 - q
            return max(float(torch.quantile(le, q)) / z for q, z in zs.items())
        e1 = e2 = s1 = 0.0
        try:
            for ci, x in enumerate(clips):
                args = (x, sigma, sqrt_abar_star, t, c_a, c_b)
                idx = torch.arange(n, dtype=torch.int64, device=self.device)
                self.set_mode(MODE_EXACT_VOTES)       # path 0 there = the loop's FIRST PASS: 16-bit WaveNet + the classifier tier it runs (ResNeXt29: split-f16)
                fast = self.eval_samples(*args, idx, path=0, seed=seed + ci)
                mid = self.eval_samples(*args, idx, path=2, seed=seed + ci)
                ref = self.eval_samples(*args, idx[:n_fp32], path=1, seed=seed + ci)
                if not (bool(torch.isfinite(fast).all()) and bool(torch.isfinite(mid).all()) and bool(torch.isfinite(ref).all())):
                    raise DmadError('calibrate_recheck: non-finite logits')
                le1 = lead_err(fast.double(), mid.double())
                e1, s1 = max(e1, float(le1.max())), max(s1, gauss_scale(le1))
                e2 = max(e2, pair_err(mid[:n_fp32].double(), ref.double()))
        finally:
            self.set_mode(mode)
        floor1, floor2 = self.floor1, self.floor2         # the committed defaults, or a wider bound the caller put in force
        new2 = max(floor2, headroom * e2)
        new1 = max(floor1, headroom * e1 + new2, TAIL_Z * s1 + new2)
        self.set_recheck_margin(new1, calibrated=True); self.set_recheck_margin2(new2, calibrated=True)
        self.calibration = {'e1': e1, 'e2': e2, 's1': s1, 'tau1': new1, 'tau2': new2, 'n': n, 'n_fp32': n_fp32, 'clips': len(clips),
                            'headroom': headroom, 'floor1': floor1, 'floor2': floor2,
                            'previous': (tau1, tau2)}
        return new1, new2, e1, e2

    def spec_eval_samples(self, clip: torch.Tensor, sigma: float, t_star: int, q_a: float, q_b: float, c_a, c_b, c_1, c_2, c_sig,
                          mel_lo: float, mel_hi: float, idx: torch.Tensor, tier: int, seed: int = 0, want_spec: bool = False):
        """dmad_spec_eval_samples: logits [len(idx), C] (and the purified dB spectrograms when asked) of the spec-domain chain for
        the Monte Carlo samples with GLOBAL indices `idx` on UNet tier 0 (exact fp32) / 1 (16-bit) / 2 (split-f16).  Nothing votes."""
        clip = clip.detach().reshape(-1).contiguous().float()
        assert clip.is_cuda and clip.numel() == self.L
        idx = idx.detach().to(device=clip.device, dtype=torch.int64).contiguous()
        n = idx.numel()
        logits = torch.empty((n, self.num_classes), device=clip.device)
        spec = torch.empty((n, 1, 32, 32), device=clip.device) if want_spec else None
        arrs = [(C.c_float * (t_star + 1))(*[float(v) for v in a]) for a in (c_a, c_b, c_1, c_2, c_sig)]
        check(self.lib.dmad_spec_eval_samples(self._h, _ptr(clip), float(sigma), int(t_star), float(q_a), float(q_b), arrs[0], arrs[1], arrs[2],
                                              arrs[3], arrs[4], float(mel_lo), float(mel_hi), int(seed), _ptr(idx), int(n), int(tier),
                                              _ptr(logits), _ptr(spec), _stream()))
        return (logits, spec) if want_spec else logits

    def calibrate_spec_recheck(self, clip, sigma: float, chain_args: tuple, n: int = 256, headroom: float = 1.5, seed: int = 0x5BECCA1,
                               n_fp32: Optional[int] = None):
        """The spec-domain counterpart of calibrate_recheck: for THE RESIDENT WEIGHTS and this (sigma, t*), run n Monte Carlo samples'
        whole chains per clip on the UNet's 16-bit tier and on its split-f16 tier, and the first n_fp32 (default n / 2) of them on the
        exact-fp32 tier, from the same Philox keys; take the leader-difference error statistic (see DEFAULT_RECHECK_MARGIN) and set
            tau_spec2 = max(floor2, headroom * e2)                                    (split-f16 tier against fp32)
            tau_spec  = max(floor, headroom * e + tau_spec2, TAIL_Z * s + tau_spec2)  (16-bit tier against the split-f16 tier)
        with e its largest value and s its Gaussian scale (q90 / q99 points), floors = the committed defaults (or wider bounds the
        caller put in force): widen-only.  chain_args = (t_star, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, mel_lo, mel_hi) as for
        spec_smooth_votes.  Returns (tau_spec, e, s); the full record is kept in self.spec_calibration."""
        if self.precision != EXACT:
            raise DmadError('calibrate_spec_recheck needs an EXACT engine')
        clips = list(clip) if isinstance(clip, (list, tuple)) else [clip]
        n_fp32 = max(1, min(n, n // 2 if n_fp32 is None else n_fp32))
        e = s = e2 = 0.0
        zs = {0.9: 2.5392, 0.99: 3.2608}                  # 18 Q(z) = 1 - q (the maximum of 9 |normal differences|)

        def lead_err(a, ref):
            d = a - ref
            return (d - d.gather(1, ref.argmax(1, keepdim=True))).abs().max(1).values
        for ci, x in enumerate(clips):
            idx = torch.arange(n, dtype=torch.int64, device=self.device)
            lo = self.spec_eval_samples(x, sigma, *chain_args, idx, tier=1, seed=seed + ci).double()
            mid = self.spec_eval_samples(x, sigma, *chain_args, idx, tier=2, seed=seed + ci).double()
            ref = self.spec_eval_samples(x, sigma, *chain_args, idx[:n_fp32], tier=0, seed=seed + ci).double()
            if not (bool(torch.isfinite(lo).all()) and bool(torch.isfinite(mid).all()) and bool(torch.isfinite(ref).all())):
                raise DmadError('calibrate_spec_recheck: non-finite logits')
            le = lead_err(lo, mid)
            e = max(e, float(le.max()))
            s = max(s, max(float(torch.quantile(le, q)) / z for q, z in zs.items()))
            e2 = max(e2, float(lead_err(mid[:n_fp32], ref).max()))
        previous, floor, floor2 = (self.spec_recheck_margin, self.spec_recheck_margin2), self.floor_spec, self.floor_spec2
        new2 = max(floor2, headroom * e2)
        new = max(floor, headroom * e + new2, TAIL_Z * s + new2)
        self.set_spec_recheck_margin(new, calibrated=True); self.set_spec_recheck_margin2(new2, calibrated=True)
        self.spec_calibration = {'e': e, 's': s, 'e2': e2, 'tau_spec': new, 'tau_spec2': new2, 'n': n, 'n_fp32': n_fp32, 'clips': len(clips),
                                 'headroom': headroom, 'floor': floor, 'floor2': floor2, 'previous': previous, 'sigma': sigma,
                                 't_star': int(chain_args[0])}
        return new, e, s

    def eval_samples(self, clip: torch.Tensor, sigma: float, sqrt_abar_star: float, t: int, c_a: float, c_b: float,
                     idx: torch.Tensor, path: int = 0, seed: int = 0, sample0: int = 0, delta: Optional[torch.Tensor] = None,
                     want_x0: bool = False):
        """dmad_eval_samples: logits [len(idx), C] (and x0 [len(idx), L] when asked) of the Monte Carlo samples with GLOBAL indices
        `idx` (int64, device) on WaveNet path 0 (the mode's default) / 1 (exact fp32) / 2 (split-f16).  Nothing votes."""
        clip = clip.detach().reshape(-1).contiguous().float()
        assert clip.is_cuda and clip.numel() == self.L
        idx = idx.detach().to(device=clip.device, dtype=torch.int64).contiguous()
        n = idx.numel()
        logits = torch.empty((n, self.num_classes), device=clip.device)
        x0 = torch.empty((n, self.L), device=clip.device) if want_x0 else None
        if delta is not None:
            delta = delta.detach().reshape(-1, self.L).contiguous().float()
            assert delta.is_cuda
        check(self.lib.dmad_eval_samples(self._h, _ptr(clip), float(sigma), float(sqrt_abar_star), int(t), float(c_a), float(c_b),
                                         int(seed), int(sample0), _ptr(delta), _ptr(idx), int(n), int(path), _ptr(logits), _ptr(x0),
                                         _stream()))
        return (logits, x0) if want_x0 else logits

    def debug_rounding(self, dil: int = 0, res: int = 0, skip: int = 0, f0: int = 0, init: int = 0):
        """dmad_debug_rounding (measurement hook): single roundings of the 16-bit path switched on inside the split-f16 tier."""
        m = (C.c_int32 * 5)(int(dil), int(res), int(skip), int(f0), int(init))
        check(self.lib.dmad_debug_rounding(self._h, m))

    def wavenet_eps_path(self, x_t: torch.Tensor, t: int, path: int) -> torch.Tensor:
        """eps-network on an explicit path of an EXACT engine: 0 mode default, 1 exact fp32, 2 split-f16 (three MFMAs per product)."""
        x = self._wave(x_t)
        out = torch.empty_like(x)
        for s, e in self._chunks(x.shape[0]):
            check(self.lib.dmad_wavenet_eps_path(self._h, _ptr(x[s:e]), int(t), e - s, int(path), _ptr(out[s:e]), _stream()))
        return out

    def load_unet(self, state_dict):
        """improved_diffusion.unet.UNetModel state dict (synth.UNET_CONFIG geometry) -> engine, names prefixed 'un.'."""
        if self.has_unet:
            raise DmadError('UNet weights are already loaded into this engine')
        self._load({'un.' + k: _as_np(v.detach().cpu().double().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float64))
                    for k, v in state_dict.items()})
        self.has_unet = True
        self.unet_owner = state_fingerprint(state_dict)

    def load_resnext29(self, state_dict):
        if self.has_classifier:
            raise DmadError('classifier weights are already loaded into this engine')
        self._load(fold_resnext29_state_dict(state_dict))
        self.has_classifier = True
        self.classifier_owner, self.classifier_kind = state_fingerprint(state_dict), 'resnext29'
        if self.precision == EXACT:            # the committed bound for this classifier kind (widen-only, like a calibration)
            self.floor1 = max(self.floor1, self._committed_margin1())
            if self.recheck_margin < self.floor1:
                self.set_recheck_margin(self.floor1, calibrated=True)

    # ------------------------------------------------------------------ helpers
    def _wave(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise DmadError('input must live on the GPU (the dmad engine has no CPU path)')
        x = x.detach()
        if x.dim() == 3:
            assert x.shape[1] == 1, 'expected [B,1,L]'
            x = x[:, 0]
        assert x.dim() == 2 and x.shape[1] == self.L, 'expected [B,%d], got %s' % (self.L, tuple(x.shape))
        return x.contiguous().float()

    def _chunks(self, B):
        for s in range(0, B, self.max_batch):
            yield s, min(B, s + self.max_batch)

    # ------------------------------------------------------------------ hot path
    def wavenet_eps(self, x_t: torch.Tensor, t: int) -> torch.Tensor:
        x = self._wave(x_t)
        out = torch.empty_like(x)
        for s, e in self._chunks(x.shape[0]):
            check(self.lib.dmad_wavenet_eps(self._h, _ptr(x[s:e]), int(t), e - s, _ptr(out[s:e]), _stream()))
        return out

    def one_shot(self, x_t: torch.Tensor, t: int, c_a: float, c_b: float) -> torch.Tensor:
        x = self._wave(x_t)
        out = torch.empty_like(x)
        for s, e in self._chunks(x.shape[0]):
            check(self.lib.dmad_one_shot(self._h, _ptr(x[s:e]), int(t), float(c_a), float(c_b), e - s, _ptr(out[s:e]), _stream()))
        return out

    def ddpm_step(self, x: torch.Tensor, t: int, c_eps: float, c_div: float, c_sig: float, z: Optional[torch.Tensor],
                  seed: int = 0, sample0: int = 0):
        """in place on x ([B, L] contiguous fp32 CUDA)."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
        for s, e in self._chunks(x.shape[0]):
            zz = None if z is None else self._wave(z)[s:e].contiguous()
            check(self.lib.dmad_ddpm_step(self._h, _ptr(x[s:e]), int(t), float(c_eps), float(c_div), float(c_sig), _ptr(zz),
                                          int(seed), int(sample0) + s, e - s, _stream()))
        return x

    def diffuse(self, x0: torch.Tensor, c_a: float, c_b: float, z: Optional[torch.Tensor], seed: int = 0, sample0: int = 0):
        x = self._wave(x0)
        out = torch.empty_like(x)
        for s, e in self._chunks(x.shape[0]):
            zz = None if z is None else self._wave(z)[s:e].contiguous()
            check(self.lib.dmad_diffuse(self._h, _ptr(x[s:e]), float(c_a), float(c_b), _ptr(zz), int(seed), int(sample0) + s,
                                        e - s, _ptr(out[s:e]), _stream()))
        return out

    def ddpm_purify(self, x0: torch.Tensor, t_star: int, c_a: float, c_b: float, c_eps, c_div, c_sig, seed: int = 0, sample0: int = 0):
        """DiffWave.forward with device noise in one library call per chunk; c_eps / c_div / c_sig: t_star floats each."""
        x = self._wave(x0)
        out = torch.empty_like(x)
        arrs = [(C.c_float * t_star)(*[float(v) for v in a]) for a in (c_eps, c_div, c_sig)]
        for s, e in self._chunks(x.shape[0]):
            check(self.lib.dmad_ddpm_purify(self._h, _ptr(x[s:e]), int(t_star), float(c_a), float(c_b), arrs[0], arrs[1], arrs[2],
                                            int(seed), int(sample0) + s, e - s, _ptr(out[s:e]), _stream()))
        return out

    def _spec(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise DmadError('input must live on the GPU (the dmad engine has no CPU path)')
        x = x.detach()
        if x.dim() == 4:
            assert x.shape[1] == 1, 'expected [B,1,32,32]'
            x = x[:, 0]
        assert x.dim() == 3 and tuple(x.shape[1:]) == (32, 32), 'expected [B,32,32], got %s' % (tuple(x.shape),)
        return x.contiguous().float()

    def unet_eps(self, x_t: torch.Tensor, t: int, tier: Optional[int] = None) -> torch.Tensor:
        """eps = UNetModel(x_t, t * ones): [B,1,32,32] or [B,32,32] -> [B,32,32].  tier: None = the mode's tier of the map-returning
        surfaces (dmad_unet_eps); 0 exact fp32 / 1 16-bit / 2 split-f16 explicitly (dmad_unet_eps_tier)."""
        x = self._spec(x_t)
        out = torch.empty_like(x)
        for s, e in self._chunks(x.shape[0]):
            if tier is None:
                check(self.lib.dmad_unet_eps(self._h, _ptr(x[s:e]), int(t), e - s, _ptr(out[s:e]), _stream()))
            else:
                check(self.lib.dmad_unet_eps_tier(self._h, _ptr(x[s:e]), int(t), e - s, int(tier), _ptr(out[s:e]), _stream()))
        return out

    def unet_p_sample(self, x: torch.Tensor, t: int, c_a: float, c_b: float, c_1: float, c_2: float, c_sig: float,
                      z: Optional[torch.Tensor] = None, seed: int = 0, sample0: int = 0, want_x0: bool = False):
        """in place on x ([B,32,32] contiguous fp32 CUDA); returns pred_xstart when asked."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and tuple(x.shape[1:]) == (32, 32)
        x0 = torch.empty_like(x) if want_x0 else None
        for s, e in self._chunks(x.shape[0]):
            zz = None if z is None else self._spec(z)[s:e].contiguous()
            check(self.lib.dmad_unet_p_sample(self._h, _ptr(x[s:e]), int(t), float(c_a), float(c_b), float(c_1), float(c_2), float(c_sig),
                                              _ptr(zz), int(seed), int(sample0) + s, e - s, _ptr(x0[s:e]) if want_x0 else None, _stream()))
        return x0

    def mel_db(self, x: torch.Tensor) -> torch.Tensor:
        x = self._wave(x)
        out = torch.empty((x.shape[0], 1, 32, 32), device=x.device, dtype=torch.float32)
        for s, e in self._chunks(x.shape[0]):
            check(self.lib.dmad_mel_db(self._h, _ptr(x[s:e]), e - s, _ptr(out[s:e]), _stream()))
        return out

    def mel_power(self, x: torch.Tensor) -> torch.Tensor:
        x = self._wave(x)
        out = torch.empty((x.shape[0], 1, 32, 32), device=x.device, dtype=torch.float32)
        for s, e in self._chunks(x.shape[0]):
            check(self.lib.dmad_mel_power(self._h, _ptr(x[s:e]), e - s, _ptr(out[s:e]), _stream()))
        return out

    def power_to_db(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise DmadError('input must live on the GPU (the dmad engine has no CPU path)')
        xc = x.detach().contiguous().float()
        out = torch.empty_like(xc)
        check(self.lib.dmad_power_to_db(self._h, _ptr(xc), xc.numel(), _ptr(out), _stream()))
        return out

    def classify(self, spec: torch.Tensor) -> torch.Tensor:
        if not spec.is_cuda:
            raise DmadError('input must live on the GPU (the dmad engine has no CPU path)')
        sp = spec.detach().reshape(spec.shape[0], 32 * 32).contiguous().float()
        out = torch.empty((sp.shape[0], self.num_classes), device=sp.device, dtype=torch.float32)
        for s, e in self._chunks(sp.shape[0]):
            check(self.lib.dmad_classify(self._h, _ptr(sp[s:e]), e - s, _ptr(out[s:e]), _stream()))
        return out

    def classify_tier(self, spec: torch.Tensor, tier: int) -> torch.Tensor:
        """dmad_classify_tier: the classifier on an explicit tier (0 fp32, 1 the 16-bit tier / 2 the split-f16 tier of ResNeXt29) — test / measurement hook."""
        if not spec.is_cuda:
            raise DmadError('input must live on the GPU (the dmad engine has no CPU path)')
        sp = spec.detach().reshape(spec.shape[0], 32 * 32).contiguous().float()
        out = torch.empty((sp.shape[0], self.num_classes), device=sp.device, dtype=torch.float32)
        for s, e in self._chunks(sp.shape[0]):
            check(self.lib.dmad_classify_tier(self._h, _ptr(sp[s:e]), e - s, int(tier), _ptr(out[s:e]), _stream()))
        return out

    def vote(self, logits: torch.Tensor, counts: torch.Tensor):
        lg = logits.detach().contiguous().float()
        assert lg.is_cuda and counts.is_cuda and counts.dtype == torch.int64 and lg.shape[1] == self.num_classes
        check(self.lib.dmad_vote(self._h, _ptr(lg), lg.shape[0], _ptr(counts), _stream()))

    def smooth_votes(self, clip: torch.Tensor, sigma: float, sqrt_abar_star: float, t: int, c_a: float, c_b: float,
                     n: int, batch: Optional[int] = None, seed: int = 0, sample0: int = 0,
                     delta: Optional[torch.Tensor] = None, want_logits: bool = False, want_x0: bool = False,
                     counts: Optional[torch.Tensor] = None):
        """The fused Monte Carlo loop (dmad_smooth_votes).  Returns (counts[int64, C] on device, logits|None, x0|None)."""
        clip = clip.detach().reshape(-1).contiguous().float()
        assert clip.is_cuda and clip.numel() == self.L
        batch = min(batch or self.max_batch, self.max_batch)
        if counts is None:
            counts = torch.zeros(self.num_classes, dtype=torch.int64, device=clip.device)
        logits = torch.empty((n, self.num_classes), device=clip.device) if want_logits else None
        x0 = torch.empty((n, self.L), device=clip.device) if want_x0 else None
        if delta is not None:
            delta = delta.detach().reshape(n, self.L).contiguous().float()
            assert delta.is_cuda
        check(self.lib.dmad_smooth_votes(self._h, _ptr(clip), float(sigma), float(sqrt_abar_star), int(t), float(c_a), float(c_b),
                                         int(n), int(batch), int(seed), int(sample0), _ptr(delta), _ptr(counts), _ptr(logits),
                                         _ptr(x0), _stream()))
        return counts, logits, x0

    def spec_smooth_votes(self, clip: torch.Tensor, sigma: float, t_star: int, q_a: float, q_b: float, c_a, c_b, c_1, c_2, c_sig,
                          mel_lo: float, mel_hi: float, n: int, batch: Optional[int] = None, seed: int = 0, sample0: int = 0,
                          want_logits: bool = False, want_spec: bool = False, counts: Optional[torch.Tensor] = None):
        """dmad_spec_smooth_votes (BASELINE C5): the vote loop with the spec-domain UNet purifier.  Coefficient sequences
        c_*: t_star + 1 floats each (p_sample at t = 0..t_star).  Returns (counts, logits|None, purified spec|None)."""
        clip = clip.detach().reshape(-1).contiguous().float()
        assert clip.is_cuda and clip.numel() == self.L
        batch = min(batch or self.max_batch, self.max_batch)
        if counts is None:
            counts = torch.zeros(self.num_classes, dtype=torch.int64, device=clip.device)
        logits = torch.empty((n, self.num_classes), device=clip.device) if want_logits else None
        spec = torch.empty((n, 1, 32, 32), device=clip.device) if want_spec else None
        arrs = [(C.c_float * (t_star + 1))(*[float(v) for v in a]) for a in (c_a, c_b, c_1, c_2, c_sig)]
        check(self.lib.dmad_spec_smooth_votes(self._h, _ptr(clip), float(sigma), int(t_star), float(q_a), float(q_b), arrs[0], arrs[1], arrs[2],
                                              arrs[3], arrs[4], float(mel_lo), float(mel_hi), int(n), int(batch), int(seed), int(sample0),
                                              _ptr(counts), _ptr(logits), _ptr(spec), _stream()))
        return counts, logits, spec

    def query_logits(self, x: torch.Tensor, repeats: int, sampler: int = 0, t_star: int = 0, c_a: float = 0.0, c_b: float = 0.0,
                     c_eps=None, c_div=None, c_sig=None, seed: int = 0, sample0: int = 0):
        """dmad_query_logits: x [B,1,L] -> (logits [repeats*B, C], decisions int32 [repeats*B]); row r*B+b is clip b."""
        xw = self._wave(x)
        B = xw.shape[0]
        logits = torch.empty((repeats * B, self.num_classes), device=xw.device, dtype=torch.float32)
        dec = torch.empty((repeats * B,), device=xw.device, dtype=torch.int32)
        arrs = [None, None, None]
        if sampler == 1:
            arrs = [(C.c_float * t_star)(*[float(v) for v in a]) for a in (c_eps, c_div, c_sig)]
        check(self.lib.dmad_query_logits(self._h, _ptr(xw), B, int(repeats), int(sampler), int(t_star), float(c_a), float(c_b),
                                         arrs[0], arrs[1], arrs[2], int(seed), int(sample0), _ptr(logits), _ptr(dec), _stream()))
        return logits, dec

    def spec_query_logits(self, x: torch.Tensor, repeats: int, t_star: int, q_a: float, q_b: float, c_a, c_b, c_1, c_2, c_sig,
                          mel_lo: float, mel_hi: float, seed: int = 0, sample0: int = 0):
        """dmad_spec_query_logits: x [B,1,L] -> (logits [repeats*B, C], decisions int32 [repeats*B]) through the spec-domain chain
        (mel dB -> standardise -> q_sample(t*) -> p_sample steps -> un-standardise -> classifier); row r*B+b is clip b."""
        xw = self._wave(x)
        B = xw.shape[0]
        logits = torch.empty((repeats * B, self.num_classes), device=xw.device, dtype=torch.float32)
        dec = torch.empty((repeats * B,), device=xw.device, dtype=torch.int32)
        arrs = [(C.c_float * (t_star + 1))(*[float(v) for v in a]) for a in (c_a, c_b, c_1, c_2, c_sig)]
        check(self.lib.dmad_spec_query_logits(self._h, _ptr(xw), B, int(repeats), int(t_star), float(q_a), float(q_b), arrs[0], arrs[1], arrs[2],
                                              arrs[3], arrs[4], float(mel_lo), float(mel_hi), int(seed), int(sample0), _ptr(logits), _ptr(dec),
                                              _stream()))
        return logits, dec

    def philox_raw(self, seed: int, sample: int, stream: int, nblocks: int) -> torch.Tensor:
        out = torch.empty(nblocks * 4, dtype=torch.int32, device=self.device)
        check(self.lib.dmad_philox_raw(self._h, int(seed), int(sample), int(stream), int(nblocks), _ptr(out), _stream()))
        return out

    def philox_normal(self, seed: int, sample0: int, stream: int, B: int) -> torch.Tensor:
        out = torch.empty((B, self.L), dtype=torch.float32, device=self.device)
        check(self.lib.dmad_philox_normal(self._h, int(seed), int(sample0), int(stream), int(B), _ptr(out), _stream()))
        return out

    def time_layer(self, layer: int, B: int, iters: int) -> float:
        ms = C.c_float(0)
        check(self.lib.dmad_time_layer(self._h, int(layer), int(B), int(iters), C.byref(ms), _stream()))
        return float(ms.value)

    def profile_layers(self, max_launches: int):
        check(self.lib.dmad_profile_layers(self._h, int(max_launches)))

    def profile_read(self):
        """-> (summed ms, launches) of the bracketed wn_layer_bf16 launches."""
        ms, n = C.c_float(0), C.c_int32(0)
        check(self.lib.dmad_profile_read(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def profile_read_final(self):
        """-> (summed ms, launches) of the bracketed wn_final launches; call before profile_read()."""
        ms, n = C.c_float(0), C.c_int32(0)
        check(self.lib.dmad_profile_read_final(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def device_bytes(self) -> int:
        return int(self.lib.dmad_device_bytes(self._h))


def conv_h16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1, groups: int = 1, relu: bool = False,
             res: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None, want32: bool = True, want16: bool = True):
    """dmad_conv_h16 — the f16 conv-GEMM family as a standalone op (test hook).  x: f16 NHWC [B,H,H,Cx] (CUDA), x2: optional second
    map [B,H,H,C2] whose channels follow x's (dense convs only); w: f16 [groups, taps, M, K] (taps 9 or 1, K per group = (Cx + C2) /
    groups); bias fp32 [groups*M]; res: optional f16 [B,Ho,Ho,groups*M].  Returns (out32 | None, out16 | None), NHWC."""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float16 and w.is_cuda and w.dtype == torch.float16 and x.dim() == 4 and w.dim() == 4
    x, w = x.contiguous(), w.contiguous()
    B, H, W_, cx = x.shape
    assert H == W_
    g_, taps, M, K = w.shape
    assert g_ == groups and taps in (1, 9)
    ksplit = 0
    if x2 is not None:
        assert groups == 1 and x2.dtype == torch.float16 and x2.shape[:3] == x.shape[:3]
        x2 = x2.contiguous()
        ksplit = cx
        assert cx + x2.shape[3] == K
    else:
        assert cx == groups * K
    Ho = (H - 1) // stride + 1
    out32 = torch.empty((B, Ho, Ho, groups * M), device=x.device, dtype=torch.float32) if want32 else None
    out16 = torch.empty((B, Ho, Ho, groups * M), device=x.device, dtype=torch.float16) if want16 else None
    if bias is not None:
        bias = bias.detach().contiguous().float()
    if res is not None:
        assert res.dtype == torch.float16 and tuple(res.shape) == (B, Ho, Ho, groups * M)
        res = res.contiguous()
    check(lib.dmad_conv_h16(_ptr(x), _ptr(x2), int(ksplit), _ptr(w), _ptr(bias), _ptr(res), B, H, M, K, taps, int(stride), int(groups),
                            1 if relu else 0, _ptr(out32), _ptr(out16), _stream()))
    return out32, out16


def conv_h16_up2(x_half: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, want_stats: bool = False):
    """dmad_conv_h16_up2 — nearest x2 upsampling + 3x3 conv in one launch (test hook of GemmH16Args::up2).  x_half: f16 NHWC
    [B,H/2,H/2,K]; w: f16 [1,9,M,K]; res: optional f16 [B,H,H,M].  Returns (out32, out16, stats | None); raises DmadError when the shape
    is not served by the fusing form."""
    lib = _lib.load()
    assert x_half.is_cuda and x_half.dtype == torch.float16 and w.dtype == torch.float16 and w.shape[0] == 1 and w.shape[1] == 9
    x_half, w = x_half.contiguous(), w.contiguous()
    B, Hh, _, K = x_half.shape
    M, H = w.shape[2], 2 * Hh
    assert w.shape[3] == K
    out32 = torch.empty((B, H, H, M), device=x_half.device, dtype=torch.float32)
    out16 = torch.empty((B, H, H, M), device=x_half.device, dtype=torch.float16)
    stats = torch.zeros((B * H * H // 64, M // 4, 2), device=x_half.device, dtype=torch.float32) if want_stats else None
    if bias is not None:
        bias = bias.detach().contiguous().float()
    if res is not None:
        assert res.dtype == torch.float16 and tuple(res.shape) == (B, H, H, M)
        res = res.contiguous()
    check(lib.dmad_conv_h16_up2(_ptr(x_half), _ptr(w), _ptr(bias), _ptr(res), B, H, M, K, _ptr(out32), _ptr(out16), _ptr(stats), _stream()))
    return out32, out16, stats


def split_f16(x: torch.Tensor) -> torch.Tensor:
    """dmad_split_f16: the split-f16 storage form (hi / lo f16 pairs in the bytes of the floats) of an fp32 CUDA tensor whose last dimension is a
    multiple of 4; returned as a float32 tensor of the same shape (its bits are NOT floats)."""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib.dmad_split_f16(_ptr(x), x.numel(), _ptr(y), _stream()))
    return y


def conv_x3(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1, relu: bool = False,
            res: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None, out_split: bool = False, groups: int = 1,
            res_split: bool = False) -> torch.Tensor:
    """dmad_conv_x3 — the split-f16 conv GEMM as a standalone op (test hook).  x: fp32 NHWC [B,H,H,Cx], x2: optional second map whose
    channels follow x's (dense only); w: fp32 [taps, M, K] (dense) or [groups, taps, M, K]; bias fp32 [groups*M]; res fp32
    [B,Ho,Ho,groups*M] (res_split: handed to the kernel in the split format).  Operands are converted with split_f16 here."""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and w.is_cuda and w.dtype == torch.float32 and x.dim() == 4 and w.dim() in (3, 4)
    if w.dim() == 3:
        w = w[None]
    B, H, W_, cx = x.shape
    assert H == W_ and w.shape[0] == groups
    _, taps, M, K = w.shape
    xs, ws = split_f16(x), split_f16(w)
    x2s, ksplit = None, 0
    if x2 is not None:
        assert groups == 1 and x2.shape[:3] == x.shape[:3] and cx + x2.shape[3] == K
        x2s, ksplit = split_f16(x2), cx
    else:
        assert cx == groups * K
    Ho = (H - 1) // stride + 1
    out = torch.empty((B, Ho, Ho, groups * M), device=x.device, dtype=torch.float32)
    if bias is not None:
        bias = bias.detach().contiguous().float()
    if res is not None:
        assert tuple(res.shape) == (B, Ho, Ho, groups * M) and res.dtype == torch.float32
        res = split_f16(res) if res_split else res.contiguous()
    check(lib.dmad_conv_x3(_ptr(xs), _ptr(x2s), int(ksplit), _ptr(ws), _ptr(bias), _ptr(res), B, H, M, K, taps, int(stride), int(groups),
                           1 if relu else 0, 1 if out_split else 0, 1 if res_split else 0, _ptr(out), _stream()))
    return out


def conv_h16_stats(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1, res: Optional[torch.Tensor] = None):
    """dmad_conv_h16_stats: dense f16 conv (x [B,H,H,K] f16, w [1,taps,M,K] f16) -> (out16 [B,Ho,Ho,M] f16, stats [B*Ho*Ho/blk, M/4, 2] fp32):
    the GroupNorm statistics the producing GEMM's epilogue leaves for groupnorm16_apply (blk = 64 pixels, 16 on 4x4 maps)."""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float16 and w.dtype == torch.float16 and w.shape[0] == 1
    x, w = x.contiguous(), w.contiguous()
    B, H, _, K = x.shape
    _, taps, M, K2 = w.shape
    assert K2 == K
    Ho = (H - 1) // stride + 1
    blk = 64 if Ho * Ho >= 64 else 16
    out16 = torch.empty((B, Ho, Ho, M), device=x.device, dtype=torch.float16)
    stats = torch.zeros((B * Ho * Ho // blk, M // 4, 2), device=x.device, dtype=torch.float32)
    if bias is not None:
        bias = bias.detach().contiguous().float()
    if res is not None:
        res = res.contiguous()
    check(lib.dmad_conv_h16_stats(_ptr(x), _ptr(w), _ptr(bias), _ptr(res), B, H, M, K, taps, int(stride), _ptr(out16), _ptr(stats), _stream()))
    return out16, stats


def groupnorm16_apply(x: torch.Tensor, st: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, silu: bool = True, ss: Optional[torch.Tensor] = None,
                      x2: Optional[torch.Tensor] = None, st2: Optional[torch.Tensor] = None, out32: bool = False):
    """dmad_groupnorm16_apply: one-pass GroupNorm32 (+ scale-shift, + SiLU) of the f16 map x [B,HW,c1] (| x2 [B,HW,C-c1]) from the
    statistics slabs of conv_h16_stats.  Returns y [B,HW,C] (f16, or fp32 with out32)."""
    lib = _lib.load()
    x, st = x.contiguous(), st.contiguous()
    B, HW, c1 = x.shape
    C = c1 + (x2.shape[2] if x2 is not None else 0)
    y = torch.empty((B, HW, C), device=x.device, dtype=torch.float32 if out32 else torch.float16)
    if x2 is not None:
        x2, st2 = x2.contiguous(), st2.contiguous()
    gamma, beta = gamma.detach().contiguous().float(), beta.detach().contiguous().float()
    if ss is not None:
        ss = ss.detach().contiguous().float()
    check(lib.dmad_groupnorm16_apply(_ptr(x), _ptr(st), _ptr(x2), _ptr(st2), int(c1 if x2 is not None else 0), _ptr(gamma), _ptr(beta), _ptr(ss),
                                     1 if silu else 0, B, HW, C, None if out32 else _ptr(y), _ptr(y) if out32 else None, _stream()))
    return y


def bind_classifier(state_dict, loader_name: str, engine: Optional[Engine] = None) -> Engine:
    """Engine that holds exactly `state_dict` as its classifier: `engine` (refused if it holds another one), else the
    shared engine, else — when the shared engine already serves a different classifier — an engine of this module's own
    (classifier-only use: mel + classify; the fused Monte Carlo loop needs denoiser and classifier in ONE engine)."""
    if engine is not None:
        engine.bind('classifier', state_dict, getattr(engine, loader_name))
        return engine
    eng = get_engine()
    if eng.has_classifier and eng.classifier_owner != state_fingerprint(state_dict):
        eng = Engine(dict(eng.wavenet_geometry), max_batch=eng.max_batch, precision=eng.precision)
    eng.bind('classifier', state_dict, getattr(eng, loader_name))
    return eng


_ENGINES: Dict[tuple, Engine] = {}
_PRECISIONS = {'bf16': BF16, 'fp32': FP32, 'exact': EXACT}


def get_engine(wavenet_config: Optional[dict] = None, precision: Optional[int] = None, max_batch: Optional[int] = None,
               fresh: bool = False) -> Engine:
    """Process-wide engine per (device, precision), shared by the denoiser, the mel transform and the classifier so that
    the Monte Carlo loop can run fused.  Defaults: DMAD_PRECISION = exact (bf16 throughput + fp32 recheck of the close
    votes: counts equal the fp32 path's; bf16 and fp32 are opt-in), DMAD_MAX_BATCH = 64.  A caller that names a WaveNet
    geometry or a max_batch the shared engine was not created with gets a DmadError, never another model's engine.
    Engines are single-stream objects (the step-embedding cache is not stream-keyed): one HIP stream at a time."""
    if precision is None:
        precision = _PRECISIONS[os.environ.get('DMAD_PRECISION', 'exact').lower()]
    key = (torch.cuda.current_device() if torch.cuda.is_available() else -1, precision)
    if fresh or key not in _ENGINES:
        eng = Engine(wavenet_config, max_batch=max_batch if max_batch is not None else int(os.environ.get('DMAD_MAX_BATCH', '64')),
                     precision=precision)
        if fresh:
            return eng
        _ENGINES[key] = eng
        return eng
    eng = _ENGINES[key]
    if wavenet_config is not None:
        want = Engine.geometry(wavenet_config)
        if want != eng.wavenet_geometry:
            raise DmadError('the shared engine was created for WaveNet geometry %s, not %s: create the denoiser first or pass '
                            'an engine of its own (get_engine(..., fresh=True))' % (eng.wavenet_geometry, want))
    if max_batch is not None and max_batch != eng.max_batch:
        raise DmadError('the shared engine has max_batch %d, not %d (set DMAD_MAX_BATCH or use fresh=True)' % (eng.max_batch, max_batch))
    return eng
