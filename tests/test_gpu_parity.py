"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes), against
(1) the golden fixtures captured from the imported reference and (2) the CPU oracle on the same seeded
inputs.  Tolerances:
  * integer work (Philox words, vote counts in fp32 mode): bit-exact;
  * fp32 engine (exact-fp32 matrix path): 2e-5 of the tensor's max (summation order differs from MKL);
  * bf16 engine: waveforms/eps within 3e-2 of the tensor's max and 2.5e-2 rms-relative
    (bf16 operands, fp32 accumulation, 36 layers); votes may flip only where the reference's own
    top-2 logit margin is below the bf16 logit error;
  * f16 operands (the exact-vote engine's 16-bit path): 4e-3 of the tensor's max (11-bit significand);
  * exact-vote mode (16-bit path + fp32 recheck below the margin bound): vote counts bit-exact against the fp32 path.
"""
import os

import numpy as np
import pytest
import torch

from dmad_hip import synth

pytestmark = pytest.mark.gpu

FP32_TOL = 2e-5
BF16_MAX_TOL, BF16_RMS_TOL = 3e-2, 2.5e-2
F16_MAX_TOL = 4e-3
UNET_X3_TOL = 1e-4          # the UNet's split-f16 middle tier vs the reference fixture (fp32 pipeline, ~22-bit products)
UNET_F16_TOL = 6e-3         # the UNet's 16-bit tier vs the reference fixture (f16 maps between ~180 ops): measured 4.1e-3 / 2.2e-3 of max|eps| at t = 3 / 40 (tools/gpu_unet_tolerance.py)
UNET_FP32_TOL = 1e-5        # the exact-fp32 UNet tier vs the reference fixture: measured 2.6e-6 / 1.7e-6 at t = 3 / 40 (tools/gpu_unet_tolerance.py)
SPLIT_TOL = 8e-5            # the split-f16 tier (three f16 MFMAs per product): what an exact-vote engine's waveform surfaces deliver


def relmax(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / np.abs(ref).max()


def relrms(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.sqrt(((got - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())


@pytest.fixture(scope='module')
def orc():
    from oracle import dmad_oracle
    return dmad_oracle


@pytest.fixture(scope='module')
def weights():
    return synth.wavenet_state_dict(1234), synth.vgg19_bn_state_dict(4321)


@pytest.fixture(scope='module')
def engines(weights):
    from dmad_hip import engine as E
    out = {}
    for name, prec in (('fp32', E.FP32), ('bf16', E.BF16), ('exact', E.EXACT)):
        eng = E.Engine(max_batch=6, precision=prec, recheck_batch=4)      # 'exact': f16 operands + fp32 recheck
        eng.load_wavenet(weights[0])
        eng.load_vgg19_bn(weights[1])
        out[name] = eng
    yield out
    for e in out.values():
        e.close()


@pytest.fixture(scope='module')
def sched(orc):
    hp = orc.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    ab = hp['Alpha_bar']
    return hp, (lambda t: (float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t])))


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def synth_vgg(seed=4321, **kw):
    """models.vgg.VGG module carrying the synthetic weights the engines of this file hold (an engine refuses a module that
    offers other weights than its resident classifier)."""
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    net = vgg19_bn(num_classes=10, in_channels=1)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(seed, **kw).items()})
    return net.eval()


# ------------------------------------------------------------------------------------------ library
def test_native_library_is_loaded(engines):
    from dmad_hip import _lib
    assert os.path.exists(_lib.LIB_PATH)
    maps = open('/proc/self/maps').read()
    assert 'libdmad_hip.so' in maps
    assert b'gfx950' in _lib.load().dmad_version()


def test_error_behaviour(engines):
    from dmad_hip._lib import DmadError
    eng = engines['fp32']
    with pytest.raises(DmadError):
        eng.wavenet_eps(torch.zeros(1, 16000), 3)                       # CPU tensor: no CPU path
    with pytest.raises(AssertionError):
        eng.wavenet_eps(torch.zeros(1, 15999, device='cuda'), 3)
    with pytest.raises(DmadError):
        eng.wavenet_eps(torch.zeros(1, 16000, device='cuda'), -1)
    with pytest.raises(DmadError):
        eng.unet_eps(torch.zeros(1, 32, 32, device='cuda'), 3)          # UNet weights were never loaded into this engine
    with pytest.raises(DmadError):
        eng.load_resnext29(synth.resnext29_state_dict(2929))            # a classifier (VGG19_bn) is already resident
    with pytest.raises(DmadError):
        eng.load_wavenet(synth.wavenet_state_dict(1234))
    from dmad_hip import engine as E
    bare = E.Engine(max_batch=2, precision=E.BF16, with_classifier=False)
    with pytest.raises(DmadError):
        bare.wavenet_eps(torch.zeros(1, 16000, device='cuda'), 3)       # nothing loaded yet
    with pytest.raises(DmadError):
        bare.classify(torch.zeros(1, 1, 32, 32, device='cuda'))         # created without the classifier stage
    with pytest.raises(DmadError):
        E.Engine(wavenet_config=dict(res_channels=128), max_batch=2)    # only the 256-channel WaveNet is built
    bare.close()
    # an empty Monte Carlo loop is legal and votes nothing (certified_robust.py:59-65 with num = 0); a negative count is refused
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
    counts, _, _ = eng.smooth_votes(clip, 0.5, 0.9, 10, 1.0, 0.5, 0)
    assert counts.tolist() == [0] * 10
    with pytest.raises(DmadError):
        eng.smooth_votes(clip, 0.5, 0.9, 10, 1.0, 0.5, -1)


# ------------------------------------------------------------------------------------------ noise
def test_philox_words_bit_exact(engines, orc):
    eng = engines['fp32']
    for seed, sample, stream in ((0, 0, 0), (0x0123456789ABCDEF, 99999, 3), (2 ** 63 + 5, 2 ** 40 + 1, 0xD1FF)):
        raw = eng.philox_raw(seed, sample, stream, 4000).cpu().numpy().view(np.uint32).reshape(-1, 4)
        ctr, key = orc.philox_counters(seed, sample, stream, 4000)
        assert np.array_equal(raw, orc.philox4x32_10(ctr, key))


def test_philox_normal_is_index_keyed(engines, orc):
    eng = engines['bf16']
    a = eng.philox_normal(42, 100, 0, 6).cpu().numpy()
    b = eng.philox_normal(42, 103, 0, 3).cpu().numpy()
    assert np.array_equal(a[3:], b)                                     # sample i's noise depends on (seed, i) only
    ref = orc.philox_normal(42, 104, 0, 16000)
    assert np.abs(a[4] - ref).max() < 1e-5
    big = eng.philox_normal(1, 0, 0, 6).cpu().numpy().reshape(-1)
    assert abs(big.mean()) < 0.02 and abs(big.std() - 1) < 0.02
    assert abs((np.abs(big) > 1.96).mean() - 0.05) < 0.006


# ------------------------------------------------------------------------------------------ stages
def test_mel_frontend(engines, orc):
    L = 16000
    t = np.arange(L) / 16000.0
    x = np.stack([synth.synthetic_clip(0)[0], 0.3 * np.sin(2 * np.pi * 1000 * t), np.eye(1, L, 8000)[0], np.zeros(L)]).astype(np.float32)
    got = engines['bf16'].mel_db(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = orc.mel_db_f64(x[:, None, :])
    assert got.shape == (4, 1, 32, 32)
    assert np.all(got[3] == -100.0)
    big = ref > -60
    assert np.abs(got - ref)[big].max() < 2e-3
    assert np.abs(got - orc.mel_db(torch.from_numpy(x[:, None, :])).numpy())[big].max() < 2e-3


def test_vgg19_bn_vs_reference_fixture(engines, golden_dir):
    z = G(golden_dir, 'classifiers.npz')
    got = engines['fp32'].classify(torch.from_numpy(z['spec_in']).cuda()).cpu().numpy()
    assert relmax(got, z['vgg_logits']) < FP32_TOL
    assert (got.argmax(1) == z['vgg_logits'].argmax(1)).all()


def test_step_embedding_table(engines, orc, weights, golden_dir):
    # the bias table is observable through eps; check the embedding MLP via a 1-layer identity: compare
    # the fp32 engine against the oracle at several steps on a short batch
    w = orc.folded_weights(weights[0])
    x = torch.from_numpy(synth.synthetic_clip(2))[None]
    for t in (0, 116):
        ref = orc.wavenet_forward(w, x, t * torch.ones((1, 1))).numpy()[:, 0]
        got = engines['fp32'].wavenet_eps(x.cuda(), t).cpu().numpy()
        assert relmax(got, ref) < FP32_TOL, t


def test_wavenet_eps_vs_reference_fixture(engines, golden_dir):
    z = G(golden_dir, 'wavenet_full.npz')
    x_t = torch.from_numpy(z['x_t']).cuda()
    ref = z['eps'][:, 0]
    got = engines['fp32'].wavenet_eps(x_t, int(z['t'])).cpu().numpy()
    assert relmax(got, ref) < FP32_TOL
    got = engines['bf16'].wavenet_eps(x_t, int(z['t'])).cpu().numpy()
    assert relmax(got, ref) < BF16_MAX_TOL and relrms(got, ref) < BF16_RMS_TOL
    from dmad_hip import engine as E
    ex = engines['exact']                                     # the exact-vote engine: f16 operands on its 16-bit path
    assert ex.half_type == 1 and ex.waveform_tier == E.WAVE_SPLIT
    got = ex.wavenet_eps(x_t, int(z['t'])).cpu().numpy()      # waveform surfaces default to the split-f16 tier: fp32-grade
    assert relmax(got, ref) < SPLIT_TOL, relmax(got, ref)
    ex.set_waveform_tier(E.WAVE_16BIT)                        # the same kernels as the bf16 engine's, instantiated on f16 operands
    got = ex.wavenet_eps(x_t, int(z['t'])).cpu().numpy()
    assert 1e-5 < relmax(got, ref) < F16_MAX_TOL and relrms(got, ref) < F16_MAX_TOL
    ex.set_waveform_tier(E.WAVE_FP32)
    got = ex.wavenet_eps(x_t, int(z['t'])).cpu().numpy()
    assert relmax(got, ref) < FP32_TOL
    ex.set_waveform_tier(E.WAVE_SPLIT)
    ex.set_mode(E.MODE_FAST)                                  # DMAD_MODE_FAST: 16-bit everywhere, whatever the waveform tier says
    got = ex.wavenet_eps(x_t, int(z['t'])).cpu().numpy()
    assert 1e-5 < relmax(got, ref) < F16_MAX_TOL
    ex.set_mode(E.MODE_FP32)                                  # DMAD_MODE_FP32: the engine's exact-fp32 path (chunks of 4)
    got = ex.wavenet_eps(x_t, int(z['t'])).cpu().numpy()
    ex.set_mode(E.MODE_EXACT_VOTES)
    assert relmax(got, ref) < FP32_TOL


def test_wavenet_batch_and_position_independence(engines):
    """clips must not bleed into each other (per-clip zero padding) and results must not depend on the batch slot."""
    x = torch.randn(5, 16000, generator=torch.Generator().manual_seed(3)).cuda() * 0.3
    for name in ('fp32', 'bf16', 'exact'):
        eng = engines[name]
        full = eng.wavenet_eps(x, 33)
        solo = eng.wavenet_eps(x[3:4], 33)
        assert torch.equal(full[3:4], solo), name
        rev = eng.wavenet_eps(x.flip(0), 33).flip(0)
        assert torch.equal(full, rev), name


def test_samplers_vs_reference_fixture(engines, sched, golden_dir):
    z = G(golden_dir, 'samplers.npz')
    hp, coef = sched
    x_t = torch.from_numpy(z['x_t']).cuda()
    for name, tol in (('fp32', FP32_TOL), ('bf16', BF16_MAX_TOL), ('exact', 1e-4)):      # 'exact': the DEFAULT engine's waveform tier
        got = engines[name].one_shot(x_t, 65, *coef(65)).cpu().numpy()
        assert relmax(got, z['one_shot_t66'][:, 0]) < tol, name


def test_default_engine_waveform_surfaces_are_fp32_grade(engines, golden_dir):
    """north_star: "purified waveforms match within a stated fp32 tolerance".  On the DEFAULT (exact-vote) engine the mirrors'
    waveform-returning surfaces — DiffWave.forward, one_shot_denoise, two_shot_denoise, compute_eps_t (diffwave_ddpm.py:36-47,
    166-182,184-226) — run the split-f16 tier: within 1e-4 of the reference fixtures (the fp32 engine holds 2e-5 / 5e-5, the 16-bit
    tier 4e-3)."""
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    z, z2 = G(golden_dir, 'samplers.npz'), G(golden_dir, 'samplers2.npz')
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    eng = engines['exact']
    den = DiffWave(WaveNetHIP(eng), hp, reverse_timestep=66)
    x_t = torch.from_numpy(z['x_t']).cuda()
    assert relmax(den.one_shot_denoise(x_t).cpu().numpy(), z['one_shot_t66']) < 1e-4
    assert relmax(den.two_shot_denoise(x_t).cpu().numpy(), z['two_shot_t66']) < 1e-4
    for tstar in (3, 5):
        d = DiffWave(WaveNetHIP(eng), hp, reverse_timestep=tstar, noise_source='torch_cpu')
        torch.manual_seed(100 + tstar)
        assert relmax(d(torch.from_numpy(z['x0']).cuda()).cpu().numpy(), z['ddpm_t%d' % tstar]) < 1e-4, tstar
    d9 = DiffWave(WaveNetHIP(eng), hp, reverse_timestep=9, noise_source='torch_cpu')
    torch.manual_seed(501)
    x9 = d9._diffusion(torch.from_numpy(z2['x0']).cuda())
    assert relmax(d9.compute_eps_t(x9, 8).cpu().numpy(), z2['eps_t9']) < 1e-4


@pytest.mark.parametrize('tstar', [3, 5])
def test_ddpm_purify_vs_reference_fixture(engines, golden_dir, tstar):
    """DiffWave.forward = _diffusion + _reverse with the reference's own CPU noise draws (BASELINE configs 1-2)."""
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    z = G(golden_dir, 'samplers.npz')
    for name, tol in (('fp32', 5e-5), ('bf16', BF16_MAX_TOL)):
        den = DiffWave(WaveNetHIP(engines[name]), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG),
                       reverse_timestep=tstar, noise_source='torch_cpu')
        torch.manual_seed(100 + tstar)
        got = den(torch.from_numpy(z['x0']).cuda()).cpu().numpy()
        assert got.shape == (1, 1, 16000)
        assert relmax(got, z['ddpm_t%d' % tstar]) < tol, name


def test_two_shot_and_coefficients(engines, golden_dir):
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    z = G(golden_dir, 'samplers.npz')
    den = DiffWave(WaveNetHIP(engines['fp32']), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=66)
    got = den.two_shot_denoise(torch.from_numpy(z['x_t']).cuda()).cpu().numpy()
    assert relmax(got, z['two_shot_t66']) < 5e-5
    got = den.one_shot_denoise(torch.from_numpy(z['x_t']).cuda()).cpu().numpy()
    assert relmax(got, z['one_shot_t66']) < FP32_TOL


def test_sampler_variants_and_reffwave_vs_reference_fixture(engines, golden_dir):
    """DiffWave.fast_reverse / _predict_x1_from_eps / _predict_x0_from_x1 and ReffWave.forward through the host
    mirror, with the reference's own CPU noise stream (torch.manual_seed + torch.normal, as the generator did)."""
    from diffusion_models.diffwave_ddpm import DiffWave, ReffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    z = G(golden_dir, 'samplers2.npz')
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    x0 = torch.from_numpy(z['x0']).cuda()
    for name, tol in (('fp32', 5e-5), ('bf16', BF16_MAX_TOL)):
        den = DiffWave(WaveNetHIP(engines[name]), hp, reverse_timestep=9, noise_source='torch_cpu')
        torch.manual_seed(501)
        x_t = den._diffusion(x0)
        assert relmax(x_t.cpu().numpy(), z['fast_t9_x_t']) < 1e-6, name
        got = den.fast_reverse(x_t)
        assert got.shape == (1, 1, 16000)
        assert relmax(got.cpu().numpy(), z['fast_t9']) < tol, name
        eps = den.compute_eps_t(x_t, 8)
        assert relmax(eps.cpu().numpy(), z['eps_t9']) < tol, name
        eps_ref = torch.from_numpy(z['eps_t9']).cuda()
        assert relmax(den._predict_x0_from_eps(x_t, 8, eps_ref).cpu().numpy(), z['x0_from_eps_t9']) < 1e-5, name
        assert relmax(den._predict_x1_from_eps(x_t, 8, eps_ref).cpu().numpy(), z['x1_t9']) < 1e-5, name
        got = den._predict_x0_from_x1(torch.from_numpy(z['x1_t9']).cuda())
        assert relmax(got.cpu().numpy(), z['x0_from_x1_t9']) < tol, name

        rw = ReffWave(WaveNetHIP(engines[name]), hp, reverse_timestep=4, num_re=3, noise_source='torch_cpu')
        torch.manual_seed(502)
        got = rw(x0)
        assert got.shape == (1, 1, 16000)
        assert relmax(got.cpu().numpy(), z['reff_t4_n3']) < tol, name
    # device-noise mode: deterministic per (seed, sample counter), finite
    rw = ReffWave(WaveNetHIP(engines['bf16']), hp, reverse_timestep=4, num_re=2, seed=9)
    a = rw(x0); rw._draws = 0
    b = rw(x0)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())


# ------------------------------------------------------------------------------------------ votes
def _ref_noise(seed, sigma, batches):
    torch.manual_seed(seed)
    return torch.cat([torch.normal(0, sigma, size=(b, 1, 16000)) for b in batches])


def test_smooth_votes_match_reference_loop_vgg(engines, sched, golden_dir):
    """counts from the reference's own smooth_predict (synthetic VGG19_bn, seeded CPU noise)."""
    z = G(golden_dir, 'smooth_predict.npz')
    hp, coef = sched
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
    delta = _ref_noise(int(z['vgg_seed']), 0.5, (16, 16, 8)).cuda()
    sc = float(torch.tensor((1 / 1.25) ** 0.5))
    ref = z['vgg_logits']
    srt = np.sort(ref, 1)
    margin = srt[:, -1] - srt[:, -2]
    counts, logits, _ = engines['fp32'].smooth_votes(clip, 0.5, sc, 65, *coef(65), 40, batch=6, delta=delta, want_logits=True)
    assert counts.cpu().tolist() == z['vgg_counts'].tolist()                  # bit-exact votes
    assert np.abs(logits.cpu().numpy() - ref).max() < 1e-3
    counts, logits, _ = engines['bf16'].smooth_votes(clip, 0.5, sc, 65, *coef(65), 40, batch=5, delta=delta, want_logits=True)
    lg = logits.cpu().numpy()
    err = np.abs(lg - ref).max(1)
    flips = lg.argmax(1) != ref.argmax(1)
    assert int(counts.sum()) == 40
    assert not (flips & (margin > 2 * err)).any()          # a vote may flip only inside the bf16 error band
    assert flips.sum() <= 3 and err.max() < 0.25
    assert np.abs(counts.cpu().numpy() - z['vgg_counts']).sum() <= 2 * flips.sum()
    # exact-vote mode (f16 path + fp32 recheck of the close votes) on the reference's own noise: the reference's counts
    ex = engines['exact']
    ex.recheck_stats(reset=True)
    counts, logits, _ = ex.smooth_votes(clip, 0.5, sc, 65, *coef(65), 40, batch=6, delta=delta, want_logits=True)
    assert counts.cpu().tolist() == z['vgg_counts'].tolist()
    voted, rechecked = ex.recheck_stats()
    assert voted == 40 and 0 <= rechecked <= 40
    lg = logits.cpu().numpy()
    srt = np.sort(lg, 1)
    close = (srt[:, -1] - srt[:, -2]) < ex.recheck_margin          # rows that were re-evaluated carry the fp32 logits
    assert (lg.argmax(1) == ref.argmax(1)).all() and np.abs(lg - ref)[close].max(initial=0) < 1e-3


def test_robust_certificate_m5_generic_path(engines, golden_dir, tmp_path):
    """RobustCertificate with a torch-module classifier (M5, real weights) behind the HIP purifier,
    noise_source='torch_cpu': counts equal the reference's (fixture smooth_predict.npz)."""
    z = G(golden_dir, 'smooth_predict.npz')
    from audio_models.ConvNets_SpeechCommands.create_model import create_model
    import M5Net  # noqa: F401  (resolved via create_model's sys.path entry)
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from robustness_eval.certified_robust import RobustCertificate
    m5 = M5Net.M5(n_input=1, first_kernel_size=160, n_output=10, stride=16, n_channel=32)
    m5.load_state_dict({k: torch.from_numpy(v) for k, v in G(golden_dir, 'm5_k160_state.npz').items()})
    path = str(tmp_path / 'm5.pth')
    torch.save(m5, path)
    clf = create_model(path).cuda()
    den = DiffWave(WaveNetHIP(engines['fp32']), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG))
    rc = RobustCertificate(classifier=clf, transform=None, denoiser=den, noise_source='torch_cpu')
    torch.manual_seed(int(z['m5_seed']))
    counts = rc.smooth_predict(torch.from_numpy(synth.synthetic_clip(0)).cuda(), num_sampling=48, sigma=0.5, batch_size=16)
    assert den.reverse_timestep == 66
    assert counts.dtype == torch.int64 and counts.device.type == 'cpu'
    assert counts.tolist() == z['m5_counts'].tolist()


def test_config1_composite_m5_behind_ddpm_purifier(engines, golden_dir):
    """BASELINE C1 as ONE composition: AcousticSystem(M5 (bundled k=160 weights), transform=None, defender=DiffWave(t*=3)) — the
    HIP DDPM purifier on the reference's CPU noise stream in front of the caller's torch classifier — against the logits /
    purified waveform of the imported reference's own AcousticSystem on the same seed (fixture c1_composite.npz,
    tests/golden/make_golden_c1.py; reference acoustic_system.py:27-51, diffwave_ddpm.py:36-104, M5Net.py:21-38); also the
    undefended call and the int16-range rescale branch."""
    import M5Net
    from acoustic_system import AcousticSystem
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    z = G(golden_dir, 'c1_composite.npz')
    m5 = M5Net.M5(n_input=1, first_kernel_size=160, n_output=10, stride=16, n_channel=32)
    m5.load_state_dict({k: torch.from_numpy(v) for k, v in G(golden_dir, 'm5_k160_state.npz').items()})
    m5 = m5.float().eval().cuda()
    x = torch.from_numpy(z['x']).cuda()
    for name, tol_w, tol_l in (('fp32', 5e-5, 2e-3), ('bf16', BF16_MAX_TOL, 0.5)):
        den = DiffWave(WaveNetHIP(engines[name]), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=int(z['t_star']),
                       noise_source='torch_cpu')
        system = AcousticSystem(classifier=m5, transform=None, defender=den, defense_type='wave').eval()
        with torch.no_grad():
            torch.manual_seed(int(z['seed']))
            pur = den(x)
            torch.manual_seed(int(z['seed']))
            logp = system(x)
            plain = system(x, defend=False)
            torch.manual_seed(int(z['seed']))
            logp_i = system(torch.from_numpy(z['x_int16']).cuda())
        assert relmax(pur.cpu().numpy(), z['purified']) < tol_w, name
        assert logp.shape == (1, 10) and float(np.abs(logp.cpu().numpy() - z['logp']).max()) < tol_l, name
        assert float(np.abs(plain.cpu().numpy() - z['logp_undefended']).max()) < 1e-4
        assert float(np.abs(logp_i.cpu().numpy() - z['logp_int16']).max()) < max(tol_l, 5e-3), name
        if name == 'fp32':
            assert logp.argmax(1).tolist() == z['logp'].argmax(1).tolist()


def test_certify_end_to_end_fused(engines, golden_dir):
    """certify() through the fused HIP loop with the reference's CPU noise -> same (y_pred, radius)."""
    z = G(golden_dir, 'smooth_predict.npz')
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    eng = engines['fp32']
    den = DiffWave(WaveNetHIP(eng), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG))
    clf = synth_vgg().bind_engine(eng)
    rc = RobustCertificate(classifier=clf, transform=MelSpectrogramDB(eng), denoiser=den, noise_source='torch_cpu')
    assert rc._fused()
    torch.manual_seed(int(z['certify_seed']))
    x = torch.from_numpy(synth.synthetic_clip(0))[None].cuda()
    y_pred, radius = rc.certify(x, torch.tensor([3]).cuda(), sigma=0.25, n_0=16, n=32, batch_size=16)
    assert den.reverse_timestep == 34
    assert y_pred.tolist() == z['certify_ypred'].tolist()
    assert abs(float(radius[0]) - float(z['certify_radius'][0])) < 1e-6
    assert radius.dtype == torch.float32 and y_pred.dtype == torch.int64


def test_votes_world_size_invariance_and_batch_invariance(engines, sched):
    """device (Philox) noise: counts are a function of (seed, sample range) only."""
    hp, coef = sched
    eng = engines['bf16']
    clip = torch.from_numpy(synth.synthetic_clip(1)).cuda()
    sc = float(torch.tensor((1 / 1.25) ** 0.5))
    whole, lg, _ = eng.smooth_votes(clip, 0.5, sc, 65, *coef(65), 24, batch=6, seed=11, sample0=0, want_logits=True)
    parts = torch.zeros_like(whole)
    lgs = []
    for lo, hi, b in ((0, 6, 2), (6, 12, 6), (12, 24, 5)):                     # 3 "ranks", different batch sizes
        c, l, _ = eng.smooth_votes(clip, 0.5, sc, 65, *coef(65), hi - lo, batch=b, seed=11, sample0=lo, want_logits=True)
        parts += c
        lgs.append(l)
    assert torch.equal(whole, parts)
    assert torch.equal(lg, torch.cat(lgs))


def test_full_size_properties(engines, sched):
    """size-independent properties at a larger batch: vote conservation, determinism, sigma -> t* plumbing."""
    hp, coef = sched
    eng = engines['bf16']
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
    sc = float(torch.tensor((1 / 1.25) ** 0.5))
    a, _, _ = eng.smooth_votes(clip, 0.5, sc, 65, *coef(65), 60, seed=5)
    b, _, _ = eng.smooth_votes(clip, 0.5, sc, 65, *coef(65), 60, seed=5)
    assert int(a.sum()) == 60 and torch.equal(a, b)
    c, _, _ = eng.smooth_votes(clip, 0.5, sc, 65, *coef(65), 0, seed=5)
    assert int(c.sum()) == 0                                                    # empty input


def test_reference_driver_surfaces_through_shims(engines, golden_dir, tmp_path, monkeypatch):
    """The certification driver's own construction sequence (certified_robustness_eval.py:52-96), with the opt-in
    torchaudio / torchvision shims: create_model -> create_diffwave_model -> Compose([MelSpectrogram, AmplitudeToDB])
    -> RobustCertificate.certify, on synthetic checkpoints written in the reference's formats."""
    import json
    import sys
    shim_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'diffusion-model-for-audio-defense_amd', 'shims')
    monkeypatch.syspath_prepend(shim_dir)
    for m in ('torchaudio', 'torchaudio.transforms', 'torchvision', 'torchvision.transforms'):
        monkeypatch.delitem(sys.modules, m, raising=False)
    import torchaudio
    from torchvision.transforms import Compose
    from audio_models.ConvNets_SpeechCommands.create_model import create_model
    from models.vgg import vgg19_bn          # the module path the reference's checkpoints were pickled under
    from diffusion_models.diffwave_ddpm import create_diffwave_model
    from dmad_hip import engine as E
    from robustness_eval.certified_robust import RobustCertificate
    eng = engines['fp32']
    for prec in (E.BF16, E.FP32, E.EXACT):
        monkeypatch.setitem(E._ENGINES, (torch.cuda.current_device(), prec), eng)   # what get_engine() hands out
    # checkpoints in the reference's on-disk formats
    d = tmp_path / 'ConvNets_SpeechCommands'
    d.mkdir()
    net = vgg19_bn(num_classes=10, in_channels=1)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(4321).items()})
    torch.save(torch.nn.DataParallel(net), str(d / 'vgg.pth'))
    torch.save({'model_state_dict': {k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(1234).items()},
                'optimizer_state_dict': {}}, str(tmp_path / '1000000.pkl'))
    cfg = str(tmp_path / 'config.json')
    json.dump({'wavenet_config': synth.WAVENET_CONFIG, 'diffusion_config': synth.DIFFUSION_CONFIG}, open(cfg, 'w'))

    Classifier = create_model(str(d / 'vgg.pth'))
    Classifier.cuda()
    DiffWave_Denoiser = create_diffwave_model(model_path=str(tmp_path / '1000000.pkl'), config_path=cfg, engine=eng)
    DiffWave_Denoiser.eval().cuda()
    MelSpecTrans = torchaudio.transforms.MelSpectrogram(n_fft=2048, hop_length=512, n_mels=32, norm='slaney', pad_mode='constant', mel_scale='slaney')
    Amp2DB = torchaudio.transforms.AmplitudeToDB(stype='power')
    Wave2Spect = Compose([MelSpecTrans.cuda(), Amp2DB.cuda()])
    x = torch.from_numpy(np.stack([synth.synthetic_clip(0), synth.synthetic_clip(3)])).cuda()
    ref = eng.mel_db(x)
    assert torch.equal(Wave2Spect(x), ref)                                # two-stage shim == fused transform
    RC = RobustCertificate(classifier=Classifier, transform=Wave2Spect, denoiser=DiffWave_Denoiser, noise_source='torch_cpu')
    assert RC._fused()
    z = G(golden_dir, 'smooth_predict.npz')
    torch.manual_seed(int(z['certify_seed']))
    y_certified, r_certified = RC.certify(x=x[:1], y=torch.tensor([3]).cuda(), sigma=0.25, n_0=16, n=32, batch_size=16)
    assert y_certified.tolist() == z['certify_ypred'].tolist()
    assert abs(float(r_certified[0]) - float(z['certify_radius'][0])) < 1e-6


# ------------------------------------------------------------------------------------------ driver + host ingest (N2)
def test_certification_driver_end_to_end(engines, tmp_path, monkeypatch):
    """certified_robustness_eval.run: WAV folder -> SC09Dataset/LoadAudio/FixAudioLength -> DataLoader ->
    RobustCertificate.certify (fused HIP loop) -> JSON records in the reference's layout; deterministic under
    torch.manual_seed; --resume continues after the last stored record."""
    import json
    import wave
    import certified_robustness_eval as drv
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from datasets.sc_dataset import SC09_CLASSES
    from diffusion_models.diffwave_ddpm import create_diffwave_model
    eng = engines['bf16']
    data = tmp_path / 'test'
    for ci, c in enumerate(SC09_CLASSES):
        (data / c).mkdir(parents=True)
        clip = synth.synthetic_clip(ci)[0][: 16000 - 500 * ci]               # ragged lengths: FixAudioLength pads
        with wave.open(str(data / c / 'a.wav'), 'wb') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(np.clip(np.round(clip * 32768.0), -32768, 32767).astype('<i2').tobytes())
    cfg = str(tmp_path / 'config.json')
    json.dump({'wavenet_config': synth.WAVENET_CONFIG, 'diffusion_config': synth.DIFFUSION_CONFIG}, open(cfg, 'w'))
    den = create_diffwave_model(None, cfg, state_dict=synth.wavenet_state_dict(1234), engine=eng)
    net = vgg19_bn(num_classes=10, in_channels=1)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(4321).items()})
    net.eval()
    args = drv.build_parser().parse_args(['--data_path', str(data), '--num_per_class', '1', '--config', cfg, '--sigma', '0.5',
                                          '--num_sampling', '28', '--batch_size', '4', '--dataload_workers_nums', '0',
                                          '--save_path', str(tmp_path / 'records')])
    torch.manual_seed(3)
    recs = drv.run(args, classifier=net, denoiser=den, log=lambda *_: None)
    path = tmp_path / 'records' / 'sigma=0.5' / 'sigma=0.5_N=28.json'
    assert json.load(open(path)) == recs and len(recs) == 10
    assert [r['id'] for r in recs] == list(range(10)) and [r['y_true'] for r in recs] == list(range(10))
    assert all(r['y_pred'] in range(-1, 10) and r['certified_radius'] >= 0 for r in recs)
    assert all((r['y_pred'] == -1) == (r['certified_radius'] == 0) for r in recs)
    torch.manual_seed(3)
    assert drv.run(args, classifier=net, denoiser=den, log=lambda *_: None) == recs
    json.dump(recs[:6], open(path, 'w'))                                     # an interrupted run: 6 of 10 stored
    args.resume = True
    again = drv.run(args, classifier=net, denoiser=den, log=lambda *_: None)
    assert len(again) == 10 and again[:6] == recs[:6] and [r['id'] for r in again] == list(range(10))
    assert [r['y_true'] for r in again] == list(range(10))


def test_certification_driver_audit_and_calibration(exact_engine, tmp_path):
    """The driver on an exact-vote engine (the drop-in default): --calibrate_margins measures the recheck bounds on the first
    --calibrate_clips clips (they can only widen) and logs what it saw; --audit k re-evaluates k tier-1 voters per example on
    the split-f16 tier and writes the outcome into each record under "audit"; the reference's four keys are unchanged."""
    import json
    import wave
    import certified_robustness_eval as drv
    from datasets.sc_dataset import SC09_CLASSES
    from diffusion_models.diffwave_ddpm import create_diffwave_model
    from dmad_hip import engine as E
    eng = exact_engine
    data = tmp_path / 'test'
    for ci, c in enumerate(SC09_CLASSES):
        (data / c).mkdir(parents=True)
        with wave.open(str(data / c / 'a.wav'), 'wb') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(np.clip(np.round(synth.synthetic_clip(ci)[0] * 32768.0), -32768, 32767).astype('<i2').tobytes())
    cfg = str(tmp_path / 'config.json')
    json.dump({'wavenet_config': synth.WAVENET_CONFIG, 'diffusion_config': synth.DIFFUSION_CONFIG}, open(cfg, 'w'))
    den = create_diffwave_model(None, cfg, state_dict=synth.wavenet_state_dict(1234), engine=eng)
    args = drv.build_parser().parse_args(['--data_path', str(data), '--num_per_class', '1', '--config', cfg, '--sigma', '0.5',
                                          '--num_sampling', '24', '--batch_size', '4', '--dataload_workers_nums', '0',
                                          '--save_path', str(tmp_path / 'records'), '--calibrate_margins', '32', '--calibrate_clips', '2',
                                          '--audit', '8'])
    old = (eng.recheck_margin, eng.recheck_margin2)
    lines = []
    try:
        recs = drv.run(args, classifier=synth_vgg(), denoiser=den, log=lines.append)
    finally:
        eng.set_recheck_margin(old[0]); eng.set_recheck_margin2(old[1])
    assert len(recs) == 10 and all(set(r) == {'id', 'y_true', 'y_pred', 'certified_radius', 'audit'} for r in recs)
    for r in recs:
        a = r['audit']
        assert a['audited'] == 8 and 0 <= a['voted_on_tier1'] <= 8 and a['disagreements'] == [] and a['tau1'] >= E.DEFAULT_RECHECK_MARGIN[E.HALF_F16]
    cal = [ln for ln in lines if ln.startswith('recheck bounds')]
    assert len(cal) == 2 and 'clip 2 of 2' in cal[1] and sum(ln.startswith('audit:') for ln in lines) == 10
    assert 'left the 16-bit tier' in [ln for ln in lines if ln.startswith('certified')][-1]
    assert json.load(open(tmp_path / 'records' / 'sigma=0.5' / 'sigma=0.5_N=24.json')) == recs


# ------------------------------------------------------------------------------------------ batched query path (N3)
def test_eot_nes_query_path(engines, orc):
    """AcousticSystem(defender = t*-step DDPM purifier) queried through the EOT / NES wrappers of the black-box drivers
    (reference _EOT.py:19-69, _NES.py:15-55): repeated batches through the HIP path, checked against the oracle's
    purify -> mel-dB -> VGG chain driven by the same CPU noise stream."""
    from acoustic_system import AcousticSystem
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval._EOT import EOT
    from robustness_eval._NES import NES
    from robustness_eval._utils import resolve_loss, resolve_prediction
    eng = engines['fp32']
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    den = DiffWave(WaveNetHIP(eng), hp, reverse_timestep=2, noise_source='torch_cpu')
    net = vgg19_bn(num_classes=10, in_channels=1)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(4321).items()})
    net.eval().bind_engine(eng)
    model = AcousticSystem(classifier=net, transform=MelSpectrogramDB(eng), defender=den, defense_type='wave')
    loss_fn, grad_sign = resolve_loss('Margin', False, 0., 'SCR', None, False)
    assert grad_sign == 1
    x = torch.from_numpy(np.stack([synth.synthetic_clip(0), synth.synthetic_clip(5)])).cuda()
    y = torch.tensor([0, 6]).cuda()

    # one EOT batch of two repeats == the oracle chain on the repeated batch with the same CPU draws
    eot = EOT(model, loss_fn, EOT_size=2, EOT_batch_size=2, use_grad=False)
    torch.manual_seed(77)
    scores, loss, grad, decisions = eot(x, y)
    assert grad is None and scores.shape == (2, 10) and loss.shape == (2,) and [len(d) for d in decisions] == [2, 2]
    torch.manual_seed(77)
    w = orc.folded_weights(synth.wavenet_state_dict(1234))
    oden = orc.DiffWaveOracle(w, orc.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=2)
    pur = oden.forward(x.cpu().repeat(2, 1, 1))
    ref_logits = orc.vgg19_bn_forward(synth.vgg19_bn_state_dict(4321), orc.mel_db(pur))
    ref_scores = ref_logits.view(2, 2, 10).mean(0)
    assert float((scores.cpu() - ref_scores).abs().max()) < 2e-3 * float(ref_scores.abs().max())
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, y.cpu().repeat(2), reduction='none').view(2, 2).mean(0)
    assert float((loss.cpu() - ref_loss).abs().max()) < 2e-3 * max(1.0, float(ref_loss.abs().max()))
    assert [list(map(int, d)) for d in decisions] == ref_logits.argmax(1).view(2, 2).t().tolist()
    assert resolve_prediction(decisions).shape == (2,)

    # several EOT batches accumulate the per-batch means (ref l.46-55)
    eot4 = EOT(model, loss_fn, EOT_size=4, EOT_batch_size=2, use_grad=False)
    torch.manual_seed(5)
    s4, l4, _, d4 = eot4(x, y)
    torch.manual_seed(5)
    parts = [model(x.repeat(2, 1, 1)) for _ in range(2)]
    want = sum(p.view(2, 2, 10).mean(0) for p in parts) / 2
    assert torch.allclose(s4, want, rtol=0, atol=1e-6) and [len(d) for d in d4] == [4, 4]
    with pytest.raises(NotImplementedError):
        EOT(model, loss_fn, 2, 2, use_grad=True)(x, y)          # this WaveNetHIP was built without weights for the gradient branch

    # NES: shapes, finiteness, the unperturbed probe first (ref l.20-21,41-47)
    nes = NES(samples_per_draw=4, samples_per_draw_batch=4, sigma=1e-3, EOT_wrapper=EOT(model, loss_fn, 1, 1, False))
    torch.manual_seed(9)
    mean_loss, g, adver_loss, adver_score, predict = nes(x, y)
    assert mean_loss.shape == (2,) and g.shape == x.shape and adver_loss.shape == (2,) and adver_score.shape == (2, 10)
    assert predict.shape == (2,) and bool(torch.isfinite(g).all()) and bool(torch.isfinite(mean_loss).all())


# ------------------------------------------------------------------------------------------ ResNeXt29 (N4)
def test_resnext29_vs_reference_fixture(golden_dir, orc):
    """CifarResNeXt 8x64d (the certification script's default classifier) on the fp32 matrix cores vs logits of the
    imported reference class on the same seeded weights; batch-size invariance; the fused Monte Carlo loop with it."""
    from audio_models.ConvNets_SpeechCommands.models.resnext import CifarResNeXt
    from dmad_hip import engine as E
    from dmad_hip.transforms import MelSpectrogramDB
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from robustness_eval.certified_robust import RobustCertificate
    z = G(golden_dir, 'resnext29.npz')
    sd = synth.resnext29_state_dict(int(z['seed']))
    eng = E.Engine(max_batch=8, precision=E.BF16)
    eng.load_wavenet(synth.wavenet_state_dict(1234))
    net = CifarResNeXt(nlabels=10, in_channels=1)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval().bind_engine(eng)
    spec = torch.from_numpy(z['spec_in']).cuda()
    got = net(spec)
    assert got.shape == (4, 10)
    assert relmax(got.cpu().numpy(), z['logits']) < 2e-4
    solo = net(spec[2:3])
    assert torch.equal(solo, got[2:3])                                   # a sample's logits do not depend on its batch
    big = net(spec.repeat(5, 1, 1, 1))                                    # 20 > max_batch: chunked
    assert torch.equal(big[:4], got) and torch.equal(big[16:], got)
    # random spectrograms against the oracle restatement
    g = torch.Generator().manual_seed(3)
    rs = (torch.randn(3, 1, 32, 32, generator=g) * 15 - 25)
    ref = orc.resnext29_forward(sd, rs).numpy()
    assert relmax(net(rs.cuda()).cpu().numpy(), ref) < 2e-4
    # fused certified-smoothing loop with the ResNeXt classifier
    den = DiffWave(WaveNetHIP(eng), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=66)
    RC = RobustCertificate(classifier=net, transform=MelSpectrogramDB(eng), denoiser=den, seed=11)
    assert RC._fused()
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()              # [1,16000], what certify() hands over
    counts = RC.smooth_predict(clip, num_sampling=24, sigma=0.5, batch_size=8)
    assert int(counts.sum()) == 24 and counts.shape == (10,)



def test_resnext29_16bit_tier_and_exact_votes(golden_dir, weights, sched):
    """The reference script's DEFAULT classifier (certified_robustness_eval.py:57; models/resnext.py:23-142) with its calibrated
    synthetic weights (tests/golden/make_classifier_calib.py: votes spread over many classes, margins of order one).
    (1) its 16-bit tier, dmad_classify_tier(1) — every conv on f16 operands through gemm_h16 (grouped 3x3, stride 2, 1x1 with the BN
    scale folded into the weights) — within the f16 tolerance of the logits of the imported reference class, batch invariant, and
    really another arithmetic than the fp32 tier; dmad_classify itself stays the fp32 tier.
    (2) the exact-vote loop: counts == the all-fp32 counts, bit for bit, on 3 clips x 3 sigmas x 512 = 4 608 samples, and NOT
    vacuously: every cell votes at least three classes, no class takes more than 70 %, samples ARE rechecked (on both recheck
    tiers over the run), the rows that reached fp32 are the fp32 path's bit for bit.  Since round 5 the first pass of the exact-vote
    mode runs the classifier's SPLIT-F16 tier (fp32-grade) and the recheck tiers the fp32 one: the f16 classifier's leader-difference
    error, measured here too, is several times the f16 WaveNet's and does NOT fit under the bound; the fast mode runs it."""
    from dmad_hip import engine as E
    z = G(golden_dir, 'resnext29.npz')
    sd = synth.resnext29_state_dict(int(z['seed']))
    eng = E.Engine(max_batch=64, precision=E.EXACT, recheck_batch=32)
    eng.load_wavenet(weights[0])
    eng.load_resnext29(sd)
    assert eng.recheck_margin == E.DEFAULT_RECHECK_MARGIN_RESNEXT29[E.HALF_F16]      # the committed bound of this classifier kind
    spec = torch.from_numpy(z['spec_in']).cuda()
    l32, l16 = eng.classify_tier(spec, 0), eng.classify_tier(spec, 1)
    assert torch.equal(l32, eng.classify(spec))                                   # dmad_classify = the fp32 tier
    assert relmax(l32.cpu().numpy(), z['logits']) < 2e-4
    e16 = relmax(l16.cpu().numpy(), z['logits'])
    assert 1e-6 < e16 < F16_MAX_TOL, e16
    assert torch.equal(eng.classify_tier(spec[1:2], 1), l16[1:2])                 # a sample's logits do not depend on its batch
    big = eng.classify_tier(spec.repeat(40, 1, 1, 1), 1)                          # 160 > max_batch: chunked; other tile shapes
    assert torch.equal(big[:4], l16) and torch.equal(big[156:], l16)
    # (1b) its split-f16 tier (tier 1 of the exact-vote loop): every conv as three f16 MFMAs per product, grouped 3x3 included — fp32-grade
    lx3 = eng.classify_tier(spec, 2)
    ex3 = relmax(lx3.cpu().numpy(), z['logits'])
    assert 0 < ex3 < 2e-4 and not torch.equal(lx3, l32), ex3                      # the fp32 tier's own tolerance
    assert torch.equal(eng.classify_tier(spec[2:3], 2), lx3[2:3])
    bigx = eng.classify_tier(spec.repeat(40, 1, 1, 1), 2)
    assert torch.equal(bigx[:4], lx3) and torch.equal(bigx[156:], lx3)
    hp, coef = sched
    ab = hp['Alpha_bar']
    N, worst, worst_fast, rechecked, reached_fp32 = 512, 0.0, 0.0, 0, 0
    for ci in (0, 1, 2):
        clip = torch.from_numpy(synth.synthetic_clip(ci)).cuda()
        for sigma in (0.25, 0.5, 1.0):
            t = int(torch.abs(ab - 1 / (1 + sigma ** 2)).min(0, keepdim=True)[1].item())
            sc = float(torch.tensor((1 / (1 + sigma ** 2)) ** 0.5, dtype=torch.float32))
            out = {}
            for mode in (E.MODE_FAST, E.MODE_FP32, E.MODE_EXACT_VOTES):
                eng.set_mode(mode)
                eng.recheck_stats(reset=True)
                c, l, _ = eng.smooth_votes(clip, sigma, sc, t, *coef(t), N, seed=700 + ci, sample0=9000, want_logits=True)
                out[mode] = (c.cpu().tolist(), l.cpu().numpy().astype(np.float64), eng.recheck_stats(detail=True))
            fast, f32, ex = out[E.MODE_FAST], out[E.MODE_FP32], out[E.MODE_EXACT_VOTES]
            assert sum(fast[0]) == sum(f32[0]) == sum(ex[0]) == N
            assert np.isfinite(fast[1]).all() and np.isfinite(f32[1]).all() and np.isfinite(ex[1]).all()      # (a NaN would slip through every max() below)
            assert sum(1 for v in f32[0] if v > 0) >= 3 and max(f32[0]) <= 0.7 * N, f32[0]          # a non-degenerate stand-in
            assert ex[0] == f32[0], (ci, sigma, ex[0], f32[0])
            assert (ex[1].argmax(1) == f32[1].argmax(1)).all()
            # tier 1 of the exact-vote mode = f16 WaveNet + split-f16 classifier: its rows where nothing was rechecked
            idx = torch.arange(N, dtype=torch.int64, device='cuda') + 9000
            eng.set_mode(E.MODE_EXACT_VOTES)
            t1 = eng.eval_samples(clip, sigma, sc, t, *coef(t), idx, path=0, seed=700 + ci).cpu().numpy().astype(np.float64)
            e = t1 - f32[1]
            worst = max(worst, float(np.abs(e - e[np.arange(N), f32[1].argmax(1)][:, None]).max()))
            ef = fast[1] - f32[1]
            worst_fast = max(worst_fast, float(np.abs(ef - ef[np.arange(N), f32[1].argmax(1)][:, None]).max()))
            srt = np.sort(t1.astype(np.float32), 1)
            low = ~((srt[:, -1] - srt[:, -2]) >= np.float32(eng.recheck_margin))
            assert ex[2][:2] == (N, int(low.sum()))
            assert (ex[1][~low] == t1[~low]).all()                                # what voted on tier 1 kept its tier-1 logits
            same = (ex[1] == f32[1]).all(1)
            assert int(same.sum()) >= ex[2][2]                                    # the rows that reached fp32 are the fp32 path's bit for bit
            rechecked += ex[2][1]
            reached_fp32 += ex[2][2]
    assert rechecked >= 0.01 * 9 * N, rechecked                                   # the recheck hand-over is exercised ...
    assert reached_fp32 >= 1                                                      # ... down to the exact-fp32 tier
    assert worst < eng.recheck_margin, (worst, eng.recheck_margin)                # the bound covers tier 1's error with the split-f16 classifier
    assert worst_fast > eng.recheck_margin, worst_fast                            # ... and would NOT cover the f16 classifier's (why it is FAST-only)
    # calibration measures the loop's FIRST PASS (f16 WaveNet + split-f16 classifier), not the fast mode's f16 classifier: the observed
    # error stays below the committed bound of this classifier kind, which therefore stands (a calibration only widens)
    t = int(torch.abs(ab - 1 / 1.25).min(0, keepdim=True)[1].item())
    tau1, tau2, e1, e2 = eng.calibrate_recheck(clip, 0.5, float(torch.tensor((1 / 1.25) ** 0.5, dtype=torch.float32)), t, *coef(t), n=128, n_fp32=64)
    floor = E.DEFAULT_RECHECK_MARGIN_RESNEXT29[E.HALF_F16]
    assert 0 < e1 < floor and floor <= tau1 < 0.07, (tau1, e1)                    # (with the f16 classifier in the measured pass e1 would be ~0.1)
    assert 0 < e2 < tau2 and tau2 >= E.DEFAULT_RECHECK_MARGIN2 and eng.mode == E.MODE_EXACT_VOTES
    eng.close()


def _conv_ref(x, x2, w, bias, res, stride, groups, relu):
    """torch fp32 NCHW convolution of the same f16-rounded operands (the fp32 reference of the f16 conv-GEMM family)."""
    xin = x if x2 is None else torch.cat([x, x2], dim=3)
    g_, taps, M, K = w.shape
    k = 3 if taps == 9 else 1
    wt = w.float().reshape(g_, k, k, M, K).permute(0, 3, 4, 1, 2).reshape(g_ * M, K, k, k)       # [groups*M, K, kh, kw]
    y = torch.nn.functional.conv2d(xin.float().permute(0, 3, 1, 2), wt, bias=None if bias is None else bias.float(), stride=stride,
                                   padding=k // 2, groups=g_)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.float()
    return torch.relu(y) if relu else y


H16_CASES = [   # B, H, cin, cout, taps, stride, c1 (two-part input: first map's channels), residual, groups, relu
    (2, 8, 64, 128, 9, 1, 0, False, 1, False), (3, 8, 128, 256, 9, 1, 0, True, 1, False), (2, 16, 128, 128, 9, 2, 0, False, 1, False),
    (5, 4, 256, 256, 9, 1, 0, True, 1, False), (2, 8, 256, 384, 1, 1, 0, False, 1, False), (2, 8, 384, 128, 1, 1, 256, False, 1, False),
    (3, 16, 256, 768, 1, 1, 0, False, 1, False), (2, 8, 512, 256, 1, 1, 256, True, 1, False), (1, 32, 128, 128, 9, 1, 0, True, 1, False),
    # the 256 x 256 register-prefetched tile (>= 256 workgroups): 3x3 with residual, N tail, stride 2, 1x1 with three row blocks
    (256, 16, 256, 256, 9, 1, 0, True, 1, False), (257, 16, 256, 256, 9, 1, 0, True, 1, False), (256, 32, 128, 256, 9, 2, 0, False, 1, False),
    (128, 16, 256, 768, 1, 1, 0, False, 1, False),
    # the 128 x 512 tile (M = 128): 3x3 with residual, N tail, stride 2, K = 384
    (128, 32, 128, 128, 9, 1, 0, True, 1, False), (129, 32, 128, 128, 9, 1, 0, False, 1, False), (512, 32, 128, 128, 9, 2, 0, False, 1, False),
    (128, 32, 384, 128, 9, 1, 0, True, 1, False),
    # two-part (concatenated) input through the persistent tiles: the 1x1 skip conv of an output block, a 3x3 with residual and N tail
    (128, 32, 384, 128, 1, 1, 256, False, 1, False), (257, 16, 512, 256, 9, 1, 256, True, 1, False),
    # the slice-resident 3x3 form on 32-, 8- and 4-pixel-wide maps (tiles spanning 1/4, 4 and 16 images; N tail on the last)
    (64, 32, 256, 256, 9, 1, 0, False, 1, False), (1024, 8, 128, 256, 9, 1, 0, True, 1, False), (4100, 4, 256, 256, 9, 1, 0, False, 1, False),
    # chip-filling 1x1 convs through the ping-pong form: proj with residual and an N tail, K = 512 two-part skip conv
    (130, 16, 256, 256, 1, 1, 0, True, 1, False), (65, 32, 512, 128, 1, 1, 256, False, 1, False),
    # ResNeXt29's forms: K = 64, ReLU, 1x1 with stride 2, grouped 3x3 (4 paired / 8 groups; stride 2; small and chip-filling launches)
    (3, 32, 64, 512, 1, 1, 0, False, 1, True), (2, 32, 256, 512, 1, 2, 0, False, 1, False), (2, 16, 512, 512, 9, 1, 0, False, 4, True),
    (3, 16, 1024, 1024, 9, 2, 0, False, 8, True), (64, 32, 512, 512, 9, 1, 0, False, 4, True), (128, 16, 2048, 2048, 9, 2, 0, False, 8, True),
    (256, 8, 2048, 1024, 1, 1, 0, True, 1, True),
]


@pytest.mark.parametrize('case', H16_CASES, ids=lambda c: 'B%d_H%d_%dto%d_t%d_s%d_c1%d_r%d_g%d_relu%d' % tuple(int(v) for v in c))
def test_gemm_h16_family_vs_torch_conv(case):
    """Every form of the f16 conv-GEMM family (csrc/gemm_h16.hip: the 384-row kernel in both splits, the 256 x 256 and 128 x 512
    persistent tiles (ping-pong and slice-resident), two-part input, stride 2, N tails, grouped convs, ReLU) through dmad_conv_h16 against a torch fp32
    convolution of the same f16-rounded operands (improved_diffusion/unet.py:107-252, models/resnext.py:23-62 are the callers' ops):
    the fp32 output within 2e-3 absolute (fp32 accumulation order only), the f16 twin within one f16 rounding of it."""
    from dmad_hip import engine as E
    B, H, cin, cout, taps, stride, c1, with_res, groups, relu = case
    g = torch.Generator().manual_seed(1000 + B + H + cin)
    Kg, Mg = cin // groups, cout // groups
    x = (torch.rand(B, H, H, cin, generator=g) * 2 - 1).half().cuda()
    w = ((torch.rand(groups, taps, Mg, Kg, generator=g) * 2 - 1) * 0.1).half().cuda()
    bias = (torch.rand(cout, generator=g) * 2 - 1).cuda()
    Ho = (H - 1) // stride + 1
    res = (torch.rand(B, Ho, Ho, cout, generator=g) * 2 - 1).half().cuda() if with_res else None
    xa, xb = (x[..., :c1].contiguous(), x[..., c1:].contiguous()) if c1 else (x, None)
    o32, o16 = E.conv_h16(xa, w, bias, stride=stride, groups=groups, relu=bool(relu), res=res, x2=xb)
    # reference in chunks of the batch (fp32 conv of 512 x 32 x 32 x 128 fits, but keep the test's footprint small)
    worst32 = worst16 = 0.0
    for b0 in range(0, B, 64):
        sl = slice(b0, min(B, b0 + 64))
        ref = _conv_ref(xa[sl], None if xb is None else xb[sl], w, bias, None if res is None else res[sl], stride, groups, bool(relu))
        worst32 = max(worst32, float((o32[sl] - ref).abs().max()))
        worst16 = max(worst16, float((o16[sl].float() - ref).abs().max()))
    assert worst32 < 2e-3, worst32
    assert worst16 < 2e-2, worst16
    assert torch.equal(o16, o32.half())                     # the f16 twin IS the rounded fp32 output



X3_CASES = [   # B, H, cin, cout, taps, stride, c1 (two-part input), residual (2: handed over in the split format), relu, groups
    (2, 32, 128, 128, 9, 1, 0, 1, False, 1), (1, 32, 128, 128, 9, 1, 0, 0, False, 1),         # the 128 x 256 tile (M = 128), one image
    (3, 16, 256, 256, 9, 1, 0, 0, True, 1), (2, 32, 256, 256, 9, 2, 0, 0, False, 1),          # the 256 x 128 tile; stride 2
    (2, 16, 384, 128, 1, 1, 256, 0, False, 1), (5, 8, 512, 256, 9, 1, 256, 1, False, 1),      # two-part input: 1x1 skip conv, 3x3 with residual
    (7, 4, 256, 768, 1, 1, 0, 0, False, 1), (65, 32, 384, 128, 9, 1, 256, 1, False, 1),       # qkv-like 1x1 on 4x4 maps; many tiles with an N tail
    # ResNeXt29's forms: grouped 3x3 (4 paired groups of 128 / 8 groups of 256, stride 2, ReLU), 1x1 with K = 64, a split-format residual
    (3, 16, 512, 512, 9, 1, 0, 0, True, 4), (2, 16, 2048, 2048, 9, 2, 0, 0, True, 8), (2, 32, 64, 512, 1, 1, 0, 0, True, 1),
    (3, 8, 1024, 512, 1, 1, 0, 2, True, 1),
]


@pytest.mark.parametrize('case', X3_CASES, ids=lambda c: 'B%d_H%d_%dto%d_t%d_s%d_c1%d_r%d_relu%d_g%d' % tuple(int(v) for v in c))
def test_gemm_x3_conv_vs_torch(case):
    """The split-f16 conv GEMM (gemm_x3_kernel in its NHWC form, csrc/gemm_f32.hip: the UNet's and ResNeXt29's middle tiers — 3x3 /
    1x1, stride 2, two-part input, grouped, fp32 or split-format residual, both tile shapes) through dmad_conv_x3 against a float64
    torch convolution of the same fp32 operands (improved_diffusion/unet.py:107-252, models/resnext.py:23-62 are the callers' ops):
    every product is three f16 MFMAs on hi / lo pairs, ~22 significant bits — the result must be fp32-grade (1e-5 of the output
    scale; the f16 family's is 2e-3), a sample's result must not depend on its batch, and out_split writes the split form of the
    same values."""
    from dmad_hip import engine as E
    B, H, cin, cout, taps, stride, c1, with_res, relu, groups = case
    g = torch.Generator().manual_seed(2000 + B + H + cin)
    Kg, Mg = cin // groups, cout // groups
    x = (torch.rand(B, H, H, cin, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(groups, taps, Mg, Kg, generator=g) * 2 - 1) * 0.1).cuda()
    bias = (torch.rand(cout, generator=g) * 2 - 1).cuda()
    Ho = (H - 1) // stride + 1
    res = (torch.rand(B, Ho, Ho, cout, generator=g) * 2 - 1).cuda() if with_res else None
    xa, xb = (x[..., :c1].contiguous(), x[..., c1:].contiguous()) if c1 else (x, None)
    kw = dict(stride=stride, relu=bool(relu), groups=groups, res_split=with_res == 2)
    out = E.conv_x3(xa, w, bias, res=res, x2=xb, **kw)
    k = 3 if taps == 9 else 1
    wt = w.double().reshape(groups, k, k, Mg, Kg).permute(0, 3, 4, 1, 2).reshape(cout, Kg, k, k)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wt, bias=bias.double(), stride=stride, padding=k // 2, groups=groups).permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    err = float((out.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-5, err
    solo = E.conv_x3(xa[:1], w, bias, res=None if res is None else res[:1], x2=None if xb is None else xb[:1], **kw)
    assert torch.equal(solo, out[:1])                                              # no split-K: batch-invariant bits
    sp = E.conv_x3(xa, w, bias, res=res, x2=xb, out_split=True, **kw)
    assert torch.equal(sp.view(torch.int32), E.split_f16(out).view(torch.int32))   # the split form of the same fp32 values


@pytest.mark.parametrize('case', [(64, 32, 256, 256, False), (260, 16, 256, 256, True), (1027, 8, 256, 512, False), (4100, 4, 128, 256, True)],
                         ids=lambda c: 'B%d_H%d_%dto%d_r%d' % tuple(int(v) for v in c))
def test_gemm_h16_fused_upsample_vs_torch(case):
    """GemmH16Args::up2 — the UNet's Upsample (F.interpolate(scale_factor=2, mode='nearest') + 3x3 conv, improved_diffusion/unet.py:72-79)
    read through the upsampling by the slice-resident form — through its own hook dmad_conv_h16_up2, against torch's interpolate +
    fp32 conv of the same f16-rounded operands: output maps 32, 16, 8 and 4 pixels wide (a tile = 1/4, 1, 4, 16 images), an N tail,
    with and without residual; the result equals the plain conv of the materialised x2 map bit for bit (the family's K order), and a
    shape the form does not serve is refused, not mis-served."""
    from dmad_hip import engine as E
    from dmad_hip._lib import DmadError
    B, H, cin, cout, with_res = case
    g = torch.Generator().manual_seed(4000 + B + H)
    xh = (torch.rand(B, H // 2, H // 2, cin, generator=g) * 2 - 1).half().cuda()
    w = ((torch.rand(1, 9, cout, cin, generator=g) * 2 - 1) * 0.1).half().cuda()
    bias = (torch.rand(cout, generator=g) * 2 - 1).cuda()
    res = (torch.rand(B, H, H, cout, generator=g) * 2 - 1).half().cuda() if with_res else None
    o32, o16, st = E.conv_h16_up2(xh, w, bias, res=res, want_stats=True)
    up = xh.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)               # nearest x2, NHWC
    worst = 0.0
    for b0 in range(0, B, 256):
        sl = slice(b0, min(B, b0 + 256))
        xr = torch.nn.functional.interpolate(xh[sl].float().permute(0, 3, 1, 2), scale_factor=2, mode='nearest').permute(0, 2, 3, 1)
        assert torch.equal(xr.half(), up[sl])
        ref = _conv_ref(xr, None, w, bias, None if res is None else res[sl], 1, 1, False)
        worst = max(worst, float((o32[sl] - ref).abs().max()))
    assert worst < 2e-3, worst
    assert torch.equal(o16, o32.half())
    p32, p16 = E.conv_h16(up, w, bias, res=res)                                    # the materialised map through the plain conv
    assert torch.equal(p32, o32) and torch.equal(p16, o16)
    blocks = o16.float().reshape(B * H * H // 64, 64, cout // 4, 4)               # statistics of the f16-rounded outputs, per 64-pixel block
    assert float((st[..., 0] - blocks.sum((1, 3))).abs().max()) < 2e-2 and float((st[..., 1] - (blocks ** 2).sum((1, 3))).abs().max()) < 0.5
    with pytest.raises(DmadError):
        E.conv_h16_up2(xh[:1], w, bias)                                            # one image: fewer tiles than CUs, not the fusing form


@pytest.mark.parametrize('case', [(3, 32, 128, 128, 0), (130, 32, 128, 128, 0), (2, 16, 256, 256, 128), (260, 16, 256, 256, 0), (5, 8, 128, 384, 256), (7, 4, 256, 256, 0),
                                  (9, 4, 256, 512, 256)],
                         ids=lambda c: 'B%d_H%d_K%d_C%d_c1_%d' % c)
def test_groupnorm_from_epilogue_statistics_vs_torch(case):
    """The 16-bit tier's GroupNorm (nn.py:15-17 GroupNorm32 in fp32, then SiLU; with the ResBlock's scale-shift, unet.py:190-194): the
    statistics a conv GEMM's epilogue leaves (dmad_conv_h16_stats: per 64-pixel block — 16 on 4x4 maps — and channel quad, of the
    f16-rounded outputs; every kernel of the family that can serve the shape) fed to the one-pass apply kernel, against
    torch.nn.functional.group_norm on the same f16 map in fp32: statistics exact to fp32 summation order, output within one f16
    rounding.  Also the two-part (concatenated) input and batch invariance."""
    from dmad_hip import engine as E
    B, H, K, C, c1 = case
    g = torch.Generator().manual_seed(77 + B + H + C)
    x = (torch.rand(B, H, H, K, generator=g) * 2 - 1).half().cuda()

    def conv(M, seed_off):
        gg = torch.Generator().manual_seed(500 + seed_off + M)
        w = ((torch.rand(1, 9, M, K, generator=gg) * 2 - 1) * 0.08).half().cuda()
        b = (torch.rand(M, generator=gg) * 4 - 1).cuda()                    # a non-zero mean: the variance is E[x^2] - mean^2 here
        return E.conv_h16_stats(x, w, b)
    HW = H * H
    if c1:
        a16, sta = conv(c1, 1)
        b16, stb = conv(C - c1, 2)
        full = torch.cat([a16, b16], dim=3)
    else:
        a16, sta = conv(C, 1)
        b16 = stb = None
        full = a16
    # the statistics themselves: block sums of the f16 outputs
    blk = 64 if HW >= 64 else 16
    ref_blocks = a16.float().reshape(B * HW // blk, blk, a16.shape[3] // 4, 4)
    assert torch.allclose(sta[..., 0], ref_blocks.sum((1, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(sta[..., 1], (ref_blocks ** 2).sum((1, 3)), rtol=1e-5, atol=1e-3)
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    beta = (torch.rand(C, generator=g) - 0.5).cuda()
    ss = (torch.rand(2 * C, generator=g) - 0.5).cuda()
    for silu, use_ss in ((True, False), (True, True), (False, False)):
        y = E.groupnorm16_apply(a16.reshape(B, HW, -1), sta, gamma, beta, silu=silu, ss=ss if use_ss else None,
                                x2=None if b16 is None else b16.reshape(B, HW, -1), st2=stb, out32=True)
        ref = torch.nn.functional.group_norm(full.float().permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-5)
        if use_ss:
            ref = ref * (1 + ss[:C, None, None]) + ss[C:, None, None]
        if silu:
            ref = torch.nn.functional.silu(ref)
        ref = ref.permute(0, 2, 3, 1).reshape(B, HW, C)
        assert float((y - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())), (silu, use_ss, float((y - ref).abs().max()))
        y16 = E.groupnorm16_apply(a16.reshape(B, HW, -1), sta, gamma, beta, silu=silu, ss=ss if use_ss else None,
                                  x2=None if b16 is None else b16.reshape(B, HW, -1), st2=stb)
        assert torch.equal(y16, y.half())
    # a sample's result does not depend on the batch it is in
    lo = B // 2
    xs = x[lo:lo + 1].contiguous()
    x_saved, x = x, xs
    if c1:
        a1, s1 = conv(c1, 1); b1, s2 = conv(C - c1, 2)
        solo = E.groupnorm16_apply(a1.reshape(1, HW, -1), s1, gamma, beta, x2=b1.reshape(1, HW, -1), st2=s2)
    else:
        a1, s1 = conv(C, 1)
        solo = E.groupnorm16_apply(a1.reshape(1, HW, -1), s1, gamma, beta)
    x = x_saved
    assert torch.equal(solo[0], y_full_batch(E, a16, sta, b16, stb, gamma, beta, B, HW)[lo])


def y_full_batch(E, a16, sta, b16, stb, gamma, beta, B, HW):
    return E.groupnorm16_apply(a16.reshape(B, HW, -1), sta, gamma, beta, x2=None if b16 is None else b16.reshape(B, HW, -1), st2=stb)


# ------------------------------------------------------------------------------------------ Improved-Diffusion UNet (N1)
def test_unet_purifier_vs_reference_fixture(golden_dir):
    """UNetModel.forward, GaussianDiffusion.q_sample / p_sample and the ImprovedDiffusion wrapper on the HIP engine vs
    outputs of the imported reference classes on the same seeded 52.5M-parameter weights (tests/golden/make_golden_unet.py)."""
    from dmad_hip import engine as E
    from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion, melspec_standardize, melspec_inv_standardize
    z = G(golden_dir, 'unet.npz')
    eng = E.Engine(max_batch=4, precision=E.EXACT, with_classifier=False, with_wavenet=False)      # both UNet tiers, no WaveNet workspace
    eng.set_mode(E.MODE_FP32)                                  # the exact-fp32 UNet: the tier pinned to the reference fixtures
    pur = create_improved_diffusion(None, reverse_timestep=3, state_dict=synth.unet_state_dict(int(z['seed'])), engine=eng)
    model, gd = pur.model, pur.diffusion
    spec = torch.from_numpy(z['spec']).cuda()
    x0 = melspec_standardize(spec)
    assert relmax(x0.cpu().numpy(), z['x0']) < 1e-6 and relmax(melspec_inv_standardize(x0).cpu().numpy(), z['inv_std']) < 1e-6
    g = torch.Generator().manual_seed(31)
    noise = torch.randn(x0.shape, generator=g).cuda()
    for t in (3, 40):
        tt = torch.full((2,), t, dtype=torch.long).cuda()
        x_t = gd.q_sample(x0, tt, noise=noise)
        assert relmax(x_t.cpu().numpy(), z['x_t%d' % t]) < 1e-6
        eps = model(torch.from_numpy(z['x_t%d' % t]).cuda(), tt)
        assert eps.shape == (2, 1, 32, 32)
        assert relmax(eps.cpu().numpy(), z['eps_t%d' % t]) < UNET_FP32_TOL, t
    solo = model(torch.from_numpy(z['x_t3'][1:]).cuda(), torch.tensor([3]))
    assert torch.equal(solo, model(torch.from_numpy(z['x_t3']).cuda(), torch.tensor([3, 3]))[1:])      # batch invariance
    big = model(torch.from_numpy(z['x_t3']).cuda().repeat(3, 1, 1, 1), torch.full((6,), 3))         # 6 > max_batch: chunked
    assert torch.equal(big[4:], big[:2])
    for t in (3, 0):
        r = gd.p_sample(model, torch.from_numpy(z['x_t3']).cuda(), torch.full((2,), t), noise=torch.from_numpy(z['p_noise_t%d' % t]).cuda())
        assert float((r['pred_xstart'].cpu() - torch.from_numpy(z['p_xstart_t%d' % t])).abs().max()) < 1e-6    # measured 1.2e-7 (values up to 0.79)
        assert float((r['sample'].cpu() - torch.from_numpy(z['p_sample_t%d' % t])).abs().max()) < 1e-6
    out = pur(x0)                                              # diffuse to t* = 3, four reverse steps, back to dB
    again = pur(x0)
    assert out.shape == spec.shape and bool(torch.isfinite(out).all())
    assert float(out.min()) >= -100.0 - 1e-3 and float(out.max()) <= 38.22 + 1e-3                  # clip_denoised at the last step
    assert out.shape == again.shape
    with pytest.raises(NotImplementedError):
        model(x0, torch.tensor([3, 4]).cuda())
    # the 16-bit tier (f16 operands in every conv / 1x1, fp32 accumulate, fp32 GroupNorm / softmax / sums): same fixtures, f16 tolerance
    eng.set_mode(E.MODE_FAST)
    for t in (3, 40):
        e16 = model(torch.from_numpy(z['x_t%d' % t]).cuda(), torch.full((2,), t, dtype=torch.long).cuda())
        assert 1e-5 < relmax(e16.cpu().numpy(), z['eps_t%d' % t]) < UNET_F16_TOL, (t, relmax(e16.cpu().numpy(), z['eps_t%d' % t]))
    solo16 = model(torch.from_numpy(z['x_t3'][1:]).cuda(), torch.tensor([3]))
    assert torch.equal(solo16, model(torch.from_numpy(z['x_t3']).cuda(), torch.tensor([3, 3]))[1:])    # batch invariance holds on this tier too
    # the split-f16 middle tier (fp32 pipeline, every conv / 1x1 as three f16 MFMAs on hi / lo pairs): the same fixtures, fp32-grade
    eng.set_mode(E.MODE_EXACT_VOTES)
    for t in (3, 40):
        xt = torch.from_numpy(z['x_t%d' % t]).cuda()
        ex3 = eng.unet_eps(xt, t, tier=2)
        err3 = relmax(ex3.cpu().numpy(), z['eps_t%d' % t][:, 0])
        assert 0 < err3 < UNET_X3_TOL, (t, err3)
        assert torch.equal(ex3, eng.unet_eps(xt, t)) and not torch.equal(ex3, eng.unet_eps(xt, t, tier=0))      # the default of an exact-vote engine's map surfaces
        assert torch.equal(eng.unet_eps(xt[1:], t, tier=2), ex3[1:])                               # batch invariance (no split-K on this tier)
    with pytest.raises(E.DmadError):
        eng.load_wavenet(synth.wavenet_state_dict(1234))       # created with with_wavenet = 0
    eng.close()
    f32 = E.Engine(max_batch=2, precision=E.FP32, with_classifier=False, with_wavenet=False)
    create_improved_diffusion(None, reverse_timestep=3, state_dict=synth.unet_state_dict(int(z['seed'])), engine=f32)
    assert relmax(f32.unet_eps(torch.from_numpy(z['x_t3']).cuda(), 3).cpu().numpy(), z['eps_t3'][:, 0]) < UNET_FP32_TOL      # FP32 engines: the fp32 tier only
    with pytest.raises(E.DmadError):
        f32.unet_eps(torch.from_numpy(z['x_t3']).cuda(), 3, tier=2)
    f32.close()


@pytest.mark.gpu
def test_unet_16bit_tier_large_batch_uses_the_256_tile_and_stays_batch_invariant():
    """At 256 spectrograms the 256-channel 16x16 convs and the M = 768 qkv convs fill the chip and run the persistent forms of
    csrc/gemm_h16.hip (slice-resident / ping-pong), at 1024 the 8x8 maps do too (and the 4 -> 8 Upsample conv reads through the
    upsampling with tiles spanning four images); smaller batches run the 384-row kernel.  All accumulate every output in the same
    order, so a sample's eps is the same bits in a batch of 1024, of 256 and of 2 (batch invariance across the kernel switches),
    and the tier stays within its f16 tolerance of the exact-fp32 tier at the large batch too.  The fp32 tier's own switch (narrow
    tiles for sub-chip launches) is checked the same way."""
    from dmad_hip import engine as E
    from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion
    eng = E.Engine(max_batch=1024, precision=E.EXACT, with_classifier=False, with_wavenet=False)
    create_improved_diffusion(None, reverse_timestep=3, state_dict=synth.unet_state_dict(31), engine=eng)
    x = torch.randn(1024, 32, 32, generator=torch.Generator().manual_seed(5)).cuda()
    eng.set_mode(E.MODE_FAST)
    huge = eng.unet_eps(x, 7)           # 1024 spectrograms: also the 8x8 maps fill the chip (persistent forms, the 4 -> 8 upsample fused: a tile = 4 images)
    big = eng.unet_eps(x[:256].contiguous(), 7)
    assert bool(torch.isfinite(huge).all()) and torch.equal(huge[:256], big)
    for lo in (0, 77, 254):
        assert torch.equal(eng.unet_eps(x[lo:lo + 2].contiguous(), 7), big[lo:lo + 2]), lo
    assert torch.equal(eng.unet_eps(x[1021:1024].contiguous(), 7), huge[1021:])
    # the split-f16 middle tier has no kernel switch (no split-K, one tile shape per layer): batch-invariant by construction, checked anyway
    mid = eng.unet_eps(x[:256].contiguous(), 7, tier=2)
    assert torch.equal(eng.unet_eps(x[100:103].contiguous(), 7, tier=2), mid[100:103])
    eng.set_mode(E.MODE_FP32)
    ref = eng.unet_eps(x[:64].contiguous(), 7)
    assert 1e-5 < relmax(big[:64].cpu().numpy(), ref.cpu().numpy()) < UNET_F16_TOL
    # the exact-fp32 tier across ITS kernel switch: one sample alone runs the 64 x 32 tiles on the 8-slot ring (launches of fewer
    # workgroups than CUs, csrc/gemm_f32.hip), 64 samples the 64 / 128 x 128 tiles — same k order per output, the same bits
    for lo in (5, 63):
        assert torch.equal(eng.unet_eps(x[lo:lo + 1].contiguous(), 7), ref[lo:lo + 1]), lo
    eng.close()


# ------------------------------------------------------------------------------------------ BASELINE.json configs at full size
@pytest.fixture(scope='module')
def big_engine(weights):
    from dmad_hip import engine as E
    eng = E.Engine(max_batch=256, precision=E.BF16)
    eng.load_wavenet(weights[0])
    eng.load_vgg19_bn(weights[1])
    yield eng
    eng.close()


def test_config2_ddpm_batch256_vgg(big_engine):
    """BASELINE config 2 — DiffWave DDPM t* = 5, batch 256, VGG19_bn forward, bf16 — through AcousticSystem, checked by
    size-independent properties: a row of the batch equals the same clip run alone with the same noise key, the
    result does not depend on the engine's chunking, repeats with the same seed are identical."""
    from acoustic_system import AcousticSystem
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip.transforms import MelSpectrogramDB
    eng = big_engine
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    net = synth_vgg().bind_engine(eng)
    den = DiffWave(WaveNetHIP(eng), hp, reverse_timestep=5, seed=123)
    model = AcousticSystem(classifier=net, transform=MelSpectrogramDB(eng), defender=den, defense_type='wave')
    x = torch.from_numpy(np.stack([synth.synthetic_clip(i % 10) for i in range(256)])).cuda()       # [256,1,16000]
    den._draws = 0
    logits = model(x)
    assert logits.shape == (256, 10) and bool(torch.isfinite(logits).all())
    den._draws = 0
    assert torch.equal(model(x), logits)                                       # same seed, same sample counters
    for i in (0, 97, 255):                                                     # row i alone, noise keyed by sample index i
        den._draws = i
        assert torch.equal(model(x[i:i + 1]), logits[i:i + 1]), i
    den._draws = 0
    pur = den(x)                                                               # the purified waveforms themselves
    assert pur.shape == (256, 1, 16000) and float(pur.abs().max()) < 50.0
    den._draws = 128
    assert torch.equal(den(x[128:]), pur[128:])                                # second half alone == second half of the batch
    assert torch.equal(model(x, defend=False), net(MelSpectrogramDB(eng)(x)))  # defend=False bypasses the purifier
    den._draws = 40                                                            # the one-call chain (dmad_ddpm_purify) == the
    x_t = den._diffusion(x[40:44])                                             # step-by-step surfaces with the same noise keys
    den._draws = 40
    assert torch.equal(den._reverse(x_t), pur[40:44])


def test_config3_certify_n1000(big_engine):
    """BASELINE config 3 — certified smoothing N = 1000, sigma = 0.5, DiffWave + VGG19_bn: vote conservation, the shard
    property (counts of [0, N) = sum of the counts of any partition of the sample range, which is what the
    multi-GPU path relies on), independence of the batch size, and certify()'s contract on top of the counts."""
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    from scipy.stats import beta, norm
    eng = big_engine
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    ab = hp['Alpha_bar']
    t = int(torch.abs(ab - 1 / 1.25).min(0, keepdim=True)[1].item())
    assert t + 1 == 66                                                         # sigma = 0.5 -> t* = 66
    ca, cb = float((1 / ab).sqrt()[t]), float((1 / ab - 1).sqrt()[t])
    sc = float(torch.tensor((1 / 1.25) ** 0.5))
    clip = torch.from_numpy(synth.synthetic_clip(4)).cuda()
    whole, _, _ = eng.smooth_votes(clip, 0.5, sc, t, ca, cb, 1000, seed=77, sample0=0)
    assert int(whole.sum()) == 1000
    parts = torch.zeros_like(whole)
    for lo, hi, b in ((0, 125, 125), (125, 500, 256), (500, 1000, 64)):        # uneven "ranks", different batch sizes
        c, _, _ = eng.smooth_votes(clip, 0.5, sc, t, ca, cb, hi - lo, batch=b, seed=77, sample0=lo)
        parts += c
    assert torch.equal(whole, parts)
    net = synth_vgg().bind_engine(eng)
    RC = RobustCertificate(classifier=net, transform=MelSpectrogramDB(eng), denoiser=DiffWave(WaveNetHIP(eng), hp), seed=5)
    assert RC._fused()
    y_pred, radius = RC.certify(clip[None], torch.tensor([0]).cuda(), sigma=0.5, n_0=100, n=1000, batch_size=256)
    RC2 = RobustCertificate(classifier=net, transform=MelSpectrogramDB(eng), denoiser=DiffWave(WaveNetHIP(eng), hp), seed=5)
    c0 = RC2.smooth_predict(clip, num_sampling=100, sigma=0.5, batch_size=50)
    c1 = RC2.smooth_predict(clip, num_sampling=1000, sigma=0.5, batch_size=100)
    cA = int(c0.argmax())
    pa = float(beta.ppf(0.001, int(c1[cA]), 1000 - int(c1[cA]) + 1)) if int(c1[cA]) > 0 else 0.0
    if pa > 0.5:
        assert int(y_pred[0]) == cA and abs(float(radius[0]) - 0.5 * norm.ppf(pa)) < 1e-5
    else:
        assert int(y_pred[0]) == -1 and float(radius[0]) == 0.0


# ------------------------------------------------------------------------------------------ two ranks, real HIP path
def _hip_rank_worker(rank, world, port, out):
    import sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, 'diffusion-model-for-audio-defense_amd'), root]
    import torch.distributed as dist
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip import engine as E
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    dist.init_process_group('gloo', rank=rank, world_size=world)        # both ranks share the one GPU of the test box
    eng = E.Engine(max_batch=6, precision=E.BF16)
    eng.load_wavenet(synth.wavenet_state_dict(1234))
    net = vgg19_bn(num_classes=10, in_channels=1)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(4321).items()})
    net.eval().bind_engine(eng)
    rc = RobustCertificate(classifier=net, transform=MelSpectrogramDB(eng), seed=21,
                           denoiser=DiffWave(WaveNetHIP(eng), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)))
    assert rc._fused()
    clip = torch.from_numpy(synth.synthetic_clip(2)).cuda()
    counts = rc.smooth_predict(clip, num_sampling=30, sigma=0.5, batch_size=6)
    y, r = rc.certify(clip[None], torch.tensor([2]).cuda(), sigma=0.5, n_0=10, n=20, batch_size=5)
    if rank == 0:
        torch.save({'counts': counts, 'y': y.cpu(), 'r': r.cpu()}, out)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_two_ranks_on_the_hip_path(engines, tmp_path):
    """One process per rank as in production (RANK / WORLD_SIZE, here gloo for the int64[10] all-reduce because both
    ranks share the test box's single GPU): the sharded counts and the certificate equal the single-process ones."""
    import torch.multiprocessing as mp
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    out = str(tmp_path / 'ranks.pt')
    mp.spawn(_hip_rank_worker, args=(2, 29600 + os.getpid() % 2000, out), nprocs=2, join=True)
    got = torch.load(out)
    eng = engines['bf16']
    net = synth_vgg().bind_engine(eng)
    rc = RobustCertificate(classifier=net, transform=MelSpectrogramDB(eng), seed=21,
                           denoiser=DiffWave(WaveNetHIP(eng), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)))
    clip = torch.from_numpy(synth.synthetic_clip(2)).cuda()
    counts = rc.smooth_predict(clip, num_sampling=30, sigma=0.5, batch_size=4)
    y, r = rc.certify(clip[None], torch.tensor([2]).cuda(), sigma=0.5, n_0=10, n=20, batch_size=6)
    assert int(got['counts'].sum()) == 30 and got['counts'].tolist() == counts.tolist()
    assert got['y'].tolist() == y.cpu().tolist() and torch.equal(got['r'], r.cpu())


def test_other_wavenet_geometry(orc):
    """A 5-layer, dilation-cycle-4 WaveNet (not a multiple of 3 layers, d up to 8): the persistent kernels are not tied to
    the 36 x 12 geometry of the shipped checkpoint; fp32 and bf16 engines against the oracle restatement."""
    from dmad_hip import engine as E
    cfg = dict(synth.WAVENET_CONFIG)
    cfg.update(num_res_layers=5, dilation_cycle=4)
    sd = synth.wavenet_state_dict(77, cfg)
    w = orc.folded_weights(sd, 5)
    x = torch.from_numpy(np.stack([synth.synthetic_clip(0), synth.synthetic_clip(7)])) * 0.8
    ref = orc.wavenet_forward(w, x, 12 * torch.ones((2, 1)), 5, 4).numpy()[:, 0]
    for prec, tol in ((E.FP32, FP32_TOL), (E.BF16, BF16_MAX_TOL)):
        eng = E.Engine(wavenet_config=cfg, max_batch=3, precision=prec, with_classifier=False)
        eng.load_wavenet(sd)
        got = eng.wavenet_eps(x.cuda(), 12).cpu().numpy()
        assert relmax(got, ref) < tol, prec
        eng.close()


def test_bench_contract():
    """bench.py prints ONE JSON line with the driver's keys; `metric` is BASELINE.json's; roofline and (at N = 1) cpu_baseline
    objects are present and consistent (tiny run: 1 step of 8 samples, 2-sample CPU leg)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '1', '--warmup', '1', '--samples-per-step', '8',
                        '--max-batch', '8', '--cpu-samples', '2', '--full-n', '64', '--c5-n', '16', '--c5-batch', '16', '--c2-iters', '1', '--c2-batch', '8', '--c3-clips', '1',
                        '--check-steps', '1', '--grid-steps', '1', '--resnext-steps', '1'], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    base = json.load(open(os.path.join(root, 'BASELINE.json')))
    # the reference script's default classifier (ResNeXt29) on its 16-bit tier, exact votes checked on its own keys
    rx = j['resnext29_mode']
    assert rx['clips_per_s'] > 0 and rx['fast_mode_clips_per_s'] > 0 and sum(rx['votes']) == 8 * rx['steps'] and rx['vs_vgg_headline'] > 0
    assert rx['exact_equals_fp32'] is True and rx['votes_exact_first_step'] == rx['votes_fp32_first_step'] and rx['check_samples'] == 8
    # BASELINE C4's whole sigma grid rides in the line: 0.25 and 1.0 beside the headline's 0.5, each with its own exactness check
    sg = {round(g['sigma'], 2): g for g in j['sigma_grid']}
    assert sorted(sg) == [0.25, 1.0] and sg[0.25]['t_star'] == 34 and sg[1.0]['t_star'] == 117
    for g in sg.values():
        assert g['clips_per_s'] > 0 and g['fast_mode_clips_per_s'] > 0 and sum(g['votes']) == 8 * g['steps']
        assert g['exact_equals_fp32'] is True and g['votes_exact_first_step'] == g['votes_fp32_first_step'] and g['check_samples'] == 8
        assert 0.0 <= g['recheck_frac_fp32'] <= g['recheck_frac'] <= 1.0
    assert j['metric'] == base['metric'] and j['unit'] == 'clips/s' and j['n_gpus'] == 1 and j['steps'] == 1 and j['warmup'] == 1
    assert j['higher_is_better'] is True and j['scaling'] == 'weak' and j['vs_baseline'] is None and j['dtype'] == 'f16' and j['data'] == 'synthetic'
    assert 'workload' in j['config'] and 'model' not in j['config']
    assert abs(j['value'] - 8 / (j['ms_per_step'] * 1e-3)) < 1e-6 * j['value']
    rf = j['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and rf['peak'] == 2500.0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12
    assert rf['launches_timed'] == 35 and rf['achieved'] > 0 and (rf['traffic'] is None or rf['traffic'] > 0)
    cb = j['cpu_baseline']
    assert cb['kind'] == 'port' and cb['unit'] == 'clips/s' and cb['value'] > 0 and 1 <= cb['cores'] <= 16 and 'sample' in cb
    assert sum(j['votes']) == 8
    # the measured mode is the exact-vote mode; the 16-bit-only and fp32-only figures ride along
    assert 'exact-vote' in j['config']['mode'] and 0.0 <= j['recheck']['frac_fp32'] <= j['recheck']['frac'] <= 1.0 and j['recheck']['margin'] > 0
    assert j['fast_mode']['clips_per_s'] > 0 and j['fp32_mode']['clips_per_s'] > 0 and 0 < j['fp32_mode']['frac_of_fp32_matrix_peak'] < 1
    assert sum(j['fast_mode']['votes']) == 8 * j['fast_mode']['steps'] and sum(j['fp32_mode']['votes']) == 8 * j['fp32_mode']['steps']
    cf = j['certify_full']                 # RobustCertificate.certify through the host mirror, timed end to end
    assert cf['n_0'] == 100 and cf['n'] == 64 and cf['clips_per_s'] > 0 and cf['y_pred'] in range(-1, 10) and cf['radius'] >= 0
    ff = j['roofline_final']
    assert ff['bound'] == 'hbm' and ff['unit'] == 'GB/s' and ff['peak'] == 8000.0 and ff['launches_timed'] == 1 and 0 < ff['frac'] < 1.5
    c5 = j['c5_spec_mode']                 # BASELINE C5 beside the headline: the spec-domain vote loop on its own fp32 engine
    assert c5['n'] == 16 and sum(c5['votes']) == 16 and c5['samples_per_s'] > 0 and c5['dtype'] == 'f16' and 0 <= c5['recheck_frac'] <= 1
    assert c5['exact_equals_fp32'] is True and sum(c5['fp32_mode']['votes']) == c5['fp32_mode']['n'] == 16 and 0 < c5['fp32_mode']['frac_of_fp32_matrix_peak'] < 1
    assert c5['fast_mode']['samples_per_s'] > 0 and sum(c5['fast_mode']['votes']) == c5['fast_mode']['n'] == 16
    # the line proves its own exactness claim: the first timed step's keys in the exact-vote mode and on the exact-fp32 path
    ck = j['exact_vs_fp32_check']
    assert j['exact_equals_fp32'] is True and ck['votes_exact'] == ck['votes_fp32'] and sum(ck['votes_fp32']) == ck['samples'] == 8
    assert ck['sample_range'] == [8, 16]
    # BASELINE.md's other single-GPU cells: C2 (DDPM t* = 5 on a bf16 engine) and C3 (certify n = 1000)
    c2 = j['c2_ddpm_mode']
    assert c2['dtype'] == 'bf16' and c2['t_star'] == 5 and c2['batch'] == 8 and c2['clips_per_s'] > 0
    assert abs(c2['network_evals_per_s'] - 5 * c2['clips_per_s']) < 1e-6 * c2['network_evals_per_s'] and sum(c2['decisions_histogram']) == 8
    assert c2['roofline']['launches_timed'] == 5 * 35 and 0 < c2['roofline']['frac'] < 1
    c3 = j['c3_certify_n1000']
    assert c3['clips_per_s'] > 0 and len(c3['y_pred']) == 1 and c3['y_pred'][0] in range(-1, 10)
    # a rank count the box cannot serve is refused loudly, never run as fewer ranks
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(torch.cuda.device_count() + 1), '--steps', '1'],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0 and 'refusing' in r.stderr and not r.stdout.strip()


def test_bench_two_ranks_rehearsal():
    """The N > 1 form of bench.py, rehearsed as two ranks sharing this box's GPU (gloo for the collectives: DMAD_BENCH_BACKEND, a
    rehearsal switch; the measured configuration is one rank per GPU over RCCL): the line carries n_gpus = 2, the aggregate
    value of both ranks' steps, `certify_full` with one clip's samples SHARDED over the ranks (strong scaling), the C3 / C2 legs
    and the C5 leg sharded the same way, and the exactness check on the timed keys of both ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(DMAD_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(29700 + os.getpid() % 200), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2',
                        '--warmup', '1', '--samples-per-step', '16', '--max-batch', '16', '--full-n', '200', '--c5-n', '16', '--c5-batch', '16', '--c2-iters', '1',
                        '--c2-batch', '8', '--c3-clips', '1', '--check-steps', '1'], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 2 and sum(j['votes']) == 2 * 16 * 2 and j['scaling'] == 'weak'
    assert abs(j['value'] - 2 * 16 * 2 / (j['ms_per_step'] * 2e-3)) < 1e-6 * j['value']
    assert j['exact_equals_fp32'] is True and j['exact_vs_fp32_check']['samples'] == 32
    cf = j['certify_full']
    assert cf['n'] == 200 and cf['n_gpus'] == 2 and 'strong' in cf['scaling'] and cf['clips_per_s'] > 0
    assert j['c3_certify_n1000']['clips_per_s'] > 0 and j['c2_ddpm_mode']['clips_per_s'] > 0
    c5 = j['c5_spec_mode']
    assert c5['n'] == 32 and c5['n_gpus'] == 2 and sum(c5['votes']) == 32 and c5['exact_equals_fp32'] is True
    assert 'cpu_baseline' not in j                                    # rank 0 at N = 1 only


# ------------------------------------------------------------------------------------------ exact-vote mode, C4's sigmas
@pytest.fixture(scope='module')
def exact_engine(weights):
    from dmad_hip import engine as E
    eng = E.Engine(max_batch=64, precision=E.EXACT, recheck_batch=32)
    eng.load_wavenet(weights[0])
    eng.load_vgg19_bn(weights[1])
    yield eng
    eng.close()


def test_exact_vote_mode_equals_fp32_votes(exact_engine, sched):
    """the same Philox keys through the 16-bit path and the exact-fp32 path, 3 clips x sigma in
    {0.25, 0.5, 1.0} (t* = 34 / 66 / 117), N = 512 each (4608 samples): records the flips and the logit-difference error
    of the 16-bit path, checks that the recheck bound covers that error, and that the exact-vote mode's counts equal
    the fp32 path's bit for bit (north_star: class-vote counts match exactly; certified_robust.py:59-65)."""
    from dmad_hip import engine as E
    eng = exact_engine
    hp, coef = sched
    ab = hp['Alpha_bar']
    N, total_flips, worst_pair, rechecked_all = 512, 0, 0.0, 0
    for ci in (0, 1, 2):
        clip = torch.from_numpy(synth.synthetic_clip(ci)).cuda()
        for sigma, tstar in ((0.25, 34), (0.5, 66), (1.0, 117)):
            t = int(torch.abs(ab - 1 / (1 + sigma ** 2)).min(0, keepdim=True)[1].item())
            assert t + 1 == tstar
            sc = float(torch.tensor((1 / (1 + sigma ** 2)) ** 0.5, dtype=torch.float32))
            out = {}
            for mode in (E.MODE_FAST, E.MODE_FP32, E.MODE_EXACT_VOTES):
                eng.set_mode(mode)
                eng.recheck_stats(reset=True)
                c, l, _ = eng.smooth_votes(clip, sigma, sc, t, *coef(t), N, seed=900 + ci, sample0=5000, want_logits=True)
                out[mode] = (c.cpu().tolist(), l.cpu().numpy().astype(np.float64), eng.recheck_stats())
            fast, f32, ex = out[E.MODE_FAST], out[E.MODE_FP32], out[E.MODE_EXACT_VOTES]
            assert sum(fast[0]) == sum(f32[0]) == sum(ex[0]) == N
            assert ex[0] == f32[0], (ci, sigma, ex[0], f32[0])                       # bit-exact vote counts
            assert (ex[1].argmax(1) == f32[1].argmax(1)).all()
            pair = np.abs((fast[1][:, :, None] - fast[1][:, None, :]) - (f32[1][:, :, None] - f32[1][:, None, :])).max()
            worst_pair = max(worst_pair, float(pair))
            total_flips += int((fast[1].argmax(1) != f32[1].argmax(1)).sum())
            srt = np.sort(fast[1].astype(np.float32), 1)                          # the kernel's own fp32 comparison
            want_recheck = int((~((srt[:, -1] - srt[:, -2]) >= np.float32(eng.recheck_margin))).sum())
            assert ex[2] == (N, want_recheck) and fast[2][1] == 0 and f32[2][1] == 0
            assert 0 <= eng.recheck_stats(detail=True)[2] <= want_recheck         # only what the split-f16 tier left open reached fp32
            rechecked_all += want_recheck
    eng.set_mode(E.MODE_EXACT_VOTES)
    # the guarantee behind the mode: the bound exceeds the largest error of any logit difference seen on 4608 samples
    assert worst_pair < eng.recheck_margin, (worst_pair, eng.recheck_margin)
    print('exact-vote study: %d samples, %d 16-bit flips, worst logit-difference error %.4f, recheck margin %.3f, rechecked %.2f %%'
          % (9 * N, total_flips, worst_pair, eng.recheck_margin, 100.0 * rechecked_all / (9 * N)))


def test_sigma_1_vote_loop_vs_oracle(exact_engine, sched, orc, weights):
    """BASELINE config 4's sigma = 1.0 (t* = 117) through dmad_smooth_votes against the CPU oracle on the same Philox noise
    (2 samples), fp32 path and exact-vote mode."""
    from dmad_hip import engine as E
    eng = exact_engine
    hp, coef = sched
    sigma, t = 1.0, 116
    assert orc.compute_t_star(hp['Alpha_bar'], sigma) == 117
    sc = float(torch.tensor((1 / 2.0) ** 0.5, dtype=torch.float32))
    clip = torch.from_numpy(synth.synthetic_clip(3))
    z = eng.philox_normal(31, 40, 0, 2).cpu()
    den = orc.DiffWaveOracle(orc.folded_weights(weights[0]), hp, reverse_timestep=117)
    x_in = sc * (clip.unsqueeze(0).repeat(2, 1, 1) + sigma * z.unsqueeze(1))
    x0_ref = den.one_shot_denoise(x_in)
    lg_ref = orc.vgg19_bn_forward(weights[1], orc.mel_db(x0_ref)).numpy()
    eng.set_mode(E.MODE_FP32)
    c, lg, x0 = eng.smooth_votes(clip.cuda(), sigma, sc, t, *coef(t), 2, seed=31, sample0=40, want_logits=True, want_x0=True)
    assert relmax(x0.cpu().numpy(), x0_ref[:, 0].numpy()) < 5e-5
    assert np.abs(lg.cpu().numpy() - lg_ref).max() < 2e-3 and c.cpu().tolist() == np.bincount(lg_ref.argmax(1), minlength=10).tolist()
    eng.set_mode(E.MODE_EXACT_VOTES)
    c2, lg2, x02 = eng.smooth_votes(clip.cuda(), sigma, sc, t, *coef(t), 2, seed=31, sample0=40, want_logits=True, want_x0=True)
    assert c2.cpu().tolist() == c.cpu().tolist()
    assert relmax(x02.cpu().numpy(), x0_ref[:, 0].numpy()) < F16_MAX_TOL


def test_query_logits_is_one_call_and_row_keyed(engines):
    """dmad_query_logits (EOT / NES query path, _EOT.py:30-64): clip batch x repeats -> purified -> mel -> logits in one call;
    row (r, b) carries noise key sample0 + r*B + b, so it equals the same rows computed through AcousticSystem.forward."""
    from acoustic_system import AcousticSystem
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval._EOT import EOT
    from robustness_eval._utils import resolve_loss
    eng = engines['exact']
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    net = vgg19_bn(num_classes=10, in_channels=1)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(4321).items()})
    net.eval().bind_engine(eng)
    den = DiffWave(WaveNetHIP(eng), hp, reverse_timestep=2, seed=17)
    model = AcousticSystem(classifier=net, transform=MelSpectrogramDB(eng), defender=den, defense_type='wave')
    assert model._engine_chain(True) == (eng, 1) and model._engine_chain(False) == (eng, 0)
    x = torch.from_numpy(np.stack([synth.synthetic_clip(0), synth.synthetic_clip(5), synth.synthetic_clip(8)])).cuda()
    den._draws = 100
    logits, dec = model.query(x, repeats=5)                      # 15 rows > max_batch 6: chunked inside the call
    assert logits.shape == (5, 3, 10) and dec.shape == (5, 3) and den._draws == 115
    assert torch.equal(dec, logits.argmax(-1))
    den._draws = 100
    ref = model(x.repeat(5, 1, 1))                               # the reference's call pattern: one repeated batch
    assert torch.equal(ref.view(5, 3, 10), logits)
    den._draws = 100 + 3 * 2 + 1
    assert torch.equal(model(x[1:2]), logits[2, 1:2])            # row (2, 1) alone, by its key
    plain, _ = model.query(x, repeats=2, defend=False)
    assert torch.equal(plain[0], plain[1]) and torch.equal(plain[0], net(MelSpectrogramDB(eng)(x)))
    # EOT over the one-call path: means over repeats, one decision per evaluation
    loss_fn, _ = resolve_loss('Margin', False, 0., 'SCR', None, False)
    den._draws = 100
    scores, loss, grad, decisions = EOT(model, loss_fn, EOT_size=4, EOT_batch_size=2, use_grad=False)(x, torch.tensor([0, 6, 1]).cuda())
    want = (logits[:2].mean(0) + logits[2:4].mean(0)) / 2
    assert torch.allclose(scores, want, atol=1e-6) and grad is None and [len(d) for d in decisions] == [4, 4, 4]
    assert [int(v) for v in decisions[1]] == dec[:4, 1].tolist()
    from dmad_hip._lib import DmadError
    with pytest.raises(DmadError):
        eng.query_logits(x, 2, sampler=1, t_star=0, c_eps=[], c_div=[], c_sig=[])


def test_spec_defense_query_is_one_call_and_row_keyed():
    """AcousticSystem(defense_type='spec') (acoustic_system.py:40-49: transform, then the defender on the spectrogram) behind the
    EOT / NES query path: with the HIP mel front-end, a SpecPurifier on the engine's UNet and the HIP classifier, `query` is ONE
    dmad_spec_query_logits call whose row (r, b) carries the key draws + r*B + b — equal to the same rows through
    AcousticSystem.forward (mel -> SpecPurifier.forward -> classifier, composed from the separately pinned q_sample / p_sample /
    classifier surfaces), for any chunking; a defender the engine does not know falls back to the forward loop."""
    from acoustic_system import AcousticSystem
    from diffusion_models.improved_diffusion_ddpm import SpecPurifier, create_improved_diffusion
    from dmad_hip import engine as E
    from dmad_hip.transforms import MelSpectrogramDB
    eng = E.Engine(max_batch=4, precision=E.FP32, with_wavenet=False)
    pur = create_improved_diffusion(None, reverse_timestep=3, state_dict=synth.unet_state_dict(31), engine=eng)
    net = synth_vgg(calibrated='c5').bind_engine(eng)
    mel = MelSpectrogramDB(eng)
    den = SpecPurifier(pur, seed=23)
    model = AcousticSystem(classifier=net, transform=mel, defender=den, defense_type='spec')
    assert model._engine_chain(True) == (eng, 3) and model._engine_chain(False) == (eng, 0)
    x = torch.from_numpy(np.stack([synth.synthetic_clip(1), synth.synthetic_clip(7), synth.synthetic_clip(9)])).cuda()
    den._draws = 40
    logits, dec = model.query(x, repeats=3)                      # 9 rows > max_batch 4: chunked inside the call
    assert logits.shape == (3, 3, 10) and dec.shape == (3, 3) and den._draws == 49
    assert torch.equal(dec, logits.argmax(-1)) and bool(torch.isfinite(logits).all())
    den._draws = 40
    ref = torch.cat([model(x.repeat(3, 1, 1)[i:i + 4]) for i in (0, 4, 8)]).view(3, 3, 10)      # the reference's call pattern, in engine-sized calls
    assert float((ref - logits).abs().max()) < 2e-3 and ref.argmax(-1).tolist() == dec.tolist()
    den._draws = 40 + 3 * 1 + 2
    assert float((model(x[2:3]) - logits[1, 2:3]).abs().max()) < 2e-3          # row (1, 2) alone, by its key
    den._draws = 40
    again, _ = model.query(x[:2], repeats=2)                     # other B, same keys for rows 0..3 of the key range -> other clips: only shape / keying
    assert again.shape == (2, 2, 10) and den._draws == 44
    assert torch.equal(again[0, 0], logits[0, 0])                # row 0 = clip 0 with key 40 in both calls
    plain, _ = model.query(x, repeats=2, defend=False)
    assert torch.equal(plain[0], plain[1]) and torch.equal(plain[0], net(mel(x)))
    # a spec defender that is not the engine's chain: the forward loop, same result type
    other = AcousticSystem(classifier=net, transform=mel, defender=torch.nn.Identity(), defense_type='spec')
    assert other._engine_chain(True) == (None, 0)
    lo, do = other.query(x, repeats=2)
    assert lo.shape == (2, 3, 10) and torch.equal(lo[0], plain[0]) and torch.equal(do, lo.argmax(-1))
    from dmad_hip._lib import DmadError
    bare = E.Engine(max_batch=2, precision=E.FP32, with_wavenet=False)
    with pytest.raises(DmadError):
        bare.spec_query_logits(x[:1], 1, *pur.purify_coefficients(), -100.0, 38.22)      # no UNet / classifier loaded
    bare.close()
    eng.close()


def test_randsmooth_with_device_noise_and_second_classifier(engines):
    """(a) `--defense_method randsmooth` (denoiser=None) with the default device noise binds the HIP classifier
    before it needs the engine; (b) a second, different classifier never runs on the first one's weights."""
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from dmad_hip import engine as E
    from dmad_hip._lib import DmadError
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    eng = engines['fp32']
    a = vgg19_bn(num_classes=10, in_channels=1)
    a.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(4321).items()})
    a.eval()
    E._ENGINES[(torch.cuda.current_device(), E.EXACT)] = eng                   # what get_engine() hands out
    try:
        rc = RobustCertificate(classifier=a, transform=MelSpectrogramDB(eng), denoiser=None, seed=3)
        assert 'engine' not in a.__dict__
        clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
        counts = rc.smooth_predict(clip, num_sampling=10, sigma=0.25, batch_size=4)
        assert int(counts.sum()) == 10 and a.engine is eng
        again = RobustCertificate(classifier=a, transform=MelSpectrogramDB(eng), denoiser=None, seed=3).smooth_predict(clip, 10, 0.25, 5)
        assert counts.tolist() == again.tolist()
        b = vgg19_bn(num_classes=10, in_channels=1)
        b.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(99, calibrated=False).items()})
        b.eval()
        with pytest.raises(DmadError):
            b.bind_engine(eng)                                                  # explicit engine holding another classifier
        b.bind_engine()                                                         # shared engine is taken: an engine of its own
        assert b.engine is not eng and b.engine.classifier_owner != eng.classifier_owner
        spec = MelSpectrogramDB(eng)(clip[None])
        assert not torch.equal(a(spec), b(spec))
        b.engine.close()
    finally:
        E._ENGINES.pop((torch.cuda.current_device(), E.EXACT), None)


def test_reference_driver_default_resnext_checkpoint(tmp_path, weights):
    """certified_robustness_eval.py:57-59 loads a pickled DataParallel(CifarResNeXt) by default: write one in the reference's
    on-disk format under a ConvNets_SpeechCommands/ path, load it with create_model and certify through the fused loop."""
    from audio_models.ConvNets_SpeechCommands.create_model import create_model
    from models.resnext import CifarResNeXt
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip import engine as E
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    d = tmp_path / 'ConvNets_SpeechCommands'
    d.mkdir()
    sd = synth.resnext29_state_dict(2929)
    rx = CifarResNeXt(nlabels=10, in_channels=1)
    rx.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    torch.save(torch.nn.DataParallel(rx), str(d / 'gaussian_aug_resnext29_8_64.pth'))
    clf = create_model(str(d / 'gaussian_aug_resnext29_8_64.pth')).cuda()
    eng = E.Engine(max_batch=8, precision=E.EXACT, recheck_batch=4)
    eng.load_wavenet(weights[0])
    den = DiffWave(WaveNetHIP(eng), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG))
    rc = RobustCertificate(classifier=clf, transform=MelSpectrogramDB(eng), denoiser=den, seed=4)
    assert rc._fused() and eng.classifier_kind == 'resnext29'
    clip = torch.from_numpy(synth.synthetic_clip(1)).cuda()
    y, r = rc.certify(clip[None], torch.tensor([1]).cuda(), sigma=0.5, n_0=8, n=24, batch_size=8)
    assert y.shape == (1,) and int(y[0]) in range(-1, 10) and float(r[0]) >= 0.0
    eng.set_mode(E.MODE_FP32)
    rc2 = RobustCertificate(classifier=clf, transform=MelSpectrogramDB(eng), denoiser=den, seed=4)
    y2, r2 = rc2.certify(clip[None], torch.tensor([1]).cuda(), sigma=0.5, n_0=8, n=24, batch_size=8)
    assert y2.tolist() == y.tolist() and torch.equal(r2, r)                    # exact-vote mode == fp32 path
    eng.close()


def _nccl_worker(_, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))        # RCCL
    counts = torch.arange(10, dtype=torch.int64, device='cuda') * (2 ** 40)    # int64 range, as the vote counts are typed
    dist.all_reduce(counts)
    seed = torch.tensor([2 ** 61 + 12345], dtype=torch.int64, device='cuda')
    dist.broadcast(seed, 0)
    tmax = torch.tensor([1.5], dtype=torch.float64, device='cuda')
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    torch.save({'counts': counts.cpu(), 'seed': seed.cpu(), 'tmax': tmax.cpu(), 'backend': dist.get_backend()}, out)
    dist.destroy_process_group()


def test_rccl_backend_runs_the_paths_collectives(tmp_path):
    """The three collectives of the multi-GPU path (int64[10] vote all-reduce, the seed broadcast of
    RobustCertificate._seed_for_call, bench.py's max-over-ranks time) on the `nccl` (= RCCL) backend with CUDA tensors,
    world_size 1 (the box has one GPU; the driver's 8-GPU run does the rest)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'nccl.pt')
    mp.spawn(_nccl_worker, args=(29700 + os.getpid() % 2000, out), nprocs=1, join=True)
    got = torch.load(out)
    assert got['backend'] == 'nccl' and got['counts'].tolist() == [i * 2 ** 40 for i in range(10)]
    assert int(got['seed'][0]) == 2 ** 61 + 12345 and float(got['tmax'][0]) == 1.5


def test_split_f16_tier_is_fp32_grade(exact_engine, golden_dir):
    """The middle tier of the exact-vote mode: the fp32 pipeline on split-f16 operands (hi + lo * 2^-11, three f16 MFMAs per
    product) against the reference fixture and the exact-fp32 path: two orders of magnitude below the f16 path's error."""
    z = G(golden_dir, 'wavenet_full.npz')
    x_t = torch.from_numpy(z['x_t']).cuda()
    ref = z['eps'][:, 0]
    eng = exact_engine
    f32 = eng.wavenet_eps_path(x_t, int(z['t']), 1).cpu().numpy()
    x3 = eng.wavenet_eps_path(x_t, int(z['t']), 2).cpu().numpy()
    f16 = eng.wavenet_eps_path(x_t, int(z['t']), 0).cpu().numpy()
    assert relmax(f32, ref) < FP32_TOL
    assert relmax(x3, ref) < 4 * FP32_TOL and relmax(x3, f32) < 4 * FP32_TOL, (relmax(x3, ref), relmax(x3, f32))
    assert relmax(f16, ref) > 20 * relmax(x3, ref)
    x = torch.randn(3, 16000, generator=torch.Generator().manual_seed(8)).cuda() * 0.4
    full = eng.wavenet_eps_path(x, 20, 2)
    assert torch.equal(full[1:2], eng.wavenet_eps_path(x[1:2], 20, 2))            # batch-slot independent, like the other paths
    from dmad_hip._lib import DmadError
    with pytest.raises(DmadError):
        eng.wavenet_eps_path(x, 20, 7)


# ------------------------------------------------------------------------------------------ BASELINE C5: spec-domain vote loop
def test_config5_spec_domain_vote_loop(golden_dir):
    """BASELINE config 5 — certified smoothing with the Improved-Diffusion UNet purifier on mel spectrograms.  The reference has
    no working composite (improved_diffusion_ddpm.py:53-59), so the loop is the one include/dmad.h defines for
    dmad_spec_smooth_votes; checked here against the SAME chain composed from the separately pinned parts (mel, q_sample,
    p_sample, classifier: each vs reference fixtures elsewhere in this file) on the same Philox keys, plus the shard /
    batch-size invariance the multi-GPU path relies on."""
    from diffusion_models.improved_diffusion_ddpm import SpecDefense, create_improved_diffusion, melspec_standardize, melspec_inv_standardize
    from dmad_hip import engine as E
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    z = G(golden_dir, 'unet.npz')
    eng = E.Engine(max_batch=16, precision=E.FP32)
    pur = create_improved_diffusion(None, reverse_timestep=3, state_dict=synth.unet_state_dict(int(z['seed'])), engine=eng)
    net = synth_vgg().bind_engine(eng)
    mel = MelSpectrogramDB(eng)
    rc = RobustCertificate(classifier=net, transform=SpecDefense(mel, pur), denoiser=None, seed=5)
    assert rc._fused_spec() is eng
    clip = torch.from_numpy(synth.synthetic_clip(6)).cuda()
    counts = rc.smooth_predict(clip, num_sampling=40, sigma=0.5, batch_size=16)
    assert counts.dtype == torch.int64 and int(counts.sum()) == 40
    # the engine call against the chain composed from its parts, same keys
    ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig = pur.purify_coefficients()
    assert ts == 3 and len(c_a) == 4 and c_sig[0] == 0.0
    seed, s0, B, sigma = 77, 10, 6, 0.5
    cnt, lg, sp = eng.spec_smooth_votes(clip, sigma, ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, -100.0, 38.22, B, seed=seed, sample0=s0,
                                        want_logits=True, want_spec=True)
    x_in = clip.reshape(1, 1, -1) + sigma * eng.philox_normal(seed, s0, 0, B).unsqueeze(1)
    x0 = melspec_standardize(mel(x_in))
    zq = eng.philox_normal(seed, s0, 0x5BEC, B)[:, :1024].reshape(B, 1, 32, 32)
    x = pur.diffusion.q_sample(x0, torch.full((B,), ts, dtype=torch.long).cuda(), noise=zq)
    for t in range(ts, -1, -1):
        x = pur.diffusion.p_sample(pur.model, x, torch.full((B,), t), seed=seed, sample0=s0)['sample']
    ref_spec = melspec_inv_standardize(x)
    ref_logits = net(ref_spec)
    assert float((sp - ref_spec).abs().max()) < 2e-3 and float((lg - ref_logits).abs().max()) < 2e-3
    assert lg.argmax(1).tolist() == ref_logits.argmax(1).tolist()
    assert cnt.cpu().tolist() == torch.bincount(ref_logits.argmax(1).cpu(), minlength=10).tolist()
    assert float(sp.min()) >= -100.0 - 1e-3 and float(sp.max()) <= 38.22 + 1e-3          # clip_denoised keeps the chain in range
    # shard / batch invariance: rows are keyed by their global sample index
    a, _, _ = eng.spec_smooth_votes(clip, sigma, ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, -100.0, 38.22, 40, batch=16, seed=3)
    b1, _, _ = eng.spec_smooth_votes(clip, sigma, ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, -100.0, 38.22, 13, batch=5, seed=3, sample0=0)
    b2, _, _ = eng.spec_smooth_votes(clip, sigma, ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, -100.0, 38.22, 27, batch=16, seed=3, sample0=13)
    assert torch.equal(a, b1 + b2)
    # unfused use of the same object: SpecDefense as a plain transform
    out = SpecDefense(mel, pur)(x_in[:2])
    assert out.shape == (2, 1, 32, 32) and bool(torch.isfinite(out).all())
    from dmad_hip._lib import DmadError
    bare = E.Engine(max_batch=2, precision=E.FP32)
    with pytest.raises(DmadError):
        bare.spec_smooth_votes(clip, sigma, ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, -100.0, 38.22, 2)     # no UNet / classifier loaded
    bare.close()
    eng.close()


def test_config5_exact_votes_on_the_16bit_unet_tier():
    """BASELINE C5 in the exact-vote mode: the whole chain on the UNet's 16-bit tier; every sample whose top-2 margin is below
    tau_spec re-runs its chain from the same Philox keys on the split-f16 tier, and on the exact-fp32 UNet if its margin there is still
    below tau_spec2.  With the VGG19_bn calibrated for the spec chain's output distribution (tests/golden/make_classifier_calib.py) the
    check is NOT vacuous: several classes vote, samples ARE rechecked.  Counts equal the fp32 engine mode's on the same keys (also
    with every sample forced through both recheck tiers, without the middle tier, and across shards), every row of logits_out carries
    the logits of the tier that settled it, the 16-bit and split-f16 tiers stay within their bounds, and the statistics count what
    was re-run."""
    from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion
    from dmad_hip import engine as E
    eng = E.Engine(max_batch=64, precision=E.EXACT, with_wavenet=False)
    eng.load_vgg19_bn(synth.vgg19_bn_state_dict(4321, calibrated='c5'))
    assert eng.spec_recheck_margin == E.DEFAULT_SPEC_RECHECK_MARGIN and eng.spec_recheck_margin2 == E.DEFAULT_SPEC_RECHECK_MARGIN2
    pur = create_improved_diffusion(None, reverse_timestep=5, state_dict=synth.unet_state_dict(31), engine=eng)
    coef = tuple(pur.purify_coefficients())
    N, total_re = 384, 0

    def margin(l):
        top2 = l.topk(2, dim=1).values
        return top2[:, 0] - top2[:, 1]
    for ci, sigma in ((4, 0.5), (5, 1.0)):
        clip = torch.from_numpy(synth.synthetic_clip(ci)).cuda()
        args = (clip, sigma) + coef + (-100.0, 38.22)
        eng.set_mode(E.MODE_FP32)
        c32, l32, _ = eng.spec_smooth_votes(*args, N, seed=11, want_logits=True)
        eng.set_mode(E.MODE_FAST)
        c16, l16, _ = eng.spec_smooth_votes(*args, N, seed=11, want_logits=True)
        assert sum(1 for v in c32.tolist() if v > 0) >= 3 and max(c32.tolist()) <= 0.8 * N, c32.tolist()      # a non-degenerate stand-in
        idx = torch.arange(N, dtype=torch.int64, device='cuda')
        l2 = eng.spec_eval_samples(clip, sigma, *coef, -100.0, 38.22, idx, tier=2, seed=11)                 # every chain on the split-f16 tier
        e = (l16 - l32).double()
        lead = float((e - e.gather(1, l32.argmax(1, keepdim=True))).abs().max())
        assert 1e-5 < lead < eng.spec_recheck_margin, lead                       # the statistic the bound covers, with room
        e2 = (l2 - l32).double()
        lead2 = float((e2 - e2.gather(1, l32.argmax(1, keepdim=True))).abs().max())
        assert 0 < lead2 < eng.spec_recheck_margin2 and lead2 < 0.05 * lead, (lead2, lead)      # fp32-grade: far below the 16-bit tier's
        eng.set_mode(E.MODE_EXACT_VOTES)
        eng.spec_recheck_stats(reset=True)
        cx, lx, _ = eng.spec_smooth_votes(*args, N, seed=11, want_logits=True)
        voted, re, re32 = eng.spec_recheck_stats(detail=True)
        assert cx.tolist() == c32.tolist() and voted == N and 0 < re < N
        low = margin(l16) < eng.spec_recheck_margin
        low2 = low & (margin(l2) < eng.spec_recheck_margin2)
        assert int(low.sum()) == re and int(low2.sum()) == re32
        want = torch.where(low2[:, None], l32, torch.where(low[:, None], l2, l16))
        assert torch.equal(lx, want)                                             # every row: the logits of the tier that settled it
        total_re += re
        a, _, _ = eng.spec_smooth_votes(*args, 150, batch=40, seed=11)           # shards of the same index range
        b, _, _ = eng.spec_smooth_votes(*args, N - 150, batch=64, seed=11, sample0=150)
        assert (a + b).tolist() == c32.tolist()
        eng.set_spec_recheck_margin2(-1.0)                                       # no middle tier: queued samples straight to fp32
        eng.spec_recheck_stats(reset=True)
        cn, ln, _ = eng.spec_smooth_votes(*args, N, seed=11, want_logits=True)
        assert cn.tolist() == c32.tolist() and eng.spec_recheck_stats(detail=True) == (N, re, re)
        assert torch.equal(ln, torch.where(low[:, None], l32, l16))
        eng.set_spec_recheck_margin2(E.DEFAULT_SPEC_RECHECK_MARGIN2)
    assert total_re >= 8, total_re
    old = eng.spec_recheck_margin
    eng.set_spec_recheck_margin(1e30)                                            # everything through the recheck tiers ...
    eng.set_spec_recheck_margin2(1e30)                                           # ... down to fp32
    cz, lz, _ = eng.spec_smooth_votes(*args, 100, batch=32, seed=11, want_logits=True)
    eng.set_spec_recheck_margin(old)
    eng.set_spec_recheck_margin2(E.DEFAULT_SPEC_RECHECK_MARGIN2)
    assert torch.equal(lz, l32[:100]) and int(cz.sum()) == 100
    # edge cases: an empty loop votes nothing; a NaN clip gives NaN logits on the 16-bit tier, is queued, stays NaN on the split-f16 tier,
    # reaches the fp32 tier and votes as torch.max does (first NaN index), exactly like the fp32 mode
    c0, _, _ = eng.spec_smooth_votes(*args, 0, seed=11)
    assert c0.tolist() == [0] * 10
    bad = clip.clone()
    bad[0, 100] = float('nan')
    bargs = (bad, sigma) + coef + (-100.0, 38.22)
    eng.spec_recheck_stats(reset=True)
    cn, ln, _ = eng.spec_smooth_votes(*bargs, 5, batch=4, seed=1, want_logits=True)
    assert bool(torch.isnan(ln).all()) and cn.tolist() == [5, 0, 0, 0, 0, 0, 0, 0, 0, 0] and eng.spec_recheck_stats(detail=True) == (5, 5, 5)
    eng.set_mode(E.MODE_FP32)
    assert eng.spec_smooth_votes(*bargs, 5, batch=4, seed=1)[0].tolist() == cn.tolist()
    eng.close()


def test_spec_tier_calibration_and_audit(tmp_path):
    """the spec-domain loop's bound is calibrated and auditable like the waveform loop's.  dmad_spec_eval_samples evaluates
    listed samples' chains on an explicit UNet tier (rows equal the vote loop's logits_out rows on the same keys);
    Engine.calibrate_spec_recheck measures the 16-bit chain's leader-difference error for the resident weights at this (sigma, t*) and
    only WIDENS tau_spec (never below the committed default, never below a wider bound the caller put in force);
    RobustCertificate(calibrate=n) runs it once per clip, certify(audit=k) re-runs k tier-1 voters on the exact-fp32 UNet and records
    the outcome; the map-returning UNet surfaces of an exact-vote engine default to the split-f16 tier (fp32-grade)."""
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn
    from diffusion_models.improved_diffusion_ddpm import create_improved_diffusion, SpecDefense
    from dmad_hip import engine as E
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    eng = E.Engine(max_batch=32, precision=E.EXACT, with_wavenet=False)
    csd = synth.vgg19_bn_state_dict(4321, calibrated='c5')
    eng.load_vgg19_bn(csd)
    pur = create_improved_diffusion(None, reverse_timestep=4, state_dict=synth.unet_state_dict(31), engine=eng)
    chain = tuple(pur.purify_coefficients()) + (-100.0, 38.22)
    clip = torch.from_numpy(synth.synthetic_clip(2)).cuda()
    # the hook reproduces the loop's rows, tier by tier
    eng.set_mode(E.MODE_FP32)
    _, l32, _ = eng.spec_smooth_votes(clip, 0.5, *chain, 40, seed=5, want_logits=True)
    eng.set_mode(E.MODE_FAST)
    _, l16, _ = eng.spec_smooth_votes(clip, 0.5, *chain, 40, seed=5, want_logits=True)
    eng.set_mode(E.MODE_EXACT_VOTES)
    idx = torch.tensor([3, 17, 39, 0], device='cuda')
    assert torch.equal(eng.spec_eval_samples(clip, 0.5, *chain, idx, tier=0, seed=5), l32[idx.cpu()])
    assert torch.equal(eng.spec_eval_samples(clip, 0.5, *chain, idx, tier=1, seed=5), l16[idx.cpu()])
    lmid = eng.spec_eval_samples(clip, 0.5, *chain, idx, tier=2, seed=5)                        # the split-f16 tier: fp32-grade
    assert float((lmid - l32[idx.cpu()]).abs().max()) < 1e-3 and not torch.equal(lmid, l32[idx.cpu()])
    # map-returning surfaces: fp32 tier by default on an exact-vote engine, the 16-bit tier is opt-in
    x = torch.randn(2, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
    e_def = eng.unet_eps(x, 3)
    eng.set_mode(E.MODE_FP32); e_32 = eng.unet_eps(x, 3)
    eng.set_mode(E.MODE_FAST); e_16 = eng.unet_eps(x, 3)
    eng.set_mode(E.MODE_EXACT_VOTES)
    # default on an exact-vote engine: the split-f16 tier (fp32-grade), like the waveform-returning surfaces; fp32 / 16-bit on request
    assert torch.equal(e_def, eng.unet_eps(x, 3, tier=2)) and not torch.equal(e_def, e_16)
    assert 0 < float((e_def - e_32).abs().max()) < 1e-4 * float(e_32.abs().max()) < float((e_16 - e_32).abs().max())
    eng.set_waveform_tier(E.WAVE_FP32)
    assert torch.equal(eng.unet_eps(x, 3), e_32)
    eng.set_waveform_tier(E.WAVE_16BIT)
    assert torch.equal(eng.unet_eps(x, 3), e_16)
    eng.set_waveform_tier(E.WAVE_SPLIT)
    # calibration: widen-only
    tau, e, s_ = eng.calibrate_spec_recheck(clip, 0.5, chain, n=96)
    assert tau >= E.DEFAULT_SPEC_RECHECK_MARGIN and 0 < e < tau and s_ > 0 and eng.spec_recheck_margin == tau
    assert eng.spec_calibration['floor'] == E.DEFAULT_SPEC_RECHECK_MARGIN and eng.spec_calibration['t_star'] == 4
    cal = eng.spec_calibration                                # the split-f16 tier's bound is calibrated with it (against fp32, widen-only)
    assert 0 < cal['e2'] < cal['tau_spec2'] and cal['tau_spec2'] >= E.DEFAULT_SPEC_RECHECK_MARGIN2 and eng.spec_recheck_margin2 == cal['tau_spec2']
    assert cal['e2'] < 0.05 * cal['e'] and cal['n_fp32'] == 48
    eng.set_spec_recheck_margin(1.25)                        # a wider bound the caller chose is a floor, too
    assert eng.calibrate_spec_recheck(clip, 0.5, chain, n=32)[0] >= 1.25
    eng.set_spec_recheck_margin(E.DEFAULT_SPEC_RECHECK_MARGIN)
    # through the mirror: one calibration per clip (certify calls smooth_predict twice per clip), an audit entry per example
    net = vgg19_bn(num_classes=10, in_channels=1).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in csd.items()})
    net.bind_engine(eng)
    lines = []
    rc = RobustCertificate(classifier=net, transform=SpecDefense(MelSpectrogramDB(eng), pur), seed=3, calibrate=64, calibrate_clips=2, log=lines.append)
    assert rc._fused_spec() is eng
    xb = torch.stack([clip.reshape(1, -1), torch.from_numpy(synth.synthetic_clip(3)).cuda().reshape(1, -1)])
    y, r = rc.certify(xb, torch.tensor([0, 0]).cuda(), sigma=0.5, n_0=16, n=64, batch_size=32, audit=24)
    cal = [l for l in lines if l.startswith('spec-tier recheck bound')]
    aud = [l for l in lines if l.startswith('audit (spec loop)')]
    assert len(cal) == 2 and 'clip 2 of 2' in cal[1] and len(aud) == 2 and len(rc.audit_log) == 2
    for rec in rc.audit_log:
        assert rec['loop'] == 'spec' and rec['audited'] == 24 and 0 <= rec['voted_on_tier1'] <= 24 and rec['disagreements'] == []
        assert rec['max_leader_diff_error'] < rec['tau_spec']
    eng.close()


def test_exact_vote_queue_drains_mid_call(weights, sched, monkeypatch):
    """The recheck queue holds a bounded number of sample indices; a call that votes more samples than fit drains it between
    batches.  With an 8-entry queue (DMAD_RECHECK_QUEUE, a test knob) and every sample forced through the recheck tiers the
    counts, logits and stats equal those of the default-sized queue; NaN clips are queued, reach the fp32 path and vote like
    torch.max does."""
    from dmad_hip import engine as E
    hp, coef = sched
    clip = torch.from_numpy(synth.synthetic_clip(2)).cuda()
    sc = float(torch.tensor((1 / 1.25) ** 0.5))
    out = {}
    for cap in ('8', None):
        if cap:
            monkeypatch.setenv('DMAD_RECHECK_QUEUE', cap)
        else:
            monkeypatch.delenv('DMAD_RECHECK_QUEUE', raising=False)
        eng = E.Engine(max_batch=4, precision=E.EXACT, recheck_batch=3, recheck_margin=1e30)     # nothing votes from the 16-bit pass
        eng.load_wavenet(weights[0])
        eng.load_vgg19_bn(weights[1])
        c, lg, _ = eng.smooth_votes(clip, 0.5, sc, 65, *coef(65), 30, batch=4, seed=12, sample0=7, want_logits=True)
        out[cap] = (c.cpu().tolist(), lg.cpu(), eng.recheck_stats(detail=True))
        if cap is None:
            eng.set_mode(E.MODE_FP32)
            c32, lg32, _ = eng.smooth_votes(clip, 0.5, sc, 65, *coef(65), 30, batch=4, seed=12, sample0=7, want_logits=True)
            assert c32.cpu().tolist() == out[None][0]
            eng.set_mode(E.MODE_EXACT_VOTES)
            eng.recheck_stats(reset=True)
            bad = clip.clone()
            bad[0, 100] = float('nan')                      # a NaN clip: NaN logits everywhere
            cn, lgn, _ = eng.smooth_votes(bad, 0.5, sc, 65, *coef(65), 5, batch=4, seed=1, want_logits=True)
            assert bool(torch.isnan(lgn).all()) and int(cn.sum()) == 5 and eng.recheck_stats(detail=True) == (5, 5, 5)
            assert cn.cpu().tolist() == [5, 0, 0, 0, 0, 0, 0, 0, 0, 0]          # first NaN wins, as torch.max picks it
        eng.close()
    assert out['8'][0] == out[None][0] and torch.equal(out['8'][1], out[None][1])
    assert out['8'][2][:2] == out[None][2][:2] == (30, 30) and sum(out[None][0]) == 30


def test_recheck_bounds_are_calibrated_for_the_resident_weights(exact_engine, sched):
    """Engine.calibrate_recheck measures the two error statistics of the exact-vote mode for the weights that are loaded (the
    defaults were measured on the synthetic VGG19_bn over 36 864 samples; a checkpoint with another logit sensitivity needs its
    own).  A calibration may only WIDEN a bound: a maximum over a few hundred samples underestimates the tail, so on the
    synthetic pair — whose errors sit below the committed defaults — the bounds stay exactly at the defaults; a small headroom
    multiple is overridden by the floor, a huge one widens.  Mode and statistics are untouched; RobustCertificate(calibrate=n)
    measures the first `calibrate_clips` clips of a sigma and keeps the widest bounds."""
    from audio_models.ConvNets_SpeechCommands.models.vgg import vgg19_bn  # noqa: F401
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from dmad_hip import engine as E
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    eng = exact_engine
    hp, coef = sched
    clips = [torch.from_numpy(synth.synthetic_clip(i)).cuda() for i in (1, 2)]
    sc = float(torch.tensor((1 / 1.25) ** 0.5))
    d1, d2 = E.DEFAULT_RECHECK_MARGIN[E.HALF_F16], E.DEFAULT_RECHECK_MARGIN2
    old = (eng.recheck_margin, eng.recheck_margin2)
    stats = eng.recheck_stats(detail=True)
    t1, t2, e1, e2 = eng.calibrate_recheck(clips, 0.5, sc, 65, *coef(65), n=256, n_fp32=64)
    assert 0.004 < e1 < d1 and 1e-5 < e2 < d2, (e1, e2)                  # the defaults' regime: 0.0244 / 2.5e-4 at N = 36 864
    cal = eng.calibration
    assert cal['clips'] == 2 and cal['e1'] == e1 and cal['n_fp32'] == 64 and 0.002 < cal['s1'] < 0.01
    # the bounds never go below the committed defaults: max(default, headroom x observed maximum, tail rule on the observed scale)
    assert t2 == d2 and t1 == pytest.approx(max(d1, 1.5 * e1 + t2, E.TAIL_Z * cal['s1'] + t2)) and d1 <= t1 < 1.25 * d1
    assert eng.recheck_margin == pytest.approx(t1)
    assert eng.mode == E.MODE_EXACT_VOTES and eng.recheck_stats(detail=True) == stats  # nothing voted, nothing was reset
    w1, w2, _, _ = eng.calibrate_recheck(clips[0], 0.5, sc, 65, *coef(65), n=128, n_fp32=32, headroom=50.0)
    assert w2 == pytest.approx(max(d2, 50 * eng.calibration['e2'])) and w1 == pytest.approx(50 * eng.calibration['e1'] + w2) and w1 > 5 * d1
    eng.set_recheck_margin(old[0]); eng.set_recheck_margin2(old[1])
    den = DiffWave(WaveNetHIP(eng), hp)
    lines = []
    rc = RobustCertificate(classifier=synth_vgg().bind_engine(eng), transform=MelSpectrogramDB(eng), denoiser=den, seed=2, calibrate=64,
                           calibrate_clips=2, log=lines.append)
    a = rc.smooth_predict(clips[0], num_sampling=32, sigma=0.5, batch_size=16)
    assert int(a.sum()) == 32 and list(rc._calibrated) == [65] and eng.recheck_margin == pytest.approx(rc._calibrated[65][0]) and rc._calibrated[65][0] >= d1
    rc.smooth_predict(clips[1], num_sampling=8, sigma=0.5, batch_size=8)
    rc.smooth_predict(clips[1], num_sampling=8, sigma=0.5, batch_size=8)
    assert {k: len(v) for k, v in rc._calibrated_clips.items()} == {65: 2} and len(lines) == 2 and 'tau1' in lines[0]     # two clips measured (each once), then no more
    eng.set_recheck_margin(old[0]); eng.set_recheck_margin2(old[1])


def test_eval_samples_and_audit(exact_engine, sched):
    """dmad_eval_samples evaluates an explicit list of Monte Carlo samples on an explicit tier from their Philox keys: its rows
    equal the rows the vote loop produced for the same global indices (bit for bit on the same tier).  RobustCertificate.audit
    re-evaluates tier-1 voters of the last smooth_predict on the split-f16 tier: no disagreement at the committed bound, and a
    bound of zero (every sample votes on tier 1) makes the audit see every flip the 16-bit tier commits."""
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from dmad_hip import engine as E
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate
    eng = exact_engine
    hp, coef = sched
    clip = torch.from_numpy(synth.synthetic_clip(3)).cuda()
    sc = float(torch.tensor((1 / 1.25) ** 0.5))
    args = (clip, 0.5, sc, 65) + coef(65)
    eng.set_mode(E.MODE_FAST)
    _, lg_fast, _ = eng.smooth_votes(*args, 48, batch=16, seed=5, sample0=100, want_logits=True)
    idx = torch.tensor([147, 100, 101, 131, 120], device='cuda')
    got = eng.eval_samples(*args, idx, path=0, seed=5)
    assert torch.equal(got, lg_fast[idx - 100])
    eng.set_mode(E.MODE_FP32)
    _, lg32, x32 = eng.smooth_votes(*args, 48, batch=16, seed=5, sample0=100, want_logits=True, want_x0=True)
    eng.set_mode(E.MODE_EXACT_VOTES)
    got32, x0 = eng.eval_samples(*args, idx, path=1, seed=5, want_x0=True)
    assert torch.equal(got32, lg32[idx - 100]) and torch.equal(x0, x32[idx - 100])
    mid = eng.eval_samples(*args, idx, path=2, seed=5)
    assert float((mid - got32).abs().max()) < 1e-3 < float((got - got32).abs().max()) + 1.0
    from dmad_hip._lib import DmadError
    with pytest.raises(DmadError):
        eng.eval_samples(*args, idx, path=5, seed=5)
    # audit through the host mirror
    den = DiffWave(WaveNetHIP(eng), hp)
    rc = RobustCertificate(classifier=synth_vgg().bind_engine(eng), transform=MelSpectrogramDB(eng), denoiser=den, seed=9)
    with pytest.raises(RuntimeError):
        rc.audit(clip, 8)                                                   # nothing to audit yet
    y, r = rc.certify(clip.reshape(1, 1, -1), torch.tensor([0], device='cuda'), sigma=0.5, n_0=16, n=192, batch_size=64, audit=96)
    rec = rc.audit_log[-1]
    assert rec['audited'] == 96 and 0 < rec['voted_on_tier1'] <= 96 and rec['disagreements'] == []
    assert 0 < rec['max_leader_diff_error'] < rec['tau1'] == eng.recheck_margin
    old = eng.recheck_margin
    eng.set_recheck_margin(0.0)                                             # every finite sample votes on the 16-bit tier
    try:
        rc.smooth_predict(clip, num_sampling=192, sigma=0.5, batch_size=64)
        rec0 = rc.audit(clip, 192)
    finally:
        eng.set_recheck_margin(old)
    assert rec0['voted_on_tier1'] == 192
    for i, a, b, m in rec0['disagreements']:                                # whatever flipped did so inside the committed bound
        assert a != b and 0 <= m < old


def test_rounding_attribution_hook(exact_engine, sched):
    """dmad_debug_rounding switches single roundings of the 16-bit path on inside the split-f16 tier (tools/gpu_error_attribution.py).
    All of them together reproduce the order of magnitude of the 16-bit tier's own logit error; each alone stays below that;
    masks off = the product tier bit for bit."""
    from dmad_hip import engine as E
    eng = exact_engine
    hp, coef = sched
    clip = torch.from_numpy(synth.synthetic_clip(0)).cuda()
    sc = float(torch.tensor((1 / 1.25) ** 0.5))
    args = (clip, 0.5, sc, 65) + coef(65)
    idx = torch.arange(32, device='cuda')
    ref = eng.eval_samples(*args, idx, path=2, seed=3)
    eng.set_mode(E.MODE_FAST)
    fast = eng.eval_samples(*args, idx, path=0, seed=3)
    eng.set_mode(E.MODE_EXACT_VOTES)

    def err(**m):
        eng.debug_rounding(**m)
        try:
            return float((eng.eval_samples(*args, idx, path=2, seed=3) - ref).abs().max())
        finally:
            eng.debug_rounding()
    e_fast = float((fast - ref).abs().max())
    e_all = err(dil=3, res=7, skip=3, f0=3, init=1)
    e_w = err(dil=1, res=1, skip=1, f0=1)
    e_h = err(res=4, init=1)
    assert 0.25 * e_fast < e_all < 4 * e_fast, (e_all, e_fast)
    assert 0 < e_w < e_all * 1.5 and 0 < e_h < e_all * 1.5, (e_w, e_h, e_all)
    assert torch.equal(eng.eval_samples(*args, idx, path=2, seed=3), ref)
    from dmad_hip._lib import DmadError
    with pytest.raises(DmadError):
        eng.debug_rounding(dil=9)


def test_f16_engine_warns_about_weights_outside_the_half_range():
    """Folded WaveNet weights below 2^-14 become f16 subnormals (fewer significant bits than the 16-bit tier's error bound was
    measured with), values above 65504 overflow: dmad_finalize_weights reports both (dmad_last_warning) and the Python engine
    raises a RuntimeWarning; the synthetic weights are inside the range and load silently."""
    import warnings
    from dmad_hip import engine as E
    sd = synth.wavenet_state_dict(1234)
    tiny = {k: (v * 1e-4 if 'res_conv.weight_g' in k else v) for k, v in sd.items()}     # weight-norm gain: scales the folded res weights
    eng = E.Engine(max_batch=1, precision=E.BF16, half_type=E.HALF_F16, with_classifier=False)
    with pytest.warns(RuntimeWarning, match='in 36 of 109 folded weight tensors f16 subnormals'):
        eng.load_wavenet(tiny)
    eng.close()
    eng = E.Engine(max_batch=1, precision=E.BF16, half_type=E.HALF_F16, with_classifier=False)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        eng.load_wavenet(sd)
    eng.close()


def test_autograd_branch(engines, weights, sched):
    """callers that differentiate through the system (x.requires_grad with gradients enabled: the white-box attack
    drivers) get a torch restatement on that branch (dmad_hip/autograd.py) instead of the inference-only HIP engine.  Checked WITH
    the HIP fp32 path: the branch's forward values equal the engine's stage by stage and end to end, and its gradient equals the
    directional finite difference of the ENGINE's output along the gradient direction.  Calls without requires_grad stay on the
    engine; CPU tensors are refused on both branches."""
    from acoustic_system import AcousticSystem
    from diffusion_models.diffwave_ddpm import DiffWave, WaveNetHIP
    from dmad_hip._lib import DmadError
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval._EOT import EOT
    from robustness_eval._utils import resolve_loss
    eng = engines['fp32']
    hp, coef = sched
    den = DiffWave(WaveNetHIP(eng, state_dict=weights[0]), hp, reverse_timestep=2, noise_source='device', seed=3)
    net = synth_vgg().cuda().bind_engine(eng)
    mel = MelSpectrogramDB(eng)
    model = AcousticSystem(classifier=net, transform=mel, defender=den, defense_type='wave')
    x = torch.from_numpy(np.stack([synth.synthetic_clip(0), synth.synthetic_clip(5)])).cuda()          # [2, 1, 16000]

    # stage by stage: torch restatement (requires_grad) against the engine (no grad), same inputs
    xg = x.clone().requires_grad_(True)
    eps_t = den.model((xg, 40 * torch.ones(2, 1)))
    eps_h = den.model((x, 40 * torch.ones(2, 1)))
    assert eps_t.requires_grad and not eps_h.requires_grad and relmax(eps_t.detach().cpu().numpy(), eps_h.cpu().numpy()) < 1e-4
    one_t, one_h = den.one_shot_denoise(xg), den.one_shot_denoise(x)
    assert one_t.requires_grad and relmax(one_t.detach().cpu().numpy(), one_h.cpu().numpy()) < 1e-4
    sp_t, sp_h = mel(xg), mel(x)
    assert sp_t.requires_grad and float((sp_t.detach() - sp_h).abs().max()) < 5e-3                    # dB
    sg = sp_h.clone().requires_grad_(True)
    lg_t, lg_h = net(sg), net(sp_h)
    assert lg_t.requires_grad and relmax(lg_t.detach().cpu().numpy(), lg_h.cpu().numpy()) < 2e-4

    # end to end (DDPM t* = 2, device noise keyed by the sample counter): same keys on both branches
    def logits_engine(xx):
        den._draws = 0
        with torch.no_grad():
            return model(xx)
    den._draws = 0
    out = model(xg)
    ref = logits_engine(x)
    assert out.requires_grad and relmax(out.detach().cpu().numpy(), ref.cpu().numpy()) < 2e-3
    cls = int(ref[0].argmax())
    (g,) = torch.autograd.grad(out[:, cls].sum(), xg)
    assert g.shape == x.shape and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    v = g / g.norm()
    h = 0.05 / float(g.norm())
    fd = float((logits_engine(x + h * v)[:, cls].sum() - logits_engine(x - h * v)[:, cls].sum()) / (2 * h))
    assert abs(fd - float(g.norm())) < 0.1 * float(g.norm()), (fd, float(g.norm()))

    # EOT with gradients (reference _EOT.py:36-66): one model call per EOT batch on a leaf that requires grad
    loss_fn, _ = resolve_loss('Margin', False, 0., 'SCR', None, False)
    y = torch.tensor([0, 6]).cuda()
    den._draws = 0
    scores, loss, grad, decisions = EOT(model, loss_fn, EOT_size=2, EOT_batch_size=1, use_grad=True)(x, y)
    assert grad.shape == x.shape and bool(torch.isfinite(grad).all()) and float(grad.abs().max()) > 0 and scores.shape == (2, 10)
    assert [len(d) for d in decisions] == [2, 2]
    # the branches refuse CPU tensors alike
    with pytest.raises(DmadError):
        den.model((x.cpu().requires_grad_(True), 40 * torch.ones(2, 1)))
    with pytest.raises(DmadError):
        mel(x.cpu().requires_grad_(True))
    with pytest.raises(NotImplementedError):
        net(sp_h.cpu().requires_grad_(True))
