"""CPU-side tests (-m "not gpu"): C-ABI surface, host logic of the reference-named Python mirrors,
checkpoint loaders, and the multi-process (gloo, world_size 2) vote sharding."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from dmad_hip import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')
LIB = os.path.join(PKG, 'libdmad_hip.so')


@pytest.fixture(scope='module')
def built_lib():
    if not os.path.exists(LIB):
        subprocess.run(['make', '-C', os.path.join(PKG, 'csrc'), '-j4'], check=True)
    return ctypes.CDLL(LIB)


def test_c_abi_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, 'include', 'dmad.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(dmad_[a-z_0-9]+)\s*\(', hdr))
    assert len(declared) >= 20
    from dmad_hip import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(built_lib, name), name


def test_c_abi_fails_loudly_without_gpu(built_lib):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from dmad_hip import _lib, engine
    with pytest.raises(_lib.DmadError):
        engine.Engine(max_batch=1)                        # python wrapper refuses: no CPU path
    lib = _lib.load()
    cfg = _lib.DmadConfig(256, 256, 36, 12, 128, 512, 512, 16000, 1, 10, 0, 1, 0, 0)
    h = ctypes.c_void_p()
    rc = lib.dmad_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -3 and lib.dmad_last_error()             # DMAD_ERR_HIP with a message
    bad = _lib.DmadConfig(128, 256, 36, 12, 128, 512, 512, 16000, 1, 10, 0, 1, 0, 0)
    assert lib.dmad_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert b'256' in lib.dmad_last_error()
    # a caller built against another revision of dmad.h (struct_size mismatch) is refused before any field is trusted
    assert cfg.struct_size == ctypes.sizeof(_lib.DmadConfig) == 16 * 4 and cfg.with_wavenet == 1
    old = _lib.DmadConfig(256, 256, 36, 12, 128, 512, 512, 16000, 1, 10, 0, 1, 0, 0)
    old.struct_size = 12 * 4                               # the round-1 layout: no struct_size, recheck_batch, half_type
    assert lib.dmad_create(ctypes.byref(old), ctypes.byref(h)) == -1 and b'struct_size' in lib.dmad_last_error()
    assert b'dmad-hip 0.5' in lib.dmad_version() and lib.dmad_last_warning() == b''


def test_c_abi_from_a_plain_c_caller(built_lib, tmp_path):
    """The boundary is a C ABI: include/dmad.h compiles as strict C99, and a plain-C host (tests/c_abi_caller.c: dlopen + the exports
    a minimal host needs) sees the revision handshake, the refusal of unsupported geometry and — without a GPU — a loud DMAD_ERR_HIP
    (on a GPU box: an engine created and destroyed)."""
    exe = str(tmp_path / 'c_abi_caller')
    cc = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(ROOT, 'include'),
                         os.path.join(ROOT, 'tests', 'c_abi_caller.c'), '-ldl', '-o', exe], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe, LIB], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert 'library: dmad-hip' in run.stdout
    assert ('engine destroyed' in run.stdout) if torch.cuda.is_available() else ('DMAD_ERR_HIP' in run.stdout)


def test_development_patches_still_apply_to_the_layer_kernel(tmp_path):
    """tools/patches/*.patch hold the layer kernel's ablation / experiment variants (profiles/r05_layer_gate_variants.md) outside the
    product source, which is under bench.py's hash guard; tools/layer_variants.sh applies them to a temporary copy.  They must not rot."""
    import glob
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which('patch') is None:
        pytest.skip('no patch(1) here')
    patches = sorted(glob.glob(os.path.join(root, 'tools', 'patches', 'wnl_*.patch')))
    assert len(patches) >= 3
    for pf in patches:
        shutil.copy(os.path.join(root, 'diffusion-model-for-audio-defense_amd', 'csrc', 'wn_layer.hip'), tmp_path / 'wn_layer.hip')
        with open(pf) as f:
            run = subprocess.run(['patch', '-s', '-p3'], stdin=f, cwd=tmp_path, capture_output=True, text=True)
        assert run.returncode == 0, (pf, run.stdout, run.stderr)
        assert 'WNL_VARIANT ==' in (tmp_path / 'wn_layer.hip').read_text()


def test_lds_layouts_are_bank_conflict_free():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'lds_bank_check.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_schedule_mirror_bit_exact_and_t_star(golden_dir):
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from robustness_eval.certified_robust import RobustCertificate
    z = np.load(os.path.join(golden_dir, 'schedule.npz'))
    hp = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    for k in ('Beta', 'Alpha', 'Alpha_bar', 'Sigma'):
        assert np.array_equal(hp[k].numpy(), z[k])

    class D:
        diffusion_hyperparams = hp
    rc = RobustCertificate(classifier=None, denoiser=D())
    for s, t in zip(z['sigmas'], z['t_star']):
        assert rc.compute_t_star(1 / (1 + float(s) ** 2)) == int(t)


def test_lower_conf_bound_known_answers(golden_dir):
    from robustness_eval.certified_robust import RobustCertificate
    rc = RobustCertificate(classifier=None)
    for r in json.load(open(os.path.join(golden_dir, 'clopper_pearson.json'))):
        assert abs(rc.lower_conf_bound(r['k'], r['n'], r['alpha']) - r['pa']) < 1e-12
    assert rc.lower_conf_bound(0, 100) == 0.0
    assert rc.lower_conf_bound(torch.tensor(990), 1000) == pytest.approx(0.976036, abs=1e-5)


class _ToyClassifier(torch.nn.Module):
    """deterministic raw-waveform classifier for host-logic tests (no HIP involved)."""
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(1)
        self.w = torch.randn(16000, 10, generator=g) * 0.05

    def forward(self, x):
        return x.reshape(x.shape[0], -1) @ self.w


def test_randsmooth_host_logic_matches_oracle_cpu():
    """denoiser=None ('randsmooth' mode of certified_robustness_eval.py:94-95) runs without the engine."""
    from oracle import dmad_oracle as orc
    from robustness_eval.certified_robust import RobustCertificate
    clf = _ToyClassifier()
    x = torch.from_numpy(synth.synthetic_clip(0))
    rc = RobustCertificate(classifier=clf, transform=None, denoiser=None, noise_source='torch_cpu')
    torch.manual_seed(5)
    got = rc.smooth_predict(x, num_sampling=70, sigma=0.5, batch_size=16)
    torch.manual_seed(5)
    ref = orc.CertifyOracle(clf, None, None).smooth_predict(x, num_sampling=70, sigma=0.5, batch_size=16)
    assert got.dtype == torch.int64 and got.tolist() == ref.tolist() and int(got.sum()) == 70
    torch.manual_seed(6)
    yp, rad = rc.certify(x[None], torch.tensor([1]), sigma=0.5, n_0=20, n=50, batch_size=16)
    torch.manual_seed(6)
    yr, rr = orc.CertifyOracle(clf, None, None).certify(x[None], torch.tensor([1]), sigma=0.5, n_0=20, n=50, batch_size=16)
    assert yp.tolist() == yr.tolist() and torch.allclose(rad, rr)
    with pytest.raises(AssertionError):
        rc.smooth_predict(torch.zeros(2, 16000), 4, 0.5, 4)


def _gloo_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path[:0] = [PKG, ROOT]
    from robustness_eval.certified_robust import RobustCertificate
    rc = RobustCertificate(classifier=_ToyClassifier(), transform=None, denoiser=None, noise_source='torch_cpu')
    torch.manual_seed(5)
    counts = rc.smooth_predict(torch.from_numpy(synth.synthetic_clip(0)), num_sampling=70, sigma=0.5, batch_size=16)
    if rank == 0:
        torch.save(counts, out)
    dist.destroy_process_group()


def test_vote_sharding_world_size_2_gloo(tmp_path):
    """N samples sharded over 2 ranks + one all_reduce == the single-process counts."""
    import torch.multiprocessing as mp
    from robustness_eval.certified_robust import RobustCertificate
    out = str(tmp_path / 'counts.pt')
    port = 29500 + os.getpid() % 2000
    mp.spawn(_gloo_worker, args=(2, port, out), nprocs=2, join=True)
    sharded = torch.load(out)
    rc = RobustCertificate(classifier=_ToyClassifier(), transform=None, denoiser=None, noise_source='torch_cpu')
    torch.manual_seed(5)
    single = rc.smooth_predict(torch.from_numpy(synth.synthetic_clip(0)), num_sampling=70, sigma=0.5, batch_size=16)
    assert sharded.tolist() == single.tolist()


def test_acoustic_system_semantics():
    from acoustic_system import AcousticSystem
    calls = []

    class Def(torch.nn.Module):
        def forward(self, x):
            calls.append('def'); return x * 2

    sys_ = AcousticSystem(classifier=lambda s: s.sum(-1), transform=lambda w: w + 1, defender=Def(), defense_type='wave')
    x = torch.full((2, 1, 4), 0.25)
    assert torch.allclose(sys_(x), torch.full((2, 1), 6.0)) and calls == ['def']
    assert torch.allclose(sys_(x, defend=False), torch.full((2, 1), 5.0))
    big = torch.tensor([[[20000.0, -20000.0, 0.0, 0.0]]])
    assert torch.allclose(sys_(big, defend=False), (big / 2 ** 15 + 1).sum(-1))     # int16-range rescale
    with pytest.raises(NotImplementedError):
        AcousticSystem(None, None, None, defense_type='image')
    spec = AcousticSystem(classifier=lambda s: s, transform=lambda w: w + 1, defender=Def(), defense_type='spec')
    assert torch.allclose(spec(x), (x + 1) * 2)


def test_spec_defense_query_routing_host_logic():
    """AcousticSystem(defense_type='spec').query routes to ONE engine call (spec_query_logits) exactly when the classifier, the mel
    transform and a SpecPurifier sit on one engine: argument block = the purifier's coefficient table + the mel-dB bounds, rows
    keyed from the purifier's draw counter, the counter advanced by repeats * B; everything else loops over forward()
    (acoustic_system.py:40-49; host logic only: the engine is a stub)."""
    from acoustic_system import AcousticSystem
    from diffusion_models.improved_diffusion_ddpm import ImprovedDiffusion, SpecPurifier
    from dmad_hip.transforms import MelSpectrogramDB

    class Eng:
        has_classifier, has_wavenet, num_classes = True, False, 10

        def __init__(self):
            self.calls = []

        def spec_query_logits(self, x, repeats, *args, seed=0, sample0=0):
            self.calls.append(('spec', tuple(x.shape), repeats, args, seed, sample0))
            n = repeats * x.shape[0]
            return torch.arange(n * 10, dtype=torch.float32).reshape(n, 10), torch.full((n,), 9, dtype=torch.int32)

        def query_logits(self, x, repeats, sampler, *a, **k):
            self.calls.append(('wave', sampler, repeats))
            n = repeats * x.shape[0]
            return torch.zeros(n, 10), torch.zeros(n, dtype=torch.int32)

    class Model(torch.nn.Module):           # stands for the HIP UNet: only `.engine` is read on this path
        pass

    class Diff:
        def _f32(self, arr, t):
            return float(arr[t])
        sqrt_alphas_cumprod = sqrt_one_minus_alphas_cumprod = sqrt_recip_alphas_cumprod = sqrt_recipm1_alphas_cumprod = [0.9, 0.8, 0.7, 0.6]
        posterior_mean_coef1 = posterior_mean_coef2 = [0.1, 0.2, 0.3, 0.4]
        model_log_variance = [-9.0, -8.0, -7.0, -6.0]
    eng = Eng()
    model = Model(); model.engine = eng
    pur = ImprovedDiffusion(model=model, diffusion=Diff(), reverse_timestep=2)
    den = SpecPurifier(pur, seed=77)
    clf = torch.nn.Identity(); clf.engine = eng
    mel = MelSpectrogramDB(eng)
    sys_ = AcousticSystem(classifier=clf, transform=mel, defender=den, defense_type='spec')
    assert sys_._engine_chain(True) == (eng, 3) and sys_._engine_chain(False) == (eng, 0)
    x = torch.zeros(3, 1, 16000)

    class FakeCuda(torch.Tensor):           # query() takes the engine route for CUDA tensors only
        @property
        def is_cuda(self):
            return True
    xc = x.as_subclass(FakeCuda)
    den._draws = 50
    logits, dec = sys_.query(xc, repeats=4)
    assert logits.shape == (4, 3, 10) and dec.shape == (4, 3) and dec.dtype == torch.int64 and den._draws == 62
    kind, shape, rep, args, seed, s0 = eng.calls[-1]
    assert (kind, shape, rep, seed, s0) == ('spec', (3, 1, 16000), 4, 77, 50)
    ts, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, lo, hi = args
    assert ts == 2 and len(c_a) == len(c_sig) == 3 and c_sig[0] == 0.0 and (lo, hi) == (-100.0, 38.22) and abs(q_a - 0.7) < 1e-6
    sys_.query(xc, repeats=2, defend=False)
    assert eng.calls[-1] == ('wave', 0, 2) and den._draws == 62                    # no defender: the plain query, no draws consumed
    other = AcousticSystem(classifier=clf, transform=mel, defender=torch.nn.Identity(), defense_type='spec')
    assert other._engine_chain(True) == (None, 0)                                   # an unknown spec defender: the forward loop
    den2 = SpecPurifier(ImprovedDiffusion(model=Model(), diffusion=Diff(), reverse_timestep=2))
    assert AcousticSystem(classifier=clf, transform=mel, defender=den2, defense_type='spec')._engine_chain(True) == (None, 0)   # a purifier on no / another engine


def test_create_model_checkpoint_layouts(tmp_path, golden_dir):
    """whole-module pickles: bare M5 and DataParallel(VGG) under a 'ConvNets_SpeechCommands' path."""
    from audio_models.ConvNets_SpeechCommands.create_model import create_model
    import M5Net
    from models.vgg import VGG, vgg19_bn
    m5 = M5Net.M5(n_input=1, first_kernel_size=160, n_output=10, stride=16, n_channel=32)
    m5.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, 'm5_k160_state.npz')).items()})
    p = str(tmp_path / 'm5.pth')
    torch.save(m5.double().train(), p)
    got = create_model(p)
    assert isinstance(got, M5Net.M5) and not got.training and next(got.parameters()).dtype == torch.float32
    z = np.load(os.path.join(golden_dir, 'classifiers.npz'))
    np.testing.assert_allclose(got(torch.from_numpy(z['wave_in'])).detach().numpy(), z['m5_logp'], rtol=1e-5, atol=1e-5)
    d = tmp_path / 'ConvNets_SpeechCommands'
    d.mkdir()
    net = vgg19_bn(num_classes=10, in_channels=1)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg19_bn_state_dict(4321).items()})
    p2 = str(d / 'vgg.pth')
    torch.save(torch.nn.DataParallel(net), p2)
    got = create_model(p2)
    assert isinstance(got, VGG) and not got.training
    assert set(got.state_dict()) == set(synth.vgg19_bn_state_dict(4321))
    with pytest.raises(Exception):
        got(torch.zeros(1, 1, 32, 32))                    # CPU tensor: the HIP module has no CPU path
    # the certification driver's DEFAULT classifier (certified_robustness_eval.py:57-59): a pickled DataParallel(CifarResNeXt)
    from models.resnext import CifarResNeXt
    rx = CifarResNeXt(nlabels=10, in_channels=1)
    rx.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.resnext29_state_dict(2929).items()})
    p3 = str(d / 'gaussian_aug_resnext29_8_64.pth')
    torch.save(torch.nn.DataParallel(rx), p3)
    got = create_model(p3)
    assert isinstance(got, CifarResNeXt) and not got.training and next(got.parameters()).dtype == torch.float32
    assert torch.equal(got.state_dict()['stage_2.stage_2_bottleneck_0.conv_conv.weight'],
                       torch.from_numpy(synth.resnext29_state_dict(2929)['stage_2.stage_2_bottleneck_0.conv_conv.weight']))
    evil = str(d / 'evil.pth')                             # anything outside the allow-list is refused, not executed
    torch.save(torch.nn.DataParallel(torch.nn.Sequential(torch.nn.Tanh())), evil)
    with pytest.raises(Exception):
        create_model(evil)


def test_weight_folding_matches_oracle():
    from dmad_hip.engine import fold_vgg19_bn_state_dict, fold_wavenet_state_dict
    from oracle import dmad_oracle as orc
    sd = synth.wavenet_state_dict(1234)
    f = fold_wavenet_state_dict(sd, 36)
    w = orc.folded_weights(sd)
    assert np.array_equal(f['dil.7.w'], w['dil.7.w'].numpy())
    assert np.array_equal(f['res.35.w'], w['res.35.w'].numpy()[:, :, 0])
    assert np.array_equal(f['init.w'], w['init.w'].numpy().reshape(256))
    assert len(f) == 6 + 36 * 8 + 4
    vsd = synth.vgg19_bn_state_dict(4321)
    fv = fold_vgg19_bn_state_dict(vsd)
    x = torch.randn(2, 64, 8, 8)
    conv = torch.nn.functional.conv2d(x, torch.from_numpy(vsd['features.3.weight']), None, padding=1)
    ref = torch.nn.functional.batch_norm(conv + torch.from_numpy(vsd['features.3.bias']).view(1, -1, 1, 1),
                                         torch.from_numpy(vsd['features.4.running_mean']), torch.from_numpy(vsd['features.4.running_var']),
                                         torch.from_numpy(vsd['features.4.weight']), torch.from_numpy(vsd['features.4.bias']), False, 0., 1e-5)
    got = conv * torch.from_numpy(fv['vgg.conv1.scale']).view(1, -1, 1, 1) + torch.from_numpy(fv['vgg.conv1.shift']).view(1, -1, 1, 1)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4)


def test_wavenet_surface_rejects_unsupported_use():
    from diffusion_models.diffwave_ddpm import WaveNetHIP

    class FakeEngine:
        def wavenet_eps(self, a, t):
            return a[:, 0] * 0 + t
    m = WaveNetHIP(FakeEngine())
    x = torch.zeros(3, 1, 8)
    assert torch.equal(m((x, 5 * torch.ones(3, 1))), torch.full((3, 1, 8), 5.0))
    with pytest.raises(NotImplementedError):
        m((x, torch.tensor([[1.0], [2.0], [1.0]])))       # per-row steps
    with pytest.raises(NotImplementedError):
        m((x.requires_grad_(), 5 * torch.ones(3, 1)))     # autograd


# ------------------------------------------------------------------------------------------ host ingest
def _write_wav(path, samples_i16, rate=16000, channels=1):
    import wave
    with wave.open(path, 'wb') as w:
        w.setnchannels(channels); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(np.asarray(samples_i16, dtype='<i2').tobytes())


def test_wav_loader_and_fix_length(tmp_path):
    """LoadAudio = PCM16 / 32768 as float32 (what librosa.load returns for a 16 kHz mono file), channel mean for
    stereo, a loud failure on another sample rate; FixAudioLength pads with zeros / truncates to time * rate."""
    from transforms import LoadAudio, FixAudioLength
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, size=12000, dtype=np.int16)
    _write_wav(str(tmp_path / 'a.wav'), pcm)
    d = LoadAudio()({'path': str(tmp_path / 'a.wav'), 'target': 3})
    assert d['sample_rate'] == 16000 and d['samples'].dtype == np.float32 and d['target'] == 3
    assert np.array_equal(d['samples'], pcm.astype(np.float32) / 32768.0)
    d = FixAudioLength()(d)
    assert d['samples'].shape == (16000,) and np.all(d['samples'][12000:] == 0)
    assert np.array_equal(d['samples'][:12000], pcm.astype(np.float32) / 32768.0)
    long = rng.integers(-1000, 1000, size=20000, dtype=np.int16)
    _write_wav(str(tmp_path / 'b.wav'), long)
    d = FixAudioLength()(LoadAudio()({'path': str(tmp_path / 'b.wav')}))
    assert np.array_equal(d['samples'], long[:16000].astype(np.float32) / 32768.0)
    st = rng.integers(-1000, 1000, size=(500, 2), dtype=np.int16)
    _write_wav(str(tmp_path / 'c.wav'), st.reshape(-1), channels=2)
    d = LoadAudio()({'path': str(tmp_path / 'c.wav')})
    np.testing.assert_allclose(d['samples'], (st.astype(np.float32) / 32768.0).mean(1), rtol=0, atol=1e-7)
    d = LoadAudio()({'path': ''})                                   # the reference's "silence" item
    assert d['samples'].shape == (16000,) and not d['samples'].any()
    _write_wav(str(tmp_path / 'd.wav'), pcm, rate=8000)
    with pytest.raises(ValueError):
        LoadAudio()({'path': str(tmp_path / 'd.wav')})


def test_sc09_dataset_index(tmp_path):
    from datasets.sc_dataset import SC09Dataset, SC09_CLASSES
    from transforms import LoadAudio, FixAudioLength
    for ci, c in enumerate(SC09_CLASSES):
        os.makedirs(tmp_path / c)
        for k in range(3):
            _write_wav(str(tmp_path / c / ('%s_%d.wav' % (c, k))), np.full(100 * (k + 1), ci * 100 + k, dtype=np.int16))
    os.makedirs(tmp_path / '_background_noise_')
    ds = SC09Dataset(str(tmp_path), transform=lambda d: FixAudioLength()(LoadAudio()(d)), num_per_class=2)
    assert len(ds) == 20
    assert [t for _, t in ds.data] == sorted([t for _, t in ds.data])          # class-major order, targets 0..9
    item = ds[5]
    assert item['target'] == 2 and item['samples'].shape == (16000,)
    assert os.path.basename(os.path.dirname(item['path'])) == 'two'
    assert len(SC09Dataset(str(tmp_path), num_per_class=100)) == 30           # fewer files than num_per_class
    # balanced-class sampler weights (reference datasets/sc_dataset.py:136-149): N / (items of the item's class)
    w = ds.make_weights_for_balanced_classes()
    assert w.dtype == np.float64 and w.shape == (20,) and np.all(w == 10.0)
    os.remove(ds.data[-1][0])                                                  # an unbalanced index: 'nine' keeps one of its files
    ub = SC09Dataset(str(tmp_path), num_per_class=2)
    wu = ub.make_weights_for_balanced_classes()
    assert len(ub) == 20 and np.all(wu == 10.0)                                # (three files per class: two are still listed)
    os.remove(ub.data[-1][0])
    ub = SC09Dataset(str(tmp_path), num_per_class=2)
    wu = ub.make_weights_for_balanced_classes()
    assert len(ub) == 19 and np.all(wu[:18] == 19 / 2) and wu[18] == 19.0
    assert ds[0].keys() >= {'path', 'target'} and SC09Dataset(str(tmp_path), num_per_class=1)[3] == {'path': ds.data[6][0], 'target': 3}
    with pytest.raises(AssertionError):                                        # a missing class folder
        os.rename(tmp_path / 'three', tmp_path / 'x3')
        SC09Dataset(str(tmp_path))


def test_certification_records_format_and_resume(tmp_path):
    import json
    from robustness_eval.records import CertificationRecords
    r = CertificationRecords(str(tmp_path), 0.5, 1000)
    r.append_batch([1, 2], [1, -1], [0.25, 0.0])
    r.flush()
    path = tmp_path / 'sigma=0.5' / 'sigma=0.5_N=1000.json'
    got = json.load(open(path))
    assert got == [{'id': 0, 'y_true': 1, 'y_pred': 1, 'certified_radius': 0.25},
                   {'id': 1, 'y_true': 2, 'y_pred': -1, 'certified_radius': 0.0}]
    assert open(path).read().startswith('[\n    {\n        "id": 0,')        # indent=4, the reference's layout
    r2 = CertificationRecords(str(tmp_path), 0.5, 1000, resume=True)
    assert len(r2) == 2
    r2.append_batch([7], [7], [1.5]); r2.flush()
    assert [d['id'] for d in json.load(open(path))] == [0, 1, 2]
    assert len(CertificationRecords(str(tmp_path), 0.5, 1000)) == 0           # no resume: starts over


def test_resnext29_mirror_layout_and_folding():
    """The module mirror keeps the reference's parameter names (checkpoints load unchanged); folding yields one
    scale/shift per conv with eval-BatchNorm semantics; no CPU forward exists."""
    from audio_models.ConvNets_SpeechCommands.models import create_model as zoo_create
    from audio_models.ConvNets_SpeechCommands.models.resnext import CifarResNeXt
    from dmad_hip.engine import fold_resnext29_state_dict
    sd = synth.resnext29_state_dict(2929)
    net = CifarResNeXt(nlabels=10, in_channels=1)
    assert isinstance(zoo_create('resnext29_8_64', 10, 1), CifarResNeXt)
    assert set(net.state_dict().keys()) == set(sd.keys())
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    f = fold_resnext29_state_dict(net.state_dict())
    assert f['rx.b0.reduce.w'].shape == (512, 64) and f['rx.b0.conv.w'].shape == (512, 64, 3, 3)
    assert f['rx.b3.conv.w'].shape == (1024, 128, 3, 3) and f['rx.b6.short.w'].shape == (1024, 512)
    assert 'rx.b1.short.w' not in f and f['rx.fc.w'].shape == (10, 1024)
    g, b = sd['stage_2.stage_2_bottleneck_1.bn.weight'].astype(np.float64), sd['stage_2.stage_2_bottleneck_1.bn.bias'].astype(np.float64)
    m, v = sd['stage_2.stage_2_bottleneck_1.bn.running_mean'].astype(np.float64), sd['stage_2.stage_2_bottleneck_1.bn.running_var'].astype(np.float64)
    np.testing.assert_allclose(f['rx.b4.conv.scale'], g / np.sqrt(v + 1e-5), rtol=1e-6)
    np.testing.assert_allclose(f['rx.b4.conv.shift'], b - m * g / np.sqrt(v + 1e-5), rtol=1e-5, atol=1e-7)
    net.eval()
    with pytest.raises(Exception):                      # no GPU here: binding the engine fails loudly, no CPU forward
        net(torch.zeros(1, 1, 32, 32))
    with pytest.raises(NotImplementedError):
        CifarResNeXt(nlabels=10, cardinality=16, in_channels=1)


def test_engine_binding_refuses_other_weights():
    """Engine.bind (host logic, no GPU needed): a part that is already resident accepts only the SAME weights; the
    fingerprint ignores BatchNorm's num_batches_tracked and distinguishes seeds."""
    from dmad_hip import engine as E
    from dmad_hip._lib import DmadError
    a, b = synth.vgg19_bn_state_dict(4321), synth.vgg19_bn_state_dict(7)
    assert E.state_fingerprint(a) == E.state_fingerprint(dict(a)) != E.state_fingerprint(b)
    withbn = dict(a)
    withbn['features.1.num_batches_tracked'] = np.array(5)
    assert E.state_fingerprint(withbn) == E.state_fingerprint(a)
    eng = E.Engine.__new__(E.Engine)                       # the bookkeeping alone (no device)
    eng.has_classifier, eng.classifier_owner, loaded = False, None, []

    def loader(sd):
        loaded.append(1); eng.has_classifier = True; eng.classifier_owner = E.state_fingerprint(sd)
    eng.bind('classifier', a, loader)
    eng.bind('classifier', a, loader)                      # same weights again: accepted, not re-uploaded
    assert loaded == [1]
    with pytest.raises(DmadError):
        eng.bind('classifier', b, loader)                  # a second, different classifier must not silently run the first
    eng._h = None


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`python bench.py --gpus N` without WORLD_SIZE starts its ranks itself; with fewer than N GPUs it fails loudly
    before touching a GPU instead of reporting a single-rank number (here: no GPU at all, or one)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and 'refusing' in r.stderr and not r.stdout.strip()
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)      # a rank count that disagrees with --gpus


def test_eot_nes_contract_on_cpu():
    """EOT / NES over a plain callable (behaviour of the reference's _EOT.py:19-69 / _NES.py:15-55, use_grad=False):
    per-call means averaged over calls, every evaluation's decision kept, NES' antithetic estimate and its extra
    division by the EOT call count."""
    from robustness_eval._EOT import EOT
    from robustness_eval._NES import NES
    from robustness_eval._utils import resolve_loss
    W = torch.randn(10, 16, generator=torch.Generator().manual_seed(0))
    calls = []

    def model(x):                                           # deterministic part + a call-dependent offset
        calls.append(x.shape[0])
        return x[:, 0, :16] @ W.t() + 0.1 * len(calls)
    loss_fn, _ = resolve_loss('Margin', False, 0., 'SCR', None, False)
    x = torch.randn(3, 1, 32, generator=torch.Generator().manual_seed(1))
    y = torch.tensor([1, 2, 3])
    scores, loss, grad, dec = EOT(model, loss_fn, EOT_size=7, EOT_batch_size=2, use_grad=False)(x, y)
    assert calls == [6, 6, 6] and grad is None                     # 7 // 2 = 3 calls of 2 repeats: the remainder is dropped
    base = x[:, 0, :16] @ W.t()
    assert torch.allclose(scores, base + 0.1 * (1 + 2 + 3) / 3, atol=1e-6)
    want_loss = sum(torch.nn.functional.cross_entropy(base + 0.1 * c, y, reduction='none') for c in (1, 2, 3)) / 3
    assert torch.allclose(loss, want_loss, atol=1e-6)
    assert [len(d) for d in dec] == [6, 6, 6] and dec[0][0] == int((base + 0.1).argmax(1)[0])
    # with gradients (reference l.36-66): one call per EOT batch on a leaf that requires grad, gradient averaged like the scores
    calls.clear()
    sg, lg, gg, dg = EOT(model, loss_fn, 4, 2, use_grad=True)(x, y)
    assert calls == [6, 6] and gg.shape == x.shape and [len(d) for d in dg] == [4, 4, 4]
    xr = x.clone().requires_grad_(True)
    want_g = torch.zeros_like(x)
    for c in (1, 2):
        (gc,) = torch.autograd.grad(torch.nn.functional.cross_entropy(xr[:, 0, :16] @ W.t() + 0.1 * c, y, reduction='sum'), xr)
        want_g += gc / 2
    assert torch.allclose(gg, want_g, atol=1e-6) and torch.allclose(sg, base + 0.1 * (1 + 2) / 2, atol=1e-6)
    calls.clear()
    torch.manual_seed(3)
    nes = NES(samples_per_draw=8, samples_per_draw_batch=4, sigma=0.01, EOT_wrapper=EOT(model, loss_fn, 2, 1, False))
    mean_loss, g, adver_loss, adver_score, predict = nes(x, y)
    assert calls == [15, 15, 12, 12]                                # (1 + 4) probes x 3 clips, then 4 x 3; two EOT calls each
    assert g.shape == x.shape and mean_loss.shape == (3,) and adver_loss.shape == (3,) and adver_score.shape == (3, 10)
    assert predict.shape == (3,) and bool(torch.isfinite(g).all())
    # the unperturbed clip sits in slot 0 of the first draw batch; EOT means over 2 calls, divided by 2 once more (ref l.33-35)
    assert torch.allclose(adver_score, (base + 0.1 * 1.5) / 2, atol=1e-5)


def test_parity_noise_is_streamed_per_batch(monkeypatch):
    """noise_source='torch_cpu' at large N: one batch of CPU noise at a time (the former implementation built the whole
    [N, 1, L] tensor: 6.4 GB at N = 100 000); the counts are unchanged by the streaming."""
    from robustness_eval.certified_robust import RobustCertificate
    clf = _ToyClassifier()
    x = torch.from_numpy(synth.synthetic_clip(0))
    rc = RobustCertificate(classifier=clf, transform=None, denoiser=None, noise_source='torch_cpu')
    biggest = [0]
    real = torch.normal

    def spy(*a, **k):
        out = real(*a, **k)
        biggest[0] = max(biggest[0], out.numel())
        return out
    monkeypatch.setattr(torch, 'normal', spy)
    torch.manual_seed(5)
    got = rc.smooth_predict(x, num_sampling=700, sigma=0.5, batch_size=16)
    assert biggest[0] == 16 * 16000 and int(got.sum()) == 700       # never more than one batch of draws alive
    monkeypatch.setattr(torch, 'normal', real)
    torch.manual_seed(5)
    parts = rc.smooth_predict(x, num_sampling=70, sigma=0.5, batch_size=16)
    from oracle import dmad_oracle as orc
    torch.manual_seed(5)
    assert parts.tolist() == orc.CertifyOracle(clf, None, None).smooth_predict(x, num_sampling=70, sigma=0.5, batch_size=16).tolist()


def test_certificate_calibration_and_audit_host_logic():
    """RobustCertificate's exact-vote housekeeping on a scripted engine (no GPU): calibration runs on the first `calibrate_clips`
    clips of a sigma and keeps the WIDEST bounds seen (never narrower from clip to clip), re-installs them when another sigma ran
    in between, logs one line per measurement; audit() re-evaluates only samples that voted on the 16-bit tier (margin >= tau1)
    and reports exactly the ones whose arg-max differs on the split-f16 tier."""
    from diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from dmad_hip.transforms import MelSpectrogramDB
    from robustness_eval.certified_robust import RobustCertificate

    class ScriptedEngine:
        has_classifier = has_wavenet = True
        precision, mode, num_classes = 2, 1, 10
        recheck_margin, recheck_margin2 = 0.034, 1e-3

        def __init__(self):
            self.cal = [(0.040, 1.0e-3, 0.020, 1e-4), (0.036, 2.0e-3, 0.018, 9e-4), (0.9, 0.9, 0.5, 0.5)]
            self.cal_calls, self.eval_calls, self.modes = [], [], []

        def set_recheck_margin(self, v, calibrated=False): self.recheck_margin = v
        def set_recheck_margin2(self, v, calibrated=False): self.recheck_margin2 = v
        def set_mode(self, m): self.modes.append(m); self.mode = m

        def calibrate_recheck(self, x, sigma, sc, t, c_a, c_b, n, n_fp32):
            self.cal_calls.append((t, n, n_fp32))
            t1, t2, e1, e2 = self.cal[len(self.cal_calls) - 1]
            self.recheck_margin, self.recheck_margin2 = t1, t2
            return t1, t2, e1, e2

        def smooth_votes(self, x, sigma, sc, t, c_a, c_b, n, seed=0, sample0=0, **kw):
            c = torch.zeros(10, dtype=torch.int64); c[3] = n
            return c, None, None

        def eval_samples(self, x, sigma, sc, t, c_a, c_b, idx, path=0, seed=0, **kw):
            self.eval_calls.append((path, idx.tolist()))
            lg = torch.zeros(len(idx), 10)
            for r, i in enumerate(idx.tolist()):
                if path == 0:                       # 16-bit tier: leader 3; sample i has margin 0.01 * (i % 8): < tau1 for i % 8 < 4
                    lg[r, 3] = 1.0; lg[r, 5] = 1.0 - 0.01 * (i % 8)
                else:                               # split-f16 tier: sample 7 flips to class 5, everything else agrees
                    lg[r, 3] = 1.0; lg[r, 5] = 1.2 if i == 7 else 0.5
            return lg

    eng = ScriptedEngine()
    cls = type('Cls', (), {})(); cls.engine = eng
    den = type('Den', (), {})(); den.engine = eng
    den.diffusion_hyperparams = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG); den.reverse_timestep = 0
    lines = []
    rc = RobustCertificate(classifier=cls, transform=MelSpectrogramDB(eng), denoiser=den, seed=4, calibrate=16, calibrate_clips=2, log=lines.append)
    assert rc._fused()
    x, x_b = torch.zeros(1, 16000), torch.full((1, 16000), 0.25)
    assert rc.smooth_predict(x, num_sampling=20, sigma=0.5, batch_size=8).tolist()[3] == 20
    assert eng.cal_calls == [(65, 16, 16)] and rc._calibrated[65] == (0.040, 1.0e-3, 0.020, 1e-4)
    rc.smooth_predict(x, num_sampling=20, sigma=0.5, batch_size=8)                 # the SAME clip again (certify's n pass after its n_0 pass):
    assert eng.cal_calls == [(65, 16, 16)] and len(lines) == 1                     # measured once per clip, not once per call
    rc.smooth_predict(x_b, num_sampling=20, sigma=0.5, batch_size=8)               # second clip: narrower tau1, wider tau2 -> elementwise max
    assert rc._calibrated[65] == (0.040, 2.0e-3, 0.020, 9e-4) and (eng.recheck_margin, eng.recheck_margin2) == (0.040, 2.0e-3)
    rc.smooth_predict(x, num_sampling=20, sigma=1.0, batch_size=8)                 # another sigma: its own calibration (third script row)
    assert eng.cal_calls[-1][0] == 116 and eng.recheck_margin == 0.9
    rc.smooth_predict(x, num_sampling=20, sigma=0.5, batch_size=8)                 # back: no new measurement, the sigma's bounds re-installed
    assert len(eng.cal_calls) == 3 and (eng.recheck_margin, eng.recheck_margin2) == (0.040, 2.0e-3) and len(lines) == 3 and 'tau1' in lines[0]
    # audit of the last smooth_predict: 16 of its 20 samples, tau1 = 0.04 -> the voters are those with i % 8 >= 4
    rec = rc.audit(x, 16)
    audited = eng.eval_calls[-2][1]
    voters = [i for i in audited if i % 8 >= 4]
    assert eng.eval_calls[-2][0] == 0 and eng.eval_calls[-1] == (2, voters) and len(audited) == 16 and audited == sorted(audited)
    assert rec['audited'] == 16 and rec['voted_on_tier1'] == len(voters) and rec['tau1'] == 0.040
    flips = [d for d in rec['disagreements']]
    assert flips == ([(7, 3, 5, pytest.approx(0.07, abs=1e-6))] if 7 in voters else [])
    assert eng.modes[-2:] == [1, 1] and rc.audit_log[-1] is rec and 'audit:' in lines[-1]      # the first pass is evaluated in the exact-vote mode (its classifier tier), the mode restored
    # certify(audit=k) audits every example
    y, r = rc.certify(x.reshape(1, 1, 16000), torch.tensor([3]), sigma=0.5, n_0=10, n=40, batch_size=8, audit=8)
    assert int(y[0]) == 3 and len(rc.audit_log) == 2 and rc.audit_log[-1]['audited'] == 8
